MODULE cdfio
  !!======================================================================
  !! cdfio : NetCDF access layer of the tools -- the module surface the PROGRAM units are written against.
  !! What is contract and what is not: the public names, their argument lists (keyword names included), defaults,
  !! messages and STOP codes are those of the reference's src/cdfio.F90 (cited per routine), for the part of its
  !! public list that the tools of this build call; everything behind them is this build's own:
  !!   - source files are not opened and closed at every call as in the reference: a small table of read handles
  !!     (src_attach) keeps the files of a run open, so a level loop of getvar costs one header parse, not npk
  !!     (CDFT_IO_NOCACHE=1 goes back to open / close per call; create and closeout drop the table);
  !!   - one routine (place_slab) decides where a (level, record) slab lies in a variable of any rank, one
  !!     (t_pack / pack_apply) does the scale_factor / add_offset / savelog10 unpacking for every reader;
  !!   - readvar3d / readvar3d_i1 (not in the reference) read a whole (kpi,kpj,kpk) record in ONE call straight
  !!     into the caller's array -- what a GPU feed wants.
  !! Kept conventions: getvar returns one (kpi,kpj) r4 slab; chkfile / chkvar return .TRUE. when the file / variable
  !! is MISSING; one output file is built at a time through module-level ids (not re-entrant, src/cdfio.F90:72-73);
  !! fatal NetCDF errors print a message and STOP 98.
  !! Not provided (no tool of this build calls them): the reference's atted, cvaratt, gettimeatt, puttimeatt, getvarid,
  !! getdimvar, getvarxz / yz, getvar3dt, getvar4d, gettimeseries, reputvar*, putvare3, ncopen.
  !!======================================================================
  USE, INTRINSIC :: ISO_C_BINDING, ONLY : C_PTR
  USE netcdf
  USE modcdfnames
  IMPLICIT NONE
  PRIVATE

  CHARACTER(LEN=256), PUBLIC :: cg_zgr_ver = 'v3.6'     ! layout of mesh_zgr: v2.0 | v3.0 | v3.6
  LOGICAL,            PUBLIC :: lg_vvl = .FALSE.        ! set by the tools' -vvl option

  TYPE, PUBLIC :: variable                              ! descriptor of an output variable (src/cdfio.F90:87-103)
     CHARACTER(LEN=256) :: cname
     CHARACTER(LEN=256) :: cunits
     REAL(KIND=4)       :: rmissing_value
     REAL(KIND=4)       :: valid_min
     REAL(KIND=4)       :: valid_max
     REAL(KIND=4)       :: scale_factor = 1.
     REAL(KIND=4)       :: add_offset = 0.
     REAL(KIND=4)       :: savelog10 = 0.
     INTEGER(KIND=4)    :: iwght = 1
     INTEGER(KIND=4), DIMENSION(4) :: ichunk = 0        ! NetCDF-4 chunk sizes (x,y,z,t); 0 = the backend chooses
     CHARACTER(LEN=256) :: clong_name
     CHARACTER(LEN=256) :: cshort_name
     CHARACTER(LEN=256) :: conline_operation
     CHARACTER(LEN=256) :: caxis
     CHARACTER(LEN=256) :: cprecision = 'r4'            ! r4 | r8 | i2 | by
  END TYPE variable

  ! ---- the output file under construction: dimension and coordinate-variable ids
  INTEGER(KIND=4) :: nid_x, nid_y, nid_z, nid_t, nid_lon, nid_lat, nid_dep, nid_tim, nid_lon1d, nid_lat1d
  ! ---- what the first getdim of a run remembers for the outputs (src/cdfio.F90:939-994)
  INTEGER(KIND=4)    :: nstart_date = -1
  CHARACTER(LEN=256) :: ctime_units = 'seconds since 0001-01-01 00:00:00', ctime_origin = 'N/A', calendar = 'N/A'
  CHARACTER(LEN=256) :: config = 'N/A', ccase = 'N/A'
  CHARACTER(LEN=10)  :: cfreq = 'N/A'
  LOGICAL            :: l_first_getdim = .TRUE.
  ! ---- CDFT_TIMING=1: wall time of the bulk reads and slab writes, printed on stderr by closeout
  INTEGER(KIND=4)    :: ntiming = -1                    ! -1 not asked yet, 0 off, 1 on
  INTEGER(KIND=8)    :: nclock0
  REAL(KIND=8)       :: dtime_read = 0.d0, dtime_write = 0.d0
  ! ---- read handles kept open (least recently used one gives way)
  INTEGER(KIND=4), PARAMETER :: jpsrc = 16
  CHARACTER(LEN=1024) :: csrc_name(jpsrc) = ' '
  INTEGER(KIND=4)     :: nsrc_ncid(jpsrc) = -1, nsrc_used(jpsrc) = 0, nsrc_clock = 0
  INTEGER(KIND=4)     :: nsrc_keep = -1                 ! -1 not asked yet, 0 open / close per call, 1 keep open

  TYPE t_pack                                           ! how the stored numbers of a variable become values
     REAL(KIND=4) :: scale = 1., offset = 0., spval = 0.
     LOGICAL      :: lscale = .FALSE., loffset = .FALSE., llog10 = .FALSE.
  END TYPE t_pack

  INTERFACE putvar
     MODULE PROCEDURE putvarr8, putvarr4, putvari2, putvarzo, reputvarr4, putvare3
  END INTERFACE
  INTERFACE putvar1d
     MODULE PROCEDURE putvar1d4, putvar1d8, reputvar1d4, reputvar1d8
  END INTERFACE
  INTERFACE putvar0d
     MODULE PROCEDURE putvar0dt, putvar0ds
  END INTERFACE
  INTERFACE getatt
     MODULE PROCEDURE getattr, getatti, getattc
  END INTERFACE

  PUBLIC :: chkfile, chkvar, getdim, getvdim, getspval, getatt, getvaratt, getnvar, getvarname, getipk
  PUBLIC :: getvar, getvar1d, getvare3, getvar3d, readvar3d, readvar3d_i1, var_is_plain_bytes, getvar_to
  PUBLIC :: create, createvar, putatt, putheadervar, putvar, putvar1d, putvar0d, closeout, ERR_HDL
  PUBLIC :: putvar_behind, wait_writes                  ! (not in the reference: the host feed's write-behind)
  PUBLIC :: cdf_phase                                   ! (not in the reference: CDFT_TIMING's phase lines)
  ! the rest of the reference's public list (src/cdfio.F90:173-180): none of the 18 programs of this build calls them; they
  ! are here so that another reference program finds the module surface whole (tests/fortran/cdfio_public.f90 calls each)
  PUBLIC :: copyatt, cvaratt, gettimeatt, puttimeatt, atted, getdimvar, getvarid
  PUBLIC :: getvarxz, getvaryz, getvar3dt, getvar4d, gettimeseries, ncopen

  INTERFACE atted
     MODULE PROCEDURE atted_char, atted_r4
  END INTERFACE

CONTAINS

  ! ================================================================================================ small tools
  SUBROUTINE ERR_HDL(kstatus)
    !! any NetCDF status other than success ends the program (src/cdfio.F90:2924-2940)
    INTEGER(KIND=4), INTENT(in) :: kstatus
    IF ( kstatus == NF90_NOERR ) RETURN
    PRINT *, 'ERROR in NETCDF routine, status=', kstatus
    PRINT *, NF90_STRERROR(kstatus)
    STOP 98
  END SUBROUTINE ERR_HDL

  SUBROUTINE give_up(kerr, cdwhere, cdwhat)
    ! ' Problem in <routine> for <variable>' then ERR_HDL, the way every reader of the reference ends
    INTEGER(KIND=4),  INTENT(in) :: kerr
    CHARACTER(LEN=*), INTENT(in) :: cdwhere, cdwhat
    IF ( kerr == NF90_NOERR ) RETURN
    PRINT *, ' Problem in ', cdwhere, ' for ', TRIM(cdwhat)
    CALL ERR_HDL(kerr)
  END SUBROUTINE give_up

  PURE INTEGER(KIND=4) FUNCTION oint(kopt, kdefault)
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kopt
    INTEGER(KIND=4),           INTENT(in) :: kdefault
    oint = kdefault
    IF ( PRESENT(kopt) ) oint = kopt
  END FUNCTION oint

  PURE LOGICAL FUNCTION olog(ldopt, lddefault)
    LOGICAL, OPTIONAL, INTENT(in) :: ldopt
    LOGICAL,           INTENT(in) :: lddefault
    olog = lddefault
    IF ( PRESENT(ldopt) ) olog = ldopt
  END FUNCTION olog

  LOGICAL FUNCTION env_is_one(cdname)
    CHARACTER(LEN=*), INTENT(in) :: cdname
    CHARACTER(LEN=8) :: cl
    CALL GET_ENVIRONMENT_VARIABLE(cdname, cl)
    env_is_one = ( cl(1:1) == '1' )
  END FUNCTION env_is_one

  FUNCTION env_value(cdname) RESULT(cl)
    CHARACTER(LEN=*), INTENT(in) :: cdname
    CHARACTER(LEN=256) :: cl
    INTEGER :: ierr
    CALL GET_ENVIRONMENT_VARIABLE(cdname, cl, STATUS=ierr)
    IF ( ierr /= 0 ) cl = ' '
  END FUNCTION env_value

  SUBROUTINE tic()
    IF ( ntiming < 0 ) ntiming = MERGE(1, 0, env_is_one('CDFT_TIMING'))
    IF ( ntiming == 1 ) CALL SYSTEM_CLOCK(nclock0)
  END SUBROUTINE tic

  SUBROUTINE cdf_phase(cdwhat)
    !! CDFT_TIMING=1: one line on stderr per call, ' CDFT_PHASE <what> <seconds since the previous call (or the first cdfio timing call)>'
    !! -- where a program's wall clock goes outside its reads, writes and library calls (not in the reference)
    CHARACTER(LEN=*), INTENT(in) :: cdwhat
    INTEGER(KIND=8), SAVE :: ilast = -1
    INTEGER(KIND=8) :: inow, irate
    IF ( ntiming < 0 ) ntiming = MERGE(1, 0, env_is_one('CDFT_TIMING'))
    IF ( ntiming /= 1 ) RETURN
    CALL SYSTEM_CLOCK(inow, irate)
    IF ( ilast >= 0 ) WRITE(0,'(a,a,f9.3)') ' CDFT_PHASE ', cdwhat, DBLE(inow - ilast) / DBLE(irate)
    ilast = inow
  END SUBROUTINE cdf_phase

  SUBROUTINE toc(dtotal)
    REAL(KIND=8), INTENT(inout) :: dtotal
    INTEGER(KIND=8) :: inow, irate
    IF ( ntiming /= 1 ) RETURN
    CALL SYSTEM_CLOCK(inow, irate)
    dtotal = dtotal + DBLE(inow - nclock0) / DBLE(irate)
  END SUBROUTINE toc

  ! ================================================================================================ read handles
  INTEGER(KIND=4) FUNCTION src_attach(cdfile, kncid)
    ! a read handle on cdfile: the one already held if there is one, else a new one that takes the place of the least
    ! recently used.  Returns the NetCDF status of the open.
    CHARACTER(LEN=*), INTENT(in)  :: cdfile
    INTEGER(KIND=4),  INTENT(out) :: kncid
    INTEGER(KIND=4) :: js, islot, ierr
    IF ( nsrc_keep < 0 ) nsrc_keep = MERGE(0, 1, env_is_one('CDFT_IO_NOCACHE'))
    IF ( nsrc_keep == 0 ) THEN
       src_attach = NF90_OPEN(cdfile, NF90_NOWRITE, kncid)
       RETURN
    ENDIF
    nsrc_clock = nsrc_clock + 1
    islot = 1
    DO js = 1, jpsrc
       IF ( nsrc_ncid(js) >= 0 .AND. csrc_name(js) == cdfile ) THEN
          nsrc_used(js) = nsrc_clock
          kncid = nsrc_ncid(js)
          src_attach = NF90_NOERR
          RETURN
       ENDIF
       IF ( nsrc_used(js) < nsrc_used(islot) ) islot = js
    END DO
    src_attach = NF90_OPEN(cdfile, NF90_NOWRITE, kncid)
    IF ( src_attach /= NF90_NOERR ) RETURN
    IF ( nsrc_ncid(islot) >= 0 ) ierr = NF90_CLOSE(nsrc_ncid(islot))
    csrc_name(islot) = cdfile ; nsrc_ncid(islot) = kncid ; nsrc_used(islot) = nsrc_clock
  END FUNCTION src_attach

  SUBROUTINE src_detach(kncid)
    ! the end of one read: nothing to do for a handle of the table
    INTEGER(KIND=4), INTENT(in) :: kncid
    INTEGER(KIND=4) :: ierr
    IF ( nsrc_keep == 0 ) ierr = NF90_CLOSE(kncid)
  END SUBROUTINE src_detach

  SUBROUTINE src_forget(cdfile)
    ! closes the handles of the table: the one on cdfile (a file about to be written), or all of them
    CHARACTER(LEN=*), OPTIONAL, INTENT(in) :: cdfile
    INTEGER(KIND=4) :: js, ierr
    DO js = 1, jpsrc
       IF ( nsrc_ncid(js) < 0 ) CYCLE
       IF ( PRESENT(cdfile) ) THEN
          IF ( csrc_name(js) /= cdfile ) CYCLE
       ENDIF
       ierr = NF90_CLOSE(nsrc_ncid(js))
       nsrc_ncid(js) = -1 ; nsrc_used(js) = 0 ; csrc_name(js) = ' '
    END DO
  END SUBROUTINE src_forget

  SUBROUTINE src_variable(cdfile, cdvar, kncid, kvar)
    ! handle and variable id, or the end of the program (the reference's CALL ERR_HDL(NF90_OPEN / NF90_INQ_VARID))
    CHARACTER(LEN=*), INTENT(in)  :: cdfile, cdvar
    INTEGER(KIND=4),  INTENT(out) :: kncid, kvar
    CALL ERR_HDL(src_attach(cdfile, kncid))
    CALL ERR_HDL(NF90_INQ_VARID(kncid, cdvar, kvar))
  END SUBROUTINE src_variable

  ! ================================================================================================ where and how
  FUNCTION zgr_name(cdvar, ldiom) RESULT(clvar)
    ! the name under which mesh_zgr holds a 3-D vertical metric in the detected layout (src/cdfio.F90:1510-1536);
    ! any other name, or ldiom absent / false, is returned as it is
    CHARACTER(LEN=*),  INTENT(in) :: cdvar
    LOGICAL, OPTIONAL, INTENT(in) :: ldiom
    CHARACTER(LEN=256) :: clvar
    CHARACTER(LEN=3)   :: clroot
    clvar = cdvar
    IF ( .NOT. olog(ldiom, .FALSE.) ) RETURN
    IF      ( cdvar == cn_ve3t ) THEN ; clroot = 'e3t'
    ELSE IF ( cdvar == cn_ve3u ) THEN ; clroot = 'e3u'
    ELSE IF ( cdvar == cn_ve3v ) THEN ; clroot = 'e3v'
    ELSE IF ( cdvar == cn_ve3w ) THEN ; clroot = 'e3w'
    ELSE ; RETURN
    ENDIF
    IF ( cg_zgr_ver == 'v2.0' ) clvar = clroot//'_ps'
    IF ( cg_zgr_ver == 'v3.0' ) clvar = clroot
    IF ( cg_zgr_ver == 'v3.6' ) clvar = clroot//'_0'
  END FUNCTION zgr_name

  SUBROUTINE place_slab(kncid, kvar, klev, ktime, kstart, ldlevels)
    ! Third and fourth start index of the (level klev, record ktime) slab of a variable.  A variable whose third
    ! dimension is the record one is (x,y,t): the record goes there and it has no levels; anything else is addressed as
    ! (x,y,z,t) (src/cdfio.F90:1544-1554; lower ranks ignore the indices they do not have).  ldlevels: the variable has a
    ! level dimension of its own, so a block of levels can be read in one piece.
    INTEGER(KIND=4), INTENT(in)    :: kncid, kvar, klev, ktime
    INTEGER(KIND=4), INTENT(inout) :: kstart(4)
    LOGICAL, OPTIONAL, INTENT(out) :: ldlevels
    INTEGER(KIND=4) :: ierr, irec, irank, idim(4)
    idim = 0
    ierr = NF90_INQUIRE(kncid, unlimitedDimId=irec)
    ierr = NF90_INQUIRE_VARIABLE(kncid, kvar, ndims=irank, dimids=idim)
    IF ( idim(3) == irec ) THEN
       kstart(3:4) = (/ ktime, 1 /)
    ELSE
       kstart(3:4) = (/ klev, ktime /)
    ENDIF
    IF ( PRESENT(ldlevels) ) ldlevels = ( irank >= 3 .AND. idim(3) /= irec )
  END SUBROUTINE place_slab

  SUBROUTINE lookup_spval(kncid, kvar, pspval, cdmissing)
    ! missing value of a variable: cn_missing_value, then the other usual spellings (src/cdfio.F90:1022-1036); 0. if none
    INTEGER(KIND=4), INTENT(in)  :: kncid, kvar
    REAL(KIND=4),    INTENT(out) :: pspval
    CHARACTER(LEN=*), OPTIONAL, INTENT(out) :: cdmissing
    CHARACTER(LEN=13), PARAMETER :: clspell(3) = (/ 'missing_value', 'Fillvalue    ', '_Fillvalue   ' /)
    CHARACTER(LEN=256) :: clname
    INTEGER :: jn
    DO jn = 0, SIZE(clspell)
       IF ( jn == 0 ) THEN ; clname = cn_missing_value
       ELSE                ; clname = clspell(jn)
       ENDIF
       IF ( NF90_GET_ATT(kncid, kvar, TRIM(clname), pspval) == NF90_NOERR ) THEN
          IF ( PRESENT(cdmissing) ) cdmissing = TRIM(clname)
          RETURN
       ENDIF
    END DO
    pspval = 0.
    IF ( PRESENT(cdmissing) ) cdmissing = cn_missing_value
  END SUBROUTINE lookup_spval

  TYPE(t_pack) FUNCTION pack_of(kncid, kvar, cdasked) RESULT(tp)
    ! scale_factor / add_offset / savelog10 of a variable (src/cdfio.F90:1563-1584) and the missing value the unpacking
    ! leaves alone.  The reference looks the missing value up under the name the caller ASKED for (:1561), which for a
    ! metric renamed by ldiom is not the name read: such a variable then has the missing value 0.
    INTEGER(KIND=4),  INTENT(in) :: kncid, kvar
    CHARACTER(LEN=*), INTENT(in) :: cdasked
    INTEGER(KIND=4) :: iflag, iasked
    IF ( NF90_INQ_VARID(kncid, cdasked, iasked) == NF90_NOERR ) CALL lookup_spval(kncid, iasked, tp%spval)
    IF ( NF90_GET_ATT(kncid, kvar, 'savelog10', iflag) == NF90_NOERR ) tp%llog10 = ( iflag /= 0 )
    IF ( NF90_GET_ATT(kncid, kvar, 'scale_factor', tp%scale) == NF90_NOERR ) THEN
       tp%lscale = ( tp%scale /= 1. )
    ELSE
       tp%scale = 1.
    ENDIF
    IF ( NF90_GET_ATT(kncid, kvar, 'add_offset', tp%offset) == NF90_NOERR ) THEN
       tp%loffset = ( tp%offset /= 0. )
    ELSE
       tp%offset = 0.
    ENDIF
  END FUNCTION pack_of

  SUBROUTINE pack_apply(tp, ptab, kn)
    ! scale, then offset, then 10**, each on the values that are not the missing value AT THAT STAGE (the reference's
    ! three WHERE statements in a row, src/cdfio.F90:1600-1603: "order does matter")
    TYPE(t_pack),    INTENT(in)    :: tp
    INTEGER(KIND=8), INTENT(in)    :: kn
    REAL(KIND=4),    INTENT(inout) :: ptab(kn)
    INTEGER(KIND=8) :: jn
    REAL(KIND=4) :: zval
    IF ( .NOT. ( tp%lscale .OR. tp%loffset .OR. tp%llog10 ) ) RETURN
    DO jn = 1, kn
       zval = ptab(jn)
       IF ( tp%lscale  .AND. zval /= tp%spval ) zval = zval * tp%scale
       IF ( tp%loffset .AND. zval /= tp%spval ) zval = zval + tp%offset
       IF ( tp%llog10  .AND. zval /= tp%spval ) zval = 10**zval
       ptab(jn) = zval
    END DO
  END SUBROUTINE pack_apply

  ! ================================================================================================ the rest of the public list
  INTEGER(KIND=4) FUNCTION ncopen(cdfile)
    !! id of cdfile opened for writing (src/cdfio.F90:2907-2922); the caller closes it with closeout
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4) :: ierr
    CALL src_forget(cdfile)                              ! (a read handle kept on it would not see what is written)
    ncopen = -1
    ierr = NF90_OPEN(cdfile, NF90_WRITE, ncopen)
  END FUNCTION ncopen

  INTEGER(KIND=4) FUNCTION copyatt(cdvar, kidvar, kcin, kcout)
    !! every attribute of variable cdvar of the open file kcin onto variable kidvar of kcout; kcin = -9999 (no file to copy
    !! from): the stock attributes of a coordinate named cdvar (src/cdfio.F90:195-257) -- what create does for its axes
    CHARACTER(LEN=*), INTENT(in) :: cdvar
    INTEGER(KIND=4),  INTENT(in) :: kidvar, kcin, kcout
    copyatt = NF90_NOERR
    CALL coordinate_attributes(cdvar, kidvar, kcin, kcout)
  END FUNCTION copyatt

  FUNCTION getvarid(cdfile, knvars)
    !! the ids of the first knvars variables of cdfile, in file order (src/cdfio.F90:462-485)
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4),  INTENT(in) :: knvars
    INTEGER(KIND=4), DIMENSION(knvars) :: getvarid
    INTEGER(KIND=4) :: ncid, jv, ierr
    CHARACTER(LEN=256) :: clname
    getvarid(:) = -1
    IF ( src_attach(cdfile, ncid) /= NF90_NOERR ) RETURN
    DO jv = 1, knvars
       IF ( NF90_INQUIRE_VARIABLE(ncid, jv, name=clname) == NF90_NOERR ) ierr = NF90_INQ_VARID(ncid, TRIM(clname), getvarid(jv))
    END DO
    CALL src_detach(ncid)
  END FUNCTION getvarid

  INTEGER(KIND=4) FUNCTION gettimeatt(cdfile, cdvartime, ctcalendar, cttitle, ctlong_name, ctaxis, ctunits, cttime_origin)
    !! the six text attributes of the time variable, 'unknown' where the file has none (src/cdfio.F90:517-552); the status
    !! returned is that of the last one looked up (time_origin), as in the reference
    CHARACTER(LEN=*), INTENT(in)  :: cdfile, cdvartime
    CHARACTER(LEN=*), INTENT(out) :: ctcalendar, cttitle, ctlong_name, ctaxis, ctunits, cttime_origin
    INTEGER(KIND=4) :: ncid, ivar
    ctcalendar = 'unknown' ; cttitle = 'unknown' ; ctlong_name = 'unknown'
    ctaxis = 'unknown' ; ctunits = 'unknown' ; cttime_origin = 'unknown'
    gettimeatt = src_attach(cdfile, ncid)
    IF ( gettimeatt /= NF90_NOERR ) RETURN
    gettimeatt = NF90_INQ_VARID(ncid, cdvartime, ivar)
    IF ( gettimeatt == NF90_NOERR ) THEN
       gettimeatt = NF90_GET_ATT(ncid, ivar, 'calendar',    ctcalendar)
       gettimeatt = NF90_GET_ATT(ncid, ivar, 'title',       cttitle)
       gettimeatt = NF90_GET_ATT(ncid, ivar, 'long_name',   ctlong_name)
       gettimeatt = NF90_GET_ATT(ncid, ivar, 'axis',        ctaxis)
       gettimeatt = NF90_GET_ATT(ncid, ivar, 'units',       ctunits)
       gettimeatt = NF90_GET_ATT(ncid, ivar, 'time_origin', cttime_origin)
    ENDIF
    CALL src_detach(ncid)
  END FUNCTION gettimeatt

  INTEGER(KIND=4) FUNCTION puttimeatt(kout, cdvartime, ctcalendar, cttitle, ctlong_name, ctaxis, ctunits, cttime_origin)
    !! the same six onto the time variable of the open output file kout (src/cdfio.F90:554-592): a missing variable or a
    !! refused attribute stops the program with 98, as there
    INTEGER(KIND=4),    INTENT(in) :: kout
    CHARACTER(LEN=20),  INTENT(in) :: cdvartime
    CHARACTER(LEN=256), INTENT(in) :: ctcalendar, cttitle, ctlong_name, ctaxis, ctunits, cttime_origin
    INTEGER(KIND=4) :: ivar
    puttimeatt = NF90_INQ_VARID(kout, cdvartime, ivar)
    IF ( puttimeatt /= NF90_NOERR ) THEN
       PRINT *, TRIM(NF90_STRERROR(puttimeatt))
       PRINT *, 'puttimeatt : ', TRIM(cdvartime), ' does not exist'
       STOP 98
    ENDIF
    puttimeatt = NF90_REDEF(kout)
    CALL one('calendar', ctcalendar) ; CALL one('title', cttitle) ; CALL one('long_name', ctlong_name)
    CALL one('axis', ctaxis) ; CALL one('units', ctunits) ; CALL one('time_origin', cttime_origin)
    puttimeatt = NF90_ENDDEF(kout)
  CONTAINS
    SUBROUTINE one(cdatt, cdval)
      CHARACTER(LEN=*), INTENT(in) :: cdatt, cdval
      puttimeatt = NF90_PUT_ATT(kout, ivar, cdatt, cdval)
      IF ( puttimeatt /= NF90_NOERR ) THEN
         PRINT *, NF90_STRERROR(puttimeatt) ; PRINT *, ' in puttimeatt ', cdatt
         STOP 98
      ENDIF
    END SUBROUTINE one
  END FUNCTION puttimeatt

  INTEGER(KIND=4) FUNCTION cvaratt(cdfile, cdvar, cdunits, pmissing_value, cdlong_name, cdshort_name)
    !! units, missing value (under the name the variable already spells it with), long_name and short_name of an EXISTING
    !! variable rewritten in place (src/cdfio.F90:595-626)
    CHARACTER(LEN=256), INTENT(in) :: cdfile, cdvar
    CHARACTER(LEN=256), INTENT(in) :: cdunits, cdlong_name, cdshort_name
    REAL(KIND=4),       INTENT(in) :: pmissing_value
    INTEGER(KIND=4) :: ncid, ivar, ierr
    REAL(KIND=4) :: zold
    CHARACTER(LEN=256) :: clmissing
    zold = getspval(cdfile, cdvar, clmissing)             ! (before the file is opened for writing: its own read handle)
    CALL src_forget(cdfile)
    cvaratt = NF90_OPEN(cdfile, NF90_WRITE, ncid)
    IF ( cvaratt /= NF90_NOERR ) RETURN
    ierr = NF90_REDEF(ncid)
    cvaratt = NF90_INQ_VARID(ncid, cdvar, ivar)
    IF ( cvaratt == NF90_NOERR ) THEN
       ierr = NF90_PUT_ATT(ncid, ivar, 'units', cdunits)
       ierr = NF90_PUT_ATT(ncid, ivar, TRIM(clmissing), pmissing_value)
       ierr = NF90_PUT_ATT(ncid, ivar, 'long_name', cdlong_name)
       ierr = NF90_PUT_ATT(ncid, ivar, 'short_name', cdshort_name)
       cvaratt = NF90_ENDDEF(ncid)
    ENDIF
    ierr = NF90_CLOSE(ncid)
  END FUNCTION cvaratt

  SUBROUTINE atted_open(cdfile, cdvar, cdatt, ktype_wanted, cdwho, kncid, kvar)
    ! the attribute editor's preamble (src/cdfio.F90:790-876): cdfile open for writing, in define mode, the variable found (or
    ! STOP 98), and an existing attribute of another type refused (STOP 98)
    CHARACTER(LEN=*), INTENT(in)  :: cdfile, cdvar, cdatt, cdwho
    INTEGER(KIND=4),  INTENT(in)  :: ktype_wanted
    INTEGER(KIND=4),  INTENT(out) :: kncid, kvar
    INTEGER(KIND=4) :: ierr, itype
    CALL src_forget(cdfile)
    ierr = NF90_OPEN(cdfile, NF90_WRITE, kncid)
    ierr = NF90_INQ_VARID(kncid, cdvar, kvar)
    IF ( ierr /= NF90_NOERR ) THEN
       PRINT *, NF90_STRERROR(ierr), ' in atted ( inq_varid)'
       STOP 98
    ENDIF
    ierr = NF90_INQUIRE_ATTRIBUTE(kncid, kvar, cdatt, xtype=itype)
    IF ( ierr /= NF90_NOERR ) THEN
       PRINT *, ' Attribute does not exist. Create it'
    ELSE IF ( itype /= ktype_wanted ) THEN
       PRINT *, ' Mismatch in attribute type in ', cdwho
       STOP 98
    ENDIF
    ierr = NF90_REDEF(kncid)
  END SUBROUTINE atted_open

  INTEGER(KIND=4) FUNCTION atted_char(cdfile, cdvar, cdatt, cdvalue)
    !! a text attribute of a variable of an existing file set or created (src/cdfio.F90:790-830)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt, cdvalue
    INTEGER(KIND=4) :: ncid, ivar, ierr
    CALL atted_open(cdfile, cdvar, cdatt, NF90_CHAR, 'atted_char', ncid, ivar)
    atted_char = NF90_PUT_ATT(ncid, ivar, cdatt, cdvalue)
    ierr = NF90_CLOSE(ncid)
  END FUNCTION atted_char

  INTEGER(KIND=4) FUNCTION atted_r4(cdfile, cdvar, cdatt, pvalue)
    !! the same for an r4 attribute (src/cdfio.F90:833-876)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt
    REAL(KIND=4),     INTENT(in) :: pvalue
    INTEGER(KIND=4) :: ncid, ivar, ierr
    CALL atted_open(cdfile, cdvar, cdatt, NF90_FLOAT, 'atted_r4', ncid, ivar)
    atted_r4 = NF90_PUT_ATT(ncid, ivar, cdatt, pvalue)
    ierr = NF90_CLOSE(ncid)
  END FUNCTION atted_r4

  FUNCTION getdimvar(cdfile, kpk, cd_depnam)
    !! the values of the file's vertical coordinate, found by trial: the first variable with ONE dimension that is not the
    !! record one (src/cdfio.F90:1085-1127); zeros and a message when there is none
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4),  INTENT(in) :: kpk
    REAL(KIND=4), DIMENSION(kpk) :: getdimvar
    CHARACTER(LEN=*), OPTIONAL, INTENT(out) :: cd_depnam
    INTEGER(KIND=4) :: ncid, jv, invar, irec, irank, idim(4), ierr
    CHARACTER(LEN=80) :: clvar
    LOGICAL :: ll_found
    clvar = 'none' ; ll_found = .FALSE.
    getdimvar(:) = 0.
    CALL ERR_HDL(src_attach(cdfile, ncid))
    ierr = NF90_INQUIRE(ncid, nVariables=invar, unlimitedDimId=irec)
    DO jv = 1, invar
       ierr = NF90_INQUIRE_VARIABLE(ncid, jv, name=clvar, ndims=irank, dimids=idim)
       IF ( irank == 1 .AND. idim(1) /= irec ) THEN
          PRINT *, ' Found vertical variable :', TRIM(clvar)
          ierr = NF90_GET_VAR(ncid, jv, getdimvar)
          ll_found = .TRUE.
          EXIT
       ENDIF
    END DO
    IF ( PRESENT(cd_depnam) ) cd_depnam = TRIM(clvar)      ! (the last name looked at when nothing was found, as there)
    IF ( .NOT. ll_found ) THEN
       PRINT *, 'Sorry, no vertical dim variables inferred ...'
       getdimvar(:) = 0.
    ENDIF
    CALL src_detach(ncid)
  END FUNCTION getdimvar

  SUBROUTINE read_block(cdwho, cdfile, cdvar, ptab, kn, kstart, kcount, ldzgr)
    ! kn values of cdvar from kstart, shape kcount, unpacked like getvar's slabs; a NetCDF error names cdwho and stops with 98
    CHARACTER(LEN=*), INTENT(in)  :: cdwho, cdfile, cdvar
    INTEGER(KIND=8),  INTENT(in)  :: kn
    REAL(KIND=4),     INTENT(out) :: ptab(kn)
    INTEGER(KIND=4),  INTENT(in)  :: kstart(4), kcount(4)
    LOGICAL,          INTENT(in)  :: ldzgr
    INTEGER(KIND=4) :: ncid, ivar, ierr
    CHARACTER(LEN=256) :: cl_name
    TYPE(t_pack) :: tp
    cl_name = cdvar
    IF ( ldzgr .AND. cdvar == cn_ve3v ) cl_name = zgr_name(cdvar, .TRUE.)      ! (the one metric these sections rename, :1985-1991)
    CALL tic()
    CALL src_variable(cdfile, cl_name, ncid, ivar)
    tp = pack_of(ncid, ivar, cl_name)
    ierr = NF90_GET_VAR(ncid, ivar, ptab, start=kstart, count=kcount)
    IF ( ierr /= NF90_NOERR ) THEN
       PRINT *, ' Problem in ', cdwho, ' for ', TRIM(cl_name)
       CALL ERR_HDL(ierr)
    ENDIF
    CALL pack_apply(tp, ptab, kn)
    CALL src_detach(ncid)
    CALL toc(dtime_read)
  END SUBROUTINE read_block

  LOGICAL FUNCTION has_y(cdfile)
    ! whether the file has a dimension called 'y' (the sections' test, src/cdfio.F90:2018, :2124)
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4) :: ncid, idum
    CALL ERR_HDL(src_attach(cdfile, ncid))
    has_y = ( NF90_INQ_DIMID(ncid, 'y', idum) == NF90_NOERR )
    CALL src_detach(ncid)
  END FUNCTION has_y

  FUNCTION getvarxz(cdfile, cdvar, kj, kpi, kpz, kimin, kkmin, ktime)
    !! the (x,z) section of cdvar at row kj of record ktime (src/cdfio.F90:1933-2043): (x,y,z,t) files; a file without a 'y'
    !! dimension is read as (x,z,t)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kj, kpi, kpz
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kkmin, ktime
    REAL(KIND=4), DIMENSION(kpi,kpz) :: getvarxz
    IF ( has_y(cdfile) ) THEN
       CALL read_block('getvarxz', cdfile, cdvar, getvarxz, INT(kpi, 8) * kpz, (/ oint(kimin, 1), kj, oint(kkmin, 1), oint(ktime, 1) /), &
            &          (/ kpi, 1, kpz, 1 /), .TRUE.)
    ELSE
       CALL read_block('getvarxz', cdfile, cdvar, getvarxz, INT(kpi, 8) * kpz, (/ oint(kimin, 1), oint(kkmin, 1), oint(ktime, 1), 1 /), &
            &          (/ kpi, kpz, 1, 1 /), .TRUE.)
    ENDIF
  END FUNCTION getvarxz

  FUNCTION getvaryz(cdfile, cdvar, ki, kpj, kpz, kjmin, kkmin, ktime)
    !! the (y,z) section at column ki (src/cdfio.F90:2046-2148); a file without an 'x' dimension is read as (y,z,t)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: ki, kpj, kpz
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kjmin, kkmin, ktime
    REAL(KIND=4), DIMENSION(kpj,kpz) :: getvaryz
    INTEGER(KIND=4) :: ncid, idum
    LOGICAL :: ll_x
    CALL ERR_HDL(src_attach(cdfile, ncid))
    ll_x = ( NF90_INQ_DIMID(ncid, 'x', idum) == NF90_NOERR )
    CALL src_detach(ncid)
    IF ( ll_x ) THEN
       CALL read_block('getvaryz', cdfile, cdvar, getvaryz, INT(kpj, 8) * kpz, (/ ki, oint(kjmin, 1), oint(kkmin, 1), oint(ktime, 1) /), &
            &          (/ 1, kpj, kpz, 1 /), .TRUE.)
    ELSE
       CALL read_block('getvaryz', cdfile, cdvar, getvaryz, INT(kpj, 8) * kpz, (/ oint(kjmin, 1), oint(kkmin, 1), oint(ktime, 1), 1 /), &
            &          (/ kpj, kpz, 1, 1 /), .TRUE.)
    ENDIF
  END FUNCTION getvaryz

  FUNCTION getvar3dt(cdfile, cdvar, kk, kpi, kpj, kpt, kimin, kjmin, ktmin)
    !! kpt records of level kk of cdvar, (x,y,t) (src/cdfio.F90:1717-1820): a 4-D variable at that level, a 3-D one (x,y,t) as it is
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kk, kpi, kpj, kpt
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kjmin, ktmin
    REAL(KIND=4), DIMENSION(kpi,kpj,kpt) :: getvar3dt
    PRINT *, '    KPI KPJ KPT, KK ', kpi, kpj, kpt, kk
    IF ( getvdim_all(cdfile, cdvar) == 4 ) THEN
       CALL read_block('getvar3dt', cdfile, cdvar, getvar3dt, INT(kpi, 8) * kpj * kpt, (/ oint(kimin, 1), oint(kjmin, 1), kk, oint(ktmin, 1) /), &
            &          (/ kpi, kpj, 1, kpt /), .FALSE.)
    ELSE
       CALL read_block('getvar3dt', cdfile, cdvar, getvar3dt, INT(kpi, 8) * kpj * kpt, (/ oint(kimin, 1), oint(kjmin, 1), oint(ktmin, 1), 1 /), &
            &          (/ kpi, kpj, kpt, 1 /), .FALSE.)
    ENDIF
  END FUNCTION getvar3dt

  FUNCTION getvar4d(cdfile, cdvar, kpi, kpj, kpz, kpt, kimin, kjmin, kkmin, ktmin)
    !! a (x,y,z,t) block in one read (src/cdfio.F90:1822-1930)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj, kpz, kpt
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kjmin, kkmin, ktmin
    REAL(KIND=4), DIMENSION(kpi,kpj,kpz,kpt) :: getvar4d
    CALL read_block('getvar4d', cdfile, cdvar, getvar4d, INT(kpi, 8) * kpj * kpz * kpt, &
         &          (/ oint(kimin, 1), oint(kjmin, 1), oint(kkmin, 1), oint(ktmin, 1) /), (/ kpi, kpj, kpz, kpt /), .FALSE.)
  END FUNCTION getvar4d

  INTEGER(KIND=4) FUNCTION getvdim_all(cdfile, cdvar)
    ! the rank of cdvar as the file declares it (getvdim answers rank - 1, the reference's convention)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4) :: ncid, ivar, ierr
    CALL src_variable(cdfile, cdvar, ncid, ivar)
    ierr = NF90_INQUIRE_VARIABLE(ncid, ivar, ndims=getvdim_all)
    CALL src_detach(ncid)
  END FUNCTION getvdim_all

  SUBROUTINE last_dim_is_record(kncid, kvar, krank, ldrec)
    ! rank of a variable and whether its LAST dimension is the record one (the re-writers' test, src/cdfio.F90:2623, :2534)
    INTEGER(KIND=4), INTENT(in)  :: kncid, kvar
    INTEGER(KIND=4), INTENT(out) :: krank
    LOGICAL,         INTENT(out) :: ldrec
    INTEGER(KIND=4) :: ierr, irec, idim(4)
    idim = 0 ; krank = 0
    ierr = NF90_INQUIRE(kncid, unlimitedDimId=irec)
    ierr = NF90_INQUIRE_VARIABLE(kncid, kvar, ndims=krank, dimids=idim)
    ldrec = .FALSE.
    IF ( krank >= 1 .AND. krank <= 4 ) ldrec = ( idim(krank) == irec )
  END SUBROUTINE last_dim_is_record

  INTEGER(KIND=4) FUNCTION reputvarr4(cdfile, cdvar, klev, kpi, kpj, kimin, kjmin, ktime, ptab, kwght)
    !! putvar given a FILE NAME: a (kpi,kpj) r4 slab written over what variable cdvar of the existing file holds at level klev,
    !! record ktime, window origin (kimin,kjmin) (src/cdfio.F90:2591-2639)
    CHARACTER(LEN=*),                 INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: klev
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: kimin, kjmin, ktime
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptab
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: kwght
    INTEGER(KIND=4) :: ncid, ivar, ierr, irank, ifirst(4), ishape(4)
    LOGICAL :: ll_rec
    CALL src_forget(cdfile)
    reputvarr4 = NF90_OPEN(cdfile, NF90_WRITE, ncid)
    IF ( reputvarr4 /= NF90_NOERR ) RETURN
    reputvarr4 = NF90_INQ_VARID(ncid, cdvar, ivar)
    IF ( reputvarr4 == NF90_NOERR ) THEN
       CALL last_dim_is_record(ncid, ivar, irank, ll_rec)
       ifirst = (/ oint(kimin, 1), oint(kjmin, 1), 1, 1 /) ; ishape = (/ kpi, kpj, 1, 1 /)
       IF ( ll_rec )     ifirst(irank) = oint(ktime, 1)
       IF ( irank == 4 ) ifirst(3) = oint(klev, 1)
       reputvarr4 = NF90_PUT_VAR(ncid, ivar, ptab, start=ifirst, count=ishape)
       IF ( PRESENT(kwght) ) THEN
          ierr = NF90_REDEF(ncid)
          reputvarr4 = NF90_PUT_ATT(ncid, ivar, 'iweight', kwght)
       ENDIF
    ENDIF
    ierr = NF90_CLOSE(ncid)
  END FUNCTION reputvarr4

  INTEGER(KIND=4) FUNCTION putvare3(kout, kid, ptab, kmax, cd_vert, ktime, kwght)
    !! a vertical profile of kmax values into variable kid (1,1,kmax[,t]) of the open file kout; cd_vert only tells this
    !! specific of putvar from the others (src/cdfio.F90:2501-2543)
    INTEGER(KIND=4),               INTENT(in) :: kout, kid, kmax
    REAL(KIND=4), DIMENSION(kmax), INTENT(in) :: ptab
    CHARACTER(LEN=*),              INTENT(in) :: cd_vert
    INTEGER(KIND=4), OPTIONAL,     INTENT(in) :: ktime, kwght
    INTEGER(KIND=4) :: irank, ifirst(4), ishape(4)
    LOGICAL :: ll_rec
    CALL last_dim_is_record(kout, kid, irank, ll_rec)
    ifirst = 1 ; ishape = (/ 1, 1, kmax, 1 /)
    IF ( ll_rec ) ifirst(irank) = oint(ktime, 1)
    putvare3 = NF90_PUT_VAR(kout, kid, ptab, start=ifirst, count=ishape)
    IF ( PRESENT(kwght) ) putvare3 = NF90_PUT_ATT(kout, kid, 'iweight', kwght)
  END FUNCTION putvare3

  INTEGER(KIND=4) FUNCTION reputvar1d4(cdfile, cdvar, ptab, kk)
    !! kk r4 values over the 1-D variable cdvar of an existing file (src/cdfio.F90:2758-2781)
    CHARACTER(LEN=*),            INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),             INTENT(in) :: kk
    REAL(KIND=4), DIMENSION(kk), INTENT(in) :: ptab
    INTEGER(KIND=4) :: ncid, ivar, ierr
    CALL src_forget(cdfile)
    reputvar1d4 = NF90_OPEN(cdfile, NF90_WRITE, ncid)
    IF ( reputvar1d4 /= NF90_NOERR ) RETURN
    reputvar1d4 = NF90_INQ_VARID(ncid, cdvar, ivar)
    IF ( reputvar1d4 == NF90_NOERR ) reputvar1d4 = NF90_PUT_VAR(ncid, ivar, ptab, start=(/ 1 /), count=(/ kk /))
    ierr = NF90_CLOSE(ncid)
  END FUNCTION reputvar1d4

  INTEGER(KIND=4) FUNCTION reputvar1d8(cdfile, cdvar, ddtab, kk)
    !! the same for r8 values (src/cdfio.F90:2783-2806)
    CHARACTER(LEN=*),            INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),             INTENT(in) :: kk
    REAL(KIND=8), DIMENSION(kk), INTENT(in) :: ddtab
    INTEGER(KIND=4) :: ncid, ivar, ierr
    CALL src_forget(cdfile)
    reputvar1d8 = NF90_OPEN(cdfile, NF90_WRITE, ncid)
    IF ( reputvar1d8 /= NF90_NOERR ) RETURN
    reputvar1d8 = NF90_INQ_VARID(ncid, cdvar, ivar)
    IF ( reputvar1d8 == NF90_NOERR ) reputvar1d8 = NF90_PUT_VAR(ncid, ivar, ddtab, start=(/ 1 /), count=(/ kk /))
    ierr = NF90_CLOSE(ncid)
  END FUNCTION reputvar1d8

  SUBROUTINE gettimeseries(cdfile, cdvar, kilook, kjlook, klev, ldncdf)
    !! two columns on stdout, time in days and the value of cdvar at point (kilook,kjlook) [level klev] for every record of the
    !! file (src/cdfio.F90:2943-3029); scale_factor and add_offset applied to every value (no missing-value test there);
    !! ldncdf is accepted and, as in the reference ("not finished"), does nothing more
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kilook, kjlook
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: klev
    LOGICAL,         OPTIONAL, INTENT(in) :: ldncdf
    INTEGER(KIND=4) :: ncid, ivar, idim, inrec, irank, ierr, jt
    REAL(KIND=4), ALLOCATABLE :: ztime(:), zval(:)
    REAL(KIND=4) :: zscale, zoffset, zatt
    CALL ERR_HDL(src_attach(cdfile, ncid))
    ierr = NF90_INQ_DIMID(ncid, cn_t, idim)
    ierr = NF90_INQUIRE_DIMENSION(ncid, idim, len=inrec)
    ALLOCATE( ztime(inrec), zval(inrec) )
    ierr = NF90_INQ_VARID(ncid, cn_vtimec, ivar)
    ierr = NF90_GET_VAR(ncid, ivar, ztime, (/ 1 /), (/ inrec /))
    ierr = NF90_INQ_VARID(ncid, cdvar, ivar)
    zscale = 1. ; zoffset = 0.
    IF ( NF90_GET_ATT(ncid, ivar, 'add_offset', zatt)   == NF90_NOERR ) zoffset = zatt
    IF ( NF90_GET_ATT(ncid, ivar, 'scale_factor', zatt) == NF90_NOERR ) zscale = zatt
    ierr = NF90_INQUIRE_VARIABLE(ncid, ivar, ndims=irank)
    SELECT CASE ( irank )
    CASE ( 3 ) ; ierr = NF90_GET_VAR(ncid, ivar, zval, (/ kilook, kjlook, 1 /), (/ 1, 1, inrec /))
    CASE ( 4 ) ; ierr = NF90_GET_VAR(ncid, ivar, zval, (/ kilook, kjlook, oint(klev, 1), 1 /), (/ 1, 1, 1, inrec /))
    CASE DEFAULT
       PRINT *, '  ERROR : variable ', TRIM(cdvar), ' has ', irank, ' dimensions !. Only 3 or 4 supported'
       STOP 98
    END SELECT
    zval(:) = zval(:) * zscale + zoffset
    DO jt = 1, inrec
       PRINT *, ztime(jt) / 86400., zval(jt)
    END DO
    CALL src_detach(ncid)
    DEALLOCATE( ztime, zval )
  END SUBROUTINE gettimeseries

  ! ================================================================================================ inquiries
  LOGICAL FUNCTION chkfile(cd_file, ld_verbose)
    !! .TRUE. when the file is MISSING ('none' never is); meeting mesh_zgr sets cg_zgr_ver (src/cdfio.F90:3032-3070)
    CHARACTER(LEN=*),  INTENT(in) :: cd_file
    LOGICAL, OPTIONAL, INTENT(in) :: ld_verbose
    LOGICAL :: ll_there
    chkfile = .FALSE.
    IF ( TRIM(cd_file) == 'none' ) RETURN
    INQUIRE(FILE=TRIM(cd_file), EXIST=ll_there)
    chkfile = .NOT. ll_there
    IF ( chkfile ) THEN
       IF ( olog(ld_verbose, .TRUE.) ) PRINT *, ' File ', TRIM(cd_file), ' is missing '
    ELSE IF ( cd_file == cn_fzgr ) THEN
       CALL SetMeshZgrVersion
    ENDIF
  END FUNCTION chkfile

  LOGICAL FUNCTION chkvar(cd_file, cd_var, ld_verbose)
    !! .TRUE. when the variable is MISSING from the file ('none' never is) (src/cdfio.F90:3072-3112)
    CHARACTER(LEN=*),  INTENT(in) :: cd_file, cd_var
    LOGICAL, OPTIONAL, INTENT(in) :: ld_verbose
    INTEGER(KIND=4) :: ierr, ncid, ivar
    chkvar = .FALSE.
    IF ( TRIM(cd_var) == 'none' ) RETURN
    ierr = src_attach(cd_file, ncid)
    IF ( ierr == NF90_NOERR ) THEN
       ierr = NF90_INQ_VARID(ncid, cd_var, ivar)
       CALL src_detach(ncid)
    ENDIF
    chkvar = ( ierr /= NF90_NOERR )
    IF ( chkvar .AND. olog(ld_verbose, .TRUE.) ) PRINT *, ' Var ', TRIM(cd_var), ' is missing in file ', TRIM(cd_file)
  END FUNCTION chkvar

  SUBROUTINE SetMeshZgrVersion()
    !! e3t_0 absent: v2.0; e3t_0(t,z): v3.0; e3t_0(t,z,y,x): v3.6 (src/cdfio.F90:3310-3335)
    CHARACTER(LEN=256) :: cl_probe
    cl_probe = 'e3t_0'
    cg_zgr_ver = 'v3.6'
    IF ( chkvar(cn_fzgr, cl_probe, .FALSE.) ) THEN
       cg_zgr_ver = 'v2.0'
    ELSE IF ( getvdim(cn_fzgr, cl_probe) == 1 ) THEN
       cg_zgr_ver = 'v3.0'
    ENDIF
    PRINT *, ' mesh_zgr version is ', TRIM(cg_zgr_ver)
  END SUBROUTINE SetMeshZgrVersion

  INTEGER(KIND=4) FUNCTION getdim(cdfile, cdim_name, cdtrue, kstatus, ldexact)
    !! Length of the first dimension whose name CONTAINS cdim_name (the exact name for cn_x or with ldexact: a missing
    !! exact name is fatal); 0, 'unknown' and kstatus = 1 when there is none.  The first call of a run also remembers the
    !! time attributes and the global attributes that the outputs re-emit (src/cdfio.F90:876-1003).
    CHARACTER(LEN=*),             INTENT(in ) :: cdfile, cdim_name
    CHARACTER(LEN=256), OPTIONAL, INTENT(out) :: cdtrue
    INTEGER(KIND=4),    OPTIONAL, INTENT(out) :: kstatus
    LOGICAL,            OPTIONAL, INTENT(in ) :: ldexact
    INTEGER(KIND=4) :: ncid, ierr, indim, jd, idim, ilen
    CHARACTER(LEN=256) :: cl_dim, cl_found
    getdim = 0 ; cl_found = 'unknown'
    IF ( src_attach(cdfile, ncid) /= NF90_NOERR ) THEN
       IF ( PRESENT(kstatus) ) kstatus = 1
       IF ( PRESENT(cdtrue) ) cdtrue = cl_found
       RETURN
    ENDIF
    IF ( cdim_name == cn_x .OR. olog(ldexact, .FALSE.) ) THEN
       ierr = NF90_INQ_DIMID(ncid, cdim_name, idim)
       IF ( ierr /= NF90_NOERR ) THEN
          PRINT *, NF90_STRERROR(ierr)
          PRINT *, ' Exact dimension name ', TRIM(cdim_name), ' not found in ', TRIM(cdfile) ; STOP 98
       ENDIF
       ierr = NF90_INQUIRE_DIMENSION(ncid, idim, len=getdim)
       cl_found = cdim_name
    ELSE
       ierr = NF90_INQUIRE(ncid, nDimensions=indim)
       DO jd = 1, indim
          ierr = NF90_INQUIRE_DIMENSION(ncid, jd, name=cl_dim, len=ilen)
          IF ( INDEX(cl_dim, TRIM(cdim_name)) == 0 ) CYCLE
          getdim = ilen ; cl_found = cl_dim
          EXIT
       END DO
    ENDIF
    IF ( PRESENT(kstatus) ) kstatus = MERGE(1, 0, cl_found == 'unknown')
    IF ( PRESENT(cdtrue) ) cdtrue = cl_found
    IF ( l_first_getdim ) CALL remember_attributes(ncid)
    l_first_getdim = .FALSE.
    CALL src_detach(ncid)
  END FUNCTION getdim

  SUBROUTINE remember_attributes(kncid)
    ! units / time_origin / calendar of the time variable, and start_date, output_frequency, CONFIG, CASE: the file's
    ! global attribute first, then the environment variable of that name, then a default (src/cdfio.F90:939-994)
    INTEGER(KIND=4), INTENT(in) :: kncid
    INTEGER(KIND=4) :: ierr, itim
    CHARACTER(LEN=256) :: cl, cl_freq
    IF ( NF90_INQ_VARID(kncid, cn_vtimec, itim) == NF90_NOERR ) THEN
       IF ( NF90_GET_ATT(kncid, itim, 'units', cl)       == NF90_NOERR ) ctime_units  = cl
       IF ( NF90_GET_ATT(kncid, itim, 'time_origin', cl) == NF90_NOERR ) ctime_origin = cl
       IF ( NF90_GET_ATT(kncid, itim, 'calendar', cl)    == NF90_NOERR ) calendar     = cl
    ENDIF
    nstart_date = -1
    IF ( NF90_INQUIRE_ATTRIBUTE(kncid, NF90_GLOBAL, 'start_date') == NF90_NOERR ) THEN
       ierr = NF90_GET_ATT(kncid, NF90_GLOBAL, 'start_date', nstart_date)
    ELSE
       cl = env_value('start_date')
       IF ( cl /= ' ' ) READ(cl, *) nstart_date
    ENDIF
    cl_freq = global_text('output_frequency') ; cfreq = TRIM(cl_freq)
    config  = global_text('CONFIG')
    ccase   = global_text('CASE')
  CONTAINS
    FUNCTION global_text(cdname) RESULT(clval)
      CHARACTER(LEN=*), INTENT(in) :: cdname
      CHARACTER(LEN=256) :: clval
      clval = ' '
      IF ( NF90_INQUIRE_ATTRIBUTE(kncid, NF90_GLOBAL, cdname) == NF90_NOERR ) THEN
         ierr = NF90_GET_ATT(kncid, NF90_GLOBAL, cdname, clval)
         RETURN
      ENDIF
      clval = env_value(cdname)
      IF ( clval == ' ' ) clval = 'N/A'
    END FUNCTION global_text
  END SUBROUTINE remember_attributes

  INTEGER(KIND=4) FUNCTION getvdim(cdfile, cdvar)
    !! number of dimensions of cdvar minus one (src/cdfio.F90:1042-1083); a missing variable is fatal here (the reference
    !! prompts interactively for another one)
    CHARACTER(LEN=*), INTENT(in)    :: cdfile
    CHARACTER(LEN=*), INTENT(inout) :: cdvar
    INTEGER(KIND=4) :: ncid, ivar, irank
    CALL ERR_HDL(src_attach(cdfile, ncid))
    IF ( NF90_INQ_VARID(ncid, cdvar, ivar) /= NF90_NOERR ) THEN
       PRINT *, ' getvdim : variable ', TRIM(cdvar), ' not found in ', TRIM(cdfile) ; STOP 98
    ENDIF
    CALL ERR_HDL(NF90_INQUIRE_VARIABLE(ncid, ivar, ndims=irank))
    getvdim = irank - 1
    CALL src_detach(ncid)
  END FUNCTION getvdim

  REAL(KIND=4) FUNCTION getspval(cdfile, cdvar, cdmissing)
    !! missing value of cdvar under any of its usual attribute names, 0. if none (src/cdfio.F90:1006-1039)
    CHARACTER(LEN=*),           INTENT(in ) :: cdfile, cdvar
    CHARACTER(LEN=*), OPTIONAL, INTENT(out) :: cdmissing
    INTEGER(KIND=4) :: ncid, ivar
    getspval = 0.
    IF ( PRESENT(cdmissing) ) cdmissing = cn_missing_value
    IF ( src_attach(cdfile, ncid) /= NF90_NOERR ) RETURN
    IF ( NF90_INQ_VARID(ncid, cdvar, ivar) == NF90_NOERR ) CALL lookup_spval(ncid, ivar, getspval, cdmissing)
    CALL src_detach(ncid)
  END FUNCTION getspval

  ! getatt, three result types (src/cdfio.F90:688-795): the attribute, or a default after a five-line report
  INTEGER(KIND=4) FUNCTION att_lookup(cdfile, cdvar, cdatt, kncid, kvar)
    CHARACTER(LEN=*), INTENT(in)  :: cdfile, cdvar, cdatt
    INTEGER(KIND=4),  INTENT(out) :: kncid, kvar
    att_lookup = src_attach(cdfile, kncid)
    att_lookup = NF90_INQ_VARID(kncid, cdvar, kvar)
    IF ( att_lookup /= NF90_NOERR ) PRINT *, TRIM(NF90_STRERROR(att_lookup)), ' when looking for ', TRIM(cdvar), ' in getatt.'
  END FUNCTION att_lookup

  SUBROUTINE att_report(kerr, cdfile, cdvar, cdatt, cddefault)
    INTEGER(KIND=4),  INTENT(in) :: kerr
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt, cddefault
    PRINT *, ' getatt problem :', NF90_STRERROR(kerr)
    PRINT *, ' attribute :', TRIM(cdatt)
    PRINT *, ' variable  :', TRIM(cdvar)
    PRINT *, ' file      :', TRIM(cdfile)
    PRINT *, ' return default ', cddefault
  END SUBROUTINE att_report

  REAL(KIND=4) FUNCTION getattr(cdfile, cdvar, cdatt)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt
    INTEGER(KIND=4) :: ierr, ncid, ivar
    ierr = att_lookup(cdfile, cdvar, cdatt, ncid, ivar)
    ierr = NF90_GET_ATT(ncid, ivar, cdatt, getattr)
    IF ( ierr /= NF90_NOERR ) THEN
       getattr = 0.
       CALL att_report(ierr, cdfile, cdvar, cdatt, '0')
    ENDIF
    CALL src_detach(ncid)
  END FUNCTION getattr

  INTEGER(KIND=4) FUNCTION getatti(cdfile, cdvar, cdatt, kidum)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt
    INTEGER(KIND=4) :: kidum                             ! only selects this specific
    INTEGER(KIND=4) :: ierr, ncid, ivar
    ierr = att_lookup(cdfile, cdvar, cdatt, ncid, ivar)
    ierr = NF90_GET_ATT(ncid, ivar, cdatt, getatti)
    IF ( ierr /= NF90_NOERR ) THEN
       getatti = 0
       CALL att_report(ierr, cdfile, cdvar, cdatt, '0')
    ENDIF
    CALL src_detach(ncid)
  END FUNCTION getatti

  CHARACTER(LEN=256) FUNCTION getattc(cdfile, cdvar, cdatt, cdum)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt
    CHARACTER(LEN=*) :: cdum                             ! only selects this specific
    INTEGER(KIND=4) :: ierr, ncid, ivar
    ierr = att_lookup(cdfile, cdvar, cdatt, ncid, ivar)
    ierr = NF90_GET_ATT(ncid, ivar, cdatt, getattc)
    IF ( ierr /= NF90_NOERR ) THEN
       getattc = 'N/A'
       CALL att_report(ierr, cdfile, cdvar, cdatt, 'N/A')
    ENDIF
    CALL src_detach(ncid)
  END FUNCTION getattc

  INTEGER(KIND=4) FUNCTION getnvar(cdfile)
    !! number of variables of a file (src/cdfio.F90:1127-1145)
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4) :: ncid, ierr
    getnvar = 0
    IF ( src_attach(cdfile, ncid) /= NF90_NOERR ) RETURN
    ierr = NF90_INQUIRE(ncid, nvariables=getnvar)
    CALL src_detach(ncid)
  END FUNCTION getnvar

  FUNCTION getipk(cdfile, knvars, cdep)
    !! levels of every variable by its rank: 4 -> the length of the depth dimension, 3 -> 1, anything else 0
    !! (src/cdfio.F90:1148-1194)
    CHARACTER(LEN=*),           INTENT(in) :: cdfile
    INTEGER(KIND=4),            INTENT(in) :: knvars
    CHARACTER(LEN=*), OPTIONAL, INTENT(in) :: cdep
    INTEGER(KIND=4), DIMENSION(knvars) :: getipk
    INTEGER(KIND=4) :: ncid, irank, jv, ilevels, ierr
    IF ( PRESENT(cdep) ) THEN ; ilevels = getdim(cdfile, TRIM(cdep), kstatus=ierr)
    ELSE                      ; ilevels = getdim(cdfile, 'dep', kstatus=ierr)
    ENDIF
    IF ( ierr /= 0 ) THEN
       PRINT *, ' getipk : vertical dim not found ...assume 1'
       ilevels = 1
    ENDIF
    getipk = 0
    IF ( src_attach(cdfile, ncid) /= NF90_NOERR ) RETURN
    DO jv = 1, knvars
       IF ( NF90_INQUIRE_VARIABLE(ncid, jv, ndims=irank) /= NF90_NOERR ) CYCLE
       IF ( irank == 4 ) getipk(jv) = ilevels
       IF ( irank == 3 ) getipk(jv) = 1
    END DO
    CALL src_detach(ncid)
  END FUNCTION getipk

  FUNCTION getvarname(cdfile, knvars, sdtypvar)
    !! names of the variables of a file, in file order, and their standard attributes into sdtypvar
    !! (src/cdfio.F90:1197-1290): text attributes 'N/A' when absent, valid range 0., weight 1
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4),  INTENT(in) :: knvars
    TYPE(variable), DIMENSION(knvars) :: sdtypvar
    CHARACTER(LEN=256), DIMENSION(knvars) :: getvarname
    INTEGER(KIND=4) :: ncid, jv, ierr
    getvarname = ' '
    IF ( src_attach(cdfile, ncid) /= NF90_NOERR ) RETURN
    DO jv = 1, knvars
       ierr = NF90_INQUIRE_VARIABLE(ncid, jv, name=getvarname(jv))
       ASSOCIATE ( sd => sdtypvar(jv) )
         sd%cname = getvarname(jv)
         CALL lookup_spval(ncid, jv, sd%rmissing_value)
         sd%valid_min = 0. ; sd%valid_max = 0. ; sd%iwght = 1
         IF ( NF90_INQUIRE_ATTRIBUTE(ncid, jv, 'valid_min') == NF90_NOERR ) ierr = NF90_GET_ATT(ncid, jv, 'valid_min', sd%valid_min)
         IF ( NF90_INQUIRE_ATTRIBUTE(ncid, jv, 'valid_max') == NF90_NOERR ) ierr = NF90_GET_ATT(ncid, jv, 'valid_max', sd%valid_max)
         IF ( NF90_INQUIRE_ATTRIBUTE(ncid, jv, 'iweight')   == NF90_NOERR ) ierr = NF90_GET_ATT(ncid, jv, 'iweight', sd%iwght)
         sd%cunits            = text_or_na('units')
         sd%clong_name        = text_or_na('long_name')
         sd%cshort_name       = text_or_na('short_name')
         sd%conline_operation = text_or_na('online_operation')
         sd%caxis             = text_or_na('axis')
       END ASSOCIATE
    END DO
    CALL src_detach(ncid)
  CONTAINS
    CHARACTER(LEN=256) FUNCTION text_or_na(cdatt)
      CHARACTER(LEN=*), INTENT(in) :: cdatt
      text_or_na = 'N/A'
      IF ( NF90_INQUIRE_ATTRIBUTE(ncid, jv, cdatt) /= NF90_NOERR ) RETURN
      text_or_na = ' '
      ierr = NF90_GET_ATT(ncid, jv, cdatt, text_or_na)
    END FUNCTION text_or_na
  END FUNCTION getvarname

  INTEGER(KIND=4) FUNCTION getvaratt(cdfile, cdvar, cdunits, pmissing_value, cdlong_name, cdshort_name)
    !! units, missing value, long and short name of a variable; an absent attribute leaves its argument untouched and
    !! the status returned is that of the last lookup only (src/cdfio.F90:488-514)
    CHARACTER(LEN=*), INTENT(in)    :: cdfile, cdvar
    REAL(KIND=4),     INTENT(out)   :: pmissing_value
    CHARACTER(LEN=*), INTENT(inout) :: cdunits, cdlong_name, cdshort_name
    INTEGER(KIND=4) :: ncid, ivar
    getvaratt = src_attach(cdfile, ncid)
    getvaratt = NF90_INQ_VARID(ncid, cdvar, ivar)
    pmissing_value = 0.
    IF ( getvaratt == NF90_NOERR ) CALL lookup_spval(ncid, ivar, pmissing_value)
    getvaratt = text_if_there('units', cdunits)
    getvaratt = text_if_there('long_name', cdlong_name)
    getvaratt = text_if_there('short_name', cdshort_name)
    CALL src_detach(ncid)
  CONTAINS
    INTEGER(KIND=4) FUNCTION text_if_there(cdatt, cdval)
      CHARACTER(LEN=*), INTENT(in)    :: cdatt
      CHARACTER(LEN=*), INTENT(inout) :: cdval
      text_if_there = NF90_INQUIRE_ATTRIBUTE(ncid, ivar, cdatt)
      IF ( text_if_there == NF90_NOERR ) text_if_there = NF90_GET_ATT(ncid, ivar, cdatt, cdval)
    END FUNCTION text_if_there
  END FUNCTION getvaratt

  ! ================================================================================================ readers
  FUNCTION getvar(cdfile, cdvar, klev, kpi, kpj, kimin, kjmin, ktime, ldiom)
    !! One (kpi,kpj) r4 slab of cdvar at level klev, record ktime, window origin (kimin,kjmin), unpacked.  A window
    !! that runs over the eastern edge continues at column 2 (the overlap column is skipped); ldiom: the name is that of a
    !! vertical metric, to be looked up under the mesh_zgr spelling of the detected layout (src/cdfio.F90:1425-1609).
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: klev
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kjmin, ktime
    LOGICAL,         OPTIONAL, INTENT(in) :: ldiom
    REAL(KIND=4), DIMENSION(kpi,kpj) :: getvar
    CALL getvar_to(cdfile, cdvar, klev, getvar, kimin, kjmin, ktime, ldiom)
  END FUNCTION getvar

  SUBROUTINE getvar_to(cdfile, cdvar, klev, ptab, kimin, kjmin, ktime, ldiom)
    !! getvar's slab read straight into the caller's array ptab (its shape is the window): what getvar returns, without the function
    !! result -- 62 MB at eORCA12, whose allocation, first touch and copy cost twice the read itself (not in the reference's cdfio).
    CHARACTER(LEN=*),          INTENT(in)  :: cdfile, cdvar
    INTEGER(KIND=4), OPTIONAL, INTENT(in)  :: klev
    REAL(KIND=4), CONTIGUOUS,  INTENT(out) :: ptab(:,:)
    INTEGER(KIND=4), OPTIONAL, INTENT(in)  :: kimin, kjmin, ktime
    LOGICAL,         OPTIONAL, INTENT(in)  :: ldiom
    INTEGER(KIND=4) :: ncid, ivar, ierr, iwest, ifirst(4), ishape(4), kpi, kpj
    CHARACTER(LEN=256) :: cl_name
    TYPE(t_pack) :: tp
    kpi = SIZE(ptab, 1) ; kpj = SIZE(ptab, 2)
    cl_name = zgr_name(cdvar, ldiom)
    iwest = kpi                                         ! columns taken before the eastern edge (kpi: the window does not wrap)
    IF ( PRESENT(kimin) ) iwest = MIN(kpi, getdim(cdfile, cn_x, ldexact=.TRUE.) - kimin)
    IF ( PRESENT(kimin) ) THEN                           ! a window that ENDS on the last column does not wrap either (:1476)
       IF ( iwest == kpi - 1 ) iwest = kpi
    ENDIF
    CALL tic()
    CALL src_variable(cdfile, cl_name, ncid, ivar)
    ifirst = (/ oint(kimin, 1), oint(kjmin, 1), 1, 1 /) ; ishape = (/ kpi, kpj, 1, 1 /)
    CALL place_slab(ncid, ivar, oint(klev, 1), oint(ktime, 1), ifirst)
    tp = pack_of(ncid, ivar, cdvar)
    IF ( iwest < kpi ) THEN
       ierr = NF90_NOERR
       ishape(1) = iwest
       IF ( iwest > 0 ) ierr = NF90_GET_VAR(ncid, ivar, ptab(1:iwest,:), start=ifirst, count=ishape)
       ifirst(1) = 2 ; ishape(1) = kpi - iwest
       IF ( ierr == NF90_NOERR ) ierr = NF90_GET_VAR(ncid, ivar, ptab(iwest+1:kpi,:), start=ifirst, count=ishape)
    ELSE
       ierr = NF90_GET_VAR(ncid, ivar, ptab, start=ifirst, count=ishape)
    ENDIF
    CALL give_up(ierr, 'getvar', cl_name)
    CALL pack_apply(tp, ptab, INT(kpi, 8) * kpj)
    CALL src_detach(ncid)
    CALL toc(dtime_read)
  END SUBROUTINE getvar_to

  FUNCTION getvar3d(cdfile, cdvar, kpi, kpj, kpk, kimin, kjmin, kkmin, ktime, ldiom)
    !! A (kpi,kpj,kpk) block of one record, window origin (kimin,kjmin,kkmin), as a function result
    !! (src/cdfio.F90:1611-1715; ldiom as in getvar)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj, kpk
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kjmin, kkmin, ktime
    LOGICAL,         OPTIONAL, INTENT(in) :: ldiom
    REAL(KIND=4), DIMENSION(kpi,kpj,kpk) :: getvar3d
    CALL readvar3d(cdfile, cdvar, getvar3d, ktime=ktime, ldiom=ldiom, kimin=kimin, kjmin=kjmin, kkmin=kkmin)
  END FUNCTION getvar3d

  SUBROUTINE readvar3d(cdfile, cdvar, ptab, ktime, ldiom, kimin, kjmin, kkmin)
    !! The (kpi,kpj,kpk) = SHAPE(ptab) block of one record, window origin (kimin,kjmin,kkmin), read straight into ptab
    !! in ONE call and unpacked like getvar's slabs: no function result to copy, none of the per-level lookups of a
    !! getvar loop.  It answers what that loop would answer for every layout of the variable:
    !!   (x,y,z,t) and a time-less (x,y,z): the levels kkmin .. kkmin+kpk-1 in one read;
    !!   (x,y,t) and (x,y): the one slab there is, repeated on every level of ptab (each getvar(.., jk, ..) returns it);
    !!   a window over the eastern edge: level by level through getvar, which knows how to wrap.
    !! Not in the reference's cdfio: the programs of this build feed the GPU with it.
    CHARACTER(LEN=*),          INTENT(in)  :: cdfile, cdvar
    REAL(KIND=4), CONTIGUOUS,  INTENT(out) :: ptab(:,:,:)
    INTEGER(KIND=4), OPTIONAL, INTENT(in)  :: ktime, kimin, kjmin, kkmin
    LOGICAL,         OPTIONAL, INTENT(in)  :: ldiom
    INTEGER(KIND=4) :: ncid, ivar, ierr, jk, ini, inj, ink, ifirst(4), ishape(4)
    LOGICAL :: ll_levels
    CHARACTER(LEN=256) :: cl_name
    TYPE(t_pack) :: tp
    ini = SIZE(ptab, 1) ; inj = SIZE(ptab, 2) ; ink = SIZE(ptab, 3)
    IF ( PRESENT(kimin) ) THEN
       IF ( kimin + ini - 1 > getdim(cdfile, cn_x, ldexact=.TRUE.) ) THEN
          DO jk = 1, ink
             ptab(:,:,jk) = getvar(cdfile, cdvar, jk + oint(kkmin, 1) - 1, ini, inj, kimin=kimin, kjmin=kjmin, ktime=ktime, ldiom=ldiom)
          END DO
          RETURN
       ENDIF
    ENDIF
    cl_name = zgr_name(cdvar, ldiom)
    CALL tic()
    CALL src_variable(cdfile, cl_name, ncid, ivar)
    tp = pack_of(ncid, ivar, cdvar)
    ifirst = (/ oint(kimin, 1), oint(kjmin, 1), 1, 1 /) ; ishape = (/ ini, inj, 1, 1 /)
    CALL place_slab(ncid, ivar, oint(kkmin, 1), oint(ktime, 1), ifirst, ll_levels)
    IF ( ll_levels ) THEN
       ishape(3) = ink
       ierr = NF90_GET_VAR(ncid, ivar, ptab, start=ifirst, count=ishape)
    ELSE
       ierr = NF90_GET_VAR(ncid, ivar, ptab(:,:,1), start=ifirst, count=ishape)
    ENDIF
    CALL give_up(ierr, 'readvar3d', cl_name)
    IF ( ll_levels ) THEN
       CALL pack_apply(tp, ptab, INT(ini, 8) * inj * ink)
    ELSE
       CALL pack_apply(tp, ptab(:,:,1), INT(ini, 8) * inj)
       DO jk = 2, ink
          ptab(:,:,jk) = ptab(:,:,1)
       END DO
    ENDIF
    CALL src_detach(ncid)
    CALL toc(dtime_read)
  END SUBROUTINE readvar3d

  LOGICAL FUNCTION var_is_plain_bytes(cdfile, cdvar)
    !! whether cdvar of cdfile is stored as bytes with levels and without scale_factor / add_offset / savelog10: its values are what
    !! readvar3d_i1 returns, widened -- so that a caller may move the bytes and widen them where they are used (module cdffeed)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4) :: ncid, ivar, itype, indims, ifirst(4)
    LOGICAL :: ll_levels
    TYPE(t_pack) :: tp
    var_is_plain_bytes = .FALSE.
    IF ( src_attach(cdfile, ncid) /= NF90_NOERR ) RETURN
    IF ( NF90_INQ_VARID(ncid, cdvar, ivar) == NF90_NOERR ) THEN
       IF ( NF90_INQUIRE_VARIABLE(ncid, ivar, xtype=itype, ndims=indims) == NF90_NOERR ) THEN
          IF ( itype == NF90_BYTE .AND. indims >= 3 ) THEN
             tp = pack_of(ncid, ivar, cdvar)
             ifirst = 1
             CALL place_slab(ncid, ivar, 1, 1, ifirst, ll_levels)
             var_is_plain_bytes = ll_levels .AND. .NOT. ( tp%lscale .OR. tp%loffset .OR. tp%llog10 )
          ENDIF
       ENDIF
    ENDIF
    CALL src_detach(ncid)
  END FUNCTION var_is_plain_bytes

  SUBROUTINE readvar3d_i1(cdfile, cdvar, ktab, ktime, kkmin)
    !! a whole 3-D mask (or its levels from kkmin on) as bytes (the backend converts when the file holds another type); a
    !! time-less (x,y,z) mask is read the same way, a mask without levels fills every level of ktab
    CHARACTER(LEN=*),            INTENT(in)  :: cdfile, cdvar
    INTEGER(KIND=1), CONTIGUOUS, INTENT(out) :: ktab(:,:,:)
    INTEGER(KIND=4), OPTIONAL,   INTENT(in)  :: ktime, kkmin
    INTEGER(KIND=4) :: ncid, ivar, ierr, jk, ifirst(4), ishape(4)
    LOGICAL :: ll_levels
    CALL tic()
    CALL src_variable(cdfile, cdvar, ncid, ivar)
    ifirst = 1 ; ishape = (/ SIZE(ktab, 1), SIZE(ktab, 2), 1, 1 /)
    CALL place_slab(ncid, ivar, oint(kkmin, 1), oint(ktime, 1), ifirst, ll_levels)
    IF ( ll_levels ) THEN
       ishape(3) = SIZE(ktab, 3)
       ierr = NF90_GET_VAR(ncid, ivar, ktab, start=ifirst, count=ishape)
    ELSE
       ierr = NF90_GET_VAR(ncid, ivar, ktab(:,:,1), start=ifirst, count=ishape)
       DO jk = 2, SIZE(ktab, 3)
          ktab(:,:,jk) = ktab(:,:,1)
       END DO
    ENDIF
    CALL give_up(ierr, 'readvar3d', cdvar)
    CALL src_detach(ncid)
    CALL toc(dtime_read)
  END SUBROUTINE readvar3d_i1

  FUNCTION getvar1d(cdfile, cdvar, kk, kstatus)
    !! a 1-D variable as r8 (time axes, depths); 99999999999.d0 and kstatus /= 0 when it is not there
    !! (src/cdfio.F90:2151-2186)
    CHARACTER(LEN=*),           INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),            INTENT(in) :: kk
    INTEGER(KIND=4), OPTIONAL, INTENT(out) :: kstatus
    REAL(KIND=8), DIMENSION(kk) :: getvar1d
    INTEGER(KIND=4) :: ncid, ivar, ierr
    IF ( PRESENT(kstatus) ) kstatus = 0
    ierr = src_attach(cdfile, ncid)
    ierr = NF90_INQ_VARID(ncid, cdvar, ivar)
    IF ( ierr /= NF90_NOERR ) THEN
       PRINT *, ' getvar1d ERROR: ', TRIM(NF90_STRERROR(ierr))
       PRINT *, '                when reading ', TRIM(cdvar)
       IF ( PRESENT(kstatus) ) kstatus = ierr
       getvar1d = 99999999999.d0
    ELSE
       ierr = NF90_GET_VAR(ncid, ivar, getvar1d, start=(/ 1 /), count=(/ kk /))
    ENDIF
    CALL src_detach(ncid)
  END FUNCTION getvar1d

  FUNCTION getvare3(cdfile, cdvar, kk)
    !! a vertical profile (gdept, gdepw, e3t, e3w) from mesh_zgr whatever its layout: v2.0 stores it (t,z,1,1) under the
    !! plain name, v3.0 as '<name>_0'(t,z), v3.6 as '<name>_1d'(t,z); other names are read as given, (t,z,1,1)
    !! (src/cdfio.F90:2189-2287)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),  INTENT(in) :: kk
    REAL(KIND=4), DIMENSION(kk) :: getvare3
    INTEGER(KIND=4) :: ncid, ivar, ierr, ishape(4)
    CHARACTER(LEN=256) :: cl_name
    CHARACTER(LEN=5)   :: cl_root
    cl_root = ' '
    IF ( cdvar == cn_gdept  ) cl_root = 'gdept'
    IF ( cdvar == cn_gdepw  ) cl_root = 'gdepw'
    IF ( cdvar == cn_ve3t1d ) cl_root = 'e3t'
    IF ( cdvar == cn_ve3w1d ) cl_root = 'e3w'
    cl_name = cdvar ; ishape = (/ 1, 1, kk, 1 /)
    IF ( cl_root /= ' ' ) THEN
       IF ( cg_zgr_ver == 'v2.0' ) cl_name = cl_root
       IF ( cg_zgr_ver == 'v3.0' ) cl_name = TRIM(cl_root)//'_0'
       IF ( cg_zgr_ver == 'v3.6' ) cl_name = TRIM(cl_root)//'_1d'
       IF ( cg_zgr_ver /= 'v2.0' ) ishape = (/ kk, 1, 1, 1 /)
    ENDIF
    ierr = src_attach(cdfile, ncid)
    IF ( ierr == NF90_NOERR ) ierr = NF90_INQ_VARID(ncid, cl_name, ivar)
    IF ( ierr == NF90_NOERR ) ierr = NF90_GET_VAR(ncid, ivar, getvare3, start=(/ 1, 1, 1, 1 /), count=ishape)
    IF ( ierr /= NF90_NOERR ) THEN
       PRINT *, ' Problem in getvare3 for ', TRIM(cdvar)
       PRINT *, TRIM(cdfile), kk
       CALL ERR_HDL(ierr)
    ENDIF
    CALL src_detach(ncid)
  END FUNCTION getvare3

  ! ================================================================================================ output file
  SUBROUTINE coordinate_attributes(cdvar, kvar, kref, kout)
    ! attributes of a coordinate variable of the output file: those the reference file gives that variable when there
    ! is a reference file (kref >= 0), a standard set otherwise (src/cdfio.F90:185-258)
    CHARACTER(LEN=*), INTENT(in) :: cdvar
    INTEGER(KIND=4),  INTENT(in) :: kvar, kref, kout
    INTEGER(KIND=4) :: ja, iref, inatt, ierr
    CHARACTER(LEN=256) :: cl_att
    IF ( kref >= 0 ) THEN
       IF ( NF90_INQ_VARID(kref, cdvar, iref) /= NF90_NOERR ) RETURN
       ierr = NF90_INQUIRE_VARIABLE(kref, iref, nAtts=inatt)
       DO ja = 1, inatt
          ierr = NF90_INQ_ATTNAME(kref, iref, ja, cl_att)
          ierr = NF90_COPY_ATT(kref, iref, TRIM(cl_att), kout, kvar)
       END DO
       RETURN
    ENDIF
    SELECT CASE ( TRIM(cdvar) )
    CASE ( 'nav_lon', 'lon', 'x', 'longitude' ) ; CALL horizontal('degrees_east', 180., 'Longitude')
    CASE ( 'nav_lat', 'lat', 'y', 'latitude' )  ; CALL horizontal('degrees_north', 90., 'Latitude')
    CASE ( 'time_counter', 'time', 't' )
       ierr = NF90_PUT_ATT(kout, kvar, 'calendar',    TRIM(calendar))
       ierr = NF90_PUT_ATT(kout, kvar, 'units',       TRIM(ctime_units))
       ierr = NF90_PUT_ATT(kout, kvar, 'time_origin', TRIM(ctime_origin))
       ierr = NF90_PUT_ATT(kout, kvar, 'title',       'Time')
       ierr = NF90_PUT_ATT(kout, kvar, 'long_name',   'Time axis')
    CASE ( 'deptht', 'depthu', 'depthv', 'depthw', 'dep', 'gdept' ) ; CALL vertical('m', 5875., 'Vertical Levels')
    CASE ( 'sigma', 'levels' )                                      ; CALL vertical('kg/m3', 40., 'Sigma bin limits')
    END SELECT
  CONTAINS
    SUBROUTINE horizontal(cdunits, pmax, cdlong)
      CHARACTER(LEN=*), INTENT(in) :: cdunits, cdlong
      REAL(KIND=4),     INTENT(in) :: pmax
      ierr = NF90_PUT_ATT(kout, kvar, 'units', cdunits)
      ierr = NF90_PUT_ATT(kout, kvar, 'valid_min', -pmax)
      ierr = NF90_PUT_ATT(kout, kvar, 'valid_max', pmax)
      ierr = NF90_PUT_ATT(kout, kvar, 'long_name', cdlong)
      ierr = NF90_PUT_ATT(kout, kvar, 'nav_model', 'Default grid')
    END SUBROUTINE horizontal
    SUBROUTINE vertical(cdunits, pmax, cdlong)
      CHARACTER(LEN=*), INTENT(in) :: cdunits, cdlong
      REAL(KIND=4),     INTENT(in) :: pmax
      ierr = NF90_PUT_ATT(kout, kvar, 'units', cdunits)
      ierr = NF90_PUT_ATT(kout, kvar, 'positive', 'unknown')
      ierr = NF90_PUT_ATT(kout, kvar, 'valid_min', 0.)
      ierr = NF90_PUT_ATT(kout, kvar, 'valid_max', pmax)
      ierr = NF90_PUT_ATT(kout, kvar, 'title', TRIM(cdvar))
      ierr = NF90_PUT_ATT(kout, kvar, 'long_name', cdlong)
    END SUBROUTINE vertical
  END SUBROUTINE coordinate_attributes

  INTEGER(KIND=4) FUNCTION create(cdfile, cdfilref, kx, ky, kz, cdep, cdepvar, cdlonvar, cdlatvar, ld_xycoo, ld_nc4)
    !! A new file with the dimensions x, y, [depth], time_counter (UNLIMITED) and the coordinate variables nav_lon,
    !! nav_lat, [depth], time_counter; returns its id (src/cdfio.F90:260-368).  Format: 64-bit offset, CDF-5 when one
    !! record of a variable would not fit that format, NetCDF-4 with ld_nc4 (src/cdfio.F90:291-299: through the HDF5
    !! plug-in of the bundled backend; without an HDF5 library the file is classic, like a reference built without
    !! key_netcdf4).  The depth dimension takes its name from the reference file: the dimension matching cdep, or cn_z.
    CHARACTER(LEN=*),           INTENT(in) :: cdfile, cdfilref
    INTEGER(KIND=4),            INTENT(in) :: kx, ky, kz
    CHARACTER(LEN=*), OPTIONAL, INTENT(in) :: cdep, cdepvar, cdlonvar, cdlatvar
    LOGICAL,          OPTIONAL, INTENT(in) :: ld_xycoo, ld_nc4
    INTEGER(KIND=4) :: ierr, iout, iref, ilen, iformat
    CHARACTER(LEN=256) :: cl_zdim, cl_zvar, cl_inref
    CALL src_forget(cdfile)
    iformat = NF90_64BIT_OFFSET
    IF ( REAL(kx, 8) * REAL(ky, 8) * REAL(MAX(kz, 1), 8) * 8.d0 > 4.0d9 ) iformat = NF90_64BIT_DATA
    IF ( olog(ld_nc4, .FALSE.) ) iformat = NF90_NETCDF4
    CALL ERR_HDL(NF90_CREATE(cdfile, IOR(NF90_CLOBBER, iformat), iout))
    ierr = NF90_DEF_DIM(iout, TRIM(cn_x), kx, nid_x)
    ierr = NF90_DEF_DIM(iout, TRIM(cn_y), ky, nid_y)
    IF ( kz /= 0 ) THEN
       IF ( PRESENT(cdep) ) THEN
          cl_zdim = cdep
          ilen = getdim(cdfilref, cl_zdim, cl_inref)
       ELSE
          ilen = getdim(cdfilref, cn_z, cl_zdim)
       ENDIF
       cl_zvar = cl_zdim
       IF ( PRESENT(cdepvar) ) cl_zvar = cdepvar
       ierr = NF90_DEF_DIM(iout, TRIM(cl_zdim), kz, nid_z)
    ENDIF
    ierr = NF90_DEF_DIM(iout, TRIM(cn_t), NF90_UNLIMITED, nid_t)
    iref = -1
    IF ( TRIM(cdfilref) /= 'none' ) THEN
       IF ( src_attach(cdfilref, iref) /= NF90_NOERR ) iref = -1
    ENDIF
    IF ( olog(ld_xycoo, .TRUE.) ) THEN
       ierr = NF90_DEF_VAR(iout, TRIM(cn_vlon2d), NF90_FLOAT, (/ nid_x, nid_y /), nid_lon)
       CALL coordinate_attributes(cn_vlon2d, nid_lon, iref, iout)
       ierr = NF90_DEF_VAR(iout, TRIM(cn_vlat2d), NF90_FLOAT, (/ nid_x, nid_y /), nid_lat)
       CALL coordinate_attributes(cn_vlat2d, nid_lat, iref, iout)
    ENDIF
    IF ( PRESENT(cdlonvar) ) ierr = NF90_DEF_VAR(iout, cdlonvar, NF90_FLOAT, (/ nid_x /), nid_lon1d)
    IF ( PRESENT(cdlatvar) ) ierr = NF90_DEF_VAR(iout, cdlatvar, NF90_FLOAT, (/ nid_y /), nid_lat1d)
    IF ( kz /= 0 ) THEN
       ierr = NF90_DEF_VAR(iout, TRIM(cl_zvar), NF90_FLOAT, (/ nid_z /), nid_dep)
       CALL coordinate_attributes(TRIM(cl_zvar), nid_dep, iref, iout)
    ENDIF
    ierr = NF90_DEF_VAR(iout, TRIM(cn_vtimec), NF90_DOUBLE, (/ nid_t /), nid_tim)
    CALL coordinate_attributes(cn_vtimec, nid_tim, iref, iout)
    ierr = NF90_PUT_ATT(iout, NF90_GLOBAL, 'start_date', nstart_date)
    ierr = NF90_PUT_ATT(iout, NF90_GLOBAL, 'output_frequency', TRIM(cfreq))
    ierr = NF90_PUT_ATT(iout, NF90_GLOBAL, 'CONFIG', TRIM(config))
    ierr = NF90_PUT_ATT(iout, NF90_GLOBAL, 'CASE', TRIM(ccase))
    IF ( iref >= 0 ) CALL src_detach(iref)
    create = iout
  END FUNCTION create

  INTEGER(KIND=4) FUNCTION createvar(kout, sdtyvar, kvar, kpk, kidvo, cdglobal, ld_nc4)
    !! Defines the kvar output variables (those named 'none' are skipped) and leaves define mode.  kpk = 1: (x,y,t);
    !! kpk > 1: (x,y,z,t); kpk < 0: (t); 0 is an error.  An r4 variable with a scale factor or an offset becomes a short
    !! (src/cdfio.F90:371-459); with ld_nc4 the variables are chunked by ichunk and deflated (:438-441).
    INTEGER(KIND=4),                  INTENT(in   ) :: kout, kvar
    TYPE(variable), DIMENSION(kvar),  INTENT(inout) :: sdtyvar
    INTEGER(KIND=4), DIMENSION(kvar), INTENT(in   ) :: kpk
    INTEGER(KIND=4), DIMENSION(kvar), INTENT(out  ) :: kidvo
    CHARACTER(LEN=*), OPTIONAL,       INTENT(in   ) :: cdglobal
    LOGICAL,          OPTIONAL,       INTENT(in   ) :: ld_nc4
    INTEGER(KIND=4) :: jv, irank, ierr, idim(4), ixtype
    createvar = NF90_NOERR
    DO jv = 1, kvar
       IF ( sdtyvar(jv)%cname == 'none' ) CYCLE
       IF ( kpk(jv) == 0 ) THEN
          PRINT *, ' ERROR: ipk = ', kpk(jv), jv, sdtyvar(jv)%cname
          STOP 98
       ENDIF
       idim = (/ nid_x, nid_y, nid_z, nid_t /) ; irank = 4
       IF ( kpk(jv) == 1 ) THEN ; idim(3) = nid_t ; irank = 3 ; ENDIF
       IF ( kpk(jv) <  0 ) THEN ; idim(1) = nid_t ; irank = 1 ; ENDIF
       SELECT CASE ( sdtyvar(jv)%cprecision )
       CASE ( 'r8' ) ; ixtype = NF90_DOUBLE
       CASE ( 'i2' ) ; ixtype = NF90_SHORT
       CASE ( 'by' ) ; ixtype = NF90_BYTE
       CASE DEFAULT
          ixtype = NF90_FLOAT
          IF ( sdtyvar(jv)%scale_factor /= 1. .OR. sdtyvar(jv)%add_offset /= 0. ) THEN
             ixtype = NF90_SHORT ; sdtyvar(jv)%cprecision = 'i2'
          ENDIF
       END SELECT
       IF ( olog(ld_nc4, .FALSE.) ) THEN
          CALL ERR_HDL(NF90_DEF_VAR(kout, TRIM(sdtyvar(jv)%cname), ixtype, idim(1:irank), kidvo(jv), &
               &                    chunksizes=sdtyvar(jv)%ichunk(1:irank), deflate_level=1))
       ELSE
          CALL ERR_HDL(NF90_DEF_VAR(kout, TRIM(sdtyvar(jv)%cname), ixtype, idim(1:irank), kidvo(jv)))
       ENDIF
       createvar = putatt(sdtyvar(jv), kout, kidvo(jv), cdglobal=cdglobal)
    END DO
    ierr = NF90_ENDDEF(kout)
    IF ( ierr /= NF90_NOERR ) THEN
       PRINT *, ' Error in CreateVar :', TRIM(NF90_STRERROR(ierr))
       STOP 98
    ENDIF
  END FUNCTION createvar

  INTEGER(KIND=4) FUNCTION putatt(sdtyvar, kout, kid, cdglobal)
    !! the twelve standard attributes of an output variable, in the reference's order, the missing value in the type of
    !! the variable (src/cdfio.F90:629-682); with cdglobal also the global attribute 'history'.  Any failure is fatal.
    TYPE(variable),             INTENT(in) :: sdtyvar
    INTEGER(KIND=4),            INTENT(in) :: kout, kid
    CHARACTER(LEN=*), OPTIONAL, INTENT(in) :: cdglobal
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'units', TRIM(sdtyvar%cunits)), 'units')
    IF      ( sdtyvar%cprecision == 'r8' ) THEN ; putatt = NF90_PUT_ATT(kout, kid, TRIM(cn_missing_value), REAL(sdtyvar%rmissing_value, 8))
    ELSE IF ( sdtyvar%cprecision == 'i2' ) THEN ; putatt = NF90_PUT_ATT(kout, kid, TRIM(cn_missing_value), INT(sdtyvar%rmissing_value, 2))
    ELSE IF ( sdtyvar%cprecision == 'by' ) THEN ; putatt = NF90_PUT_ATT(kout, kid, TRIM(cn_missing_value), INT(sdtyvar%rmissing_value, 1))
    ELSE                                        ; putatt = NF90_PUT_ATT(kout, kid, TRIM(cn_missing_value), REAL(sdtyvar%rmissing_value, 4))
    ENDIF
    CALL att_or_stop(putatt, 'missing value')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'valid_min', sdtyvar%valid_min), 'valid_min')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'valid_max', sdtyvar%valid_max), 'valid_max')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'long_name', TRIM(sdtyvar%clong_name)), 'long_name')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'short_name', TRIM(sdtyvar%cshort_name)), 'short_name')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'iweight', sdtyvar%iwght), 'iweight')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'online_operation', TRIM(sdtyvar%conline_operation)), 'online_operation')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'axis', TRIM(sdtyvar%caxis)), 'axis')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'scale_factor', sdtyvar%scale_factor), 'scale_factor')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'add_offset', sdtyvar%add_offset), 'add_offset')
    CALL att_or_stop(NF90_PUT_ATT(kout, kid, 'savelog10', sdtyvar%savelog10), 'savelog10')
    IF ( PRESENT(cdglobal) ) CALL att_or_stop(NF90_PUT_ATT(kout, NF90_GLOBAL, 'history', TRIM(cdglobal)), 'history')
    putatt = NF90_NOERR
  CONTAINS
    SUBROUTINE att_or_stop(kerr, cdwhat)
      INTEGER(KIND=4),  INTENT(in) :: kerr
      CHARACTER(LEN=*), INTENT(in) :: cdwhat
      IF ( kerr == NF90_NOERR ) RETURN
      PRINT *, NF90_STRERROR(kerr) ; PRINT *, ' putatt for ', cdwhat ; STOP 98
    END SUBROUTINE att_or_stop
  END FUNCTION putatt

  INTEGER(KIND=4) FUNCTION putheadervar(kout, cdfile, kpi, kpj, kpk, pnavlon, pnavlat, pdep, cdep, ld_xycoo)
    !! nav_lon, nav_lat and the depth of the output file: the arrays given, else what cdfile holds -- the 2-D
    !! coordinate, else the 1-D one spread over the plane, else zeros with a warning; the depth under cdep or any of the
    !! usual names, and no depth at all is fatal (src/cdfio.F90:2290-2407)
    INTEGER(KIND=4),                            INTENT(in) :: kout, kpi, kpj, kpk
    CHARACTER(LEN=*),                           INTENT(in) :: cdfile
    REAL(KIND=4), OPTIONAL, DIMENSION(kpi,kpj), INTENT(in) :: pnavlon, pnavlat
    REAL(KIND=4), OPTIONAL, DIMENSION(kpk),     INTENT(in) :: pdep
    CHARACTER(LEN=*), OPTIONAL,                 INTENT(in) :: cdep
    LOGICAL,          OPTIONAL,                 INTENT(in) :: ld_xycoo
    REAL(KIND=4), ALLOCATABLE :: zplane(:,:), zdepth(:)
    CHARACTER(LEN=256) :: cl_try(7)
    INTEGER(KIND=4) :: jn, ierr
    putheadervar = NF90_NOERR
    IF ( olog(ld_xycoo, .TRUE.) ) THEN
       ALLOCATE( zplane(kpi,kpj) )
       IF ( PRESENT(pnavlon) ) THEN ; zplane = pnavlon
       ELSE                         ; CALL from_file(cn_vlon2d, cn_vlon1d, 1)
       ENDIF
       putheadervar = putvar(kout, nid_lon, zplane, 1, kpi, kpj)
       IF ( PRESENT(pnavlat) ) THEN ; zplane = pnavlat
       ELSE                         ; CALL from_file(cn_vlat2d, cn_vlat1d, 2)
       ENDIF
       putheadervar = putvar(kout, nid_lat, zplane, 1, kpi, kpj)
    ENDIF
    IF ( kpk == 0 ) RETURN
    ALLOCATE( zdepth(kpk) )
    IF ( PRESENT(pdep) ) THEN
       zdepth = pdep
    ELSE
       cl_try(:) = ' '
       IF ( PRESENT(cdep) ) cl_try(1) = cdep
       cl_try(2:7) = (/ cn_vdeptht, cn_vdepthu, cn_vdepthv, cn_vdepthw, 'nav_lev'//REPEAT(' ', 249), 'z'//REPEAT(' ', 255) /)
       ierr = -1
       DO jn = 1, SIZE(cl_try)
          IF ( cl_try(jn) == ' ' ) CYCLE
          IF ( jn > 1 ) THEN
             IF ( chkvar(cdfile, cl_try(jn), .FALSE.) ) CYCLE
          ENDIF
          zdepth = REAL(getvar1d(cdfile, cl_try(jn), kpk, ierr), 4)
          IF ( ierr == NF90_NOERR ) EXIT
       END DO
       IF ( ierr /= NF90_NOERR ) THEN
          PRINT *, ' No depth variable found in ', TRIM(cdfile)
          STOP 98
       ENDIF
    ENDIF
    putheadervar = putvar1d(kout, zdepth, kpk, 'D')
  CONTAINS
    SUBROUTINE from_file(cd2d, cd1d, kaxis)
      ! kaxis = 1: a 1-D coordinate runs along i, 2: along j
      CHARACTER(LEN=*), INTENT(in) :: cd2d, cd1d
      INTEGER(KIND=4),  INTENT(in) :: kaxis
      REAL(KIND=8), ALLOCATABLE :: dl_line(:)
      INTEGER(KIND=4) :: ist
      IF ( .NOT. chkvar(cdfile, cd2d, .FALSE.) ) THEN
         zplane = getvar(cdfile, cd2d, 1, kpi, kpj)
      ELSE IF ( .NOT. chkvar(cdfile, cd1d, .FALSE.) ) THEN
         ALLOCATE( dl_line(MERGE(kpi, kpj, kaxis == 1)) )
         dl_line = getvar1d(cdfile, cd1d, SIZE(dl_line), ist)
         zplane = SPREAD(REAL(dl_line, 4), DIM=3-kaxis, NCOPIES=MERGE(kpj, kpi, kaxis == 1))
      ELSE
         PRINT *, '... dummy value used!'
         zplane = 0.
      ENDIF
    END SUBROUTINE from_file
  END FUNCTION putheadervar

  FUNCTION slab_origin(kout, kid, klev, ktime) RESULT(ifirst)
    ! where a (level, record) slab of an output variable starts: its last dimension, when it is the record one, takes
    ! the record; a 4-D variable takes the level in its third (src/cdfio.F90:2437-2441)
    INTEGER(KIND=4), INTENT(in) :: kout, kid, klev
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: ktime
    INTEGER(KIND=4) :: ifirst(4), ierr, irec, irank, idim(4)
    idim = 0 ; ifirst = 1
    ierr = NF90_INQUIRE(kout, unlimitedDimId=irec)
    ierr = NF90_INQUIRE_VARIABLE(kout, kid, ndims=irank, dimids=idim)
    IF ( irank < 1 .OR. irank > 4 ) RETURN
    IF ( idim(irank) == irec ) ifirst(irank) = oint(ktime, 1)
    IF ( irank == 4 ) ifirst(3) = klev
  END FUNCTION slab_origin

  ! putvar: one (kpi,kpj) slab at level klev of record ktime, converted by the backend to the type of the variable;
  ! kwght rewrites the attribute 'iweight' (src/cdfio.F90:2410-2498, :2501-2640)
  INTEGER(KIND=4) FUNCTION putvarr8(kout, kid, ptab, klev, kpi, kpj, ktime, kwght)
    INTEGER(KIND=4),                  INTENT(in) :: kout, kid, klev, kpi, kpj
    REAL(KIND=8), DIMENSION(kpi,kpj), INTENT(in) :: ptab
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: ktime, kwght
    CALL tic()
    putvarr8 = NF90_PUT_VAR(kout, kid, ptab, start=slab_origin(kout, kid, klev, ktime), count=(/ kpi, kpj, 1, 1 /))
    IF ( PRESENT(kwght) ) putvarr8 = NF90_PUT_ATT(kout, kid, 'iweight', kwght)
    CALL toc(dtime_write)
  END FUNCTION putvarr8

  ! putvar of the (kpi,kpj) slab at the ADDRESS pdata (r4, or r8 with ld_r8 -- a page-locked array the GPU filled) that returns
  ! before the slab is in the file: a thread of the output file writes the queued slabs in order while the program goes on
  ! READING the next ones (the NetCDF layer's write-behind, ncclassic.h).  The memory at pdata must not be touched before
  ! wait_writes(kout) -- or any other call on kout, which all wait -- has returned.  (Not in the reference, whose level loops
  ! read, compute and write in turn; the programs of the host feed use it where the result is as large as the input.)
  INTEGER(KIND=4) FUNCTION putvar_behind(kout, kid, pdata, ld_r8, klev, kpi, kpj, ktime, kwght)
    INTEGER(KIND=4),           INTENT(in) :: kout, kid, klev, kpi, kpj
    TYPE(C_PTR),               INTENT(in) :: pdata
    LOGICAL,                   INTENT(in) :: ld_r8
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: ktime, kwght
    CALL tic()
    IF ( PRESENT(kwght) ) putvar_behind = NF90_PUT_ATT(kout, kid, 'iweight', kwght)      ! (waits for the queue: give it once)
    putvar_behind = nf90x_put_var_behind(kout, kid, pdata, ld_r8, (/ kpi, kpj /), start=slab_origin(kout, kid, klev, ktime), &
         &                                count=(/ kpi, kpj, 1, 1 /))
    CALL toc(dtime_write)
  END FUNCTION putvar_behind

  SUBROUTINE wait_writes(kout)
    !! every slab queued by putvar_behind is in the file (the time spent waiting counts as write time); a failed write stops
    !! the program as a failed putvar's NetCDF error would
    INTEGER(KIND=4), INTENT(in) :: kout
    INTEGER(KIND=4) :: ierr
    CALL tic()
    ierr = nf90x_wait_writes(kout)
    CALL toc(dtime_write)
    IF ( ierr /= NF90_NOERR ) THEN
       PRINT *, ' ERROR in a queued write : ', TRIM(NF90_STRERROR(ierr))
       STOP 98
    ENDIF
  END SUBROUTINE wait_writes

  INTEGER(KIND=4) FUNCTION putvarr4(kout, kid, ptab, klev, kpi, kpj, ktime, kwght)
    INTEGER(KIND=4),                  INTENT(in) :: kout, kid, klev, kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptab
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: ktime, kwght
    CALL tic()
    putvarr4 = NF90_PUT_VAR(kout, kid, ptab, start=slab_origin(kout, kid, klev, ktime), count=(/ kpi, kpj, 1, 1 /))
    IF ( PRESENT(kwght) ) putvarr4 = NF90_PUT_ATT(kout, kid, 'iweight', kwght)
    CALL toc(dtime_write)
  END FUNCTION putvarr4

  INTEGER(KIND=4) FUNCTION putvari2(kout, kid, ktab, klev, kpi, kpj, ktime, kwght)
    INTEGER(KIND=4),                     INTENT(in) :: kout, kid, klev, kpi, kpj
    INTEGER(KIND=2), DIMENSION(kpi,kpj), INTENT(in) :: ktab
    INTEGER(KIND=4), OPTIONAL,           INTENT(in) :: ktime, kwght
    CALL tic()
    putvari2 = NF90_PUT_VAR(kout, kid, ktab, start=slab_origin(kout, kid, klev, ktime), count=(/ kpi, kpj, 1, 1 /))
    IF ( PRESENT(kwght) ) putvari2 = NF90_PUT_ATT(kout, kid, 'iweight', kwght)
    CALL toc(dtime_write)
  END FUNCTION putvari2

  INTEGER(KIND=4) FUNCTION putvarzo(kout, kid, ptab, klev, kpi, kpj, ktime)
    !! a zonal (kpi = 1) or meridional (kpj = 1) line given as a 1-D r4 array (src/cdfio.F90:2642-2681)
    INTEGER(KIND=4),            INTENT(in) :: kout, kid, klev, kpi, kpj
    REAL(KIND=4), DIMENSION(:), INTENT(in) :: ptab
    INTEGER(KIND=4), OPTIONAL,  INTENT(in) :: ktime
    CALL tic()
    putvarzo = NF90_PUT_VAR(kout, kid, ptab, start=slab_origin(kout, kid, klev, ktime), count=(/ kpi, kpj, 1, 1 /))
    CALL toc(dtime_write)
  END FUNCTION putvarzo

  INTEGER(KIND=4) FUNCTION putvar0dt(kout, kid, pvalue, ktime)
    !! one value, given as a (1,1) array, at record ktime (src/cdfio.F90:2809-2848)
    INTEGER(KIND=4),              INTENT(in) :: kout, kid
    REAL(KIND=4), DIMENSION(1,1), INTENT(in) :: pvalue
    INTEGER(KIND=4), OPTIONAL,    INTENT(in) :: ktime
    putvar0dt = NF90_PUT_VAR(kout, kid, pvalue, start=slab_origin(kout, kid, 1, ktime), count=(/ 1, 1, 1, 1 /))
  END FUNCTION putvar0dt

  INTEGER(KIND=4) FUNCTION putvar0ds(kout, kid, pvalue, ktime)
    !! the same for a scalar (src/cdfio.F90:2850-2892)
    INTEGER(KIND=4),           INTENT(in) :: kout, kid
    REAL(KIND=4),              INTENT(in) :: pvalue
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: ktime
    putvar0ds = putvar0dt(kout, kid, RESHAPE((/ pvalue /), (/ 1, 1 /)), ktime)
  END FUNCTION putvar0ds

  INTEGER(KIND=4) FUNCTION axis_variable(cdtype)
    ! 'T' time, 'D' depth, 'X' / 'Y' the 1-D longitude / latitude of create's cdlonvar / cdlatvar
    CHARACTER(LEN=1), INTENT(in) :: cdtype
    axis_variable = -1
    IF ( SCAN(cdtype, 'Tt') == 1 ) axis_variable = nid_tim
    IF ( SCAN(cdtype, 'Dd') == 1 ) axis_variable = nid_dep
    IF ( SCAN(cdtype, 'Xx') == 1 ) axis_variable = nid_lon1d
    IF ( SCAN(cdtype, 'Yy') == 1 ) axis_variable = nid_lat1d
  END FUNCTION axis_variable

  INTEGER(KIND=4) FUNCTION putvar1d4(kout, ptab, kk, cdtype)
    !! the values of an axis of the output file, r4 or r8 (src/cdfio.F90:2684-2756)
    INTEGER(KIND=4),             INTENT(in) :: kout, kk
    REAL(KIND=4), DIMENSION(kk), INTENT(in) :: ptab
    CHARACTER(LEN=1),            INTENT(in) :: cdtype
    putvar1d4 = NF90_PUT_VAR(kout, axis_variable(cdtype), ptab, start=(/ 1 /), count=(/ kk /))
  END FUNCTION putvar1d4

  INTEGER(KIND=4) FUNCTION putvar1d8(kout, ddtab, kk, cdtype)
    INTEGER(KIND=4),             INTENT(in) :: kout, kk
    REAL(KIND=8), DIMENSION(kk), INTENT(in) :: ddtab
    CHARACTER(LEN=1),            INTENT(in) :: cdtype
    putvar1d8 = NF90_PUT_VAR(kout, axis_variable(cdtype), ddtab, start=(/ 1 /), count=(/ kk /))
  END FUNCTION putvar1d8

  INTEGER(KIND=4) FUNCTION closeout(kout)
    !! closes the output file (src/cdfio.F90:2894-2906); the read handles of the run go with it
    INTEGER(KIND=4), INTENT(in) :: kout
    CALL tic()
    closeout = NF90_CLOSE(kout)
    CALL toc(dtime_write)
    ! the close waits for the slabs putvar_behind queued and reports a failed one: stop as the synchronous putvar's
    ! NetCDF error would have (no program looks at closeout's value, the reference's included)
    IF ( closeout /= NF90_NOERR ) THEN
       PRINT *, ' ERROR while closing the output file : ', TRIM(NF90_STRERROR(closeout))
       STOP 98
    ENDIF
    CALL src_forget()
    IF ( ntiming == 1 ) WRITE(0,'(a,f9.3,a,f9.3)') ' CDFT_TIMING cdfio (s): reads', dtime_read, '  writes', dtime_write
  END FUNCTION closeout

END MODULE cdfio
