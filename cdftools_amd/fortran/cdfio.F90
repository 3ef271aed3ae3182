MODULE cdfio
  !!======================================================================
  !! cdfio : NetCDF access layer of the tools -- the module surface the PROGRAM units are written
  !! against.  Same public names, argument lists and conventions as the reference's src/cdfio.F90
  !! for the part of the surface the hot-path tools use (cited per routine); the implementation is
  !! new and sits on the NF90 API (module netcdf: the bundled classic-format backend, or a real
  !! netcdf-fortran).  Conventions kept from the reference:
  !!   - input files are opened and closed on every read call; one output file is built at a time
  !!     through the module-level ids set by create (not re-entrant, src/cdfio.F90:72-73);
  !!   - getvar returns one (kpi,kpj) r4 slab, unpacking scale_factor / add_offset / savelog10;
  !!   - chkfile / chkvar return .TRUE. when the file / variable is MISSING;
  !!   - fatal NetCDF errors print a message and STOP 98.
  !! Extension: getvar3d / readvar3d read a whole (kpi,kpj,kpk) record in one call (what the HIP path wants).
  !!======================================================================
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

  ! ids of the output file under construction
  INTEGER(KIND=4) :: nid_x, nid_y, nid_z, nid_t, nid_lon, nid_lat, nid_dep, nid_tim, nid_lon1d, nid_lat1d
  ! time / global attributes captured from the first file seen by getdim (src/cdfio.F90:939-994)
  INTEGER(KIND=4)    :: nstart_date = -1
  CHARACTER(LEN=256) :: ctime_units = 'seconds since 0001-01-01 00:00:00', ctime_origin = 'N/A', calendar = 'N/A'
  CHARACTER(LEN=256) :: config = 'N/A', ccase = 'N/A'
  CHARACTER(LEN=10)  :: cfreq = 'N/A'
  LOGICAL            :: l_first_getdim = .TRUE.
  ! CDFT_TIMING=1: wall time of the bulk reads (getvar, getvar3d, readvar3d) and slab writes (putvar), summed over the
  ! run and printed on stderr by closeout (stdout stays what the reference prints)
  INTEGER(KIND=4)    :: ntiming = -1                    ! -1 not asked yet, 0 off, 1 on
  INTEGER(KIND=8)    :: nclock0
  REAL(KIND=8)       :: dtime_read = 0.d0, dtime_write = 0.d0

  INTERFACE putvar
     MODULE PROCEDURE putvarr8, putvarr4, putvari2, putvarzo, reputvarr4, putvare3
  END INTERFACE
  INTERFACE putvar1d
     MODULE PROCEDURE putvar1d4, putvar1d8, reputvar1d4, reputvar1d8
  END INTERFACE
  INTERFACE atted
     MODULE PROCEDURE atted_char, atted_r4
  END INTERFACE
  INTERFACE putvar0d
     MODULE PROCEDURE putvar0dt, putvar0ds
  END INTERFACE
  INTERFACE getatt
     MODULE PROCEDURE getattr, getatti, getattc
  END INTERFACE

  PUBLIC :: getvaratt, getnvar, getvarname, getipk, getvarid, getdimvar, cvaratt, gettimeatt, puttimeatt, atted
  PUBLIC :: getvarxz, getvaryz, getvar3dt, getvar4d, gettimeseries
  PUBLIC :: chkfile, chkvar, getdim, getvdim, getspval, getatt, getvar, getvar1d, getvare3, getvar3d, readvar3d, readvar3d_i1
  PUBLIC :: create, createvar, putatt, putheadervar, putvar, putvar1d, putvar0d, closeout, ncopen, copyatt, ERR_HDL

CONTAINS

  SUBROUTINE ERR_HDL(kstatus)
    INTEGER(KIND=4), INTENT(in) :: kstatus
    IF ( kstatus /= NF90_NOERR ) THEN
       PRINT *, 'ERROR in NETCDF routine, status=', kstatus
       PRINT *, NF90_STRERROR(kstatus)
       STOP 98
    ENDIF
  END SUBROUTINE ERR_HDL

  SUBROUTINE tic()
    CHARACTER(LEN=8) :: cl
    IF ( ntiming < 0 ) THEN
       CALL GET_ENVIRONMENT_VARIABLE('CDFT_TIMING', cl)
       ntiming = MERGE(1, 0, cl(1:1) == '1')
    ENDIF
    IF ( ntiming == 1 ) CALL SYSTEM_CLOCK(nclock0)
  END SUBROUTINE tic

  SUBROUTINE toc(dtotal)
    REAL(KIND=8), INTENT(inout) :: dtotal
    INTEGER(KIND=8) :: iclock, irate
    IF ( ntiming /= 1 ) RETURN
    CALL SYSTEM_CLOCK(iclock, irate)
    dtotal = dtotal + DBLE(iclock - nclock0) / DBLE(irate)
  END SUBROUTINE toc

  FUNCTION iom_name(cdvar) RESULT(clvar)
    ! name of a vertical metric in mesh_zgr for the detected layout (src/cdfio.F90:1510-1536)
    CHARACTER(LEN=*), INTENT(in) :: cdvar
    CHARACTER(LEN=256) :: clvar, clbase
    clvar = cdvar
    IF      ( cdvar == cn_ve3t ) THEN ; clbase = 'e3t'
    ELSE IF ( cdvar == cn_ve3u ) THEN ; clbase = 'e3u'
    ELSE IF ( cdvar == cn_ve3v ) THEN ; clbase = 'e3v'
    ELSE IF ( cdvar == cn_ve3w ) THEN ; clbase = 'e3w'
    ELSE ; RETURN
    ENDIF
    SELECT CASE ( cg_zgr_ver )
    CASE ( 'v2.0' ) ; clvar = TRIM(clbase)//'_ps'
    CASE ( 'v3.0' ) ; clvar = clbase
    CASE ( 'v3.6' ) ; clvar = TRIM(clbase)//'_0'
    END SELECT
  END FUNCTION iom_name

  FUNCTION env_value(cdname) RESULT(cl)
    CHARACTER(LEN=*), INTENT(in) :: cdname
    CHARACTER(LEN=256) :: cl
    INTEGER :: istat
    cl = ' '
    CALL GET_ENVIRONMENT_VARIABLE(cdname, cl, STATUS=istat)
    IF ( istat /= 0 ) cl = ' '
  END FUNCTION env_value

  ! ------------------------------------------------------------------------------------------------
  LOGICAL FUNCTION chkfile(cd_file, ld_verbose)
    !! .TRUE. when the file is missing; seeing mesh_zgr sets cg_zgr_ver (src/cdfio.F90:3032-3070)
    CHARACTER(LEN=*),  INTENT(in) :: cd_file
    LOGICAL, OPTIONAL, INTENT(in) :: ld_verbose
    LOGICAL :: ll_exist, ll_verbose
    ll_verbose = .TRUE. ; IF ( PRESENT(ld_verbose) ) ll_verbose = ld_verbose
    chkfile = .FALSE.
    IF ( TRIM(cd_file) == 'none' ) RETURN
    INQUIRE(FILE=TRIM(cd_file), EXIST=ll_exist)
    IF ( ll_exist ) THEN
       IF ( cd_file == cn_fzgr ) CALL SetMeshZgrVersion
    ELSE
       IF ( ll_verbose ) PRINT *, ' File ', TRIM(cd_file), ' is missing '
       chkfile = .TRUE.
    ENDIF
  END FUNCTION chkfile

  LOGICAL FUNCTION chkvar(cd_file, cd_var, ld_verbose)
    !! .TRUE. when the variable is missing from the file (src/cdfio.F90:3072-3112)
    CHARACTER(LEN=*),  INTENT(in) :: cd_file, cd_var
    LOGICAL, OPTIONAL, INTENT(in) :: ld_verbose
    INTEGER(KIND=4) :: istatus, incid, id_var
    LOGICAL :: ll_verbose
    ll_verbose = .TRUE. ; IF ( PRESENT(ld_verbose) ) ll_verbose = ld_verbose
    chkvar = .FALSE.
    IF ( TRIM(cd_var) == 'none' ) RETURN
    istatus = NF90_OPEN(cd_file, NF90_NOWRITE, incid)
    IF ( istatus == NF90_NOERR ) THEN
       istatus = NF90_INQ_VARID(incid, cd_var, id_var)
       id_var  = NF90_CLOSE(incid)
    ENDIF
    IF ( istatus /= NF90_NOERR ) THEN
       IF ( ll_verbose ) PRINT *, ' Var ', TRIM(cd_var), ' is missing in file ', TRIM(cd_file)
       chkvar = .TRUE.
    ENDIF
  END FUNCTION chkvar

  SUBROUTINE SetMeshZgrVersion()
    !! e3t_0 absent: v2.0; e3t_0(t,z): v3.0; e3t_0(t,z,y,x): v3.6 (src/cdfio.F90:3310-3335)
    CHARACTER(LEN=256) :: clvar
    clvar = 'e3t_0'
    IF ( chkvar(cn_fzgr, clvar, .FALSE.) ) THEN
       cg_zgr_ver = 'v2.0'
    ELSE IF ( getvdim(cn_fzgr, clvar) == 1 ) THEN
       cg_zgr_ver = 'v3.0'
    ELSE
       cg_zgr_ver = 'v3.6'
    ENDIF
    PRINT *, ' mesh_zgr version is ', TRIM(cg_zgr_ver)
  END SUBROUTINE SetMeshZgrVersion

  INTEGER(KIND=4) FUNCTION getdim(cdfile, cdim_name, cdtrue, kstatus, ldexact)
    !! Length of the first dimension whose name CONTAINS cdim_name (exact match for cn_x or with
    !! ldexact); 0 / 'unknown' / kstatus=1 when not found.  The first call also captures the time
    !! attributes and the global attributes that outputs re-emit (src/cdfio.F90:876-1003).
    CHARACTER(LEN=*),             INTENT(in ) :: cdfile, cdim_name
    CHARACTER(LEN=256), OPTIONAL, INTENT(out) :: cdtrue
    INTEGER(KIND=4),    OPTIONAL, INTENT(out) :: kstatus
    LOGICAL,            OPTIONAL, INTENT(in ) :: ldexact
    INTEGER(KIND=4) :: incid, istatus, idims, jdim, id_dim, ilen
    CHARACTER(LEN=256) :: clnam
    LOGICAL :: ll_exact, ll_found
    getdim = 0
    IF ( PRESENT(kstatus) ) kstatus = 0
    IF ( PRESENT(cdtrue) ) cdtrue = 'unknown'
    ll_exact = ( cdim_name == cn_x )
    IF ( PRESENT(ldexact) ) ll_exact = ll_exact .OR. ldexact
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    IF ( istatus /= NF90_NOERR ) THEN
       IF ( PRESENT(kstatus) ) kstatus = 1
       RETURN
    ENDIF
    ll_found = .FALSE.
    IF ( ll_exact ) THEN
       istatus = NF90_INQ_DIMID(incid, cdim_name, id_dim)
       IF ( istatus /= NF90_NOERR ) THEN
          PRINT *, NF90_STRERROR(istatus)
          PRINT *, ' Exact dimension name ', TRIM(cdim_name), ' not found in ', TRIM(cdfile) ; STOP 98
       ENDIF
       istatus = NF90_INQUIRE_DIMENSION(incid, id_dim, len=getdim)
       IF ( PRESENT(cdtrue) ) cdtrue = cdim_name
       ll_found = .TRUE.
    ELSE
       istatus = NF90_INQUIRE(incid, nDimensions=idims)
       DO jdim = 1, idims
          istatus = NF90_INQUIRE_DIMENSION(incid, jdim, name=clnam, len=ilen)
          IF ( INDEX(clnam, TRIM(cdim_name)) /= 0 ) THEN
             getdim = ilen ; ll_found = .TRUE.
             IF ( PRESENT(cdtrue) ) cdtrue = clnam
             EXIT
          ENDIF
       END DO
    ENDIF
    IF ( .NOT. ll_found ) THEN
       getdim = 0
       IF ( PRESENT(kstatus) ) kstatus = 1
    ENDIF
    IF ( l_first_getdim ) THEN
       l_first_getdim = .FALSE.
       CALL capture_file_attributes(incid)
    ENDIF
    istatus = NF90_CLOSE(incid)
  END FUNCTION getdim

  SUBROUTINE capture_file_attributes(kncid)
    ! time_counter attributes and start_date / output_frequency / CONFIG / CASE: file attribute first,
    ! then the environment variable of the same name, then the default (src/cdfio.F90:939-994)
    INTEGER(KIND=4), INTENT(in) :: kncid
    INTEGER(KIND=4) :: istatus, idv
    CHARACTER(LEN=256) :: cl
    istatus = NF90_INQ_VARID(kncid, cn_vtimec, idv)
    IF ( istatus == NF90_NOERR ) THEN
       istatus = NF90_GET_ATT(kncid, idv, 'units', cl)       ; IF ( istatus == NF90_NOERR ) ctime_units = cl
       istatus = NF90_GET_ATT(kncid, idv, 'time_origin', cl) ; IF ( istatus == NF90_NOERR ) ctime_origin = cl
       istatus = NF90_GET_ATT(kncid, idv, 'calendar', cl)    ; IF ( istatus == NF90_NOERR ) calendar = cl
    ENDIF
    cl = env_value('start_date')
    IF ( NF90_INQUIRE_ATTRIBUTE(kncid, NF90_GLOBAL, 'start_date') == NF90_NOERR ) THEN
       istatus = NF90_GET_ATT(kncid, NF90_GLOBAL, 'start_date', nstart_date)
    ELSE IF ( cl /= ' ' ) THEN ; READ(cl, *) nstart_date
    ELSE ; nstart_date = -1
    ENDIF
    cl = env_value('output_frequency')
    IF ( NF90_INQUIRE_ATTRIBUTE(kncid, NF90_GLOBAL, 'output_frequency') == NF90_NOERR ) THEN
       istatus = NF90_GET_ATT(kncid, NF90_GLOBAL, 'output_frequency', cfreq)
    ELSE IF ( cl /= ' ' ) THEN ; cfreq = TRIM(cl)
    ELSE ; cfreq = 'N/A'
    ENDIF
    cl = env_value('CONFIG')
    IF ( NF90_INQUIRE_ATTRIBUTE(kncid, NF90_GLOBAL, 'CONFIG') == NF90_NOERR ) THEN
       istatus = NF90_GET_ATT(kncid, NF90_GLOBAL, 'CONFIG', config)
    ELSE IF ( cl /= ' ' ) THEN ; config = cl
    ELSE ; config = 'N/A'
    ENDIF
    cl = env_value('CASE')
    IF ( NF90_INQUIRE_ATTRIBUTE(kncid, NF90_GLOBAL, 'CASE') == NF90_NOERR ) THEN
       istatus = NF90_GET_ATT(kncid, NF90_GLOBAL, 'CASE', ccase)
    ELSE IF ( cl /= ' ' ) THEN ; ccase = cl
    ELSE ; ccase = 'N/A'
    ENDIF
  END SUBROUTINE capture_file_attributes

  INTEGER(KIND=4) FUNCTION getvdim(cdfile, cdvar)
    !! number of dimensions of cdvar minus one (src/cdfio.F90:1042-1083); a missing variable is fatal
    !! here (the reference prompts interactively for another one)
    CHARACTER(LEN=*), INTENT(in)    :: cdfile
    CHARACTER(LEN=*), INTENT(inout) :: cdvar
    INTEGER(KIND=4) :: incid, id_var, idi
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    IF ( NF90_INQ_VARID(incid, cdvar, id_var) /= NF90_NOERR ) THEN
       PRINT *, ' getvdim : variable ', TRIM(cdvar), ' not found in ', TRIM(cdfile) ; STOP 98
    ENDIF
    CALL ERR_HDL(NF90_INQUIRE_VARIABLE(incid, id_var, ndims=idi))
    getvdim = idi - 1
    CALL ERR_HDL(NF90_CLOSE(incid))
  END FUNCTION getvdim

  SUBROUTINE find_spval(kncid, kvar, pspval, cdmissing)
    ! missing value of a variable: cn_missing_value, then the usual spellings (src/cdfio.F90:1022-1036)
    INTEGER(KIND=4), INTENT(in) :: kncid, kvar
    REAL(KIND=4), INTENT(out) :: pspval
    CHARACTER(LEN=*), OPTIONAL, INTENT(out) :: cdmissing
    CHARACTER(LEN=13), PARAMETER :: cl_try(3) = (/ 'missing_value', 'Fillvalue    ', '_Fillvalue   ' /)
    INTEGER :: jt
    IF ( PRESENT(cdmissing) ) cdmissing = cn_missing_value
    IF ( NF90_GET_ATT(kncid, kvar, cn_missing_value, pspval) == NF90_NOERR ) RETURN
    DO jt = 1, SIZE(cl_try)
       IF ( NF90_GET_ATT(kncid, kvar, TRIM(cl_try(jt)), pspval) == NF90_NOERR ) THEN
          IF ( PRESENT(cdmissing) ) cdmissing = TRIM(cl_try(jt))
          RETURN
       ENDIF
    END DO
    pspval = 0.
  END SUBROUTINE find_spval

  REAL(KIND=4) FUNCTION getspval(cdfile, cdvar, cdmissing)
    CHARACTER(LEN=*),           INTENT(in ) :: cdfile, cdvar
    CHARACTER(LEN=*), OPTIONAL, INTENT(out) :: cdmissing
    INTEGER(KIND=4) :: incid, id_var, istatus
    getspval = 0.
    IF ( PRESENT(cdmissing) ) cdmissing = cn_missing_value
    IF ( NF90_OPEN(cdfile, NF90_NOWRITE, incid) /= NF90_NOERR ) RETURN
    IF ( NF90_INQ_VARID(incid, cdvar, id_var) == NF90_NOERR ) CALL find_spval(incid, id_var, getspval, cdmissing)
    istatus = NF90_CLOSE(incid)
  END FUNCTION getspval

  SUBROUTINE att_problem(kstatus, cdfile, cdvar, cdatt, cddefault)
    INTEGER(KIND=4), INTENT(in) :: kstatus
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt, cddefault
    PRINT *, ' getatt problem :', NF90_STRERROR(kstatus)
    PRINT *, ' attribute :', TRIM(cdatt)
    PRINT *, ' variable  :', TRIM(cdvar)
    PRINT *, ' file      :', TRIM(cdfile)
    PRINT *, ' return default ', cddefault
  END SUBROUTINE att_problem

  REAL(KIND=4) FUNCTION getattr(cdfile, cdvar, cdatt)
    !! r4 attribute of a variable, 0. (with a message) when absent (src/cdfio.F90:688-716)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt
    INTEGER(KIND=4) :: istatus, incid, idv
    getattr = 0.
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, idv)
    IF ( istatus /= NF90_NOERR ) PRINT *, TRIM(NF90_STRERROR(istatus)), ' when looking for ', TRIM(cdvar), ' in getatt.'
    istatus = NF90_GET_ATT(incid, idv, cdatt, getattr)
    IF ( istatus /= NF90_NOERR ) THEN
       CALL att_problem(istatus, cdfile, cdvar, cdatt, '0') ; getattr = 0.
    ENDIF
    istatus = NF90_CLOSE(incid)
  END FUNCTION getattr

  INTEGER(KIND=4) FUNCTION getatti(cdfile, cdvar, cdatt, kidum)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt
    INTEGER(KIND=4) :: kidum                             ! only selects this specific
    INTEGER(KIND=4) :: istatus, incid, idv
    getatti = 0
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, idv)
    IF ( istatus /= NF90_NOERR ) PRINT *, TRIM(NF90_STRERROR(istatus)), ' when looking for ', TRIM(cdvar), ' in getatt.'
    istatus = NF90_GET_ATT(incid, idv, cdatt, getatti)
    IF ( istatus /= NF90_NOERR ) THEN
       CALL att_problem(istatus, cdfile, cdvar, cdatt, '0') ; getatti = 0
    ENDIF
    istatus = NF90_CLOSE(incid)
  END FUNCTION getatti

  INTEGER(KIND=4) FUNCTION getnvar(cdfile)
    !! number of variables of a file (src/cdfio.F90:1127-1145)
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4) :: incid, istatus
    getnvar = 0
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQUIRE(incid, nvariables=getnvar)
    istatus = NF90_CLOSE(incid)
  END FUNCTION getnvar

  FUNCTION getipk(cdfile, knvars, cdep)
    !! levels of every variable: 4-D -> size of the depth dimension, 3-D -> 1, else 0 (src/cdfio.F90:1148-1194)
    CHARACTER(LEN=*),           INTENT(in) :: cdfile
    INTEGER(KIND=4),            INTENT(in) :: knvars
    CHARACTER(LEN=*), OPTIONAL, INTENT(in) :: cdep
    INTEGER(KIND=4), DIMENSION(knvars) :: getipk
    INTEGER(KIND=4) :: incid, inb, jv, iipk, istatus
    CHARACTER(LEN=256) :: cldep
    cldep = 'dep' ; IF ( PRESENT(cdep) ) cldep = cdep
    iipk = getdim(cdfile, TRIM(cldep), kstatus=istatus)
    IF ( istatus /= 0 ) THEN
       PRINT *, ' getipk : vertical dim not found ...assume 1'
       iipk = 1
    ENDIF
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    DO jv = 1, knvars
       istatus = NF90_INQUIRE_VARIABLE(incid, jv, ndims=inb)
       IF ( inb == 4 ) THEN ; getipk(jv) = iipk
       ELSE IF ( inb == 3 ) THEN ; getipk(jv) = 1
       ELSE ; getipk(jv) = 0
       ENDIF
    END DO
    istatus = NF90_CLOSE(incid)
  END FUNCTION getipk

  FUNCTION getvarname(cdfile, knvars, sdtypvar)
    !! names of the variables of a file, and their standard attributes into sdtypvar (src/cdfio.F90:1197-1290)
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4),  INTENT(in) :: knvars
    TYPE(variable), DIMENSION(knvars) :: sdtypvar
    CHARACTER(LEN=256), DIMENSION(knvars) :: getvarname
    INTEGER(KIND=4) :: incid, jv, istatus, iatt
    REAL(KIND=4) :: zatt
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    DO jv = 1, knvars
       istatus = NF90_INQUIRE_VARIABLE(incid, jv, name=getvarname(jv))
       sdtypvar(jv)%cname = getvarname(jv)
       sdtypvar(jv)%cunits            = catt('units')
       sdtypvar(jv)%rmissing_value    = getspval(cdfile, sdtypvar(jv)%cname)
       sdtypvar(jv)%valid_min = 0. ; sdtypvar(jv)%valid_max = 0. ; sdtypvar(jv)%iwght = 1
       IF ( NF90_INQUIRE_ATTRIBUTE(incid, jv, 'valid_min') == NF90_NOERR ) THEN
          istatus = NF90_GET_ATT(incid, jv, 'valid_min', zatt) ; sdtypvar(jv)%valid_min = zatt
       ENDIF
       IF ( NF90_INQUIRE_ATTRIBUTE(incid, jv, 'valid_max') == NF90_NOERR ) THEN
          istatus = NF90_GET_ATT(incid, jv, 'valid_max', zatt) ; sdtypvar(jv)%valid_max = zatt
       ENDIF
       IF ( NF90_INQUIRE_ATTRIBUTE(incid, jv, 'iweight') == NF90_NOERR ) THEN
          istatus = NF90_GET_ATT(incid, jv, 'iweight', iatt) ; sdtypvar(jv)%iwght = iatt
       ENDIF
       sdtypvar(jv)%clong_name        = catt('long_name')
       sdtypvar(jv)%cshort_name       = catt('short_name')
       sdtypvar(jv)%conline_operation = catt('online_operation')
       sdtypvar(jv)%caxis             = catt('axis')
    END DO
    istatus = NF90_CLOSE(incid)
  CONTAINS
    CHARACTER(LEN=256) FUNCTION catt(cdatt)
      CHARACTER(LEN=*), INTENT(in) :: cdatt
      INTEGER(KIND=4) :: ilen
      catt = 'N/A'
      IF ( NF90_INQUIRE_ATTRIBUTE(incid, jv, cdatt, len=ilen) == NF90_NOERR ) THEN
         catt = ' '
         istatus = NF90_GET_ATT(incid, jv, cdatt, catt)
      ENDIF
    END FUNCTION catt
  END FUNCTION getvarname

  INTEGER(KIND=4) FUNCTION getvaratt(cdfile, cdvar, cdunits, pmissing_value, cdlong_name, cdshort_name)
    !! units, missing value, long and short name of a variable; an absent attribute leaves its argument
    !! untouched and only the status of the last lookup is returned (src/cdfio.F90:488-514)
    CHARACTER(LEN=*), INTENT(in)    :: cdfile, cdvar
    REAL(KIND=4),     INTENT(out)   :: pmissing_value
    CHARACTER(LEN=*), INTENT(inout) :: cdunits, cdlong_name, cdshort_name
    INTEGER(KIND=4) :: istatus, incid, idv
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, idv)
    IF ( NF90_INQUIRE_ATTRIBUTE(incid, idv, 'units') == NF90_NOERR ) istatus = NF90_GET_ATT(incid, idv, 'units', cdunits)
    pmissing_value = getspval(cdfile, cdvar)
    IF ( NF90_INQUIRE_ATTRIBUTE(incid, idv, 'long_name') == NF90_NOERR ) istatus = NF90_GET_ATT(incid, idv, 'long_name', cdlong_name)
    getvaratt = NF90_INQUIRE_ATTRIBUTE(incid, idv, 'short_name')
    IF ( getvaratt == NF90_NOERR ) getvaratt = NF90_GET_ATT(incid, idv, 'short_name', cdshort_name)
    istatus = NF90_CLOSE(incid)
  END FUNCTION getvaratt

  CHARACTER(LEN=256) FUNCTION getattc(cdfile, cdvar, cdatt, cdum)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt
    CHARACTER(LEN=*) :: cdum                             ! only selects this specific
    INTEGER(KIND=4) :: istatus, incid, idv
    getattc = 'N/A'
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, idv)
    IF ( istatus /= NF90_NOERR ) PRINT *, TRIM(NF90_STRERROR(istatus)), ' when looking for ', TRIM(cdvar), ' in getatt.'
    istatus = NF90_GET_ATT(incid, idv, cdatt, getattc)
    IF ( istatus /= NF90_NOERR ) THEN
       CALL att_problem(istatus, cdfile, cdvar, cdatt, 'N/A') ; getattc = 'N/A'
    ENDIF
    istatus = NF90_CLOSE(incid)
  END FUNCTION getattc

  ! ------------------------------------------------------------------------------------------------
  SUBROUTINE packing(kncid, kvar, psf, pao, ldlog, ldsf, ldao)
    ! scale_factor / add_offset / savelog10 of a variable (src/cdfio.F90:1563-1584)
    INTEGER(KIND=4), INTENT(in) :: kncid, kvar
    REAL(KIND=4), INTENT(out) :: psf, pao
    LOGICAL, INTENT(out) :: ldlog, ldsf, ldao
    INTEGER(KIND=4) :: ilog
    psf = 1. ; pao = 0. ; ldlog = .FALSE. ; ldsf = .FALSE. ; ldao = .FALSE.
    IF ( NF90_INQUIRE_ATTRIBUTE(kncid, kvar, 'savelog10') == NF90_NOERR ) THEN
       IF ( NF90_GET_ATT(kncid, kvar, 'savelog10', ilog) == NF90_NOERR ) ldlog = ( ilog /= 0 )
    ENDIF
    IF ( NF90_INQUIRE_ATTRIBUTE(kncid, kvar, 'scale_factor') == NF90_NOERR ) THEN
       IF ( NF90_GET_ATT(kncid, kvar, 'scale_factor', psf) == NF90_NOERR ) ldsf = ( psf /= 1. )
    ENDIF
    IF ( NF90_INQUIRE_ATTRIBUTE(kncid, kvar, 'add_offset') == NF90_NOERR ) THEN
       IF ( NF90_GET_ATT(kncid, kvar, 'add_offset', pao) == NF90_NOERR ) ldao = ( pao /= 0. )
    ENDIF
  END SUBROUTINE packing

  SUBROUTINE level_start(kncid, kvar, klev, ktime, kstart)
    ! start vector of a (level, time) slab: a variable whose 3rd dimension is the unlimited one is
    ! (x,y,t), anything else is read as (x,y,z,t) (src/cdfio.F90:1544-1554)
    INTEGER(KIND=4), INTENT(in) :: kncid, kvar, klev, ktime
    INTEGER(KIND=4), INTENT(inout) :: kstart(4)
    INTEGER(KIND=4) :: istatus, iunlim, inb, idd(4)
    idd = 0
    istatus = NF90_INQUIRE(kncid, unlimitedDimId=iunlim)
    istatus = NF90_INQUIRE_VARIABLE(kncid, kvar, ndims=inb, dimids=idd)
    IF ( idd(3) == iunlim ) THEN
       kstart(3) = ktime ; kstart(4) = 1
    ELSE
       kstart(3) = klev ; kstart(4) = ktime
    ENDIF
  END SUBROUTINE level_start

  FUNCTION getvar(cdfile, cdvar, klev, kpi, kpj, kimin, kjmin, ktime, ldiom)
    !! One (kpi,kpj) r4 slab of cdvar at level klev, record ktime, window origin (kimin,kjmin);
    !! a window crossing the E-W boundary wraps skipping the overlap column; ldiom maps e3t/e3u/e3v/e3w
    !! to their mesh_zgr names (src/cdfio.F90:1425-1609).
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: klev
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kjmin, ktime
    LOGICAL,         OPTIONAL, INTENT(in) :: ldiom
    REAL(KIND=4), DIMENSION(kpi,kpj) :: getvar
    INTEGER(KIND=4) :: incid, id_var, istatus, ilev, imin, jmin, itime, ipiglo, ileft, istart(4), icount(4)
    REAL(KIND=4) :: zsf, zao, zspval
    LOGICAL :: llog, lsf, lao, llperio
    CHARACTER(LEN=256) :: clvar
    ilev = 1  ; IF ( PRESENT(klev ) ) ilev  = klev
    imin = 1  ; IF ( PRESENT(kimin) ) imin  = kimin
    jmin = 1  ; IF ( PRESENT(kjmin) ) jmin  = kjmin
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    clvar = cdvar
    IF ( PRESENT(ldiom) ) THEN
       IF ( ldiom ) clvar = iom_name(cdvar)
    ENDIF
    llperio = .FALSE.
    IF ( PRESENT(kimin) ) THEN
       ipiglo = getdim(cdfile, cn_x, ldexact=.TRUE.)
       llperio = ( imin + kpi - 1 > ipiglo )
    ENDIF
    CALL tic()
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    CALL ERR_HDL(NF90_INQ_VARID(incid, clvar, id_var))
    istart = (/ imin, jmin, 1, 1 /) ; icount = (/ kpi, kpj, 1, 1 /)
    CALL level_start(incid, id_var, ilev, itime, istart)
    CALL find_spval(incid, id_var, zspval)
    CALL packing(incid, id_var, zsf, zao, llog, lsf, lao)
    IF ( llperio ) THEN
       ileft = ipiglo - imin                          ! columns imin .. ipiglo-1, then from column 2 on
       istart(1) = imin ; icount(1) = ileft
       istatus = NF90_GET_VAR(incid, id_var, getvar(1:ileft,:), start=istart, count=icount)
       istart(1) = 2 ; icount(1) = kpi - ileft
       IF ( istatus == NF90_NOERR ) istatus = NF90_GET_VAR(incid, id_var, getvar(ileft+1:kpi,:), start=istart, count=icount)
    ELSE
       istatus = NF90_GET_VAR(incid, id_var, getvar, start=istart, count=icount)
    ENDIF
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in getvar for ', TRIM(clvar)
       CALL ERR_HDL(istatus)
    ENDIF
    IF ( lsf  ) WHERE ( getvar /= zspval ) getvar = getvar * zsf
    IF ( lao  ) WHERE ( getvar /= zspval ) getvar = getvar + zao
    IF ( llog ) WHERE ( getvar /= zspval ) getvar = 10**getvar
    istatus = NF90_CLOSE(incid)
    CALL toc(dtime_read)
  END FUNCTION getvar

  FUNCTION getvar3d(cdfile, cdvar, kpi, kpj, kpk, kimin, kjmin, kkmin, ktime, ldiom)
    !! A (kpi,kpj,kpk) block of one record in one read, window origin (kimin,kjmin,kkmin) (src/cdfio.F90:1611-1715; ldiom
    !! as in getvar): same unpacking as getvar level by level.
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj, kpk
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kjmin, kkmin, ktime
    LOGICAL,         OPTIONAL, INTENT(in) :: ldiom
    REAL(KIND=4), DIMENSION(kpi,kpj,kpk) :: getvar3d
    INTEGER(KIND=4) :: incid, id_var, istatus, itime, iimin, ijmin, ikmin
    REAL(KIND=4) :: zsf, zao, zspval
    LOGICAL :: llog, lsf, lao
    CHARACTER(LEN=256) :: clvar
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    iimin = 1 ; IF ( PRESENT(kimin) ) iimin = kimin
    ijmin = 1 ; IF ( PRESENT(kjmin) ) ijmin = kjmin
    ikmin = 1 ; IF ( PRESENT(kkmin) ) ikmin = kkmin
    clvar = cdvar
    IF ( PRESENT(ldiom) ) THEN
       IF ( ldiom ) clvar = iom_name(cdvar)
    ENDIF
    CALL tic()
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    CALL ERR_HDL(NF90_INQ_VARID(incid, clvar, id_var))
    CALL find_spval(incid, id_var, zspval)
    CALL packing(incid, id_var, zsf, zao, llog, lsf, lao)
    istatus = NF90_GET_VAR(incid, id_var, getvar3d, start=(/ iimin, ijmin, ikmin, itime /), count=(/ kpi, kpj, kpk, 1 /))
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in getvar3d for ', TRIM(clvar)
       CALL ERR_HDL(istatus)
    ENDIF
    IF ( lsf  ) WHERE ( getvar3d /= zspval ) getvar3d = getvar3d * zsf
    IF ( lao  ) WHERE ( getvar3d /= zspval ) getvar3d = getvar3d + zao
    IF ( llog ) WHERE ( getvar3d /= zspval ) getvar3d = 10**getvar3d
    istatus = NF90_CLOSE(incid)
    CALL toc(dtime_read)
  END FUNCTION getvar3d

  SUBROUTINE readvar3d(cdfile, cdvar, ptab, ktime, ldiom, kimin, kjmin, kkmin)
    !! getvar3d without the function result: the (kpi,kpj,kpk) block of one record, window origin (kimin,kjmin,kkmin), is
    !! read straight into ptab in ONE call (no copy of hundreds of MB, none of the per-level open / inquire / close of a
    !! getvar loop), same unpacking.  A (x,y,t) variable gives its one slab; a window that crosses the E-W boundary is
    !! read level by level through getvar, which knows how to wrap.  Not in the reference's cdfio: used by the programs
    !! of this build only.
    CHARACTER(LEN=*),          INTENT(in)  :: cdfile, cdvar
    REAL(KIND=4), CONTIGUOUS,  INTENT(out) :: ptab(:,:,:)
    INTEGER(KIND=4), OPTIONAL, INTENT(in)  :: ktime, kimin, kjmin, kkmin
    LOGICAL,         OPTIONAL, INTENT(in)  :: ldiom
    INTEGER(KIND=4) :: incid, id_var, istatus, itime, iimin, ijmin, ikmin, jk, indim, istart(4), icount(4)
    REAL(KIND=4) :: zsf, zao, zspval
    LOGICAL :: llog, lsf, lao, lliom
    CHARACTER(LEN=256) :: clvar
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    iimin = 1 ; IF ( PRESENT(kimin) ) iimin = kimin
    ijmin = 1 ; IF ( PRESENT(kjmin) ) ijmin = kjmin
    ikmin = 1 ; IF ( PRESENT(kkmin) ) ikmin = kkmin
    lliom = .FALSE. ; IF ( PRESENT(ldiom) ) lliom = ldiom
    clvar = cdvar
    IF ( lliom ) clvar = iom_name(cdvar)
    IF ( PRESENT(kimin) ) THEN
       IF ( iimin + SIZE(ptab,1) - 1 > getdim(cdfile, cn_x, ldexact=.TRUE.) ) THEN       ! wraps: getvar's business
          DO jk = 1, SIZE(ptab,3)
             ptab(:,:,jk) = getvar(cdfile, cdvar, jk + ikmin - 1, SIZE(ptab,1), SIZE(ptab,2), kimin=iimin, kjmin=ijmin, &
                  &                ktime=itime, ldiom=lliom)
          END DO
          RETURN
       ENDIF
    ENDIF
    CALL tic()
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    CALL ERR_HDL(NF90_INQ_VARID(incid, clvar, id_var))
    CALL find_spval(incid, id_var, zspval)
    CALL packing(incid, id_var, zsf, zao, llog, lsf, lao)
    istart = (/ iimin, ijmin, 1, 1 /) ; icount = (/ SIZE(ptab,1), SIZE(ptab,2), 1, 1 /)
    CALL level_start(incid, id_var, ikmin, itime, istart)                    ! (x,y,t): istart(3) = the record
    istatus = NF90_INQUIRE_VARIABLE(incid, id_var, ndims=indim)
    IF ( indim == 4 ) icount(3) = SIZE(ptab,3)                                ! (x,y,z,t): all the levels at once
    istatus = NF90_GET_VAR(incid, id_var, ptab, start=istart, count=icount)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in readvar3d for ', TRIM(clvar)
       CALL ERR_HDL(istatus)
    ENDIF
    IF ( lsf  ) WHERE ( ptab /= zspval ) ptab = ptab * zsf
    IF ( lao  ) WHERE ( ptab /= zspval ) ptab = ptab + zao
    IF ( llog ) WHERE ( ptab /= zspval ) ptab = 10**ptab
    istatus = NF90_CLOSE(incid)
    CALL toc(dtime_read)
  END SUBROUTINE readvar3d

  SUBROUTINE readvar3d_i1(cdfile, cdvar, ktab, ktime)
    !! a whole 3-D mask as bytes (values converted by the backend when the file holds another type)
    CHARACTER(LEN=*),          INTENT(in)  :: cdfile, cdvar
    INTEGER(KIND=1), CONTIGUOUS, INTENT(out) :: ktab(:,:,:)
    INTEGER(KIND=4), OPTIONAL, INTENT(in)  :: ktime
    INTEGER(KIND=4) :: incid, id_var, istatus, itime
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    CALL tic()
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    CALL ERR_HDL(NF90_INQ_VARID(incid, cdvar, id_var))
    istatus = NF90_GET_VAR(incid, id_var, ktab, start=(/ 1, 1, 1, itime /), count=(/ SIZE(ktab,1), SIZE(ktab,2), SIZE(ktab,3), 1 /))
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in readvar3d for ', TRIM(cdvar)
       CALL ERR_HDL(istatus)
    ENDIF
    istatus = NF90_CLOSE(incid)
    CALL toc(dtime_read)
  END SUBROUTINE readvar3d_i1

  FUNCTION getvar1d(cdfile, cdvar, kk, kstatus)
    !! 1-D variable as r8 (time axes); 99999999999.d0 and kstatus /= 0 when absent (src/cdfio.F90:2151-2186)
    CHARACTER(LEN=*),           INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),            INTENT(in) :: kk
    INTEGER(KIND=4), OPTIONAL, INTENT(out) :: kstatus
    REAL(KIND=8), DIMENSION(kk) :: getvar1d
    INTEGER(KIND=4) :: incid, id_var, istatus
    IF ( PRESENT(kstatus) ) kstatus = 0
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, id_var)
    IF ( istatus == NF90_NOERR ) THEN
       istatus = NF90_GET_VAR(incid, id_var, getvar1d, start=(/ 1 /), count=(/ kk /))
    ELSE
       PRINT *, ' getvar1d ERROR: ', TRIM(NF90_STRERROR(istatus))
       PRINT *, '                when reading ', TRIM(cdvar)
       IF ( PRESENT(kstatus) ) kstatus = istatus
       getvar1d = 99999999999.d0
    ENDIF
    istatus = NF90_CLOSE(incid)
  END FUNCTION getvar1d

  FUNCTION getvare3(cdfile, cdvar, kk)
    !! 1-D vertical metric (gdept, gdepw, e3t, e3w) from mesh_zgr whatever its layout: v2.0 stores
    !! (t,z,1,1), v3.0 'xxx_0'(t,z), v3.6 'xxx_1d'(t,z) (src/cdfio.F90:2189-2287)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),  INTENT(in) :: kk
    REAL(KIND=4), DIMENSION(kk) :: getvare3
    INTEGER(KIND=4) :: incid, id_var, istatus, icount(4)
    CHARACTER(LEN=256) :: clvar, clbase
    LOGICAL :: ll_1d
    clvar = cdvar ; clbase = ' ' ; ll_1d = .TRUE.
    IF      ( cdvar == cn_gdept  ) THEN ; clbase = 'gdept'
    ELSE IF ( cdvar == cn_gdepw  ) THEN ; clbase = 'gdepw'
    ELSE IF ( cdvar == cn_ve3t1d ) THEN ; clbase = 'e3t'
    ELSE IF ( cdvar == cn_ve3w1d ) THEN ; clbase = 'e3w'
    ELSE ; ll_1d = .FALSE.
    ENDIF
    icount = (/ 1, 1, kk, 1 /)
    IF ( ll_1d ) THEN
       SELECT CASE ( cg_zgr_ver )
       CASE ( 'v2.0' ) ; clvar = clbase
       CASE ( 'v3.0' ) ; clvar = TRIM(clbase)//'_0'  ; icount = (/ kk, 1, 1, 1 /)
       CASE ( 'v3.6' ) ; clvar = TRIM(clbase)//'_1d' ; icount = (/ kk, 1, 1, 1 /)
       END SELECT
    ENDIF
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    IF ( istatus == NF90_NOERR ) istatus = NF90_INQ_VARID(incid, clvar, id_var)
    IF ( istatus == NF90_NOERR ) istatus = NF90_GET_VAR(incid, id_var, getvare3, start=(/ 1, 1, 1, 1 /), count=icount)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in getvare3 for ', TRIM(cdvar)
       PRINT *, TRIM(cdfile), kk
       CALL ERR_HDL(istatus)
    ENDIF
    istatus = NF90_CLOSE(incid)
  END FUNCTION getvare3

  ! ------------------------------------------------------------------------------------------------
  INTEGER(KIND=4) FUNCTION copyatt(cdvar, kidvar, kcin, kcout)
    !! attributes of a coordinate variable: copied from the reference file when there is one
    !! (kcin /= -9999), standard ones otherwise (src/cdfio.F90:185-258)
    CHARACTER(LEN=*), INTENT(in) :: cdvar
    INTEGER(KIND=4),  INTENT(in) :: kidvar, kcin, kcout
    INTEGER(KIND=4) :: ja, idvar, iatt
    CHARACTER(LEN=256) :: clatt
    copyatt = NF90_NOERR
    IF ( kcin /= -9999 ) THEN
       IF ( NF90_INQ_VARID(kcin, cdvar, idvar) /= NF90_NOERR ) RETURN
       copyatt = NF90_INQUIRE_VARIABLE(kcin, idvar, nAtts=iatt)
       DO ja = 1, iatt
          copyatt = NF90_INQ_ATTNAME(kcin, idvar, ja, clatt)
          copyatt = NF90_COPY_ATT(kcin, idvar, TRIM(clatt), kcout, kidvar)
       END DO
       RETURN
    ENDIF
    SELECT CASE ( TRIM(cdvar) )
    CASE ( 'nav_lon', 'lon', 'x', 'longitude' )
       CALL coord_atts('degrees_east', -180., 180., 'Longitude', 'nav_model', 'Default grid')
    CASE ( 'nav_lat', 'lat', 'y', 'latitude' )
       CALL coord_atts('degrees_north', -90., 90., 'Latitude', 'nav_model', 'Default grid')
    CASE ( 'time_counter', 'time', 't' )
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'calendar',    TRIM(calendar))
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'units',       TRIM(ctime_units))
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'time_origin', TRIM(ctime_origin))
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'title',       'Time')
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'long_name',   'Time axis')
    CASE ( 'deptht', 'depthu', 'depthv', 'depthw', 'dep', 'gdept' )
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'units', 'm')
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'positive', 'unknown')
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'valid_min', 0.)
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'valid_max', 5875.)
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'title', TRIM(cdvar))
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'long_name', 'Vertical Levels')
    CASE ( 'sigma', 'levels' )
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'units', 'kg/m3')
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'positive', 'unknown')
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'valid_min', 0.)
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'valid_max', 40.)
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'title', TRIM(cdvar))
       copyatt = NF90_PUT_ATT(kcout, kidvar, 'long_name', 'Sigma bin limits')
    END SELECT
  CONTAINS
    SUBROUTINE coord_atts(cdunits, pmin, pmax, cdlong, cdextra, cdextraval)
      CHARACTER(LEN=*), INTENT(in) :: cdunits, cdlong, cdextra, cdextraval
      REAL(KIND=4), INTENT(in) :: pmin, pmax
      copyatt = NF90_PUT_ATT(kcout, kidvar, 'units', cdunits)
      copyatt = NF90_PUT_ATT(kcout, kidvar, 'valid_min', pmin)
      copyatt = NF90_PUT_ATT(kcout, kidvar, 'valid_max', pmax)
      copyatt = NF90_PUT_ATT(kcout, kidvar, 'long_name', cdlong)
      copyatt = NF90_PUT_ATT(kcout, kidvar, cdextra, cdextraval)
    END SUBROUTINE coord_atts
  END FUNCTION copyatt

  INTEGER(KIND=4) FUNCTION create(cdfile, cdfilref, kx, ky, kz, cdep, cdepvar, cdlonvar, cdlatvar, ld_xycoo, ld_nc4)
    !! New 64-bit-offset file with dims x, y, [depth], time_counter(UNLIMITED) and the coordinate
    !! variables nav_lon, nav_lat, [depth], time_counter; returns its ncid (src/cdfio.F90:260-368).
    !! ld_nc4: a NetCDF-4 file (src/cdfio.F90:291-299) through the HDF5 plug-in of the bundled backend; where no
    !! HDF5 library is found the file is classic, as with a reference built without key_netcdf4.
    CHARACTER(LEN=*),           INTENT(in) :: cdfile, cdfilref
    INTEGER(KIND=4),            INTENT(in) :: kx, ky, kz
    CHARACTER(LEN=*), OPTIONAL, INTENT(in) :: cdep, cdepvar, cdlonvar, cdlatvar
    LOGICAL,          OPTIONAL, INTENT(in) :: ld_xycoo, ld_nc4
    INTEGER(KIND=4) :: istatus, icout, incid, idum, icmode
    CHARACTER(LEN=256) :: cldep, cldepref, cldepvar
    LOGICAL :: ll_xycoo
    ll_xycoo = .TRUE. ; IF ( PRESENT(ld_xycoo) ) ll_xycoo = ld_xycoo
    icmode = IOR(NF90_CLOBBER, NF90_64BIT_OFFSET)
    ! a slab of more than 2^32-4 bytes per record does not fit the 64-bit-offset format: use CDF-5
    IF ( REAL(kx, 8) * REAL(ky, 8) * REAL(MAX(kz, 1), 8) * 8.d0 > 4.0d9 ) icmode = IOR(NF90_CLOBBER, NF90_64BIT_DATA)
    IF ( PRESENT(ld_nc4) ) THEN
       IF ( ld_nc4 ) icmode = IOR(NF90_CLOBBER, NF90_NETCDF4)
    ENDIF
    CALL ERR_HDL(NF90_CREATE(cdfile, icmode, icout))
    istatus = NF90_DEF_DIM(icout, TRIM(cn_x), kx, nid_x)
    istatus = NF90_DEF_DIM(icout, TRIM(cn_y), ky, nid_y)
    IF ( kz /= 0 ) THEN                                  ! name of the depth dimension: from the reference file
       IF ( PRESENT(cdep) ) THEN
          cldep = cdep
          idum = getdim(cdfilref, cldep, cldepref)
          IF ( cldepref == 'unknown' ) cldepref = cdep
       ELSE
          idum = getdim(cdfilref, cn_z, cldep)
          cldepref = cldep
       ENDIF
       cldepvar = cldep
       IF ( PRESENT(cdepvar) ) cldepvar = cdepvar
       istatus = NF90_DEF_DIM(icout, TRIM(cldep), kz, nid_z)
    ENDIF
    istatus = NF90_DEF_DIM(icout, TRIM(cn_t), NF90_UNLIMITED, nid_t)
    incid = -9999
    IF ( TRIM(cdfilref) /= 'none' ) istatus = NF90_OPEN(cdfilref, NF90_NOWRITE, incid)
    IF ( ll_xycoo ) THEN
       istatus = NF90_DEF_VAR(icout, TRIM(cn_vlon2d), NF90_FLOAT, (/ nid_x, nid_y /), nid_lon)
       istatus = copyatt(cn_vlon2d, nid_lon, incid, icout)
       istatus = NF90_DEF_VAR(icout, TRIM(cn_vlat2d), NF90_FLOAT, (/ nid_x, nid_y /), nid_lat)
       istatus = copyatt(cn_vlat2d, nid_lat, incid, icout)
    ENDIF
    IF ( PRESENT(cdlonvar) ) istatus = NF90_DEF_VAR(icout, cdlonvar, NF90_FLOAT, (/ nid_x /), nid_lon1d)
    IF ( PRESENT(cdlatvar) ) istatus = NF90_DEF_VAR(icout, cdlatvar, NF90_FLOAT, (/ nid_y /), nid_lat1d)
    IF ( kz /= 0 ) THEN
       istatus = NF90_DEF_VAR(icout, TRIM(cldepvar), NF90_FLOAT, (/ nid_z /), nid_dep)
       istatus = copyatt(TRIM(cldepvar), nid_dep, incid, icout)
    ENDIF
    istatus = NF90_DEF_VAR(icout, TRIM(cn_vtimec), NF90_DOUBLE, (/ nid_t /), nid_tim)
    istatus = copyatt(cn_vtimec, nid_tim, incid, icout)
    istatus = NF90_PUT_ATT(icout, NF90_GLOBAL, 'start_date', nstart_date)
    istatus = NF90_PUT_ATT(icout, NF90_GLOBAL, 'output_frequency', TRIM(cfreq))
    istatus = NF90_PUT_ATT(icout, NF90_GLOBAL, 'CONFIG', TRIM(config))
    istatus = NF90_PUT_ATT(icout, NF90_GLOBAL, 'CASE', TRIM(ccase))
    IF ( incid /= -9999 ) istatus = NF90_CLOSE(incid)
    create = icout
  END FUNCTION create

  INTEGER(KIND=4) FUNCTION createvar(kout, sdtyvar, kvar, kpk, kidvo, cdglobal, ld_nc4)
    !! Defines the kvar output variables ('none' skipped) and leaves define mode.  kpk == 1: (x,y,t);
    !! kpk > 1: (x,y,z,t); kpk < 0: (t).  An r4 variable with a scale factor or offset is stored as
    !! short (src/cdfio.F90:371-459).
    INTEGER(KIND=4),                  INTENT(in   ) :: kout, kvar
    TYPE(variable), DIMENSION(kvar),  INTENT(inout) :: sdtyvar
    INTEGER(KIND=4), DIMENSION(kvar), INTENT(in   ) :: kpk
    INTEGER(KIND=4), DIMENSION(kvar), INTENT(out  ) :: kidvo
    CHARACTER(LEN=*), OPTIONAL,       INTENT(in   ) :: cdglobal
    LOGICAL,          OPTIONAL,       INTENT(in   ) :: ld_nc4
    INTEGER(KIND=4) :: jv, idims, istatus, iidims(4), itype
    LOGICAL :: ll_nc4
    ll_nc4 = .FALSE. ; IF ( PRESENT(ld_nc4) ) ll_nc4 = ld_nc4
    createvar = NF90_NOERR
    DO jv = 1, kvar
       IF ( sdtyvar(jv)%cname == 'none' ) CYCLE
       IF ( kpk(jv) == 1 ) THEN
          idims = 3 ; iidims(1:3) = (/ nid_x, nid_y, nid_t /)
       ELSE IF ( kpk(jv) > 1 ) THEN
          idims = 4 ; iidims(1:4) = (/ nid_x, nid_y, nid_z, nid_t /)
       ELSE IF ( kpk(jv) < 0 ) THEN
          idims = 1 ; iidims(1) = nid_t
       ELSE
          PRINT *, ' ERROR: ipk = ', kpk(jv), jv, sdtyvar(jv)%cname
          STOP 98
       ENDIF
       SELECT CASE ( sdtyvar(jv)%cprecision )
       CASE ( 'r8' ) ; itype = NF90_DOUBLE
       CASE ( 'i2' ) ; itype = NF90_SHORT
       CASE ( 'by' ) ; itype = NF90_BYTE
       CASE DEFAULT
          itype = NF90_FLOAT
          IF ( sdtyvar(jv)%scale_factor /= 1. .OR. sdtyvar(jv)%add_offset /= 0. ) THEN
             itype = NF90_SHORT ; sdtyvar(jv)%cprecision = 'i2'
          ENDIF
       END SELECT
       IF ( ll_nc4 ) THEN                                  ! chunked and deflated (src/cdfio.F90:438-441)
          CALL ERR_HDL(NF90_DEF_VAR(kout, TRIM(sdtyvar(jv)%cname), itype, iidims(1:idims), kidvo(jv), &
               &                    chunksizes=sdtyvar(jv)%ichunk(1:idims), deflate_level=1))
       ELSE
          CALL ERR_HDL(NF90_DEF_VAR(kout, TRIM(sdtyvar(jv)%cname), itype, iidims(1:idims), kidvo(jv)))
       ENDIF
       createvar = putatt(sdtyvar(jv), kout, kidvo(jv), cdglobal=cdglobal)
    END DO
    istatus = NF90_ENDDEF(kout)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Error in CreateVar :', TRIM(NF90_STRERROR(istatus))
       STOP 98
    ENDIF
  END FUNCTION createvar

  INTEGER(KIND=4) FUNCTION putatt(sdtyvar, kout, kid, cdglobal)
    !! the 12 standard attributes of an output variable, in the reference's order
    !! (src/cdfio.F90:629-682); any failure is fatal
    TYPE(variable),             INTENT(in) :: sdtyvar
    INTEGER(KIND=4),            INTENT(in) :: kout, kid
    CHARACTER(LEN=*), OPTIONAL, INTENT(in) :: cdglobal
    CALL must(NF90_PUT_ATT(kout, kid, 'units', TRIM(sdtyvar%cunits)), 'units')
    SELECT CASE ( sdtyvar%cprecision )                   ! the missing value has the variable's type
    CASE ( 'r8' ) ; putatt = NF90_PUT_ATT(kout, kid, TRIM(cn_missing_value), REAL(sdtyvar%rmissing_value, 8))
    CASE ( 'i2' ) ; putatt = NF90_PUT_ATT(kout, kid, TRIM(cn_missing_value), INT(sdtyvar%rmissing_value, 2))
    CASE ( 'by' ) ; putatt = NF90_PUT_ATT(kout, kid, TRIM(cn_missing_value), INT(sdtyvar%rmissing_value, 1))
    CASE DEFAULT  ; putatt = NF90_PUT_ATT(kout, kid, TRIM(cn_missing_value), REAL(sdtyvar%rmissing_value, 4))
    END SELECT
    CALL must(putatt, 'missing value')
    CALL must(NF90_PUT_ATT(kout, kid, 'valid_min', sdtyvar%valid_min), 'valid_min')
    CALL must(NF90_PUT_ATT(kout, kid, 'valid_max', sdtyvar%valid_max), 'valid_max')
    CALL must(NF90_PUT_ATT(kout, kid, 'long_name', TRIM(sdtyvar%clong_name)), 'long_name')
    CALL must(NF90_PUT_ATT(kout, kid, 'short_name', TRIM(sdtyvar%cshort_name)), 'short_name')
    CALL must(NF90_PUT_ATT(kout, kid, 'iweight', sdtyvar%iwght), 'iweight')
    CALL must(NF90_PUT_ATT(kout, kid, 'online_operation', TRIM(sdtyvar%conline_operation)), 'online_operation')
    CALL must(NF90_PUT_ATT(kout, kid, 'axis', TRIM(sdtyvar%caxis)), 'axis')
    CALL must(NF90_PUT_ATT(kout, kid, 'scale_factor', sdtyvar%scale_factor), 'scale_factor')
    CALL must(NF90_PUT_ATT(kout, kid, 'add_offset', sdtyvar%add_offset), 'add_offset')
    CALL must(NF90_PUT_ATT(kout, kid, 'savelog10', sdtyvar%savelog10), 'savelog10')
    IF ( PRESENT(cdglobal) ) CALL must(NF90_PUT_ATT(kout, NF90_GLOBAL, 'history', TRIM(cdglobal)), 'history')
    putatt = NF90_NOERR
  CONTAINS
    SUBROUTINE must(kstatus, cdwhat)
      INTEGER(KIND=4), INTENT(in) :: kstatus
      CHARACTER(LEN=*), INTENT(in) :: cdwhat
      IF ( kstatus /= NF90_NOERR ) THEN
         PRINT *, NF90_STRERROR(kstatus) ; PRINT *, ' putatt for ', cdwhat ; STOP 98
      ENDIF
    END SUBROUTINE must
  END FUNCTION putatt

  INTEGER(KIND=4) FUNCTION putheadervar(kout, cdfile, kpi, kpj, kpk, pnavlon, pnavlat, pdep, cdep, ld_xycoo)
    !! fills nav_lon / nav_lat / depth of the output file from the arrays given, else from cdfile
    !! (2-D, then 1-D lon/lat, then zeros; depth under any of its usual names) (src/cdfio.F90:2290-2407)
    INTEGER(KIND=4),                            INTENT(in) :: kout, kpi, kpj, kpk
    CHARACTER(LEN=*),                           INTENT(in) :: cdfile
    REAL(KIND=4), OPTIONAL, DIMENSION(kpi,kpj), INTENT(in) :: pnavlon, pnavlat
    REAL(KIND=4), OPTIONAL, DIMENSION(kpk),     INTENT(in) :: pdep
    CHARACTER(LEN=*), OPTIONAL,                 INTENT(in) :: cdep
    LOGICAL,          OPTIONAL,                 INTENT(in) :: ld_xycoo
    REAL(KIND=4), ALLOCATABLE :: z2d(:,:)
    REAL(KIND=4) :: z1d(MAX(kpk, 1))
    CHARACTER(LEN=256) :: cldept(6)
    INTEGER(KIND=4) :: istatus, idep, jj
    LOGICAL :: ll_xycoo
    ll_xycoo = .TRUE. ; IF ( PRESENT(ld_xycoo) ) ll_xycoo = ld_xycoo
    istatus = NF90_NOERR
    IF ( ll_xycoo ) THEN
       ALLOCATE( z2d(kpi,kpj) )
       IF ( PRESENT(pnavlon) ) THEN ; z2d = pnavlon
       ELSE ; CALL horizontal_coordinate(cn_vlon2d, cn_vlon1d, .TRUE.)
       ENDIF
       istatus = putvar(kout, nid_lon, z2d, 1, kpi, kpj)
       IF ( PRESENT(pnavlat) ) THEN ; z2d = pnavlat
       ELSE ; CALL horizontal_coordinate(cn_vlat2d, cn_vlat1d, .FALSE.)
       ENDIF
       istatus = putvar(kout, nid_lat, z2d, 1, kpi, kpj)
       DEALLOCATE( z2d )
    ENDIF
    IF ( kpk /= 0 ) THEN
       IF ( PRESENT(pdep) ) THEN
          z1d(1:kpk) = pdep
       ELSE
          idep = -1
          IF ( PRESENT(cdep) ) z1d(1:kpk) = getvar1d(cdfile, cdep, kpk, idep)
          IF ( idep /= NF90_NOERR ) THEN
             cldept = (/ cn_vdeptht, cn_vdepthu, cn_vdepthv, cn_vdepthw, 'nav_lev'//REPEAT(' ', 249), 'z'//REPEAT(' ', 255) /)
             DO jj = 1, SIZE(cldept)
                IF ( .NOT. chkvar(cdfile, cldept(jj), .FALSE.) ) THEN
                   z1d(1:kpk) = getvar1d(cdfile, cldept(jj), kpk, idep)
                   IF ( idep == NF90_NOERR ) EXIT
                ENDIF
             END DO
             IF ( idep /= NF90_NOERR ) THEN
                PRINT *, ' No depth variable found in ', TRIM(cdfile)
                STOP 98
             ENDIF
          ENDIF
       ENDIF
       istatus = putvar1d(kout, z1d(1:kpk), kpk, 'D')
    ENDIF
    putheadervar = istatus
  CONTAINS
    SUBROUTINE horizontal_coordinate(cd2d, cd1d, ldlon)
      CHARACTER(LEN=*), INTENT(in) :: cd2d, cd1d
      LOGICAL, INTENT(in) :: ldlon
      INTEGER(KIND=4) :: ji, ist
      IF ( .NOT. chkvar(cdfile, cd2d, .FALSE.) ) THEN
         z2d = getvar(cdfile, cd2d, 1, kpi, kpj)
      ELSE IF ( .NOT. chkvar(cdfile, cd1d, .FALSE.) ) THEN
         IF ( ldlon ) THEN
            z2d(:,1) = REAL(getvar1d(cdfile, cd1d, kpi, ist), 4)
            DO ji = 2, kpj ; z2d(:,ji) = z2d(:,1) ; END DO
         ELSE
            z2d(1,:) = REAL(getvar1d(cdfile, cd1d, kpj, ist), 4)
            DO ji = 2, kpi ; z2d(ji,:) = z2d(1,:) ; END DO
         ENDIF
      ELSE
         PRINT *, '... dummy value used!'
         z2d = 0.
      ENDIF
    END SUBROUTINE horizontal_coordinate
  END FUNCTION putheadervar

  SUBROUTINE slab_start(kout, kid, klev, ktime, kstart)
    ! where a (level, time) slab of an output variable goes: the last dimension is the record one,
    ! a 4-D variable also has a level (src/cdfio.F90:2437-2441)
    INTEGER(KIND=4), INTENT(in) :: kout, kid, klev, ktime
    INTEGER(KIND=4), INTENT(out) :: kstart(4)
    INTEGER(KIND=4) :: istatus, iunlim, inb, idd(4)
    idd = 0 ; kstart = 1
    istatus = NF90_INQUIRE(kout, unlimitedDimId=iunlim)
    istatus = NF90_INQUIRE_VARIABLE(kout, kid, ndims=inb, dimids=idd)
    IF ( inb >= 1 .AND. inb <= 4 ) THEN
       IF ( idd(inb) == iunlim ) kstart(inb) = ktime
    ENDIF
    IF ( inb == 4 ) kstart(3) = klev
  END SUBROUTINE slab_start

  INTEGER(KIND=4) FUNCTION putvarr8(kout, kid, ptab, klev, kpi, kpj, ktime, kwght)
    !! r8 slab (converted by NetCDF to the variable's type) (src/cdfio.F90:2410-2451)
    INTEGER(KIND=4),                  INTENT(in) :: kout, kid, klev, kpi, kpj
    REAL(KIND=8), DIMENSION(kpi,kpj), INTENT(in) :: ptab
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: ktime, kwght
    INTEGER(KIND=4) :: itime, istart(4)
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    CALL tic()
    CALL slab_start(kout, kid, klev, itime, istart)
    putvarr8 = NF90_PUT_VAR(kout, kid, ptab, start=istart, count=(/ kpi, kpj, 1, 1 /))
    IF ( PRESENT(kwght) ) putvarr8 = NF90_PUT_ATT(kout, kid, 'iweight', kwght)
    CALL toc(dtime_write)
  END FUNCTION putvarr8

  INTEGER(KIND=4) FUNCTION putvarr4(kout, kid, ptab, klev, kpi, kpj, ktime, kwght)
    INTEGER(KIND=4),                  INTENT(in) :: kout, kid, klev, kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptab
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: ktime, kwght
    INTEGER(KIND=4) :: itime, istart(4)
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    CALL tic()
    CALL slab_start(kout, kid, klev, itime, istart)
    putvarr4 = NF90_PUT_VAR(kout, kid, ptab, start=istart, count=(/ kpi, kpj, 1, 1 /))
    IF ( PRESENT(kwght) ) putvarr4 = NF90_PUT_ATT(kout, kid, 'iweight', kwght)
    CALL toc(dtime_write)
  END FUNCTION putvarr4

  INTEGER(KIND=4) FUNCTION putvari2(kout, kid, ktab, klev, kpi, kpj, ktime, kwght)
    INTEGER(KIND=4),                     INTENT(in) :: kout, kid, klev, kpi, kpj
    INTEGER(KIND=2), DIMENSION(kpi,kpj), INTENT(in) :: ktab
    INTEGER(KIND=4), OPTIONAL,           INTENT(in) :: ktime, kwght
    INTEGER(KIND=4) :: itime, istart(4)
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    CALL tic()
    CALL slab_start(kout, kid, klev, itime, istart)
    putvari2 = NF90_PUT_VAR(kout, kid, ktab, start=istart, count=(/ kpi, kpj, 1, 1 /))
    IF ( PRESENT(kwght) ) putvari2 = NF90_PUT_ATT(kout, kid, 'iweight', kwght)
    CALL toc(dtime_write)
  END FUNCTION putvari2

  INTEGER(KIND=4) FUNCTION putvarzo(kout, kid, ptab, klev, kpi, kpj, ktime)
    !! a zonal (kpi = 1) or meridional (kpj = 1) line given as a 1-D r4 array (src/cdfio.F90:2642-2681)
    INTEGER(KIND=4),            INTENT(in) :: kout, kid, klev, kpi, kpj
    REAL(KIND=4), DIMENSION(:), INTENT(in) :: ptab
    INTEGER(KIND=4), OPTIONAL,  INTENT(in) :: ktime
    INTEGER(KIND=4) :: itime, istart(4)
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    CALL tic()
    CALL slab_start(kout, kid, klev, itime, istart)
    putvarzo = NF90_PUT_VAR(kout, kid, ptab, start=istart, count=(/ kpi, kpj, 1, 1 /))
    CALL toc(dtime_write)
  END FUNCTION putvarzo

  INTEGER(KIND=4) FUNCTION putvar0dt(kout, kid, pvalue, ktime)
    !! one value, given as a (1,1) array, at record ktime (src/cdfio.F90:2809-2848)
    INTEGER(KIND=4),              INTENT(in) :: kout, kid
    REAL(KIND=4), DIMENSION(1,1), INTENT(in) :: pvalue
    INTEGER(KIND=4), OPTIONAL,    INTENT(in) :: ktime
    INTEGER(KIND=4) :: itime, istart(4)
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    CALL slab_start(kout, kid, 1, itime, istart)
    putvar0dt = NF90_PUT_VAR(kout, kid, pvalue, start=istart, count=(/ 1, 1, 1, 1 /))
  END FUNCTION putvar0dt

  INTEGER(KIND=4) FUNCTION putvar0ds(kout, kid, pvalue, ktime)
    !! the same for a scalar argument (src/cdfio.F90:2850-2892)
    INTEGER(KIND=4),           INTENT(in) :: kout, kid
    REAL(KIND=4),              INTENT(in) :: pvalue
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: ktime
    REAL(KIND=4) :: ztab(1,1)
    ztab = pvalue
    IF ( PRESENT(ktime) ) THEN ; putvar0ds = putvar0dt(kout, kid, ztab, ktime)
    ELSE                       ; putvar0ds = putvar0dt(kout, kid, ztab)
    ENDIF
  END FUNCTION putvar0ds

  INTEGER(KIND=4) FUNCTION axis_id(cdtype)
    CHARACTER(LEN=1), INTENT(in) :: cdtype
    SELECT CASE ( cdtype )
    CASE ( 'T', 't' ) ; axis_id = nid_tim
    CASE ( 'D', 'd' ) ; axis_id = nid_dep
    CASE ( 'X', 'x' ) ; axis_id = nid_lon1d
    CASE ( 'Y', 'y' ) ; axis_id = nid_lat1d
    CASE DEFAULT ; axis_id = -1
    END SELECT
  END FUNCTION axis_id

  INTEGER(KIND=4) FUNCTION putvar1d4(kout, ptab, kk, cdtype)
    !! time ('T'), depth ('D') or 1-D lon/lat ('X','Y') axis values (src/cdfio.F90:2684-2756)
    INTEGER(KIND=4),             INTENT(in) :: kout, kk
    REAL(KIND=4), DIMENSION(kk), INTENT(in) :: ptab
    CHARACTER(LEN=1),            INTENT(in) :: cdtype
    putvar1d4 = NF90_PUT_VAR(kout, axis_id(cdtype), ptab, start=(/ 1 /), count=(/ kk /))
  END FUNCTION putvar1d4

  INTEGER(KIND=4) FUNCTION putvar1d8(kout, ddtab, kk, cdtype)
    INTEGER(KIND=4),             INTENT(in) :: kout, kk
    REAL(KIND=8), DIMENSION(kk), INTENT(in) :: ddtab
    CHARACTER(LEN=1),            INTENT(in) :: cdtype
    putvar1d8 = NF90_PUT_VAR(kout, axis_id(cdtype), ddtab, start=(/ 1 /), count=(/ kk /))
  END FUNCTION putvar1d8

  ! ------------------------------------------------------------------------------------------------------------------
  ! The rest of the reference's public list (src/cdfio.F90:173-180).  No program of the hot path calls these; they are
  ! here so that any reference program compiles against this module set unchanged.  Same argument lists, same defaults,
  ! same messages and STOP codes; reading goes through the helpers above (find_spval, packing), writing through the
  ! bundled backend's redef (attributes may be added to a file that already holds data).
  ! ------------------------------------------------------------------------------------------------------------------
  FUNCTION getvarid(cdfile, knvars)
    !! the variable ids of cdfile, in file order (src/cdfio.F90:462-485)
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4),  INTENT(in) :: knvars
    INTEGER(KIND=4), DIMENSION(knvars) :: getvarid
    INTEGER(KIND=4) :: jv, incid, istatus
    CHARACTER(LEN=256) :: clvar
    getvarid(:) = 0
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    DO jv = 1, knvars
       istatus = NF90_INQUIRE_VARIABLE(incid, jv, clvar)
       IF ( istatus == NF90_NOERR ) istatus = NF90_INQ_VARID(incid, clvar, getvarid(jv))
    END DO
    istatus = NF90_CLOSE(incid)
  END FUNCTION getvarid

  INTEGER(KIND=4) FUNCTION gettimeatt(cdfile, cdvartime, ctcalendar, cttitle, ctlong_name, ctaxis, ctunits, cttime_origin)
    !! the six attributes of the time variable, 'unknown' where absent (src/cdfio.F90:517-552)
    CHARACTER(LEN=*), INTENT(in)  :: cdfile, cdvartime
    CHARACTER(LEN=*), INTENT(out) :: ctcalendar, cttitle, ctlong_name, ctaxis, ctunits, cttime_origin
    INTEGER(KIND=4) :: istatus, incid, ivarid
    ctcalendar = 'unknown' ; cttitle = 'unknown' ; ctlong_name = 'unknown'
    ctaxis = 'unknown' ; ctunits = 'unknown' ; cttime_origin = 'unknown'
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvartime, ivarid)
    istatus = NF90_GET_ATT(incid, ivarid, 'calendar',    ctcalendar)
    istatus = NF90_GET_ATT(incid, ivarid, 'title',       cttitle)
    istatus = NF90_GET_ATT(incid, ivarid, 'long_name',   ctlong_name)
    istatus = NF90_GET_ATT(incid, ivarid, 'axis',        ctaxis)
    istatus = NF90_GET_ATT(incid, ivarid, 'units',       ctunits)
    istatus = NF90_GET_ATT(incid, ivarid, 'time_origin', cttime_origin)
    gettimeatt = istatus                                   ! the status of the last attribute, as in the reference
    istatus = NF90_CLOSE(incid)
  END FUNCTION gettimeatt

  INTEGER(KIND=4) FUNCTION puttimeatt(kout, cdvartime, ctcalendar, cttitle, ctlong_name, ctaxis, ctunits, cttime_origin)
    !! the same six attributes onto the time variable of an output file in data mode (src/cdfio.F90:554-593)
    INTEGER(KIND=4),    INTENT(in) :: kout
    CHARACTER(LEN=20),  INTENT(in) :: cdvartime
    CHARACTER(LEN=256), INTENT(in) :: ctcalendar, cttitle, ctlong_name, ctaxis, ctunits, cttime_origin
    INTEGER(KIND=4) :: ivarid
    puttimeatt = NF90_INQ_VARID(kout, cdvartime, ivarid)
    IF ( puttimeatt /= 0 ) THEN
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
      puttimeatt = NF90_PUT_ATT(kout, ivarid, cdatt, cdval)
      IF ( puttimeatt /= 0 ) THEN
         PRINT *, NF90_STRERROR(puttimeatt) ; PRINT *, ' in puttimeatt ', cdatt ; STOP 98
      ENDIF
    END SUBROUTINE one
  END FUNCTION puttimeatt

  INTEGER(KIND=4) FUNCTION cvaratt(cdfile, cdvar, cdunits, pmissing_value, cdlong_name, cdshort_name)
    !! change units, the missing-value attribute (under whichever name the file uses), long_name and short_name of a
    !! variable of an existing file (src/cdfio.F90:595-627)
    CHARACTER(LEN=256), INTENT(in) :: cdfile, cdvar, cdunits, cdlong_name, cdshort_name
    REAL(KIND=4),       INTENT(in) :: pmissing_value
    INTEGER(KIND=4) :: istatus, incid, ivarid
    REAL(KIND=4) :: zspval
    CHARACTER(LEN=256) :: clmissing
    zspval = getspval(cdfile, cdvar, clmissing)              ! (read before the file is opened for writing)
    istatus = NF90_OPEN(cdfile, NF90_WRITE, incid)
    istatus = NF90_REDEF(incid)
    istatus = NF90_INQ_VARID(incid, cdvar, ivarid)
    istatus = NF90_PUT_ATT(incid, ivarid, 'units',      cdunits)
    istatus = NF90_PUT_ATT(incid, ivarid, clmissing,    pmissing_value)
    istatus = NF90_PUT_ATT(incid, ivarid, 'long_name',  cdlong_name)
    istatus = NF90_PUT_ATT(incid, ivarid, 'short_name', cdshort_name)
    istatus = NF90_ENDDEF(incid)
    cvaratt = istatus
    istatus = NF90_CLOSE(incid)
  END FUNCTION cvaratt

  INTEGER(KIND=4) FUNCTION atted_char(cdfile, cdvar, cdatt, cdvalue)
    !! attribute editor, text value: create the attribute or overwrite one of the same type (src/cdfio.F90:790-830)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt, cdvalue
    INTEGER(KIND=4) :: incid, istatus, idvar, ityp
    istatus = NF90_OPEN(cdfile, NF90_WRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, idvar)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, NF90_STRERROR(istatus), ' in atted ( inq_varid)'
       STOP 98
    ENDIF
    istatus = NF90_INQUIRE_ATTRIBUTE(incid, idvar, cdatt, xtype=ityp)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Attribute does not exist. Create it'
    ELSE IF ( ityp /= NF90_CHAR ) THEN
       PRINT *, ' Mismatch in attribute type in atted_char'
       STOP 98
    ENDIF
    istatus = NF90_REDEF(incid)
    atted_char = NF90_PUT_ATT(incid, idvar, cdatt, cdvalue)
    istatus = NF90_CLOSE(incid)
  END FUNCTION atted_char

  INTEGER(KIND=4) FUNCTION atted_r4(cdfile, cdvar, cdatt, pvalue)
    !! attribute editor, REAL*4 value (src/cdfio.F90:833-873)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar, cdatt
    REAL(KIND=4),     INTENT(in) :: pvalue
    INTEGER(KIND=4) :: incid, istatus, idvar, ityp
    istatus = NF90_OPEN(cdfile, NF90_WRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, idvar)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, NF90_STRERROR(istatus), ' in atted ( inq_varid)'
       STOP 98
    ENDIF
    istatus = NF90_INQUIRE_ATTRIBUTE(incid, idvar, cdatt, xtype=ityp)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Attribute does not exist. Create it'
    ELSE IF ( ityp /= NF90_FLOAT ) THEN
       PRINT *, ' Mismatch in attribute type in atted_r4'
       STOP 98
    ENDIF
    istatus = NF90_REDEF(incid)
    atted_r4 = NF90_PUT_ATT(incid, idvar, cdatt, pvalue)
    istatus = NF90_CLOSE(incid)
  END FUNCTION atted_r4

  FUNCTION getdimvar(cdfile, kpk, cd_depnam)
    !! the values of the first 1-D variable that does not run along the unlimited dimension: the vertical axis of a
    !! file whose depth variable has an unknown name (src/cdfio.F90:1085-1127)
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4),  INTENT(in) :: kpk
    REAL(KIND=4), DIMENSION(kpk) :: getdimvar
    CHARACTER(LEN=*), OPTIONAL, INTENT(out) :: cd_depnam
    INTEGER(KIND=4) :: ji, incid, idims, iuldid, idimv, ivars, istatus, idimt(4)
    CHARACTER(LEN=80) :: clvar
    LOGICAL :: llfound
    clvar = 'none' ; llfound = .FALSE.
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQUIRE(incid, nDimensions=idims, unlimitedDimId=iuldid, nVariables=ivars)
    DO ji = 1, ivars
       istatus = NF90_INQUIRE_VARIABLE(incid, ji, name=clvar, ndims=idimv, dimids=idimt)
       IF ( idimv == 1 ) THEN
          IF ( idimt(1) /= iuldid ) THEN
             PRINT *, ' Found vertical variable :', TRIM(clvar)
             istatus = NF90_GET_VAR(incid, ji, getdimvar(:))
             llfound = .TRUE.
             EXIT
          ENDIF
       ENDIF
    END DO
    IF ( PRESENT(cd_depnam) ) cd_depnam = TRIM(clvar)
    IF ( .NOT. llfound ) THEN
       PRINT *, 'Sorry, no vertical dim variables inferred ...'
       getdimvar = 0.
    ENDIF
    istatus = NF90_CLOSE(incid)
  END FUNCTION getdimvar

  SUBROUTINE unpack_in_place(kncid, kvar, cdfile, cdvar, ptab, kn)
    ! scale_factor, add_offset, savelog10 applied to the kn values just read, missing values left alone -- the order
    ! matters (src/cdfio.F90:1806-1808)
    INTEGER(KIND=4),  INTENT(in) :: kncid, kvar, kn
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar
    REAL(KIND=4), INTENT(inout) :: ptab(kn)
    REAL(KIND=4) :: zsf, zao, zspval
    LOGICAL :: llog, lsf, lao
    CALL find_spval(kncid, kvar, zspval)
    CALL packing(kncid, kvar, zsf, zao, llog, lsf, lao)
    IF ( lsf  ) WHERE ( ptab /= zspval ) ptab = ptab * zsf
    IF ( lao  ) WHERE ( ptab /= zspval ) ptab = ptab + zao
    IF ( llog ) WHERE ( ptab /= zspval ) ptab = 10**ptab
  END SUBROUTINE unpack_in_place

  FUNCTION getvar3dt(cdfile, cdvar, kk, kpi, kpj, kpt, kimin, kjmin, ktmin)
    !! kpt records of level kk: (kpi,kpj,kpt) (src/cdfio.F90:1717-1820; a 3-D variable is taken as x, y, t)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj, kk, kpt
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kjmin, ktmin
    REAL(KIND=4), DIMENSION(kpi,kpj,kpt) :: getvar3dt
    INTEGER(KIND=4) :: istart(4), icount(4), incid, id_var, iid, istatus, iimin, ijmin, itmin
    iimin = 1 ; IF ( PRESENT(kimin) ) iimin = kimin
    ijmin = 1 ; IF ( PRESENT(kjmin) ) ijmin = kjmin
    itmin = 1 ; IF ( PRESENT(ktmin) ) itmin = ktmin
    PRINT *, ' GETVAR3DT '
    PRINT *, '  ', TRIM(cdfile)
    PRINT *, '  ', TRIM(cdvar)
    PRINT *, '    KPI KPJ KPT, KK ', kpi, kpj, kpt, kk
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    CALL ERR_HDL(NF90_INQ_VARID(incid, cdvar, id_var))
    istatus = NF90_INQUIRE_VARIABLE(incid, id_var, ndims=iid)
    istart = (/ iimin, ijmin, kk, itmin /) ; icount = (/ kpi, kpj, 1, kpt /)
    IF ( iid == 3 ) THEN
       istart = (/ iimin, ijmin, itmin, 1 /) ; icount = (/ kpi, kpj, kpt, 1 /)
    ENDIF
    istatus = NF90_GET_VAR(incid, id_var, getvar3dt, start=istart, count=icount)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in getvar3dt for ', TRIM(cdvar)
       CALL ERR_HDL(istatus)
       STOP 98
    ENDIF
    CALL unpack_in_place(incid, id_var, cdfile, cdvar, getvar3dt, kpi * kpj * kpt)
    istatus = NF90_CLOSE(incid)
  END FUNCTION getvar3dt

  FUNCTION getvar4d(cdfile, cdvar, kpi, kpj, kpz, kpt, kimin, kjmin, kkmin, ktmin)
    !! a (kpi,kpj,kpz,kpt) block (src/cdfio.F90:1822-1931)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj, kpz, kpt
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kjmin, kkmin, ktmin
    REAL(KIND=4), DIMENSION(kpi,kpj,kpz,kpt) :: getvar4d
    INTEGER(KIND=4) :: istart(4), icount(4), incid, id_var, iid, istatus, iimin, ijmin, ikmin, itmin
    iimin = 1 ; IF ( PRESENT(kimin) ) iimin = kimin
    ijmin = 1 ; IF ( PRESENT(kjmin) ) ijmin = kjmin
    ikmin = 1 ; IF ( PRESENT(kkmin) ) ikmin = kkmin
    itmin = 1 ; IF ( PRESENT(ktmin) ) itmin = ktmin
    PRINT *, ' GETVAR4D '
    PRINT *, '  ', TRIM(cdfile)
    PRINT *, '  ', TRIM(cdvar)
    PRINT *, '    KPI KPJ KPZ, KPT ', kpi, kpj, kpz, kpt
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    CALL ERR_HDL(NF90_INQ_VARID(incid, cdvar, id_var))
    istatus = NF90_INQUIRE_VARIABLE(incid, id_var, ndims=iid)
    istart = (/ iimin, ijmin, ikmin, itmin /) ; icount = (/ kpi, kpj, kpz, kpt /)
    IF ( iid == 3 ) THEN
       istart = (/ iimin, ijmin, itmin, 1 /) ; icount = (/ kpi, kpj, kpt, 1 /)
    ENDIF
    istatus = NF90_GET_VAR(incid, id_var, getvar4d, start=istart, count=icount)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in getvar4d for ', TRIM(cdvar)
       CALL ERR_HDL(istatus)
       STOP 98
    ENDIF
    CALL unpack_in_place(incid, id_var, cdfile, cdvar, getvar4d, kpi * kpj * kpz * kpt)
    istatus = NF90_CLOSE(incid)
  END FUNCTION getvar4d

  FUNCTION getvarxz(cdfile, cdvar, kj, kpi, kpz, kimin, kkmin, ktime)
    !! the x-z section at row kj (src/cdfio.F90:1933-2044; cn_ve3v is mapped to the mesh_zgr name of the layout; a file
    !! without a y dimension is read as x, z, t)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kj, kpi, kpz
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kimin, kkmin, ktime
    REAL(KIND=4), DIMENSION(kpi,kpz) :: getvarxz
    INTEGER(KIND=4) :: istart(4), icount(4), incid, id_var, istatus, imin, kmin, itime, idum
    CHARACTER(LEN=256) :: clvar
    imin = 1  ; IF ( PRESENT(kimin) ) imin  = kimin
    kmin = 1  ; IF ( PRESENT(kkmin) ) kmin  = kkmin
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    clvar = cdvar
    IF ( clvar == cn_ve3v ) clvar = iom_name(cdvar)
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    CALL ERR_HDL(NF90_INQ_VARID(incid, clvar, id_var))
    IF ( NF90_INQ_DIMID(incid, 'y', idum) == NF90_NOERR ) THEN
       istart = (/ imin, kj, kmin, itime /) ; icount = (/ kpi, 1, kpz, 1 /)
    ELSE
       istart = (/ imin, kmin, itime, 1 /)  ; icount = (/ kpi, kpz, 1, 1 /)
    ENDIF
    istatus = NF90_GET_VAR(incid, id_var, getvarxz, start=istart, count=icount)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in getvarxz for ', TRIM(clvar)
       CALL ERR_HDL(istatus)
       STOP 98
    ENDIF
    CALL unpack_in_place(incid, id_var, cdfile, clvar, getvarxz, kpi * kpz)
    istatus = NF90_CLOSE(incid)
  END FUNCTION getvarxz

  FUNCTION getvaryz(cdfile, cdvar, ki, kpj, kpz, kjmin, kkmin, ktime)
    !! the y-z section at column ki (src/cdfio.F90:2046-2149; cn_ve3u mapped like getvarxz's cn_ve3v; a file without an
    !! x dimension is read as y, z, t)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: ki, kpj, kpz
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: kjmin, kkmin, ktime
    REAL(KIND=4), DIMENSION(kpj,kpz) :: getvaryz
    INTEGER(KIND=4) :: istart(4), icount(4), incid, id_var, istatus, jmin, kmin, itime, idum
    CHARACTER(LEN=256) :: clvar
    jmin = 1  ; IF ( PRESENT(kjmin) ) jmin  = kjmin
    kmin = 1  ; IF ( PRESENT(kkmin) ) kmin  = kkmin
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    clvar = cdvar
    IF ( clvar == cn_ve3u ) clvar = iom_name(cdvar)
    CALL ERR_HDL(NF90_OPEN(cdfile, NF90_NOWRITE, incid))
    CALL ERR_HDL(NF90_INQ_VARID(incid, clvar, id_var))
    IF ( NF90_INQ_DIMID(incid, 'x', idum) == NF90_NOERR ) THEN
       istart = (/ ki, jmin, kmin, itime /) ; icount = (/ 1, kpj, kpz, 1 /)
    ELSE
       istart = (/ jmin, kmin, itime, 1 /)  ; icount = (/ kpj, kpz, 1, 1 /)
    ENDIF
    istatus = NF90_GET_VAR(incid, id_var, getvaryz, start=istart, count=icount)
    IF ( istatus /= NF90_NOERR ) THEN
       PRINT *, ' Problem in getvaryz for ', TRIM(clvar)
       CALL ERR_HDL(istatus)
       STOP 98
    ENDIF
    CALL unpack_in_place(incid, id_var, cdfile, clvar, getvaryz, kpj * kpz)
    istatus = NF90_CLOSE(incid)
  END FUNCTION getvaryz

  SUBROUTINE gettimeseries(cdfile, cdvar, kilook, kjlook, klev, ldncdf)
    !! prints the time series of cdvar at (kilook,kjlook[,klev]): days, value (src/cdfio.F90:2943-3029; the ldncdf
    !! branch only builds a file name in the reference -- "not finished" -- and does the same here)
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4),           INTENT(in) :: kilook, kjlook
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: klev
    LOGICAL,         OPTIONAL, INTENT(in) :: ldncdf
    INTEGER(KIND=4) :: jt, jk, iint, istatus, incid, id_t, id_var, indim
    REAL(KIND=4), DIMENSION(:), ALLOCATABLE :: ztime, zval
    REAL(KIND=4) :: ztmp, zao, zsf
    CHARACTER(LEN=256) :: clname
    jk = 1 ; IF ( PRESENT(klev) ) jk = klev
    zao = 0. ; zsf = 1.0
    istatus = NF90_OPEN(cdfile, NF90_NOWRITE, incid)
    istatus = NF90_INQ_DIMID(incid, cn_t, id_t)
    istatus = NF90_INQUIRE_DIMENSION(incid, id_t, len=iint)
    ALLOCATE ( ztime(iint), zval(iint) )
    istatus = NF90_INQ_VARID(incid, cn_vtimec, id_var)
    istatus = NF90_GET_VAR(incid, id_var, ztime, (/ 1 /), (/ iint /))
    istatus = NF90_INQ_VARID(incid, cdvar, id_var)
    IF ( NF90_GET_ATT(incid, id_var, 'add_offset', ztmp) == NF90_NOERR ) zao = ztmp
    IF ( NF90_GET_ATT(incid, id_var, 'scale_factor', ztmp) == NF90_NOERR ) zsf = ztmp
    istatus = NF90_INQUIRE_VARIABLE(incid, id_var, ndims=indim)
    IF ( indim == 3 ) THEN
       istatus = NF90_GET_VAR(incid, id_var, zval, (/ kilook, kjlook, 1 /), (/ 1, 1, iint /))
    ELSE IF ( indim == 4 ) THEN
       istatus = NF90_GET_VAR(incid, id_var, zval, (/ kilook, kjlook, jk, 1 /), (/ 1, 1, 1, iint /))
    ELSE
       PRINT *, '  ERROR : variable ', TRIM(cdvar), ' has ', indim, ' dimensions !. Only 3 or 4 supported'
       STOP 98
    ENDIF
    zval = zval * zsf + zao
    DO jt = 1, iint
       PRINT *, ztime(jt) / 86400., zval(jt)
    END DO
    istatus = NF90_CLOSE(incid)
    IF ( PRESENT(ldncdf) ) THEN
       IF ( ldncdf ) WRITE (clname, '("probe_",i4.4,"_",i4.4,"_",a,".nc")') kilook, kjlook, TRIM(cdvar)
    ENDIF
    DEALLOCATE ( ztime, zval )
  END SUBROUTINE gettimeseries

  INTEGER(KIND=4) FUNCTION putvare3(kout, kid, ptab, kmax, cd_vert, ktime, kwght)
    !! a vertical profile of kmax values into variable kid, record ktime (src/cdfio.F90:2501-2543; cd_vert only selects
    !! this member of the generic putvar)
    INTEGER(KIND=4),               INTENT(in) :: kout, kid, kmax
    REAL(KIND=4), DIMENSION(kmax), INTENT(in) :: ptab
    CHARACTER(LEN=*),              INTENT(in) :: cd_vert
    INTEGER(KIND=4), OPTIONAL,     INTENT(in) :: ktime, kwght
    INTEGER(KIND=4) :: istatus, itime, id_dimunlim, inbdim, istart(4), icount(4), inldim(4)
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    inldim = 0
    istatus = NF90_INQUIRE(kout, unlimitedDimId=id_dimunlim)
    istatus = NF90_INQUIRE_VARIABLE(kout, kid, ndims=inbdim, dimids=inldim)
    istart(:) = 1 ; icount(:) = 1
    icount(3) = kmax
    IF ( inldim(inbdim) == id_dimunlim ) istart(inbdim) = itime
    istatus = NF90_PUT_VAR(kout, kid, ptab, start=istart, count=icount)
    IF ( PRESENT(kwght) ) istatus = NF90_PUT_ATT(kout, kid, 'iweight', kwght)
    putvare3 = istatus
  END FUNCTION putvare3

  INTEGER(KIND=4) FUNCTION reputvarr4(cdfile, cdvar, klev, kpi, kpj, kimin, kjmin, ktime, ptab, kwght)
    !! overwrite a (kpi,kpj) window of level klev, record ktime of a variable of an EXISTING file
    !! (src/cdfio.F90:2591-2640)
    CHARACTER(LEN=*),                 INTENT(in) :: cdfile, cdvar
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: klev
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: kimin, kjmin, ktime
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptab
    INTEGER(KIND=4), OPTIONAL,        INTENT(in) :: kwght
    INTEGER(KIND=4) :: incid, id_var, id_dimunlim, inbdim, istatus, ilev, iimin, ijmin, itime, istart(4), icount(4), inldim(4)
    ilev  = 1 ; IF ( PRESENT(klev ) ) ilev  = klev
    iimin = 1 ; IF ( PRESENT(kimin) ) iimin = kimin
    ijmin = 1 ; IF ( PRESENT(kjmin) ) ijmin = kjmin
    itime = 1 ; IF ( PRESENT(ktime) ) itime = ktime
    inldim = 0
    istatus = NF90_OPEN(cdfile, NF90_WRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, id_var)
    istatus = NF90_INQUIRE(incid, unlimitedDimId=id_dimunlim)
    istatus = NF90_INQUIRE_VARIABLE(incid, id_var, ndims=inbdim, dimids=inldim)
    istart(:) = 1 ; icount(:) = 1
    istart(1) = iimin ; istart(2) = ijmin
    icount(1) = kpi   ; icount(2) = kpj
    IF ( inldim(inbdim) == id_dimunlim ) istart(inbdim) = itime
    IF ( inbdim == 4 ) istart(3) = ilev
    istatus = NF90_PUT_VAR(incid, id_var, ptab, start=istart, count=icount)
    IF ( PRESENT(kwght) ) THEN
       istatus = NF90_REDEF(incid)
       istatus = NF90_PUT_ATT(incid, id_var, 'iweight', kwght)
       istatus = NF90_ENDDEF(incid)
    ENDIF
    reputvarr4 = istatus
    istatus = NF90_CLOSE(incid)
  END FUNCTION reputvarr4

  INTEGER(KIND=4) FUNCTION reputvar1d4(cdfile, cdvar, ptab, kk)
    !! overwrite a 1-D variable of an existing file (src/cdfio.F90:2758-2781)
    CHARACTER(LEN=*),            INTENT(in) :: cdfile, cdvar
    REAL(KIND=4), DIMENSION(kk), INTENT(in) :: ptab
    INTEGER(KIND=4),             INTENT(in) :: kk
    INTEGER :: istatus, incid, id
    istatus = NF90_OPEN(cdfile, NF90_WRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, id)
    istatus = NF90_PUT_VAR(incid, id, ptab, start=(/ 1 /), count=(/ kk /))
    reputvar1d4 = istatus
    istatus = NF90_CLOSE(incid)
  END FUNCTION reputvar1d4

  INTEGER(KIND=4) FUNCTION reputvar1d8(cdfile, cdvar, ddtab, kk)
    CHARACTER(LEN=*),            INTENT(in) :: cdfile, cdvar
    REAL(KIND=8), DIMENSION(kk), INTENT(in) :: ddtab
    INTEGER(KIND=4),             INTENT(in) :: kk
    INTEGER :: istatus, incid, id
    istatus = NF90_OPEN(cdfile, NF90_WRITE, incid)
    istatus = NF90_INQ_VARID(incid, cdvar, id)
    istatus = NF90_PUT_VAR(incid, id, ddtab, start=(/ 1 /), count=(/ kk /))
    reputvar1d8 = istatus
    istatus = NF90_CLOSE(incid)
  END FUNCTION reputvar1d8

  INTEGER(KIND=4) FUNCTION closeout(kout)
    INTEGER(KIND=4), INTENT(in) :: kout
    CALL tic()
    closeout = NF90_CLOSE(kout)
    CALL toc(dtime_write)
    IF ( ntiming == 1 ) WRITE(0,'(a,f9.3,a,f9.3)') ' CDFT_TIMING cdfio (s): reads', dtime_read, '  writes', dtime_write
  END FUNCTION closeout

  INTEGER(KIND=4) FUNCTION ncopen(cdfile)
    CHARACTER(LEN=*), INTENT(in) :: cdfile
    INTEGER(KIND=4) :: istatus
    istatus = NF90_OPEN(cdfile, NF90_WRITE, ncopen)
  END FUNCTION ncopen

END MODULE cdfio
