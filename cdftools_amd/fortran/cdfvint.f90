PROGRAM cdfvint
  !!======================================================================
  !!                     ***  PROGRAM  cdfvint  ***
  !! Vertical integral of temperature (heat content) or salinity from the surface down, cumulated and
  !! written at every output depth (model levels, -GSOP or -OCCI depths).
  !!
  !! Drop-in for the reference tool (src/cdfvint.f90): same options, same <var>.nc (voheatc | vohsalt |
  !! votemper | vosaline float (t,depthw,y,x), :318-346).  The column integration (:266-311) runs on the GPU
  !! for all levels of a record at once (cdft_vint_cumul, bit-identical); the scalar bookkeeping of the level
  !! loop (depth counters, which level is written, clipping depth) is done here exactly as the reference does.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: narg, ijarg, ierr, jk, jt, it, iko, npiglo, npjglo, npk, npko, npt, ncout
  INTEGER(KIND=4) :: iimin = 0, iimax = 0, ijmin = 0, ijmax = 0, ipk(1), id_varout(1)
  INTEGER(C_INT), ALLOCATABLE :: kout(:), khit(:)
  REAL(KIND=4), PARAMETER :: pprho0 = 1020., ppcp = 4000.
  REAL(KIND=4) :: rdep1, rdep2, tol = 1.0, sclf = 1.0
  REAL(KIND=4), ALLOCATABLE, TARGET :: e3t(:,:,:), e31d(:)
  REAL(KIND=4), ALLOCATABLE :: gdepw(:), gdepo(:), zt(:,:,:), tmask(:,:,:), rlon(:,:), rlat(:,:), zout(:,:,:), zclip(:), zdiv(:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:)
  CHARACTER(LEN=256) :: cf_in = 'none', cf_out, cv_in, cv_out, cunits, clongname, cldum
  TYPE(variable) :: stypvar(1)
  TYPE(C_PTR) :: pe3, pe31d
  TYPE(C_PTR) :: pd_zt = C_NULL_PTR, pd_msk = C_NULL_PTR, pd_e3 = C_NULL_PTR, pd_e31 = C_NULL_PTR
  REAL(KIND=4), POINTER :: zt_d(:), tmask_d(:), zout_d(:)             ! device arrays of the host feed, never touched by the host
  TYPE(C_PTR) :: pd_out = C_NULL_PTR, ph_lev(2) = C_NULL_PTR          ! the result in HBM, two page-locked levels on its way to the file
  INTEGER(KIND=4) :: il, ib
  INTEGER(KIND=8) :: n38
  LOGICAL :: lfeed
  INTEGER(KIND=8) :: iclk(4), iclk_rate
  REAL(KIND=8) :: dl_ph(3) = 0.d0                                   ! CDFT_TIMING: the three steps of a level on its way out
  LOGICAL :: lfull = .FALSE., lgsop = .FALSE., locci = .FALSE., lchk, lout, ltmean = .FALSE., lsmean = .FALSE.
  LOGICAL :: lnc4 = .FALSE., lfout = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  cv_in = cn_votemper
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfvint.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-f'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_in) ; ijarg = ijarg + 1
     CASE ( '-v'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cv_in) ; ijarg = ijarg + 1
     CASE ( '-GSOP'  ) ; lgsop  = .TRUE.
     CASE ( '-OCCI'  ) ; locci  = .TRUE.
     CASE ( '-full'  ) ; lfull  = .TRUE.
     CASE ( '-tmean' ) ; ltmean = .TRUE.
     CASE ( '-smean' ) ; lsmean = .TRUE.
     CASE ( '-vvl'   ) ; lg_vvl = .TRUE.
     CASE ( '-w'     ) ; CALL geti(iimin) ; CALL geti(iimax) ; CALL geti(ijmin) ; CALL geti(ijmax)
     CASE ( '-nc4'   ) ; lnc4   = .TRUE.
     CASE ( '-o'     ) ; lfout  = .TRUE. ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out) ; ijarg = ijarg + 1
     CASE DEFAULT      ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  lchk = chkfile(cf_in)
  lchk = chkfile(cn_fmsk) .OR. lchk
  lchk = chkfile(cn_fhgr) .OR. lchk
  lchk = chkfile(cn_fzgr) .OR. lchk
  IF ( lchk ) STOP 99
  IF ( lg_vvl ) THEN
     cn_fe3t = cf_in
     cn_ve3t = cn_ve3tvvl
  ENDIF
  IF ( ltmean .AND. cv_in == cn_vosaline ) THEN
     PRINT *, ' WARNING : flag -tmean is useless with variable', TRIM(cv_in)
     ltmean = .FALSE.
  ENDIF
  IF ( lsmean .AND. cv_in == cn_votemper ) THEN
     PRINT *, ' WARNING : flag -smean is useless with variable', TRIM(cv_in)
     lsmean = .FALSE.
  ENDIF
  IF ( cv_in == cn_votemper ) THEN
     IF ( ltmean ) THEN
        cv_out = cn_votemper ; clongname = 'Mean temperature ' ; cunits = 'Deg Celsius' ; sclf = 1.
     ELSE
        cv_out = 'voheatc' ; clongname = 'Heat Content per unit area' ; cunits = '10^6 J/m2' ; sclf = pprho0*ppcp/1.e6
     ENDIF
  ELSEIF ( cv_in == cn_vosaline ) THEN
     IF ( lsmean ) THEN
        cv_out = cn_vosaline ; clongname = 'Mean salinity' ; cunits = 'psu'
     ELSE
        cv_out = 'vohsalt' ; clongname = 'Vertically Integrated Salinity' ; cunits = 'psu.m'
     ENDIF
     sclf = 1.
  ELSE
     PRINT *, '  ERROR: Variable ', TRIM(cv_in), ' not pre-defined ...'
     PRINT *, '     Accepted variables are ', TRIM(cn_votemper), ' and ', TRIM(cn_vosaline)
     STOP 99
  ENDIF
  IF ( .NOT. lfout ) cf_out = TRIM(cv_out)//'.nc'
  PRINT *, ' INPUT VARIABLE  : ', TRIM(cv_in)
  PRINT *, ' OUTPUT VARIABLE : ', TRIM(cv_out)
  PRINT *, ' OUTPUT FILE     : ', TRIM(cf_out)

  npiglo = getdim(cf_in, cn_x)
  npjglo = getdim(cf_in, cn_y)
  npk    = getdim(cf_in, cn_z)
  npt    = getdim(cf_in, cn_t)
  IF ( iimin /= 0 ) THEN ; npiglo = iimax - iimin + 1 ; ELSE ; iimin = 1
  ENDIF
  IF ( ijmin /= 0 ) THEN ; npjglo = ijmax - ijmin + 1 ; ELSE ; ijmin = 1
  ENDIF
  IF ( lgsop ) THEN ; PRINT *, ' using GSOP depths' ; npko = 7
  ENDIF
  IF ( locci ) THEN ; PRINT *, ' using OCCIPUT depths' ; npko = 3
  ENDIF
  IF ( (.NOT. lgsop) .AND. (.NOT. locci) ) THEN ; PRINT *, ' using model depths' ; npko = npk
  ENDIF
  PRINT *, ' NPIGLO = ', npiglo
  PRINT *, ' NPJGLO = ', npjglo
  PRINT *, ' NPK    = ', npk
  PRINT *, ' NPKO   = ', npko
  PRINT *, ' NPT    = ', npt

  ! large fields: mask, e3t and every record from the file to HBM through the page-locked ring (module cdffeed); the result
  ! (npko levels) stays in HBM and leaves level by level through two page-locked arrays, each queued for the output file's
  ! writing thread (putvar_behind) -- the last levels are written while the next record is read
  n38 = INT(npiglo, 8) * npjglo * npk
  lfeed = feed_wanted(4 * n38)
  ALLOCATE( dtim(npt), rlon(npiglo,npjglo), rlat(npiglo,npjglo) )
  IF ( .NOT. lfeed ) ALLOCATE( tmask(npiglo,npjglo,npk), zt(npiglo,npjglo,npk) )
  ALLOCATE( e31d(npk), gdepw(npk), gdepo(npko), kout(npk), khit(npk), zclip(npk), zdiv(npk) )
  IF ( .NOT. lfeed ) ALLOCATE( zout(npiglo,npjglo,npko) )
  IF ( .NOT. ( lfull .OR. lfeed ) ) ALLOCATE( e3t(npiglo,npjglo,npk) )
  gdepw(:) = getvare3(cn_fzgr, cn_gdepw, npk)
  e31d(:)  = getvare3(cn_fzgr, cn_ve3t1d, npk)
  IF ( lgsop ) gdepo(:) = (/ 100., 300., 500., 700., 800., 2000., 6000. /)
  IF ( locci ) gdepo(:) = (/ 700., 2000., 6000. /)
  IF ( (.NOT. lgsop) .AND. (.NOT. locci) ) THEN
     gdepo(1:npk-1) = gdepw(2:npk)
     gdepo(npk)     = 6000.
  ENDIF
  CALL getvar_to(cf_in, cn_vlon2d, 1, rlon, kimin=iimin, kjmin=ijmin)
  CALL getvar_to(cf_in, cn_vlat2d, 1, rlat, kimin=iimin, kjmin=ijmin)
  CALL CreateOutput
  PRINT *, 'OUTPUT DEPTHS ARE : ', gdepo

  IF ( lfeed ) THEN
     pd_zt = feed_device(4 * n38) ; pd_msk = feed_device(4 * n38) ; pd_out = feed_device(4_8 * npiglo * npjglo * npko)
     CALL C_F_POINTER(pd_zt, zt_d, (/ n38 /)) ; CALL C_F_POINTER(pd_msk, tmask_d, (/ n38 /))
     CALL C_F_POINTER(pd_out, zout_d, (/ INT(npiglo, 8) * npjglo * npko /))
     CALL hip_check(cdft_host_alloc(ph_lev(1), INT(4_8 * npiglo * npjglo, C_SIZE_T)), 'cdft_host_alloc')
     CALL hip_check(cdft_host_alloc(ph_lev(2), INT(4_8 * npiglo * npjglo, C_SIZE_T)), 'cdft_host_alloc')
     CALL feed_field(cn_fmsk, cn_tmask, pd_msk, npiglo, npjglo, npk, kimin=iimin, kjmin=ijmin)
     IF ( lfull ) THEN
        pd_e31 = feed_device(4_8 * npk) ; CALL feed_host(C_LOC(e31d), 4_8 * npk, pd_e31)
     ELSE
        pd_e3 = feed_device(4 * n38)
     ENDIF
  ELSE
     CALL readvar3d(cn_fmsk, cn_tmask, tmask, kimin=iimin, kjmin=ijmin)      ! (every level in one read)
  ENDIF

  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     ! the scalar part of the reference's level loop (:266-311): which levels are written, and clipped where
     iko = 1 ; rdep1 = 0.0 ; rdep2 = 0.0 ; lout = .TRUE.
     DO jk = 1, npk
        IF ( lgsop .OR. locci ) lout = .FALSE.
        rdep1 = rdep2
        rdep2 = rdep1 + e31d(jk)
        khit(jk) = 0 ; kout(jk) = -1 ; zclip(jk) = 0. ; zdiv(jk) = rdep2
        IF ( iko <= npko ) THEN
           IF ( rdep2 >= (gdepo(iko) - tol) ) THEN
              lout = .TRUE.
              khit(jk) = 1 ; zclip(jk) = gdepo(iko) - rdep1
           ENDIF
           IF ( lout ) THEN
              IF ( jt == 1 ) THEN
                 PRINT *, 'Output for level ', iko
                 PRINT *, 'rdep1, rdep2, depo ', rdep1, rdep2, gdepo(iko)
              ENDIF
              kout(jk) = iko - 1
              iko = iko + 1
           ENDIF
        ENDIF
     END DO
     pe3 = C_NULL_PTR ; pe31d = C_NULL_PTR
     IF ( lfeed ) THEN
        IF ( .NOT. lfull .AND. ( jt == 1 .OR. lg_vvl ) )  &
             &  CALL feed_field(cn_fe3t, cn_ve3t, pd_e3, npiglo, npjglo, npk, ktime=it, ldiom=.NOT. lg_vvl, kimin=iimin, kjmin=ijmin)
        CALL feed_field(cf_in, cv_in, pd_zt, npiglo, npjglo, npk, ktime=jt, kimin=iimin, kjmin=ijmin)
        CALL hip_check(cdft_vint_cumul(zt_d, pd_e3, pd_e31, tmask_d, npiglo, npjglo, npk, kout, khit, zclip, zdiv, npko, sclf, &
             &         MERGE(1_C_INT, 0_C_INT, ltmean .OR. lsmean), CDFT_MEM_DEVICE, zout_d, CDFT_MEM_DEVICE, feed_stream), 'cdft_vint_cumul')
        il = 0
        DO jk = 1, npk
           IF ( kout(jk) < 0 ) CYCLE
           il = il + 1 ; ib = 1 + MOD(il - 1, 2)
           CALL SYSTEM_CLOCK(iclk(1), iclk_rate)
           CALL wait_writes(ncout)                        ! (the page-locked array of this parity is in the file)
           CALL SYSTEM_CLOCK(iclk(2))
           CALL hip_check(cdft_memcpy_d2h(ph_lev(ib), C_LOC(zout_d(1 + INT(kout(jk), 8) * npiglo * npjglo)),  &
                &         INT(4_8 * npiglo * npjglo, C_SIZE_T), feed_stream), 'cdft_memcpy_d2h')
           CALL SYSTEM_CLOCK(iclk(3))
           ierr = putvar_behind(ncout, id_varout(1), ph_lev(ib), .FALSE., kout(jk) + 1, npiglo, npjglo, ktime=jt)
           CALL SYSTEM_CLOCK(iclk(4))
           dl_ph(1:3) = dl_ph(1:3) + REAL(iclk(2:4) - iclk(1:3), 8) / iclk_rate
        END DO
        CYCLE
     ELSE IF ( lfull ) THEN
        pe31d = C_LOC(e31d)
     ELSE
        IF ( jt == 1 .OR. lg_vvl ) THEN
           CALL readvar3d(cn_fe3t, cn_ve3t, e3t, kimin=iimin, kjmin=ijmin, ktime=it, ldiom=.NOT. lg_vvl)
        ENDIF
        pe3 = C_LOC(e3t)
     ENDIF
     IF ( .NOT. lfeed ) CALL readvar3d(cf_in, cv_in, zt, kimin=iimin, kjmin=ijmin, ktime=jt)
     IF ( .NOT. lfeed ) CALL hip_check(cdft_vint_cumul(zt, pe3, pe31d, tmask, npiglo, npjglo, npk, kout, khit, zclip, zdiv, npko, sclf, &
          &         MERGE(1_C_INT, 0_C_INT, ltmean .OR. lsmean), CDFT_MEM_HOST, zout, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_vint_cumul')
     DO jk = 1, npk
        IF ( kout(jk) >= 0 ) ierr = putvar(ncout, id_varout(1), zout(:,:,kout(jk)+1), kout(jk)+1, npiglo, npjglo, ktime=jt)
     END DO
  END DO
  IF ( lfeed ) CALL wait_writes(ncout)                    ! the last levels of the last record (a failed write stops here)
  CALL GET_ENVIRONMENT_VARIABLE('CDFT_TIMING', cldum)
  IF ( lfeed .AND. LEN_TRIM(cldum) > 0 ) WRITE(0, '(a,3f9.3)') ' CDFT_TIMING cdfvint levels out (s): wait for the file, D2H, queue', dl_ph
  ierr = closeout(ncout)
  IF ( lfeed ) THEN
     CALL feed_release(pd_zt) ; CALL feed_release(pd_msk) ; CALL feed_release(pd_e3) ; CALL feed_release(pd_e31) ; CALL feed_release(pd_out)
     ierr = cdft_host_free(ph_lev(1)) ; ierr = cdft_host_free(ph_lev(2))
     CALL feed_close()
  ENDIF

CONTAINS

  SUBROUTINE geti(kval)
    INTEGER(KIND=4), INTENT(out) :: kval
    CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) kval
  END SUBROUTINE geti

  SUBROUTINE CreateOutput
    !! src/cdfvint.f90:318-346
    ipk(:)                       = npko
    stypvar(1)%ichunk            = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
    stypvar(1)%cname             = TRIM(cv_out)
    stypvar(1)%cunits            = TRIM(cunits)
    stypvar(1)%rmissing_value    = 0.
    stypvar(1)%valid_min         = -1.e15
    stypvar(1)%valid_max         =  1.e15
    stypvar(1)%clong_name        = TRIM(clongname)
    stypvar(1)%cshort_name       = TRIM(cv_out)
    stypvar(1)%conline_operation = 'N/A'
    stypvar(1)%caxis             = 'TZYX'
    ncout = create      (cf_out, 'none', npiglo, npjglo, npko, cdep=cn_vdepthw, ld_xycoo=.FALSE., ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, 1, ipk, id_varout, ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_in, npiglo, npjglo, npko, pnavlon=rlon, pnavlat=rlat, pdep=gdepo, ld_xycoo=.FALSE.)
    dtim  = getvar1d(cf_in, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfvint
