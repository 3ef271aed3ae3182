PROGRAM cdfvertmean
  !!======================================================================
  !!                     ***  PROGRAM  cdfvertmean  ***
  !! Vertical mean of a list of variables between two depths.
  !!
  !! Drop-in for the reference tool (src/cdfvertmean.f90): same options, same vertmean.nc (<var>_vert_mean float
  !! (t,y,x), :313-346).  The level loop (:236-270) runs on the GPU for the levels ik1..ik2 of a variable and
  !! record at once (cdft_vertmean, bit-identical).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE modutils
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: jk, jvar, jvarin, jt, it, ik1, ik2, nlev, narg, ijarg, ierr, ncout, nvars, nvaro = 1
  INTEGER(KIND=4) :: iimin = 0, iimax = 0, ijmin = 0, ijmax = 0, npiglo, npjglo, npk, npt
  INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:)
  REAL(KIND=4) :: rdep_up = 0., rdep_down = 0., rdep(1)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e3(:,:,:), e31d(:)
  REAL(KIND=4), ALLOCATABLE :: gdep(:), zv(:,:,:), zmask(:,:,:), rlon(:,:), rlat(:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:), dvertmean(:,:)
  CHARACTER(LEN=256) :: cf_in = 'none', cf_out = 'vertmean.nc', cf_e3, cv_cur, cv_dep, cv_e3, cv_e31d, cv_msk, ctype = 'T'
  CHARACTER(LEN=256) :: cglobal, cldum
  CHARACTER(LEN=256), ALLOCATABLE :: cv_names(:), cv_in(:)
  TYPE(variable), ALLOCATABLE :: stypvarin(:), stypvar(:)
  TYPE(C_PTR) :: pe3, pe31d
  TYPE(C_PTR) :: pd_zv = C_NULL_PTR, pd_msk = C_NULL_PTR, pd_e3 = C_NULL_PTR, pd_e31 = C_NULL_PTR
  REAL(KIND=4), POINTER :: zv_d(:), zmask_d(:)                        ! device arrays of the host feed, never touched by the host
  INTEGER(KIND=8) :: n38
  LOGICAL :: lfeed
  LOGICAL :: lfull = .FALSE., ldebug = .FALSE., lnc4 = .FALSE., lchk
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfvertmean.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-f'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_in) ; ijarg = ijarg + 1
     CASE ( '-l'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; CALL ParseVars(cldum)
     CASE ( '-p'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, ctype) ; ijarg = ijarg + 1
     CASE ( '-zlim'  ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) rdep_up
        ;                CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) rdep_down
     CASE ( '-w'     ) ; CALL geti(iimin) ; CALL geti(iimax) ; CALL geti(ijmin) ; CALL geti(ijmax)
     CASE ( '-full'  ) ; lfull  = .TRUE.
     CASE ( '-debug' ) ; ldebug = .TRUE.
     CASE ( '-vvl'   ) ; lg_vvl = .TRUE.
     CASE ( '-nc4'   ) ; lnc4   = .TRUE.
     CASE ( '-o'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out) ; ijarg = ijarg + 1
     CASE DEFAULT      ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO
  lchk = chkfile(cn_fzgr)
  lchk = chkfile(cn_fmsk) .OR. lchk
  lchk = chkfile(cf_in  ) .OR. lchk
  IF ( lchk ) STOP 99
  CALL SetGlobalAtt(cglobal)
  IF ( rdep_down < rdep_up ) THEN
     PRINT *, 'Give depth limits in increasing order !'
     STOP 99
  ENDIF
  IF ( .NOT. ALLOCATED(cv_in) ) THEN
     PRINT *, ' ERROR : option -l LST-var is mandatory.' ; STOP 99
  ENDIF

  npiglo = getdim(cf_in, cn_x)
  npjglo = getdim(cf_in, cn_y)
  npk    = getdim(cf_in, cn_z)
  npt    = getdim(cf_in, cn_t)
  IF ( iimin /= 0 ) THEN ; npiglo = iimax - iimin + 1 ; ELSE ; iimin = 1
  ENDIF
  IF ( ijmin /= 0 ) THEN ; npjglo = ijmax - ijmin + 1 ; ELSE ; ijmin = 1
  ENDIF
  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt

  nvars = getnvar(cf_in)
  ALLOCATE( cv_names(nvars), stypvarin(nvars), stypvar(nvaro), ipk(nvaro), id_varout(nvaro) )
  cv_names(:) = getvarname(cf_in, nvars, stypvarin)
  DO jvar = 1, nvaro
     IF ( chkvar(cf_in, cv_in(jvar)) ) STOP 99
  END DO
  ALLOCATE( dvertmean(npiglo,npjglo), rlon(npiglo,npjglo), rlat(npiglo,npjglo), gdep(npk), dtim(npt) )
  IF ( lfull ) ALLOCATE( e31d(npk) )
  CALL getvar_to(cf_in, cn_vlon2d, 1, rlon, kimin=iimin, kjmin=ijmin)
  CALL getvar_to(cf_in, cn_vlat2d, 1, rlat, kimin=iimin, kjmin=ijmin)
  CALL CreateOutput

  SELECT CASE ( ctype )
  CASE ( 'T', 't' ) ; cv_dep = cn_gdepw ; cv_e3 = cn_ve3t ; cv_e31d = cn_ve3t1d ; cv_msk = cn_tmask ; cf_e3 = cn_fe3t
     IF ( lg_vvl ) cv_e3 = cn_ve3tvvl
  CASE ( 'U', 'u' ) ; cv_dep = cn_gdepw ; cv_e3 = cn_ve3u ; cv_e31d = cn_ve3t1d ; cv_msk = cn_umask ; cf_e3 = cn_fe3u
     IF ( lg_vvl ) cv_e3 = cn_ve3uvvl
  CASE ( 'V', 'v' ) ; cv_dep = cn_gdepw ; cv_e3 = cn_ve3v ; cv_e31d = cn_ve3t1d ; cv_msk = cn_vmask ; cf_e3 = cn_fe3v
     IF ( lg_vvl ) cv_e3 = cn_ve3vvvl
  CASE ( 'W', 'w' ) ; cv_dep = cn_gdept ; cv_e3 = cn_ve3w ; cv_e31d = cn_ve3w1d ; cv_msk = cn_tmask ; cf_e3 = cn_fe3w
     IF ( lg_vvl ) cv_e3 = cn_ve3wvvl
  CASE DEFAULT ; PRINT *, 'Point type ', TRIM(ctype), ' not known! ' ; STOP 99
  END SELECT
  IF ( lg_vvl ) cf_e3 = cf_in
  gdep(:) = getvare3(cn_fzgr, cv_dep, npk)
  IF ( lfull ) e31d(:) = getvare3(cn_fzgr, cv_e31d, npk)

  ik1 = 1 ; ik2 = npk                                        ! nearest levels of rdep_up and rdep_down (:219-224)
  DO jk = 1, npk
     IF ( gdep(jk) <= rdep_up   ) ik1 = jk
     IF ( gdep(jk) <= rdep_down ) ik2 = jk
  END DO
  PRINT '(a,2f10.3)', 'depth limit of integration : ', rdep_up, rdep_down
  PRINT '(a,2i10  )', 'nearest level found        : ', ik1, ik2
  PRINT '(a,2f10.3)', 'corresponding depth        : ', gdep(ik1), gdep(MIN(ik2+1, npk))
  nlev = ik2 - ik1 + 1
  ! large fields: the levels ik1..ik2 of mask, e3 and every field from the file to HBM through the page-locked ring (cdffeed)
  n38 = INT(npiglo, 8) * npjglo * nlev
  lfeed = feed_wanted(4 * n38)
  IF ( lfeed ) THEN
     pd_zv = feed_device(4 * n38) ; pd_msk = feed_device(4 * n38)
     CALL C_F_POINTER(pd_zv, zv_d, (/ n38 /)) ; CALL C_F_POINTER(pd_msk, zmask_d, (/ n38 /))
     CALL feed_field(cn_fmsk, cv_msk, pd_msk, npiglo, npjglo, nlev, kimin=iimin, kjmin=ijmin, kkmin=ik1)
     IF ( lfull ) THEN
        pd_e31 = feed_device(4_8 * nlev) ; CALL feed_host(C_LOC(e31d(ik1)), 4_8 * nlev, pd_e31)
     ELSE
        pd_e3 = feed_device(4 * n38)
     ENDIF
  ELSE
     ALLOCATE( zv(npiglo,npjglo,nlev), zmask(npiglo,npjglo,nlev) )
     IF ( .NOT. lfull ) ALLOCATE( e3(npiglo,npjglo,nlev) )
     DO jk = ik1, ik2
        CALL getvar_to(cn_fmsk, cv_msk, jk, zmask(:,:,jk-ik1+1), kimin=iimin, kjmin=ijmin)
     END DO
  ENDIF

  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     pe3 = C_NULL_PTR ; pe31d = C_NULL_PTR
     IF ( lfeed ) THEN
        IF ( .NOT. lfull .AND. ( jt == 1 .OR. lg_vvl ) )  &
             &  CALL feed_field(cf_e3, cv_e3, pd_e3, npiglo, npjglo, nlev, ktime=it, ldiom=.NOT. lg_vvl, kimin=iimin, kjmin=ijmin, kkmin=ik1)
        DO jvar = 1, nvaro
           CALL feed_field(cf_in, cv_in(jvar), pd_zv, npiglo, npjglo, nlev, ktime=jt, kimin=iimin, kjmin=ijmin, kkmin=ik1)
           CALL hip_check(cdft_vertmean(zv_d, pd_e3, pd_e31, zmask_d, npiglo, npjglo, nlev, gdep(ik1), gdep(ik2), rdep_up, rdep_down, &
                &         CDFT_MEM_DEVICE, dvertmean, CDFT_MEM_HOST, feed_stream), 'cdft_vertmean')
           ierr = putvar(ncout, id_varout(jvar), REAL(dvertmean), 1, npiglo, npjglo, ktime=jt)
        END DO
        CYCLE
     ENDIF
     IF ( lfull ) THEN
        pe31d = C_LOC(e31d(ik1))
     ELSE
        IF ( jt == 1 .OR. lg_vvl ) THEN
           DO jk = ik1, ik2
              CALL getvar_to(cf_e3, cv_e3, jk, e3(:,:,jk-ik1+1), kimin=iimin, kjmin=ijmin, ktime=it, ldiom=.NOT. lg_vvl)
           END DO
        ENDIF
        pe3 = C_LOC(e3)
     ENDIF
     DO jvar = 1, nvaro
        cv_cur = cv_in(jvar)
        DO jk = ik1, ik2
           CALL getvar_to(cf_in, cv_cur, jk, zv(:,:,jk-ik1+1), kimin=iimin, kjmin=ijmin, ktime=jt)
        END DO
        CALL hip_check(cdft_vertmean(zv, pe3, pe31d, zmask, npiglo, npjglo, nlev, gdep(ik1), gdep(ik2), rdep_up, rdep_down, &
             &         CDFT_MEM_HOST, dvertmean, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_vertmean')
        ierr = putvar(ncout, id_varout(jvar), REAL(dvertmean), 1, npiglo, npjglo, ktime=jt)
     END DO
  END DO
  ierr = closeout(ncout)
  IF ( lfeed ) THEN
     CALL feed_release(pd_zv) ; CALL feed_release(pd_msk) ; CALL feed_release(pd_e3) ; CALL feed_release(pd_e31) ; CALL feed_close()
  ENDIF

CONTAINS

  SUBROUTINE geti(kval)
    INTEGER(KIND=4), INTENT(out) :: kval
    CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) kval
  END SUBROUTINE geti

  SUBROUTINE ParseVars(cdum)
    CHARACTER(LEN=*), INTENT(in) :: cdum
    CHARACTER(LEN=80) :: cl_dum(100)
    INTEGER :: jc, inchar, i1
    i1 = 1
    inchar = LEN(TRIM(cdum))
    DO jc = 1, inchar
       IF ( cdum(jc:jc) == ',' ) THEN
          cl_dum(nvaro) = cdum(i1:jc-1)
          i1 = jc + 1
          nvaro = nvaro + 1
       ENDIF
    END DO
    cl_dum(nvaro) = cdum(i1:inchar)
    ALLOCATE( cv_in(nvaro) )
    IF ( ldebug ) PRINT *, ' SELECTED VARIABLES :'
    DO jc = 1, nvaro
       cv_in(jc) = cl_dum(jc)
       IF ( ldebug ) PRINT *, '    ', TRIM(cv_in(jc))
    END DO
  END SUBROUTINE ParseVars

  SUBROUTINE CreateOutput
    !! src/cdfvertmean.f90:313-346
    rdep(1) = 0.
    ipk(:)  = 1
    DO jvar = 1, nvaro
       DO jvarin = 1, nvars
          IF ( cv_in(jvar) == stypvarin(jvarin)%cname ) EXIT
       END DO
       stypvar(jvar)%ichunk            = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
       stypvar(jvar)%cname             = TRIM(cv_in(jvar))//'_vert_mean'
       stypvar(jvar)%cunits            = stypvarin(jvarin)%cunits
       stypvar(jvar)%rmissing_value    = stypvarin(jvarin)%rmissing_value
       stypvar(jvar)%valid_min         = stypvarin(jvarin)%valid_min
       stypvar(jvar)%valid_max         = stypvarin(jvarin)%valid_max
       stypvar(jvar)%clong_name        = 'vertical average of '//TRIM(stypvarin(jvarin)%clong_name)
       stypvar(jvar)%cshort_name       = TRIM(cv_in(jvar))//'_vert_mean'
       stypvar(jvar)%conline_operation = 'N/A'
       stypvar(jvar)%caxis             = 'TYX'
    END DO
    ncout = create      (cf_out, cf_in, npiglo, npjglo, 1, ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, nvaro, ipk, id_varout, cdglobal=cglobal, ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_in, npiglo, npjglo, 1, pnavlon=rlon, pnavlat=rlat, pdep=rdep)
    dtim  = getvar1d(cf_in, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfvertmean
