PROGRAM cdfheatc
  !!======================================================================
  !!                     ***  PROGRAM  cdfheatc  ***
  !! Heat content of the ocean (or of a sub-domain): rho0 * cp * sum( e1t*e2t*e3t * T * tmask ).
  !!
  !! Drop-in for the reference tool (src/cdfheatc.f90): same options (-zoom, -full, -vvl, -mxloption,
  !! -mxlf, -M, -o), same stdout lines per level and for the totals (:233-243), same heatc.nc
  !! (heatc3d, heatc2d, heatc3dpervol double, :251-306).  The three SUMs of every level (:225-229) run
  !! on the GPU for all levels of a record at once (cdft_heatc_*): r4 products as written, summed in r4 in the
  !! reference's sequential order (bit-identical to a gfortran -O / flang -O1 build of the reference), or in r8 with
  !! CDFT_HEATC_SUM=r8 (the more accurate value, at the HBM roofline).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4), PARAMETER :: jp_hc3d = 1, jp_hc2d = 2, jp_hcvol = 3
  REAL(KIND=4),    PARAMETER :: pprho0 = 1020., ppcp = 4000.       ! src/cdfheatc.f90:39-40
  INTEGER(C_INT)  :: ngpu, igpus(64)                          ! row blocks and their HIP devices (CDFT_GPUS, CDFT_GPU_MAP)
  INTEGER(KIND=4) :: narg, ijarg, ierr, jt, jk, ik, it
  INTEGER(KIND=4) :: iimin = 0, iimax = 0, ijmin = 0, ijmax = 0, ikmin = 0, ikmax = 0, mxloption = 0
  INTEGER(KIND=4) :: npiglo, npjglo, npk, npkk, npt, nvpk, ncout
  INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e1t(:,:), e2t(:,:), e3t(:,:,:), temp(:,:,:), tmask(:,:,:), rmxldep(:,:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: gdepw(:), e31d(:), gdepw_sel(:), e31d_sel(:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:), dsums(:,:)
  REAL(KIND=8) :: dvol, dsum, dvol2d, dsum2d, dsurf, dl_dum(1,1)
  CHARACTER(LEN=256) :: cf_tfil = 'none', cf_mfil = 'none', cf_out = 'heatc.nc', cv_msk = 'tmask', cldum
  TYPE(variable), ALLOCATABLE :: stypvar(:)
  TYPE(C_PTR) :: plan, pe3t, pe31d, pmxl, pdepw
  ! the same arrays in HBM when they are fed there through the ring (module cdffeed): never touched by the host
  TYPE(C_PTR) :: pd_e1 = C_NULL_PTR, pd_e2 = C_NULL_PTR, pd_e3 = C_NULL_PTR, pd_e31d = C_NULL_PTR, pd_msk = C_NULL_PTR
  TYPE(C_PTR) :: pd_t = C_NULL_PTR, pd_mxl = C_NULL_PTR
  REAL(KIND=4), POINTER :: e1_d(:), e2_d(:), temp_d(:)
  INTEGER(KIND=8) :: nij8, n38
  LOGICAL :: lfull = .FALSE., lchk, lfeed
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfheatc.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-f'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_tfil) ; ijarg = ijarg + 1
     CASE ( '-full'      ) ; lfull = .TRUE.
     CASE ( '-vvl'       ) ; lg_vvl = .TRUE.
     CASE ( '-mxloption' ) ; mxloption = int_arg()
     CASE ( '-mxlf'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_mfil) ; ijarg = ijarg + 1
     CASE ( '-o'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out ) ; ijarg = ijarg + 1
     CASE ( '-zoom'      ) ; iimin = int_arg() ; iimax = int_arg() ; ijmin = int_arg() ; ijmax = int_arg()
        ;                    ikmin = int_arg() ; ikmax = int_arg()
     CASE ( '-M'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cn_fmsk) ; ijarg = ijarg + 1
        ;                    CALL GET_COMMAND_ARGUMENT(ijarg, cv_msk ) ; ijarg = ijarg + 1
     CASE DEFAULT          ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO
  IF ( cf_mfil == 'none' ) cf_mfil = cf_tfil

  lchk = chkfile(cn_fhgr)
  lchk = chkfile(cn_fzgr) .OR. lchk
  lchk = chkfile(cn_fmsk) .OR. lchk
  lchk = chkfile(cf_tfil) .OR. lchk
  lchk = chkfile(cf_mfil) .OR. lchk
  IF ( lchk ) STOP 99
  IF ( lg_vvl ) THEN
     cn_fe3t = cf_tfil
     cn_ve3t = cn_ve3tvvl
  ENDIF

  npiglo = getdim(cf_tfil, cn_x)
  npjglo = getdim(cf_tfil, cn_y)
  npk    = getdim(cf_tfil, cn_z)
  npt    = getdim(cf_tfil, cn_t)
  npkk = npk
  IF ( iimin /= 0 ) THEN ; npiglo = iimax - iimin + 1 ; ELSE ; iimin = 1 ; ENDIF
  IF ( ijmin /= 0 ) THEN ; npjglo = ijmax - ijmin + 1 ; ELSE ; ijmin = 1 ; ENDIF
  IF ( ikmin /= 0 ) THEN ; npkk   = ikmax - ikmin + 1 ; ELSE ; ikmin = 1 ; ikmax = npk ; ENDIF
  nvpk = getvdim(cf_tfil, cn_votemper)
  IF ( nvpk == 2 ) nvpk = 1
  IF ( nvpk == 3 ) nvpk = npkk

  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt
  PRINT *, 'nvpk   = ', nvpk

  CALL cdf_phase('start -> arguments, dimensions')
  CALL hip_shard(ngpu, igpus)                                ! CDFT_GPUS / CDFT_GPU_MAP: row blocks over the node's GPUs
  lfeed = feed_wanted(4_8 * npiglo * npjglo * nvpk) .AND. ngpu == 1                      ! one device: file -> HBM through the ring (module cdffeed)
  nij8 = INT(npiglo, 8) * npjglo ; n38 = nij8 * nvpk
  CALL cdf_phase('device count (the HIP runtime comes up)')
  ALLOCATE( e1t(npiglo,npjglo), e2t(npiglo,npjglo), gdepw(npk), gdepw_sel(nvpk), dtim(npt), dsums(3,nvpk) )
  IF ( .NOT. lfeed ) ALLOCATE( temp(npiglo,npjglo,nvpk), tmask(npiglo,npjglo,nvpk) )
  IF ( mxloption /= 0 ) ALLOCATE( rmxldep(npiglo,npjglo) )
  IF ( lfull ) THEN ; ALLOCATE( e31d(npk), e31d_sel(nvpk) ) ; ELSE IF ( .NOT. lfeed ) THEN ; ALLOCATE( e3t(npiglo,npjglo,nvpk) ) ; ENDIF

  CALL getvar_to(cn_fhgr, cn_ve1t, 1, e1t(:,:), kimin=iimin, kjmin=ijmin)
  CALL getvar_to(cn_fhgr, cn_ve2t, 1, e2t(:,:), kimin=iimin, kjmin=ijmin)
  gdepw(:) = getvare3(cn_fzgr, cn_gdepw, npk)
  gdepw_sel(:) = gdepw(ikmin:ikmin + nvpk - 1)
  IF ( lfull ) THEN
     e31d(:) = getvare3(cn_fzgr, cn_ve3t1d, npk)
     e31d_sel(:) = e31d(ikmin:ikmin + nvpk - 1)
  ENDIF
  CALL cdf_phase('allocations, e1t e2t gdepw read')
  IF ( lfeed ) THEN
     pd_e1 = feed_device(4 * nij8) ; pd_e2 = feed_device(4 * nij8) ; pd_msk = feed_device(4 * n38) ; pd_t = feed_device(4 * n38)
     CALL C_F_POINTER(pd_e1, e1_d, (/ nij8 /)) ; CALL C_F_POINTER(pd_e2, e2_d, (/ nij8 /)) ; CALL C_F_POINTER(pd_t, temp_d, (/ n38 /))
     CALL feed_host(C_LOC(e1t), 4 * nij8, pd_e1) ; CALL feed_host(C_LOC(e2t), 4 * nij8, pd_e2)
     CALL feed_field(cn_fmsk, cv_msk, pd_msk, npiglo, npjglo, nvpk, kimin=iimin, kjmin=ijmin, kkmin=ikmin)
     IF ( lfull ) THEN
        pd_e31d = feed_device(4_8 * nvpk) ; CALL feed_host(C_LOC(e31d_sel), 4_8 * nvpk, pd_e31d)
     ELSE
        pd_e3 = feed_device(4 * n38)
     ENDIF
     IF ( mxloption /= 0 ) pd_mxl = feed_device(4 * nij8)
  ELSE
     CALL readvar3d(cn_fmsk, cv_msk, tmask, kimin=iimin, kjmin=ijmin, kkmin=ikmin)      ! (every level in one read)
  ENDIF

  CALL cdf_phase('device arrays, ring, mask file -> HBM')
  CALL CreateOutput
  IF ( ngpu > 1 ) THEN
     CALL hip_check(cdft_heatc_create_sharded(plan, npiglo, npjglo, nvpk, ngpu, igpus), 'cdft_heatc_create_sharded')
  ELSE
     CALL hip_check(cdft_heatc_create(plan, npiglo, npjglo, nvpk), 'cdft_heatc_create')
  ENDIF
  CALL cdf_phase('CreateOutput, plan')
  ! The reference's SUMs are REAL(KIND=4), added sequentially in memory order (src/cdfheatc.f90:225-229 as gfortran -O and
  ! flang -O1 compile them): the program returns exactly those values.  CDFT_HEATC_SUM=r8 asks for r8 sums instead -- the
  ! more accurate number (the two differ by 1e-4 relative at eORCA025) at the HBM roofline.
  CALL GET_ENVIRONMENT_VARIABLE('CDFT_HEATC_SUM', cldum)
  IF ( TRIM(cldum) /= 'r8' ) CALL hip_check(cdft_heatc_set_sum_mode(plan, CDFT_HEATC_SUM_R4SEQ), 'cdft_heatc_set_sum_mode')

  CALL cdf_phase('sum mode')
  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     dvol = 0.d0
     dsum = 0.d0
     PRINT *, 'TIME : ', dtim(jt)
     pmxl = C_NULL_PTR ; pdepw = C_NULL_PTR
     IF ( mxloption /= 0 ) THEN
        CALL getvar_to(cf_mfil, cn_somxl010, 1, rmxldep(:,:), kimin=iimin, kjmin=ijmin, ktime=jt)
        pmxl = C_LOC(rmxldep) ; pdepw = C_LOC(gdepw_sel)
        IF ( lfeed ) THEN ; CALL feed_host(pmxl, 4 * nij8, pd_mxl) ; pmxl = pd_mxl ; ENDIF
     ENDIF
     IF ( lfeed ) THEN                                    ! the record, and the metric when it is due, slab by slab into HBM
        CALL feed_field(cf_tfil, cn_votemper, pd_t, npiglo, npjglo, nvpk, kimin=iimin, kjmin=ijmin, kkmin=ikmin, ktime=jt)
        IF ( jt == 1 .OR. lg_vvl ) THEN
           IF ( .NOT. lfull ) CALL feed_field(cn_fe3t, cn_ve3t, pd_e3, npiglo, npjglo, nvpk, kimin=iimin, kjmin=ijmin, &
                &                              kkmin=ikmin, ktime=it, ldiom=.NOT. lg_vvl)
           CALL hip_check(cdft_heatc_set_mesh(plan, e1_d, e2_d, pd_e3, pd_e31d, pd_msk, 1_C_INT, CDFT_MEM_DEVICE, feed_stream), &
                &         'cdft_heatc_set_mesh')
        ENDIF
        CALL hip_check(cdft_heatc_compute(plan, temp_d, CDFT_MEM_DEVICE, pmxl, pdepw, INT(mxloption, C_INT), dsums, CDFT_MEM_HOST, &
             &         feed_stream), 'cdft_heatc_compute')
     ELSE
        CALL readvar3d(cf_tfil, cn_votemper, temp, kimin=iimin, kjmin=ijmin, kkmin=ikmin, ktime=jt)
        IF ( .NOT. lfull .AND. (jt == 1 .OR. lg_vvl) ) &
           &  CALL readvar3d(cn_fe3t, cn_ve3t, e3t, kimin=iimin, kjmin=ijmin, kkmin=ikmin, ktime=it, ldiom=.NOT. lg_vvl)
        IF ( jt == 1 .OR. lg_vvl ) THEN
           pe3t = C_NULL_PTR ; pe31d = C_NULL_PTR
           IF ( lfull ) THEN ; pe31d = C_LOC(e31d_sel) ; ELSE ; pe3t = C_LOC(e3t) ; ENDIF
           CALL hip_check(cdft_heatc_set_mesh(plan, e1t, e2t, pe3t, pe31d, C_LOC(tmask), 1_C_INT, CDFT_MEM_HOST, C_NULL_PTR), &
                &         'cdft_heatc_set_mesh')
        ENDIF
        CALL hip_check(cdft_heatc_compute(plan, temp, CDFT_MEM_HOST, pmxl, pdepw, INT(mxloption, C_INT), dsums, CDFT_MEM_HOST, &
             &         C_NULL_PTR), 'cdft_heatc_compute')
     ENDIF

     DO jk = 1, nvpk                                       ! src/cdfheatc.f90:225-238
        ik = jk + ikmin - 1
        dsurf = dsums(1,jk) ; dvol2d = dsums(2,jk) ; dsum2d = dsums(3,jk)
        dvol = dvol + dvol2d
        dsum = dsum + dsum2d
        IF ( dvol2d /= 0 ) THEN
           PRINT *, ' Heat Content  at level ', ik, '(', gdepw(ik), ' m) ', pprho0 * ppcp * dsum2d, 'surface = ', dsurf / 1.e6, ' km^2'
        ELSE
           PRINT *, ' No points in the water at level ', ik, '(', gdepw(ik), ' m) '
        ENDIF
        dl_dum(1,1) = pprho0 * ppcp * dsum2d
        ierr = putvar(ncout, id_varout(jp_hc2d), dl_dum(:,:), jk, 1, 1, ktime=jt)
     END DO
     PRINT *, ' Total Heat content        : ', pprho0 * ppcp * dsum, ' Joules'
     PRINT *, ' Total Heat content/volume : ', pprho0 * ppcp * dsum / dvol, ' Joules/m3 '
     dl_dum(1,1) = pprho0 * ppcp * dsum
     ierr = putvar(ncout, id_varout(jp_hc3d), dl_dum(:,:), 1, 1, 1, ktime=jt)
     dl_dum(1,1) = dl_dum(1,1) / dvol
     ierr = putvar(ncout, id_varout(jp_hcvol), dl_dum(:,:), 1, 1, 1, ktime=jt)
  END DO
  CALL cdf_phase('records')
  ierr = cdft_heatc_destroy(plan)
  IF ( lfeed ) THEN
     CALL feed_release(pd_e1) ; CALL feed_release(pd_e2) ; CALL feed_release(pd_msk) ; CALL feed_release(pd_t)
     CALL feed_release(pd_e3) ; CALL feed_release(pd_e31d) ; CALL feed_release(pd_mxl)
     CALL feed_close()
  ENDIF
  ierr = closeout(ncout)
  CALL cdf_phase('destroy, release, close')

CONTAINS

  INTEGER FUNCTION int_arg()
    CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
    READ(cldum, *) int_arg
  END FUNCTION int_arg

  SUBROUTINE CreateOutput
    !! three scalar-like variables stored as (1,1[,k],t) doubles (src/cdfheatc.f90:251-306)
    REAL(KIND=4) :: zdumlon(1,1), zdumlat(1,1)
    ALLOCATE( stypvar(3), ipk(3), id_varout(3) )
    zdumlon = 0. ; zdumlat = 0.
    ipk(:) = 1
    stypvar%scale_factor      = 1.
    stypvar%add_offset        = 0.
    stypvar%savelog10         = 0.
    stypvar%conline_operation = 'N/A'
    stypvar%caxis             = 'T'
    stypvar%cprecision        = 'r8'
    stypvar%rmissing_value    = 0.                         ! left unset by the reference
    stypvar%valid_min         = 0.
    stypvar%valid_max         = 0.
    stypvar(jp_hc3d)%cname  = 'heatc3d'        ; stypvar(jp_hc3d)%cunits  = 'Joules'
    stypvar(jp_hc3d)%clong_name  = 'Total Heat Content' ; stypvar(jp_hc3d)%cshort_name = 'heatc3d'
    stypvar(jp_hcvol)%cname = 'heatc3dpervol'  ; stypvar(jp_hcvol)%cunits = 'Joules/m3'
    stypvar(jp_hcvol)%clong_name = 'Total Heat Content per unit volume' ; stypvar(jp_hcvol)%cshort_name = 'heatc3dpervol'
    ipk(jp_hc2d) = npkk
    stypvar(jp_hc2d)%cname  = 'heatc2d'        ; stypvar(jp_hc2d)%cunits  = 'Joules'
    stypvar(jp_hc2d)%clong_name  = 'Heat Content at each selected level' ; stypvar(jp_hc2d)%cshort_name = 'heatc2d'
    ncout = create      (cf_out, 'none', 1, 1, npkk, cdep='depthw')
    ierr  = createvar   (ncout, stypvar, 3, ipk, id_varout)
    ierr  = putheadervar(ncout, cf_tfil, 1, 1, npkk, pnavlon=zdumlon, pnavlat=zdumlat, pdep=gdepw(ikmin:ikmax), cdep='depthw')
    dtim  = getvar1d(cf_tfil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfheatc
