PROGRAM cdfvtrp
  !!======================================================================
  !!                     ***  PROGRAM  cdfvtrp  ***
  !! Vertically integrated transports at each grid cell, optionally rotated along / across the
  !! bathymetric slope.
  !!
  !! Drop-in for the reference tool (src/cdfvtrp.f90): same options, same trp.nc (sozoutrp, somevtrp
  !! [soastrp, socstrp] float (t,y,x), :262-291).  The level loop (:196-219) runs on the GPU for all levels
  !! of a record at once (cdft_vint_transport, bit-identical); the -bathy rotation (:224-259) is 2-D work
  !! through the compiler's ATAN2 / COS / SIN and stays in the program, as written in the reference.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: narg, ijarg, ierr, jk, jt, it, npiglo, npjglo, npk, npt, ncout, nvarout = 2
  INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e1v(:,:), e2u(:,:), e3u(:,:,:), e3v(:,:,:), e31d(:)
  REAL(KIND=4), ALLOCATABLE :: zu(:,:,:), zv(:,:,:)
  REAL(KIND=4), ALLOCATABLE :: e1u(:,:), e2v(:,:), tmask(:,:), hdepw(:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:), dwku(:,:), dwkv(:,:), dtrpu(:,:), dtrpv(:,:)
  TYPE(variable), ALLOCATABLE :: stypvar(:)
  TYPE(C_PTR) :: pe3u, pe3v, pe31d
  TYPE(C_PTR) :: pd_u = C_NULL_PTR, pd_v = C_NULL_PTR, pd_e1v = C_NULL_PTR, pd_e2u = C_NULL_PTR, pd_e3u = C_NULL_PTR, pd_e3v = C_NULL_PTR
  TYPE(C_PTR) :: pd_e31 = C_NULL_PTR
  REAL(KIND=4), POINTER :: zu_d(:), zv_d(:), e1v_d(:), e2u_d(:)       ! device arrays of the host feed, never touched by the host
  INTEGER(KIND=8) :: nij8, n38
  LOGICAL :: lfeed
  CHARACTER(LEN=256) :: cf_ufil = 'none', cf_vfil = 'none', cf_out = 'trp.nc', cldum
  CHARACTER(LEN=256) :: cv_soastrp = 'soastrp', cv_socstrp = 'socstrp'
  LOGICAL :: lfull = .FALSE., lbathy = .FALSE., lchk = .FALSE., lnc4 = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfvtrp.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-u'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_ufil) ; ijarg = ijarg + 1
     CASE ( '-v'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vfil) ; ijarg = ijarg + 1
     CASE ( '-full'  ) ; lfull  = .TRUE.
     CASE ( '-vvl'   ) ; lg_vvl = .TRUE.
     CASE ( '-o'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out) ; ijarg = ijarg + 1
     CASE ( '-nc4'   ) ; lnc4   = .TRUE.
     CASE ( '-bathy' ) ; lbathy = .TRUE. ; nvarout = 4
     CASE DEFAULT      ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  lchk = lchk .OR. chkfile(cn_fzgr)
  lchk = lchk .OR. chkfile(cn_fhgr)
  lchk = lchk .OR. chkfile(cf_ufil)
  lchk = lchk .OR. chkfile(cf_vfil)
  IF ( lbathy ) lchk = lchk .OR. chkfile(cn_fmsk)
  IF ( lchk ) STOP 99
  IF ( lg_vvl ) THEN
     cn_fe3u = cf_ufil ; cn_fe3v = cf_vfil
     cn_ve3u = cn_ve3uvvl ; cn_ve3v = cn_ve3vvvl
  ENDIF
  ALLOCATE( ipk(nvarout), id_varout(nvarout), stypvar(nvarout) )

  npiglo = getdim(cf_ufil, cn_x)
  npjglo = getdim(cf_ufil, cn_y)
  npk    = getdim(cf_ufil, cn_z)
  npt    = getdim(cf_ufil, cn_t)
  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt

  ! large fields: everything 3-D goes from the file to HBM through the page-locked ring (module cdffeed)
  nij8 = INT(npiglo, 8) * npjglo ; n38 = nij8 * npk
  lfeed = feed_wanted(4 * n38)
  ALLOCATE( e1v(npiglo,npjglo), e2u(npiglo,npjglo) )
  IF ( .NOT. lfeed ) ALLOCATE( zu(npiglo,npjglo,npk), zv(npiglo,npjglo,npk) )
  ALLOCATE( dwku(npiglo,npjglo), dwkv(npiglo,npjglo), dtrpu(npiglo,npjglo), dtrpv(npiglo,npjglo), e31d(npk), dtim(npt) )
  IF ( .NOT. ( lfull .OR. lfeed ) ) ALLOCATE( e3u(npiglo,npjglo,npk), e3v(npiglo,npjglo,npk) )
  IF ( lbathy ) THEN
     ALLOCATE( e1u(npiglo,npjglo), e2v(npiglo,npjglo), tmask(npiglo,npjglo), hdepw(npiglo,npjglo) )
  ENDIF
  CALL CreateOutput
  CALL getvar_to(cn_fhgr, cn_ve1v, 1, e1v(:,:))
  CALL getvar_to(cn_fhgr, cn_ve2u, 1, e2u(:,:))
  IF ( lfull ) e31d(:) = getvare3(cn_fzgr, cn_ve3t1d, npk)
  IF ( lbathy ) THEN
     CALL getvar_to(cn_fhgr, cn_ve1u, 1, e1u(:,:))
     CALL getvar_to(cn_fhgr, cn_ve2v, 1, e2v(:,:))
     CALL getvar_to(cn_fmsk, cn_tmask, 1, tmask(:,:))
     CALL getvar_to(cn_fzgr, cn_hdepw, 1, hdepw(:,:))
  ENDIF

  IF ( lfeed ) THEN
     pd_u = feed_device(4 * n38) ; pd_v = feed_device(4 * n38) ; pd_e1v = feed_device(4 * nij8) ; pd_e2u = feed_device(4 * nij8)
     CALL C_F_POINTER(pd_u, zu_d, (/ n38 /)) ; CALL C_F_POINTER(pd_v, zv_d, (/ n38 /))
     CALL C_F_POINTER(pd_e1v, e1v_d, (/ nij8 /)) ; CALL C_F_POINTER(pd_e2u, e2u_d, (/ nij8 /))
     CALL feed_host(C_LOC(e1v), 4 * nij8, pd_e1v) ; CALL feed_host(C_LOC(e2u), 4 * nij8, pd_e2u)
     IF ( lfull ) THEN
        pd_e31 = feed_device(4_8 * npk) ; CALL feed_host(C_LOC(e31d), 4_8 * npk, pd_e31)
     ELSE
        pd_e3u = feed_device(4 * n38) ; pd_e3v = feed_device(4 * n38)
     ENDIF
  ENDIF

  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     pe3u = C_NULL_PTR ; pe3v = C_NULL_PTR ; pe31d = C_NULL_PTR
     IF ( lfeed ) THEN
        IF ( .NOT. lfull .AND. ( jt == 1 .OR. lg_vvl ) ) THEN
           CALL feed_field(cn_fe3v, cn_ve3v, pd_e3v, npiglo, npjglo, npk, ktime=it, ldiom=.NOT. lg_vvl)
           CALL feed_field(cn_fe3u, cn_ve3u, pd_e3u, npiglo, npjglo, npk, ktime=it, ldiom=.NOT. lg_vvl)
        ENDIF
     ELSE IF ( lfull ) THEN
        pe31d = C_LOC(e31d)
     ELSE
        IF ( jt == 1 .OR. lg_vvl ) THEN
           CALL readvar3d(cn_fe3v, cn_ve3v, e3v, ktime=it, ldiom=.NOT. lg_vvl)      ! (every level in one read)
           CALL readvar3d(cn_fe3u, cn_ve3u, e3u, ktime=it, ldiom=.NOT. lg_vvl)
        ENDIF
        pe3u = C_LOC(e3u) ; pe3v = C_LOC(e3v)
     ENDIF
     DO jk = 1, npk
        PRINT *, 'level ', jk
     END DO
     IF ( lfeed ) THEN
        CALL feed_field(cf_ufil, cn_vozocrtx, pd_u, npiglo, npjglo, npk, ktime=jt)
        CALL hip_check(cdft_vint_transport(zu_d, e2u_d, pd_e3u, pd_e31, npiglo, npjglo, npk, CDFT_MEM_DEVICE, dtrpu, CDFT_MEM_HOST, &
             &         feed_stream), 'cdft_vint_transport')
        CALL feed_field(cf_vfil, cn_vomecrty, pd_v, npiglo, npjglo, npk, ktime=jt)
        CALL hip_check(cdft_vint_transport(zv_d, e1v_d, pd_e3v, pd_e31, npiglo, npjglo, npk, CDFT_MEM_DEVICE, dtrpv, CDFT_MEM_HOST, &
             &         feed_stream), 'cdft_vint_transport')
     ELSE
        CALL readvar3d(cf_ufil, cn_vozocrtx, zu, ktime=jt)
        CALL readvar3d(cf_vfil, cn_vomecrty, zv, ktime=jt)
        CALL hip_check(cdft_vint_transport(zu, e2u, pe3u, pe31d, npiglo, npjglo, npk, CDFT_MEM_HOST, dtrpu, CDFT_MEM_HOST, &
             &         C_NULL_PTR), 'cdft_vint_transport')
        CALL hip_check(cdft_vint_transport(zv, e1v, pe3v, pe31d, npiglo, npjglo, npk, CDFT_MEM_HOST, dtrpv, CDFT_MEM_HOST, &
             &         C_NULL_PTR), 'cdft_vint_transport')
     ENDIF
     ierr = putvar(ncout, id_varout(1), REAL(dtrpu(:,:)), 1, npiglo, npjglo, ktime=jt)
     ierr = putvar(ncout, id_varout(2), REAL(dtrpv(:,:)), 1, npiglo, npjglo, ktime=jt)

     IF ( lbathy ) THEN                                   ! along- and cross-slope components (src/cdfvtrp.f90:210-252): on the GPU
        CALL hip_check(cdft_vint_slope_rotate(dtrpu, dtrpv, hdepw, e1u, e2v, tmask, npiglo, npjglo, CDFT_MEM_HOST, dwku, dwkv, &
             &         CDFT_MEM_HOST, C_NULL_PTR), 'cdft_vint_slope_rotate')
        ierr = putvar(ncout, id_varout(3), REAL(dwku(:,:)), 1, npiglo, npjglo, ktime=jt)
        ierr = putvar(ncout, id_varout(4), REAL(dwkv(:,:)), 1, npiglo, npjglo, ktime=jt)
     ENDIF
  END DO
  ierr = closeout(ncout)
  IF ( lfeed ) THEN
     CALL feed_release(pd_u) ; CALL feed_release(pd_v) ; CALL feed_release(pd_e1v) ; CALL feed_release(pd_e2u)
     CALL feed_release(pd_e3u) ; CALL feed_release(pd_e3v) ; CALL feed_release(pd_e31) ; CALL feed_close()
  ENDIF

CONTAINS

  SUBROUTINE CreateOutput
    !! src/cdfvtrp.f90:262-291
    ipk(:) = 1
    stypvar%rmissing_value    = 0.
    stypvar%valid_min         = -100.
    stypvar%valid_max         = 100.
    stypvar%cunits            = 'm3/s'
    stypvar%conline_operation = 'N/A'
    stypvar%caxis             = 'TYX'
    stypvar(1)%ichunk = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /) ; stypvar(2)%ichunk = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
    stypvar(1)%cname       = cn_sozoutrp                  ; stypvar(2)%cname       = cn_somevtrp
    stypvar(1)%clong_name  = 'Zonal_barotropic_transport' ; stypvar(2)%clong_name  = 'Meridional_barotropic_transport'
    stypvar(1)%cshort_name = cn_sozoutrp                  ; stypvar(2)%cshort_name = cn_somevtrp
    IF ( lbathy ) THEN
       stypvar(3)%ichunk = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /) ; stypvar(4)%ichunk = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
       stypvar(3)%cname       = cv_soastrp                      ; stypvar(4)%cname       = cv_socstrp
       stypvar(3)%clong_name  = 'Along_Slope_Barotropic_Transp' ; stypvar(4)%clong_name  = 'Cross_Slope_Barotropic_Transp'
       stypvar(3)%cshort_name = cv_soastrp                      ; stypvar(4)%cshort_name = cv_socstrp
    ENDIF
    ncout = create      (cf_out, cf_ufil, npiglo, npjglo, 1, ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, nvarout, ipk, id_varout, ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_ufil, npiglo, npjglo, 1)
    dtim  = getvar1d(cf_ufil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfvtrp
