PROGRAM cdfvhst
  !!======================================================================
  !!                     ***  PROGRAM  cdfvhst  ***
  !! Vertically integrated heat and salt transports at each grid cell (the vertical integral of a VT file).
  !!
  !! Drop-in for the reference tool (src/cdfvhst.f90): same options, same trp.nc (somevt, somevs, sozout,
  !! sozous float (t,y,x), :199-236).  The level loop (:155-179) runs on the GPU for all levels of a
  !! record at once (cdft_vhst_transport, bit-identical: r4 products as written, r8 sums in level order).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: narg, ijarg, ierr, jt, jk, it, npiglo, npjglo, npk, npt, ncout
  INTEGER(KIND=4) :: ipk(4), id_varout(4)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e1v(:,:), e2u(:,:), e3u(:,:,:), e3v(:,:,:), e31d(:)
  REAL(KIND=4), ALLOCATABLE :: zt(:,:,:), zs(:,:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:), dtrpt(:,:), dtrps(:,:)
  TYPE(variable) :: stypvar(4)
  TYPE(C_PTR) :: pe3u, pe3v, pe31d
  TYPE(C_PTR) :: pd_t = C_NULL_PTR, pd_s = C_NULL_PTR, pd_e1v = C_NULL_PTR, pd_e2u = C_NULL_PTR, pd_e3u = C_NULL_PTR, pd_e3v = C_NULL_PTR
  TYPE(C_PTR) :: pd_e31 = C_NULL_PTR
  REAL(KIND=4), POINTER :: zt_d(:), zs_d(:), e1v_d(:), e2u_d(:)       ! device arrays of the host feed, never touched by the host
  INTEGER(KIND=8) :: nij8, n38
  LOGICAL :: lfeed
  CHARACTER(LEN=256) :: cf_vtfil = 'none', cf_out = 'trp.nc', cldum
  LOGICAL :: lfull = .FALSE., lnc4 = .FALSE., lchk = .FALSE., lchkvar = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfvhst.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-f'    ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vtfil) ; ijarg = ijarg + 1
     CASE ( '-full' ) ; lfull  = .TRUE.
     CASE ( '-vvl'  ) ; lg_vvl = .TRUE.
     CASE ( '-o'    ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out) ; ijarg = ijarg + 1
     CASE ( '-nc4'  ) ; lnc4   = .TRUE.
     CASE DEFAULT     ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  ! the reference combines these with .AND. on a flag that starts .FALSE. (src/cdfvhst.f90:103-107), so a
  ! missing file is reported by chkfile but never stops the program there; kept: the first read fails (STOP 98)
  lchk = lchk .AND. chkfile(cn_fhgr)
  lchk = lchk .AND. chkfile(cn_fzgr)
  lchk = lchk .AND. chkfile(cf_vtfil)
  IF ( lchk ) STOP 99
  IF ( lg_vvl ) THEN
     cn_fe3u = cf_vtfil ; cn_fe3v = cf_vtfil
     cn_ve3u = cn_ve3uvvl ; cn_ve3v = cn_ve3vvvl
     lchkvar = lchkvar .AND. chkvar(cn_fe3u, cn_ve3u)
     lchkvar = lchkvar .AND. chkvar(cn_fe3v, cn_ve3v)
     IF ( lchkvar ) THEN ; PRINT *, 'no vertical metrics for vvl' ; STOP 99
     ENDIF
  ENDIF

  npiglo = getdim(cf_vtfil, cn_x)
  npjglo = getdim(cf_vtfil, cn_y)
  npk    = getdim(cf_vtfil, cn_z)
  npt    = getdim(cf_vtfil, cn_t)
  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt

  ! large fields: everything 3-D goes from the file to HBM through the page-locked ring (module cdffeed)
  nij8 = INT(npiglo, 8) * npjglo ; n38 = nij8 * npk
  lfeed = feed_wanted(4 * n38)
  ALLOCATE( e1v(npiglo,npjglo), e2u(npiglo,npjglo) )
  IF ( .NOT. lfeed ) ALLOCATE( zt(npiglo,npjglo,npk), zs(npiglo,npjglo,npk) )
  ALLOCATE( dtrpt(npiglo,npjglo), dtrps(npiglo,npjglo), dtim(npt), e31d(npk) )
  IF ( .NOT. ( lfull .OR. lfeed ) ) ALLOCATE( e3u(npiglo,npjglo,npk), e3v(npiglo,npjglo,npk) )
  CALL CreateOutput
  CALL getvar_to(cn_fhgr, cn_ve1v, 1, e1v(:,:))
  CALL getvar_to(cn_fhgr, cn_ve2u, 1, e2u(:,:))
  e31d(:)  = getvare3(cn_fzgr, cn_ve3t1d, npk)

  IF ( lfeed ) THEN
     pd_t = feed_device(4 * n38) ; pd_s = feed_device(4 * n38) ; pd_e1v = feed_device(4 * nij8) ; pd_e2u = feed_device(4 * nij8)
     CALL C_F_POINTER(pd_t, zt_d, (/ n38 /)) ; CALL C_F_POINTER(pd_s, zs_d, (/ n38 /))
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
     IF ( lfeed ) THEN
        IF ( .NOT. lfull .AND. ( jt == 1 .OR. lg_vvl ) ) THEN
           CALL feed_field(cn_fe3v, cn_ve3v, pd_e3v, npiglo, npjglo, npk, ktime=it, ldiom=.NOT. lg_vvl)
           CALL feed_field(cn_fe3u, cn_ve3u, pd_e3u, npiglo, npjglo, npk, ktime=it, ldiom=.NOT. lg_vvl)
        ENDIF
        DO jk = 1, npk
           PRINT *, 'level ', jk, ' time ', jt
        END DO
        CALL feed_field(cf_vtfil, cn_vomevt, pd_t, npiglo, npjglo, npk, ktime=jt)
        CALL feed_field(cf_vtfil, cn_vomevs, pd_s, npiglo, npjglo, npk, ktime=jt)
        CALL hip_check(cdft_vhst_transport(zt_d, zs_d, e1v_d, pd_e3v, pd_e31, npiglo, npjglo, npk, CDFT_MEM_DEVICE, dtrpt, dtrps, &
             &         CDFT_MEM_HOST, feed_stream), 'cdft_vhst_transport')
        ierr = putvar(ncout, id_varout(1), SNGL(dtrpt), 1, npiglo, npjglo, ktime=jt)
        ierr = putvar(ncout, id_varout(2), SNGL(dtrps), 1, npiglo, npjglo, ktime=jt)
        CALL feed_field(cf_vtfil, cn_vozout, pd_t, npiglo, npjglo, npk, ktime=jt)
        CALL feed_field(cf_vtfil, cn_vozous, pd_s, npiglo, npjglo, npk, ktime=jt)
        CALL hip_check(cdft_vhst_transport(zt_d, zs_d, e2u_d, pd_e3u, pd_e31, npiglo, npjglo, npk, CDFT_MEM_DEVICE, dtrpt, dtrps, &
             &         CDFT_MEM_HOST, feed_stream), 'cdft_vhst_transport')
        ierr = putvar(ncout, id_varout(3), SNGL(dtrpt), 1, npiglo, npjglo, ktime=jt)
        ierr = putvar(ncout, id_varout(4), SNGL(dtrps), 1, npiglo, npjglo, ktime=jt)
        CYCLE
     ENDIF
     pe3u = C_NULL_PTR ; pe3v = C_NULL_PTR ; pe31d = C_NULL_PTR
     IF ( lfull ) THEN
        pe31d = C_LOC(e31d)
     ELSE
        IF ( jt == 1 .OR. lg_vvl ) THEN
           CALL readvar3d(cn_fe3v, cn_ve3v, e3v, ktime=it, ldiom=.NOT. lg_vvl)      ! (every level in one read)
           CALL readvar3d(cn_fe3u, cn_ve3u, e3u, ktime=it, ldiom=.NOT. lg_vvl)
        ENDIF
        pe3u = C_LOC(e3u) ; pe3v = C_LOC(e3v)
     ENDIF
     DO jk = 1, npk
        PRINT *, 'level ', jk, ' time ', jt
     END DO
     ! meridional components with e1v, e3v
     CALL readvar3d(cf_vtfil, cn_vomevt, zt, ktime=jt)
     CALL readvar3d(cf_vtfil, cn_vomevs, zs, ktime=jt)
     CALL hip_check(cdft_vhst_transport(zt, zs, e1v, pe3v, pe31d, npiglo, npjglo, npk, CDFT_MEM_HOST, dtrpt, dtrps, &
          &         CDFT_MEM_HOST, C_NULL_PTR), 'cdft_vhst_transport')
     ierr = putvar(ncout, id_varout(1), SNGL(dtrpt), 1, npiglo, npjglo, ktime=jt)
     ierr = putvar(ncout, id_varout(2), SNGL(dtrps), 1, npiglo, npjglo, ktime=jt)
     ! zonal components with e2u, e3u
     CALL readvar3d(cf_vtfil, cn_vozout, zt, ktime=jt)
     CALL readvar3d(cf_vtfil, cn_vozous, zs, ktime=jt)
     CALL hip_check(cdft_vhst_transport(zt, zs, e2u, pe3u, pe31d, npiglo, npjglo, npk, CDFT_MEM_HOST, dtrpt, dtrps, &
          &         CDFT_MEM_HOST, C_NULL_PTR), 'cdft_vhst_transport')
     ierr = putvar(ncout, id_varout(3), SNGL(dtrpt), 1, npiglo, npjglo, ktime=jt)
     ierr = putvar(ncout, id_varout(4), SNGL(dtrps), 1, npiglo, npjglo, ktime=jt)
  END DO
  ierr = closeout(ncout)
  IF ( lfeed ) THEN
     CALL feed_release(pd_t) ; CALL feed_release(pd_s) ; CALL feed_release(pd_e1v) ; CALL feed_release(pd_e2u)
     CALL feed_release(pd_e3u) ; CALL feed_release(pd_e3v) ; CALL feed_release(pd_e31) ; CALL feed_close()
  ENDIF

CONTAINS

  SUBROUTINE CreateOutput
    !! src/cdfvhst.f90:199-236
    INTEGER(KIND=4) :: ji
    ipk(:)                    = 1
    stypvar%rmissing_value    = 0.
    stypvar%valid_min         = -100.
    stypvar%valid_max         = 100.
    stypvar%conline_operation = 'N/A'
    stypvar%caxis             = 'TYX'
    DO ji = 1, 4
       stypvar(ji)%ichunk     = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
    END DO
    stypvar(1)%cname = cn_somevt ; stypvar(2)%cname = cn_somevs ; stypvar(3)%cname = cn_sozout ; stypvar(4)%cname = cn_sozous
    stypvar(1)%cunits = 'W' ; stypvar(2)%cunits = 'kg.s-1' ; stypvar(3)%cunits = 'W' ; stypvar(4)%cunits = 'kg.s-1'
    stypvar(1)%clong_name = 'Meridional_heat_transport'
    stypvar(2)%clong_name = 'Meridional_salt_transport'
    stypvar(3)%clong_name = 'Zonal_heat_transport'
    stypvar(4)%clong_name = 'Zonal_salt_transport'
    DO ji = 1, 4
       stypvar(ji)%cshort_name = stypvar(ji)%cname
    END DO
    ncout = create      (cf_out, cf_vtfil, npiglo, npjglo, 1, ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, 4, ipk, id_varout, ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_vtfil, npiglo, npjglo, 1)
    dtim  = getvar1d(cf_vtfil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfvhst
