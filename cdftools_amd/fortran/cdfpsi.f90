PROGRAM cdfpsi
  !!======================================================================
  !!                     ***  PROGRAM  cdfpsi  ***
  !! Barotropic stream function from the vertically integrated transports.
  !!
  !! Drop-in for the reference tool (src/cdfpsi.f90): same options, same option report on stdout
  !! (:208-224), same psi.nc (sobarstf float (t,y,x), m3/s, history = command line, :444-505).  The
  !! vertical integration (:271-298), the integration along j or i (:377-402), -mean, the reference point
  !! and the SNGL/mask at output (:412-438) run on the GPU (cdft_psi_*, bit-identical: every sum keeps the
  !! reference's sequential order).
  !! -open (:316-366, psi started from a line inside the domain), --ssh-file (:260-267, :280, :299: the SSH
  !! contribution, same scans on other operands) and -lev (:282-307: one psi per level) use the staged
  !! entry points cdft_psi_integrate / _scan_u / _scan_open / _finish, the last two with a 1-level plan.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE modutils
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(C_INT)  :: ngpu, igpus(64)                          ! row blocks and their HIP devices (CDFT_GPUS, CDFT_GPU_MAP)
  INTEGER(KIND=4) :: narg, ijarg, ierr, jt, jk, it, npiglo, npjglo, npk, npt, ncout, iiref = -1, ijref = -1
  INTEGER(KIND=4) :: ipk(3), id_varout(3), nvout = 1
  REAL(KIND=4), ALLOCATABLE, TARGET :: zu(:,:,:), zv(:,:,:), e3u(:,:,:), e3v(:,:,:), e1v(:,:), e2u(:,:), zmask(:,:), e31d(:)
  REAL(KIND=4), ALLOCATABLE :: glamf(:,:), gphif(:,:), zpsi(:,:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: zssh(:,:), zsshu(:,:), zsshv(:,:), zvel1(:,:), zlev(:,:), ze3(:,:)
  REAL(KIND=8), ALLOCATABLE, TARGET :: dpsi(:,:), dpsissh(:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:)
  CHARACTER(LEN=256) :: cf_ufil = 'none', cf_vfil = 'none', cf_out = 'psi.nc', cv_out = 'sobarstf', cglobal, cldum
  CHARACTER(LEN=256) :: cf_sshfil = 'none', cv_outssh = 'sobarstfssh', cv_outotal = 'sobarstftotal'
  TYPE(variable) :: stypvar(3)
  TYPE(C_PTR) :: plan, plan1, pzu, pe2u, pe3u, pzv, pe1v, pe3v, pe31d, pmask, pmask1, pdpsi
  LOGICAL :: lchk = .FALSE., lfull = .FALSE., lmask = .FALSE., lmean = .FALSE., lopen = .FALSE., lssh = .FALSE.
  LOGICAL :: llev = .FALSE., lnc4 = .FALSE., ll_u = .TRUE., ll_v = .FALSE.
  ! the 3-D operands in HBM when they are fed there through the ring (module cdffeed); pstream: the stream the plan's calls
  ! are queued on (the feed's, so that they follow the copies)
  TYPE(C_PTR) :: pd_zu = C_NULL_PTR, pd_zv = C_NULL_PTR, pd_e3u = C_NULL_PTR, pd_e3v = C_NULL_PTR, pd_e2u = C_NULL_PTR
  TYPE(C_PTR) :: pd_e1v = C_NULL_PTR, pd_e31d = C_NULL_PTR, pd_mask = C_NULL_PTR, pstream = C_NULL_PTR
  INTEGER(C_INT)  :: nspace = CDFT_MEM_HOST
  INTEGER(KIND=8) :: nij8, n38
  LOGICAL :: lfeed = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfpsi.inc"
     STOP
  ENDIF

  CALL SetGlobalAtt(cglobal)
  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-u'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_ufil) ; ijarg = ijarg + 1
     CASE ( '-v'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vfil) ; ijarg = ijarg + 1
     CASE ( '-V'         ) ; ll_v  = .TRUE. ; ll_u = .FALSE.
     CASE ( '-full'      ) ; lfull = .TRUE.
     CASE ( '-mask'      ) ; lmask = .TRUE.
     CASE ( '-mean'      ) ; lmean = .TRUE. ; ll_v = .TRUE. ; ll_u = .TRUE.
     CASE ( '--ssh-file' ) ; lssh  = .TRUE. ; nvout = 3
        ;                    CALL GET_COMMAND_ARGUMENT(ijarg, cf_sshfil) ; ijarg = ijarg + 1
     CASE ( '-open'      ) ; lopen = .TRUE. ; ll_v = .TRUE. ; ll_u = .TRUE.
     CASE ( '-o'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out) ; ijarg = ijarg + 1
     CASE ( '-nc4'       ) ; lnc4  = .TRUE.
     CASE ( '-ref'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) iiref
        ;                    CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) ijref
     CASE ( '-vvl'       ) ; lg_vvl = .TRUE.
     CASE ( '-lev'       ) ; llev  = .TRUE. ; cv_out = 'vobarstf'
     CASE DEFAULT          ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  lchk = lchk .OR. chkfile(cn_fhgr)
  lchk = lchk .OR. chkfile(cn_fzgr)
  IF ( lmask ) lchk = lchk .OR. chkfile(cn_fmsk)
  lchk = lchk .OR. chkfile(cf_ufil)
  lchk = lchk .OR. chkfile(cf_vfil)
  IF ( lssh ) lchk = lchk .OR. chkfile(cf_sshfil)
  IF ( lchk ) STOP 99
  IF ( llev .AND. lopen ) THEN
     PRINT *, ' +++ error : cannot use both -lev and -open options.' ; STOP 99
  ENDIF
  IF ( llev .AND. lssh ) THEN
     PRINT *, ' +++ error : cannot use both -lev and -ssh options.' ; STOP 99
  ENDIF
  IF ( llev .AND. lmean ) THEN
     PRINT *, ' +++ error : cannot use both -lev and -mean options.' ; STOP 99
  ENDIF
  IF ( llev .AND. lmask ) THEN
     PRINT *, ' +++ error : cannot use both -lev and -mask options.' ; STOP 99
  ENDIF
  IF ( lg_vvl ) THEN
     cn_fe3u = cf_ufil ; cn_fe3v = cf_vfil
     cn_ve3u = cn_ve3uvvl ; cn_ve3v = cn_ve3vvvl
  ENDIF

  npiglo = getdim(cf_ufil, cn_x)
  npjglo = getdim(cf_ufil, cn_y)
  npk    = getdim(cf_ufil, cn_z)
  npt    = getdim(cf_ufil, cn_t)
  IF ( iiref == -1 .OR. ijref == -1 ) THEN
     iiref = npiglo
     ijref = npjglo
  ENDIF

  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt
  PRINT *, ' Option is use :'
  PRINT *, '    -full : ', lfull
  PRINT *, '    -mask : ', lmask
  PRINT *, '    -mean : ', lmean
  PRINT *, '    -ssh  : ', lssh
  PRINT *, '    -open : ', lopen
  PRINT *, '    -ref  : ', iiref, ijref
  PRINT *, '  U-comp  : ', ll_u
  PRINT *, '  V-comp  : ', ll_v
  PRINT *, '      -o  : ', TRIM(cf_out)
  PRINT *, '    -lnc4 : ', lnc4
  PRINT *, '    -vvl  : ', lg_vvl

  ALLOCATE( glamf(npiglo,npjglo), gphif(npiglo,npjglo), zpsi(npiglo,npjglo), dtim(npt) )
  ALLOCATE( e1v(npiglo,npjglo), e2u(npiglo,npjglo) )
  CALL hip_shard(ngpu, igpus)                                ! CDFT_GPUS / CDFT_GPU_MAP: row blocks over the node's GPUs
  lfeed = feed_wanted(4_8 * npiglo * npjglo * npk) .AND. ngpu == 1 .AND. .NOT. llev     ! one device: file -> HBM through the ring (module cdffeed)
  nij8 = INT(npiglo, 8) * npjglo ; n38 = nij8 * npk
  IF ( ll_u .AND. .NOT. ( llev .OR. lfeed ) ) ALLOCATE( zu(npiglo,npjglo,npk) )      ! -lev works level by level
  IF ( ll_v .AND. .NOT. ( llev .OR. lfeed ) ) ALLOCATE( zv(npiglo,npjglo,npk) )
  IF ( lfull ) THEN ; ALLOCATE( e31d(npk) )
  ELSE IF ( .NOT. ( llev .OR. lfeed ) ) THEN
     IF ( ll_u ) ALLOCATE( e3u(npiglo,npjglo,npk) )
     IF ( ll_v ) ALLOCATE( e3v(npiglo,npjglo,npk) )
  ENDIF

  CALL getvar_to(cn_fhgr, cn_glamf, 1, glamf(:,:))
  CALL getvar_to(cn_fhgr, cn_gphif, 1, gphif(:,:))
  CALL CreateOutput
  CALL getvar_to(cn_fhgr, cn_ve1v, 1, e1v(:,:))
  CALL getvar_to(cn_fhgr, cn_ve2u, 1, e2u(:,:))
  pmask = C_NULL_PTR
  IF ( lmask ) THEN
     ALLOCATE( zmask(npiglo,npjglo) )
     CALL getvar_to(cn_fmsk, cn_fmask, 1, zmask(:,:))
     WHERE ( zmask >= 2 ) zmask = 1                       ! get rid of the free-slip / no-slip coding
     pmask = C_LOC(zmask)
  ENDIF
  pe31d = C_NULL_PTR
  IF ( lfull ) THEN
     e31d(:) = getvare3(cn_fzgr, cn_ve3t1d, npk)
     pe31d = C_LOC(e31d)
  ENDIF

  IF ( lfeed ) THEN                                          ! the 2-D operands and the 3-D buffers, in HBM
     nspace = CDFT_MEM_DEVICE
     IF ( ll_u ) THEN
        pd_e2u = feed_device(4 * nij8) ; CALL feed_host(C_LOC(e2u), 4 * nij8, pd_e2u) ; pd_zu = feed_device(4 * n38)
        IF ( .NOT. lfull ) pd_e3u = feed_device(4 * n38)
     ENDIF
     IF ( ll_v ) THEN
        pd_e1v = feed_device(4 * nij8) ; CALL feed_host(C_LOC(e1v), 4 * nij8, pd_e1v) ; pd_zv = feed_device(4 * n38)
        IF ( .NOT. lfull ) pd_e3v = feed_device(4 * n38)
     ENDIF
     IF ( lfull ) THEN ; pd_e31d = feed_device(4_8 * npk) ; CALL feed_host(pe31d, 4_8 * npk, pd_e31d) ; pe31d = pd_e31d ; ENDIF
     IF ( lmask ) THEN ; pd_mask = feed_device(4 * nij8) ; CALL feed_host(pmask, 4 * nij8, pd_mask) ; pmask = pd_mask ; ENDIF
     pstream = feed_stream
  ENDIF
  IF ( ngpu > 1 .AND. .NOT. lopen ) THEN                     ! (-open integrates along one row of the whole domain)
     CALL hip_check(cdft_psi_create_sharded(plan, npiglo, npjglo, npk, ngpu, igpus), 'cdft_psi_create_sharded')
  ELSE
     CALL hip_check(cdft_psi_create(plan, npiglo, npjglo, npk), 'cdft_psi_create')
  ENDIF
  IF ( lssh .OR. llev ) CALL hip_check(cdft_psi_create(plan1, npiglo, npjglo, 1), 'cdft_psi_create')
  IF ( lssh ) THEN
     ALLOCATE( zssh(npiglo,npjglo), zsshu(npiglo,npjglo), zsshv(npiglo,npjglo), zvel1(npiglo,npjglo) )
     ALLOCATE( dpsi(npiglo,npjglo), dpsissh(npiglo,npjglo) )
  ENDIF
  IF ( llev ) ALLOCATE( zlev(npiglo,npjglo), ze3(npiglo,npjglo) )
  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     IF ( llev ) THEN                                     ! src/cdfpsi.f90:282-315: one psi per level, first record only
        DO jk = 1, npk
           IF ( ll_v ) CALL level_psi(cf_vfil, cn_vomecrty, cn_fe3v, cn_ve3v, .FALSE.)
           IF ( ll_u ) CALL level_psi(cf_ufil, cn_vozocrtx, cn_fe3u, cn_ve3u, .TRUE. )
        END DO
        ierr = closeout(ncout)
        PRINT *, ' cdfpsi_level completed.'
        STOP 0
     ENDIF
     pzu = C_NULL_PTR ; pe2u = C_NULL_PTR ; pe3u = C_NULL_PTR ; pzv = C_NULL_PTR ; pe1v = C_NULL_PTR ; pe3v = C_NULL_PTR
     IF ( ll_v .AND. lfeed ) THEN                         ! slab by slab into HBM, the copy of one under the read of the next
        CALL feed_field(cf_vfil, cn_vomecrty, pd_zv, npiglo, npjglo, npk, ktime=jt)
        IF ( .NOT. lfull .AND. ( jt == 1 .OR. lg_vvl ) ) CALL feed_field(cn_fe3v, cn_ve3v, pd_e3v, npiglo, npjglo, npk, ktime=it, &
             &                                                             ldiom=.NOT. lg_vvl)
        pzv = pd_zv ; pe1v = pd_e1v ; pe3v = pd_e3v
     ELSE IF ( ll_v ) THEN
        CALL readvar3d(cf_vfil, cn_vomecrty, zv, ktime=jt)
        pzv = C_LOC(zv) ; pe1v = C_LOC(e1v)
        IF ( .NOT. lfull ) THEN
           IF ( jt == 1 .OR. lg_vvl ) THEN
              CALL readvar3d(cn_fe3v, cn_ve3v, e3v, ktime=it, ldiom=.NOT. lg_vvl)      ! (every level in one read)
           ENDIF
           pe3v = C_LOC(e3v)
        ENDIF
     ENDIF
     IF ( ll_u .AND. lfeed ) THEN
        CALL feed_field(cf_ufil, cn_vozocrtx, pd_zu, npiglo, npjglo, npk, ktime=jt)
        IF ( .NOT. lfull .AND. ( jt == 1 .OR. lg_vvl ) ) CALL feed_field(cn_fe3u, cn_ve3u, pd_e3u, npiglo, npjglo, npk, ktime=it, &
             &                                                             ldiom=.NOT. lg_vvl)
        pzu = pd_zu ; pe2u = pd_e2u ; pe3u = pd_e3u
     ELSE IF ( ll_u ) THEN
        CALL readvar3d(cf_ufil, cn_vozocrtx, zu, ktime=jt)
        pzu = C_LOC(zu) ; pe2u = C_LOC(e2u)
        IF ( .NOT. lfull ) THEN
           IF ( jt == 1 .OR. lg_vvl ) THEN
              CALL readvar3d(cn_fe3u, cn_ve3u, e3u, ktime=it, ldiom=.NOT. lg_vvl)
           ENDIF
           pe3u = C_LOC(e3u)
        ENDIF
     ENDIF
     pdpsi = C_NULL_PTR ; IF ( lssh ) pdpsi = C_LOC(dpsi)
     CALL hip_check(cdft_psi_integrate(plan, pzu, pe2u, pe3u, pzv, pe1v, pe3v, pe31d, nspace, pstream), 'cdft_psi_integrate')
     CALL scan_and_finish(plan, 0_C_INT, pdpsi, pmask, nspace, pstream)   ! dpsi = dpsi - dpsi(iiref,ijref), SNGL, mask (:423-438)
     IF ( lmask ) THEN ; PRINT *, ' Write masked BSF'
     ELSE              ; PRINT *, ' Write BSF'
     ENDIF
     ierr = putvar(ncout, id_varout(1), zpsi, 1, npiglo, npjglo, ktime=jt)
     IF ( lssh ) THEN                                     ! :260-267, :280, :299, :323-333, :386-408
        CALL getvar_to(cf_sshfil, cn_sossheig, 1, zssh(:,:), ktime=jt)
        zsshu(:,:) = 0.0 ; zsshv(:,:) = 0.0
        zsshu(1:npiglo-1, :) = 0.5*( zssh(2:npiglo,:) + zssh(1:npiglo-1,:) )
        zsshv(:, 1:npjglo-1) = 0.5*( zssh(:,2:npjglo) + zssh(:,1:npjglo-1) )
        ! the reference multiplies BOTH contributions by zv(:,:), the meridional velocity of level 1
        ! (:299 uses zv where zu was meant); zv is never assigned without the V component: taken as 0
        zvel1(:,:) = 0.0
        IF ( ll_v .AND. lfeed ) THEN ; CALL getvar_to(cf_vfil, cn_vomecrty, 1, zvel1(:,:), ktime=jt)
        ELSE IF ( ll_v )        THEN ; zvel1(:,:) = zv(:,:,1)
        ENDIF
        pzu = C_NULL_PTR ; pe2u = C_NULL_PTR ; pe3u = C_NULL_PTR ; pzv = C_NULL_PTR ; pe1v = C_NULL_PTR ; pe3v = C_NULL_PTR
        IF ( ll_u ) THEN ; pzu = C_LOC(zvel1) ; pe2u = C_LOC(e2u) ; pe3u = C_LOC(zsshu) ; ENDIF
        IF ( ll_v ) THEN ; pzv = C_LOC(zvel1) ; pe1v = C_LOC(e1v) ; pe3v = C_LOC(zsshv) ; ENDIF
        CALL hip_check(cdft_psi_integrate(plan1, pzu, pe2u, pe3u, pzv, pe1v, pe3v, C_NULL_PTR, CDFT_MEM_HOST, C_NULL_PTR), &
             &         'cdft_psi_integrate')
        pmask1 = C_NULL_PTR ; IF ( lmask ) pmask1 = C_LOC(zmask)
        CALL scan_and_finish(plan1, 1_C_INT, C_LOC(dpsissh), pmask1, CDFT_MEM_HOST, C_NULL_PTR)   ! the SSH part is not referenced to (iiref,ijref)
        ierr = putvar(ncout, id_varout(2), zpsi, 1, npiglo, npjglo, ktime=jt)
        IF ( lmask ) THEN ; zpsi(:,:) = SNGL(dpsissh + dpsi) * zmask(:,:)
        ELSE              ; zpsi(:,:) = SNGL(dpsissh + dpsi)
        ENDIF
        ierr = putvar(ncout, id_varout(3), zpsi, 1, npiglo, npjglo, ktime=jt)
     ENDIF
  END DO
  IF ( lssh .OR. llev ) ierr = cdft_psi_destroy(plan1)
  ierr = cdft_psi_destroy(plan)
  IF ( lfeed ) THEN
     CALL feed_release(pd_zu) ; CALL feed_release(pd_zv) ; CALL feed_release(pd_e3u) ; CALL feed_release(pd_e3v)
     CALL feed_release(pd_e2u) ; CALL feed_release(pd_e1v) ; CALL feed_release(pd_e31d) ; CALL feed_release(pd_mask)
     CALL feed_close()
  ENDIF
  ierr = closeout(ncout)

CONTAINS

  SUBROUTINE scan_and_finish(kplan, knoref, pd8, pd_msk, kmskspace, pd_stream)
    !! scans of the transports held by kplan (closed domain or -open), then reference point, SNGL and mask
    !! into zpsi; knoref = 1: no reference point (value 0 is subtracted); pd8: optional r8 copy of psi; pd_msk: the mask
    !! (NULL: none) in memory space kmskspace; pd_stream: the stream of this plan's calls
    TYPE(C_PTR),    INTENT(in) :: kplan, pd8, pd_msk, pd_stream
    INTEGER(C_INT), INTENT(in) :: knoref, kmskspace
    IF ( lopen ) THEN
       CALL hip_check(cdft_psi_scan_open(kplan, MERGE(1_C_INT, 0_C_INT, lmean), pd_stream), 'cdft_psi_scan_open')
    ELSE
       CALL hip_check(cdft_psi_scan_u(kplan, C_NULL_PTR, CDFT_MEM_HOST, 1_C_INT, C_NULL_PTR, CDFT_MEM_HOST, pd_stream), &
            &         'cdft_psi_scan_u')
    ENDIF
    CALL hip_check(cdft_psi_finish(kplan, knoref, 0.d0, INT(iiref, C_INT), INT(ijref, C_INT), pd_msk, kmskspace, zpsi, &
         &         pd8, CDFT_MEM_HOST, pd_stream), 'cdft_psi_finish')
  END SUBROUTINE scan_and_finish

  SUBROUTINE level_psi(cd_file, cd_var, cd_fe3, cd_ve3, ld_u)
    !! -lev: transport of level jk alone, scanned and written without reference point (:275-307)
    CHARACTER(LEN=*), INTENT(in) :: cd_file, cd_var, cd_fe3, cd_ve3
    LOGICAL,          INTENT(in) :: ld_u
    TYPE(C_PTR) :: pl_e3, pl_1d
    CALL getvar_to(cd_file, cd_var, jk, zlev(:,:), ktime=jt)
    pl_e3 = C_NULL_PTR ; pl_1d = C_NULL_PTR
    IF ( lfull ) THEN ; pl_1d = C_LOC(e31d(jk))
    ELSE
       CALL getvar_to(cd_fe3, cd_ve3, jk, ze3(:,:), ktime=it, ldiom=.NOT. lg_vvl)
       pl_e3 = C_LOC(ze3)
    ENDIF
    IF ( ld_u ) THEN
       CALL hip_check(cdft_psi_integrate(plan1, C_LOC(zlev), C_LOC(e2u), pl_e3, C_NULL_PTR, C_NULL_PTR, C_NULL_PTR, pl_1d, &
            &         CDFT_MEM_HOST, C_NULL_PTR), 'cdft_psi_integrate')
    ELSE
       CALL hip_check(cdft_psi_integrate(plan1, C_NULL_PTR, C_NULL_PTR, C_NULL_PTR, C_LOC(zlev), C_LOC(e1v), pl_e3, pl_1d, &
            &         CDFT_MEM_HOST, C_NULL_PTR), 'cdft_psi_integrate')
    ENDIF
    pmask1 = C_NULL_PTR ; IF ( lmask ) pmask1 = C_LOC(zmask)
    CALL scan_and_finish(plan1, 1_C_INT, C_NULL_PTR, pmask1, CDFT_MEM_HOST, C_NULL_PTR)
    ! the reference creates the file with ONE level (:497) and ignores the status of the writes that
    ! do not fit: only level 1 reaches the file there, and here
    ierr = putvar(ncout, id_varout(1), zpsi, jk, npiglo, npjglo, ktime=jt)
  END SUBROUTINE level_psi

  SUBROUTINE CreateOutput
    !! src/cdfpsi.f90:444-505
    INTEGER(KIND=4) :: ji
    ipk(:) = 1
    IF ( llev ) ipk(1) = npk
    stypvar(:)%cunits            = 'm3/s'
    stypvar(:)%valid_min         = -300.e6
    stypvar(:)%valid_max         = 300.e6
    stypvar(:)%conline_operation = 'N/A'
    stypvar(:)%caxis             = 'TYX'
    DO ji = 1, 3
       stypvar(ji)%ichunk        = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
    END DO
    stypvar(1)%cname             = cv_out
    stypvar(1)%rmissing_value    = 0.
    stypvar(1)%clong_name        = 'Barotropic_Stream_Function'
    IF ( llev ) stypvar(1)%clong_name = 'Barotropic_Stream_Function_per_level'
    stypvar(1)%cshort_name       = cv_out
    stypvar(2)%cname             = cv_outssh
    stypvar(2)%rmissing_value    = 0.
    stypvar(2)%clong_name        = 'Barotropic_Stream_Function SSH contribution'
    stypvar(2)%cshort_name       = cv_outssh
    stypvar(3)%cname             = cv_outotal
    stypvar(3)%rmissing_value    = 0.
    stypvar(3)%clong_name        = 'Barotropic_Stream_Function SSH total'
    stypvar(3)%cshort_name       = cv_outotal
    ncout = create      (cf_out, cf_ufil, npiglo, npjglo, 1, ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, nvout, ipk, id_varout, cdglobal=TRIM(cglobal), ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_ufil, npiglo, npjglo, 1, glamf, gphif)
    dtim  = getvar1d(cf_ufil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfpsi
