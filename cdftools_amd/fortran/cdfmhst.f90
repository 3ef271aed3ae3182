PROGRAM cdfmhst
  !!======================================================================
  !!                     ***  PROGRAM  cdfmhst  ***
  !! Meridional heat (and salt) transport, global and per basin, from V.T and V.S.
  !!
  !! Drop-in for the reference tool (src/cdfmhst.f90): same options (-vt | -v -t [-s], -vtvar, -MST, -full,
  !! -Zdim, -b, -o, -vvl), same mhst.nc (zomht_* [zomst_*] float, PW / T/sec, fill 9999.99, :504-550) and
  !! the same two ASCII files zonal_heat_trp.dat / zonal_salt_trp.dat (:449-492).  The vertical
  !! integration and the masked zonal sums (:308-367) run on the GPU (cdft_mhst_*).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  REAL(KIND=4), PARAMETER :: ppspval = 9999.99
  INTEGER(C_INT)  :: ngpu, igpus(64)                          ! row blocks and their HIP devices (CDFT_GPUS, CDFT_GPU_MAP)
  INTEGER(KIND=4) :: narg, ijarg, ierr, jt, jk, jj, jvar, jb, it, ik0, ivar, iloc(2)
  INTEGER(KIND=4) :: npiglo, npjglo, npk, npko, npt, nbasins, nbasinso, npvar = 1, ncout, nmask, nlevo
  INTEGER(KIND=4) :: numouth = 10, numouts = 11
  INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e1v(:,:), e3v(:,:,:), e31d(:), zmasks(:,:,:), zvt(:,:,:), zvs(:,:,:), zt(:,:,:)
  REAL(KIND=4), ALLOCATABLE :: gphiv(:,:), rdumlon(:,:), rdumlat(:,:), gdep(:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:), dheat(:,:,:), dsalt(:,:,:), dmtrp(:), dz(:,:)
  CHARACTER(LEN=256) :: cf_vtfil = 'none', cf_vfil = 'none', cf_tfil = 'none', cf_sfil = 'none'
  CHARACTER(LEN=256) :: cf_outh = 'zonal_heat_trp.dat', cf_outs = 'zonal_salt_trp.dat', cf_outnc = 'mhst.nc'
  CHARACTER(LEN=256) :: cv_zomht = 'zomht', cv_zomst = 'zomst', cldimension, cldum
  CHARACTER(LEN=5)   :: cbasin(6) = (/ '_glo ', '_atl ', '_inp ', '_ind ', '_pac ', '_inp0' /)
  INTEGER(KIND=4) :: jbasins                               ! (the usage text lists the basins' variables)
  TYPE(variable), ALLOCATABLE :: stypvar(:)
  TYPE(C_PTR) :: plan, pe3v, pe31d, p1, p2, p3, p4, p5
  LOGICAL :: llglo, lchk = .FALSE., lfull = .FALSE., lsepf = .TRUE., lzdim = .FALSE.
  ! the operands in HBM when they are fed there through the ring (module cdffeed): never touched by the host
  TYPE(C_PTR) :: pd_a = C_NULL_PTR, pd_b = C_NULL_PTR, pd_c = C_NULL_PTR, pd_e3 = C_NULL_PTR, pd_e31d = C_NULL_PTR
  TYPE(C_PTR) :: pd_e1 = C_NULL_PTR, pd_msk = C_NULL_PTR
  REAL(KIND=4), POINTER :: e1_d(:), msk_d(:)
  INTEGER(KIND=8) :: nij8, n38
  TYPE(C_PTR) :: pstream = C_NULL_PTR
  LOGICAL :: lfeed = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfmhst.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-vt'    ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vtfil ) ; ijarg = ijarg + 1
     CASE ( '-vtvar' ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cn_vomevt) ; ijarg = ijarg + 1
        ;                CALL GET_COMMAND_ARGUMENT(ijarg, cn_vomevs) ; ijarg = ijarg + 1
     CASE ( '-v'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vfil  ) ; ijarg = ijarg + 1
     CASE ( '-t'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_tfil  ) ; ijarg = ijarg + 1
     CASE ( '-s'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_sfil  ) ; ijarg = ijarg + 1
     CASE ( '-MST'   ) ; npvar = 2
     CASE ( '-full'  ) ; lfull = .TRUE.
     CASE ( '-Zdim'  ) ; lzdim = .TRUE.
     CASE ( '-b'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cn_fbasins) ; ijarg = ijarg + 1
     CASE ( '-o'     ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_outnc ) ; ijarg = ijarg + 1
     CASE ( '-vvl'   ) ; lg_vvl = .TRUE.
     CASE DEFAULT      ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  IF ( TRIM(cf_vtfil) /= 'none' ) lsepf = .FALSE.
  IF ( lsepf ) THEN
     lchk = chkfile(cf_vfil)
     lchk = lchk .OR. chkfile(cf_tfil)
     IF ( lchk ) STOP 99
     IF ( TRIM(cf_sfil) == 'none' ) cf_sfil = cf_tfil
  ENDIF
  lchk = lchk .OR. chkfile(cn_fhgr)
  lchk = lchk .OR. chkfile(cn_fzgr)
  lchk = lchk .OR. chkfile(cn_fmsk)
  lchk = lchk .OR. chkfile(cf_vtfil)
  IF ( lsepf ) THEN
     lchk = lchk .OR. chkfile(cf_tfil)
     lchk = lchk .OR. chkfile(cf_sfil)
  ENDIF
  IF ( lchk ) STOP 99
  IF ( lsepf ) cf_vtfil = cf_vfil                          ! dimensions (and e3v with -vvl) come from there
  IF ( lg_vvl ) THEN
     cn_fe3v = cf_vtfil
     cn_ve3v = cn_ve3vvvl
  ENDIF

  IF ( .NOT. chkfile(cn_fbasins) ) THEN
     llglo = .TRUE. ; nbasins = 5 ; nbasinso = 6 ; nmask = 4
  ELSE
     PRINT *, ' Only compute for GLOBAL basin'
     llglo = .FALSE. ; nbasins = 1 ; nbasinso = 1 ; nmask = 1
  ENDIF

  npiglo = getdim(cf_vtfil, cn_x)
  npjglo = getdim(cf_vtfil, cn_y)
  npk    = getdim(cf_vtfil, cn_z)
  npt    = getdim(cf_vtfil, cn_t)
  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt
  IF ( lzdim ) THEN ; npko = npk ; cldimension = 'TZY'
  ELSE              ; npko = 1   ; cldimension = 'TY'
  ENDIF
  nlevo = npko

  CALL hip_shard(ngpu, igpus)                                ! CDFT_GPUS / CDFT_GPU_MAP: row blocks over the node's GPUs
  lfeed = feed_wanted(4_8 * npiglo * npjglo * npk) .AND. ngpu == 1                      ! one device: file -> HBM through the ring (module cdffeed)
  nij8 = INT(npiglo, 8) * npjglo ; n38 = nij8 * npk
  ALLOCATE( e1v(npiglo,npjglo), gphiv(npiglo,npjglo), zmasks(npiglo,npjglo,nmask) )
  IF ( lfull ) ALLOCATE( e31d(npk) )
  IF ( .NOT. lfeed ) THEN
     ALLOCATE( zvt(npiglo,npjglo,npk), zvs(npiglo,npjglo,npk) )
     IF ( lsepf ) ALLOCATE( zt(npiglo,npjglo,npk) )
     IF ( .NOT. lfull ) ALLOCATE( e3v(npiglo,npjglo,npk) )
  ENDIF
  ALLOCATE( rdumlon(1,npjglo), rdumlat(1,npjglo), gdep(npko), dtim(npt), dmtrp(npjglo), dz(npjglo,6) )
  ALLOCATE( dheat(nmask,npjglo,nlevo), dsalt(nmask,npjglo,nlevo) )
  ALLOCATE( stypvar(nbasinso*npvar), ipk(nbasinso*npvar), id_varout(nbasinso*npvar) )

  CALL getvar_to(cn_fhgr, cn_ve1v, 1, e1v(:,:))
  CALL getvar_to(cn_fhgr, cn_gphiv, 1, gphiv(:,:))
  IF ( lzdim ) THEN ; gdep(:) = getvare3(cn_fzgr, cn_gdept, npk)
  ELSE              ; gdep(:) = 0.
  ENDIF
  IF ( lfull ) e31d = getvare3(cn_fzgr, cn_ve3t1d, npk)
  iloc = MAXLOC(gphiv)
  rdumlat(1,:) = gphiv(iloc(1),:)
  rdumlon(:,:) = 0.
  CALL CreateOutput

  ! masks: slab 1 global (vmask level 1), then Atlantic, Pacific, Indian (src/cdfmhst.f90:340-363)
  CALL getvar_to(cn_fmsk, cn_vmask, 1, zmasks(:,:,1))
  IF ( llglo ) THEN
     CALL getvar_to(cn_fbasins, cn_tmaskatl, 1, zmasks(:,:,2))
     CALL getvar_to(cn_fbasins, cn_tmaskpac, 1, zmasks(:,:,3))
     CALL getvar_to(cn_fbasins, cn_tmaskind, 1, zmasks(:,:,4))
  ENDIF

  OPEN(numouth, FILE=cf_outh, FORM='FORMATTED', RECL=256)
  OPEN(numouts, FILE=cf_outs, FORM='FORMATTED', RECL=256)
  IF ( lfeed ) THEN
     pd_e1 = feed_device(4 * nij8) ; pd_msk = feed_device(4 * nij8 * nmask) ; pd_a = feed_device(4 * n38) ; pd_b = feed_device(4 * n38)
     IF ( lsepf ) pd_c = feed_device(4 * n38)
     CALL C_F_POINTER(pd_e1, e1_d, (/ nij8 /)) ; CALL C_F_POINTER(pd_msk, msk_d, (/ nij8 * nmask /))
     CALL feed_host(C_LOC(e1v), 4 * nij8, pd_e1) ; CALL feed_host(C_LOC(zmasks), 4 * nij8 * nmask, pd_msk)
     IF ( lfull ) THEN ; pd_e31d = feed_device(4_8 * npk) ; CALL feed_host(C_LOC(e31d), 4_8 * npk, pd_e31d)
     ELSE              ; pd_e3 = feed_device(4 * n38)
     ENDIF
  ENDIF
  IF ( ngpu > 1 ) THEN
     CALL hip_check(cdft_mhst_create_sharded(plan, npiglo, npjglo, npk, nmask, ngpu, igpus), 'cdft_mhst_create_sharded')
  ELSE
     CALL hip_check(cdft_mhst_create(plan, npiglo, npjglo, npk, nmask), 'cdft_mhst_create')
  ENDIF

  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     IF ( lfeed .AND. ( jt == 1 .OR. lg_vvl ) ) THEN     ! the metric slab by slab into HBM, the mesh set from there
        IF ( .NOT. lfull ) CALL feed_field(cn_fe3v, cn_ve3v, pd_e3, npiglo, npjglo, npk, ktime=it, ldiom=.NOT. lg_vvl)
        CALL hip_check(cdft_mhst_set_mesh(plan, e1_d, pd_e3, pd_e31d, msk_d, CDFT_MEM_DEVICE, feed_stream), 'cdft_mhst_set_mesh')
     ELSE IF ( jt == 1 .OR. lg_vvl ) THEN
        pe3v = C_NULL_PTR ; pe31d = C_NULL_PTR
        IF ( lfull ) THEN
           pe31d = C_LOC(e31d)
        ELSE
           CALL readvar3d(cn_fe3v, cn_ve3v, e3v, ktime=it, ldiom=.NOT. lg_vvl)      ! (every level in one read)
           pe3v = C_LOC(e3v)
        ENDIF
        CALL hip_check(cdft_mhst_set_mesh(plan, e1v, pe3v, pe31d, zmasks, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_mhst_set_mesh')
     ENDIF
     DO jk = 1, npk
        PRINT *, 'level ', jk
     END DO
     p1 = C_NULL_PTR ; p2 = C_NULL_PTR ; p3 = C_NULL_PTR ; p4 = C_NULL_PTR ; p5 = C_NULL_PTR
     IF ( lfeed ) pstream = feed_stream
     IF ( lfeed ) THEN                                     ! the records slab by slab into HBM
        IF ( lsepf ) THEN
           CALL feed_field(cf_vfil, cn_vomecrty, pd_a, npiglo, npjglo, npk, ktime=jt)
           CALL feed_field(cf_tfil, cn_votemper, pd_b, npiglo, npjglo, npk, ktime=jt)
           CALL feed_field(cf_sfil, cn_vosaline, pd_c, npiglo, npjglo, npk, ktime=jt)
           p3 = pd_a ; p4 = pd_b ; p5 = pd_c
        ELSE
           CALL feed_field(cf_vtfil, cn_vomevt, pd_a, npiglo, npjglo, npk, ktime=jt)
           CALL feed_field(cf_vtfil, cn_vomevs, pd_b, npiglo, npjglo, npk, ktime=jt)
           p1 = pd_a ; p2 = pd_b
        ENDIF
     ELSE IF ( lsepf ) THEN                                ! V, T, S: the kernel forms 0.5*(t(j)+t(j+1))*v
        CALL readvar3d(cf_vfil, cn_vomecrty, zvt, ktime=jt)
        CALL readvar3d(cf_tfil, cn_votemper, zt, ktime=jt)
        CALL readvar3d(cf_sfil, cn_vosaline, zvs, ktime=jt)
        p3 = C_LOC(zvt) ; p4 = C_LOC(zt) ; p5 = C_LOC(zvs)
     ELSE
        CALL readvar3d(cf_vtfil, cn_vomevt, zvt, ktime=jt)
        CALL readvar3d(cf_vtfil, cn_vomevs, zvs, ktime=jt)
        p1 = C_LOC(zvt) ; p2 = C_LOC(zvs)
     ENDIF
     CALL hip_check(cdft_mhst_compute(plan, p1, p2, p3, p4, p5, MERGE(CDFT_MEM_DEVICE, CDFT_MEM_HOST, lfeed),                &
          &         MERGE(1_C_INT, 0_C_INT, lzdim), dheat, dsalt, CDFT_MEM_HOST, pstream), 'cdft_mhst_compute')

     DO ik0 = 1, nlevo                                     ! src/cdfmhst.f90:389-445
        DO jvar = 1, npvar
           IF ( jvar == 1 ) THEN ; CALL basin_columns(dheat(:,:,ik0), 1.d15)
           ELSE                  ; CALL basin_columns(dsalt(:,:,ik0), 1.d6)
           ENDIF
           DO jb = 1, nbasinso
              ivar = (jvar - 1) * nbasinso + jb
              dmtrp(:) = dz(:,jb)
              WHERE ( dmtrp == 0 ) dmtrp = ppspval
              ierr = putvar(ncout, id_varout(ivar), REAL(dmtrp), ik0, 1, npjglo, ktime=jt)
           END DO
        END DO
     END DO

     ! ASCII dumps of the totals (last level), north to south (:449-492)
     WRITE(numouth,*) '! Zonal heat transport (integrated alon I-model coordinate) (in Pw)'
     CALL ascii_dump(numouth, dheat(:,:,nlevo), 1.d15, 9000)
     WRITE(numouts,*) ' ! Zonal salt transport (integrated alon I-model coordinate) (in 10^6 kg/s)'
     CALL ascii_dump(numouts, dsalt(:,:,nlevo), 1.d6, 9001)
  END DO
  ierr = cdft_mhst_destroy(plan)
  IF ( lfeed ) THEN
     CALL feed_release(pd_a) ; CALL feed_release(pd_b) ; CALL feed_release(pd_c) ; CALL feed_release(pd_e3)
     CALL feed_release(pd_e31d) ; CALL feed_release(pd_e1) ; CALL feed_release(pd_msk)
     CALL feed_close()
  ENDIF
  ierr = closeout(ncout)
  CLOSE(numouth)
  CLOSE(numouts)

CONTAINS

  SUBROUTINE basin_columns(dsum, ddiv)
    !! glo, atl, inp = (ind + pac), ind, pac, inp0 = (glo - atl), each divided by ddiv (:395-443)
    REAL(KIND=8), INTENT(in) :: dsum(:,:), ddiv
    dz(:,1) = dsum(1,:) / ddiv
    IF ( nbasinso == 6 ) THEN
       dz(:,2) = dsum(2,:) / ddiv
       dz(:,3) = (dsum(4,:) + dsum(3,:)) / ddiv
       dz(:,4) = dsum(4,:) / ddiv
       dz(:,5) = dsum(3,:) / ddiv
       dz(:,6) = (dsum(1,:) - dsum(2,:)) / ddiv
    ENDIF
  END SUBROUTINE basin_columns

  SUBROUTINE ascii_dump(knum, dsum, ddiv, kfmt)
    INTEGER(KIND=4), INTENT(in) :: knum, kfmt
    REAL(KIND=8), INTENT(in) :: dsum(:,:), ddiv
    IF ( llglo ) THEN
       ! (the two files differ by a blank in front of the column titles: src/cdfmhst.f90:451 and :474)
       IF ( kfmt == 9000 ) THEN
          WRITE(knum,*) '! J        Global          Atlantic         Pacific          Indian           Mediteranean     Austral  '
       ELSE
          WRITE(knum,*) ' ! J        Global          Atlantic         Pacific          Indian           Mediteranean     Austral  '
       ENDIF
       WRITE(knum,*) ' ! time : ', jt
       DO jj = npjglo, 1, -1
          IF ( kfmt == 9000 ) THEN
             WRITE(knum, 9000) jj, rdumlat(1,jj), dsum(1,jj)/ddiv, dsum(2,jj)/ddiv, dsum(3,jj)/ddiv, dsum(4,jj)/ddiv, 0.d0, 0.d0
          ELSE
             WRITE(knum, 9001) jj, rdumlat(1,jj), dsum(1,jj)/ddiv, dsum(2,jj)/ddiv, dsum(3,jj)/ddiv, dsum(4,jj)/ddiv, 0.d0, 0.d0
          ENDIF
       END DO
    ELSE
       IF ( kfmt == 9000 ) THEN ; WRITE(knum,*) '! J        Global        '          ! (:464)
       ELSE                     ; WRITE(knum,*) ' J        Global  '                 ! (:487)
       ENDIF
       WRITE(knum,*) ' ! time : ', jt
       DO jj = npjglo, 1, -1
          IF ( kfmt == 9000 ) THEN ; WRITE(knum, 9000) jj, rdumlat(1,jj), dsum(1,jj)/ddiv
          ELSE                     ; WRITE(knum, 9001) jj, rdumlat(1,jj), dsum(1,jj)/ddiv
          ENDIF
       END DO
    ENDIF
9000 FORMAT(I4,6(1x,f9.3,1x,f8.4))
9001 FORMAT(I4,6(1x,f9.2,1x,f9.3))
  END SUBROUTINE ascii_dump

  SUBROUTINE CreateOutput
    !! src/cdfmhst.f90:504-550
    INTEGER(KIND=4) :: jbasins, iv
    ipk(:) = npko
    DO jbasins = 1, nbasinso
       stypvar(jbasins)%cname             = TRIM(cv_zomht)//TRIM(cbasin(jbasins))
       stypvar(jbasins)%cunits            = 'PW'
       stypvar(jbasins)%rmissing_value    = ppspval
       stypvar(jbasins)%valid_min         = -10.
       stypvar(jbasins)%valid_max         = 20
       stypvar(jbasins)%clong_name        = 'Meridional Heat Transport '//TRIM(cbasin(jbasins))
       stypvar(jbasins)%cshort_name       = stypvar(jbasins)%cname
       stypvar(jbasins)%conline_operation = 'N/A'
       stypvar(jbasins)%caxis             = cldimension
       IF ( npvar == 2 ) THEN
          iv = nbasinso + jbasins
          stypvar(iv)%cname             = TRIM(cv_zomst)//TRIM(cbasin(jbasins))
          stypvar(iv)%cunits            = 'T/sec'
          stypvar(iv)%rmissing_value    = ppspval
          stypvar(iv)%valid_min         = -10.e9
          stypvar(iv)%valid_max         = 20.e9
          stypvar(iv)%clong_name        = 'Meridional Salt Transport '//TRIM(cbasin(jbasins))
          stypvar(iv)%cshort_name       = stypvar(iv)%cname
          stypvar(iv)%conline_operation = 'N/A'
          stypvar(iv)%caxis             = cldimension
       ENDIF
    END DO
    ncout = create      (cf_outnc, cf_vtfil, 1, npjglo, npko, cdep='depthv')
    ierr  = createvar   (ncout, stypvar, nbasinso*npvar, ipk, id_varout)
    ierr  = putheadervar(ncout, cf_vtfil, 1, npjglo, npko, pnavlon=rdumlon, pnavlat=rdumlat, pdep=gdep)
    dtim  = getvar1d(cf_vtfil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfmhst
