PROGRAM cdfmoc
  !!======================================================================
  !!                     ***  PROGRAM  cdfmoc  ***
  !! Meridional overturning stream function, global and (with new_maskglo.nc) per ocean basin.
  !!
  !! Drop-in for the reference tool (src/cdfmoc.f90): same command line, same auxiliary files
  !! (mesh_hgr.nc, mesh_zgr.nc, mask.nc, new_maskglo.nc through modcdfnames), same stdout lines for the
  !! dimensions, same moc.nc (dims x=1,y,depthw,time_counter; zomsfglo [atl inp ind pac inp0] float,
  !! units Sverdrup, :1006-1188).  The two hot loops (:355-379 zonal sum, :382-388 vertical cumulated
  !! sum) run on the GPU through cdft_moc_* (libcdftools_hip.so); several records are sent per call so
  !! that the time-invariant weights are read once for all of them.
  !! -decomp (:390-517: geostrophic-shear, barotropic and ageostrophic parts) runs on the GPU as well
  !! (cdft_moc_decomp_*), one record per call.  -rapid (:598-1004: the RAPID-MOCHA diagnostics of one (i,k)
  !! section at 26.5 N) has nothing data-parallel in it and is host code as in the reference
  !! (module cdfmoc_rapid, cdf_findij from module cdftools).
  !! Host feed (module cdffeed): on one device, without -decomp / -full, nothing 3-D is held in host memory -- the metric,
  !! its mask and every velocity record go from the file to HBM slab by slab through a page-locked ring, the copy of one
  !! slab under the read of the next, so a record costs its read, not its read plus its transfer (CDFT_FEED=0: the old way).
  !! With the feed on nothing 3-D is held in HBM either (-vvl: a window holds the metric and the velocity of one record): the STREAM route (run_stream below) takes the domain
  !! through HBM a window of levels at a time -- the window's e3v, vmask and the same levels of EVERY record of the run, each
  !! (jj,jk) row's metric and records in one run -- and one launch per window streams all the records past the row's weights:
  !! every byte read from the files is read from HBM once, 4*(1+1/npt) bytes per cell and record as on bench.py's resident
  !! pool, with 1.6 GB of HBM instead of 61 GB at eORCA12 x 12 (allocating HBM costs about what reading the file does).
  !! CDFT_MOC_ROUTE = stream (default) | pool (all records resident in the plan's pool, ONE launch: bench.py's very call)
  !! | record (a launch per record on dense arrays, the weights read again for each).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE eos
  USE cdftools_hip
  USE cdffeed
  USE cdfmoc_rapid
  IMPLICIT NONE

  INTEGER(C_INT)  :: ngpu, igpus(64)                          ! row blocks and their HIP devices (CDFT_GPUS, CDFT_GPU_MAP)
  INTEGER(KIND=4) :: narg, ijarg, ierr, ii, iv
  INTEGER(KIND=4) :: npiglo, npjglo, npk, npt, nbasins, nbasinso, nvarout
  INTEGER(KIND=4) :: ji, jj, jt, jk, jbasin, jr, it0, nrec, nbatch, ijvar, ncout
  INTEGER(KIND=4) :: npglo = 1, npatl = 2, npinp = 3, npind = 4, nppac = 5
  INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:)
  INTEGER(KIND=4) :: iloc(2)
  INTEGER(KIND=2), ALLOCATABLE, TARGET :: ibmask(:,:,:)         ! (nbasins, npiglo, npjglo)
  INTEGER(KIND=1), ALLOCATABLE :: ivmask(:,:,:)                 ! vmask of every level, as bytes
  REAL(KIND=4),    ALLOCATABLE, TARGET :: e1v(:,:)
  REAL(KIND=4),    ALLOCATABLE :: gphiv(:,:), e3v(:,:,:), zv(:,:,:,:), zmsk(:,:), zbas(:,:,:)
  REAL(KIND=4),    ALLOCATABLE :: rdumlon(:,:), rdumlat(:,:), gdepw(:), e31d(:)
  REAL(KIND=8),    ALLOCATABLE :: dtim(:), dmoc(:,:,:,:)          ! (nbasins, npjglo, npk, nbatch)
  REAL(KIND=8) :: dl_budget, dl_t(5) = 0.d0                     ! CDFT_TIMING: mesh reads, data reads, GPU calls, output
  INTEGER(KIND=8) :: iclk0, iclk1, iclk_rate
  LOGICAL :: ltiming = .FALSE.
  ! -decomp
  INTEGER(KIND=1), ALLOCATABLE, TARGET :: itmask(:,:,:), iumask(:,:,:)   ! (npiglo, npjglo, npk) 0/1
  REAL(KIND=4),    ALLOCATABLE, TARGET :: e1u(:,:), zcoef(:,:), gdept(:), ztemp(:,:,:), zsal(:,:,:)
  REAL(KIND=8),    ALLOCATABLE :: dmoc_sh(:,:,:), dmoc_bt(:,:,:), dmoc_ag(:,:,:)   ! (nbasins, npjglo, npk)
  REAL(KIND=4) :: rpi, grav = 9.81, rau0 = 1025.
  CHARACTER(LEN=256) :: cf_vfil = 'none', cf_tfil = 'none', cf_sfil = 'none', cf_ufil = 'none'
  CHARACTER(LEN=256) :: cf_moc = 'moc.nc', cglobal, cldum
  TYPE(variable), ALLOCATABLE :: stypvar(:)
  TYPE(C_PTR) :: plan
  TYPE(C_PTR) :: pd_zv = C_NULL_PTR, pd_e3v = C_NULL_PTR, pd_vm = C_NULL_PTR, pd_e1v = C_NULL_PTR, pd_ib = C_NULL_PTR
  TYPE(C_PTR) :: pd_t = C_NULL_PTR, pd_s = C_NULL_PTR     ! -decomp: T and S of the record in HBM
  REAL(KIND=4), POINTER :: zt_d(:), zs_d(:)
  LOGICAL :: lfeed_dec
  REAL(KIND=4), POINTER :: zv_d(:), e3v_d(:), e1v_d(:)       ! the same device arrays seen as arrays (never touched by the host)
  INTEGER(KIND=1), POINTER :: ivm_d(:)
  INTEGER(KIND=2), POINTER :: ib_d(:)
  INTEGER(KIND=8) :: nij8, n38
  LOGICAL :: lfeed
  LOGICAL :: lpool = .FALSE.                                  ! the records of a batch go into the plan's pool: one zonal launch for all
  LOGICAL :: lstream = .FALSE.                                ! the stream route: windows of levels, nothing 3-D resident
  INTEGER(KIND=4) :: nlevw = 1                                ! levels per window
  CHARACTER(LEN=16) :: croute
  INTEGER(C_SIZE_T) :: ifree, itotal
  INTEGER(C_INT64_T) :: ilaunch, irecs
  REAL(KIND=8) :: dl_field
  CHARACTER(KIND=C_CHAR) :: cl_plan(512)
  CHARACTER(LEN=512) :: cplan
  LOGICAL :: lbas, lfull = .FALSE., lchk = .FALSE., ldec = .FALSE., lrap = .FALSE., ll_teos10 = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()

  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfmoc.inc"
     STOP
  ENDIF

  cglobal = 'Partial step computation'
  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-v'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vfil) ; ijarg = ijarg + 1
     CASE ( '-t'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_tfil) ; ijarg = ijarg + 1
     CASE ( '-s'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_sfil) ; ijarg = ijarg + 1
     CASE ( '-u'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_ufil) ; ijarg = ijarg + 1
     CASE ( '-o'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_moc ) ; ijarg = ijarg + 1
     CASE ( '-full'   ) ; lfull  = .TRUE. ; cglobal = 'Full step computation'
     CASE ( '-decomp' ) ; ldec   = .TRUE.
     CASE ( '-rapid'  ) ; lrap   = .TRUE.
     CASE ( '-vvl'    ) ; lg_vvl = .TRUE.
     CASE ( '-teos10' ) ; ll_teos10 = .TRUE.
     CASE DEFAULT       ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  CALL eos_init( ll_teos10 )
  IF ( cf_sfil == 'none' ) cf_sfil = cf_tfil

  lchk = lchk .OR. chkfile( cn_fhgr )
  lchk = lchk .OR. chkfile( cn_fzgr )
  lchk = lchk .OR. chkfile( cn_fmsk )
  lchk = lchk .OR. chkfile( cf_vfil )
  IF ( ldec ) lchk = lchk .OR. chkfile( cf_tfil ) .OR. chkfile( cf_sfil )
  IF ( lrap ) lchk = lchk .OR. chkfile( cf_tfil ) .OR. chkfile( cf_sfil ) .OR. chkfile( cf_ufil )
  IF ( lchk ) STOP 99                                     ! missing file(s)
  IF ( lg_vvl ) THEN
     cn_fe3v = cf_vfil
     cn_ve3v = cn_ve3vvvl
  ENDIF
  IF ( lrap ) THEN
     CALL rapid_amoc(cf_vfil, cf_tfil, cf_sfil, cf_ufil, cf_moc, lfull)      ! all of it in a separate routine
     STOP 99                                              ! the reference ends here, with this code (:255)
  ENDIF

  npiglo = getdim(cf_vfil, cn_x)
  npjglo = getdim(cf_vfil, cn_y)
  npk    = getdim(cf_vfil, cn_z)
  npt    = getdim(cf_vfil, cn_t)

  PRINT *, 'Working with cdfmoc ...'
  PRINT *, '  npiglo =', npiglo
  PRINT *, '  npjglo =', npjglo
  PRINT *, '  npk    =', npk
  PRINT *, '  npt    =', npt

  lbas = .NOT. chkfile( cn_fbasins )                      ! sub-basin file present ?
  IF ( lbas ) THEN ; nbasins = 5 ; nbasinso = 6
  ELSE             ; nbasins = 1 ; nbasinso = 1
  ENDIF
  nvarout = nbasinso
  IF ( ldec ) nvarout = nbasinso * 4                      ! total, _sh, _bt, _ag (src/cdfmoc.f90:276)

  ! records sent to the GPU per call: as many as fit a host budget (default 8 GB, CDFT_HOST_GB)
  dl_budget = 8.d0
  CALL GET_ENVIRONMENT_VARIABLE('CDFT_HOST_GB', cldum)
  IF ( LEN_TRIM(cldum) > 0 ) READ(cldum, *) dl_budget
  nbatch = MAX(1, MIN(npt, INT(dl_budget * 1.d9 / (4.d0 * npiglo * npjglo * npk))))
  CALL GET_ENVIRONMENT_VARIABLE('CDFT_TIMING', cldum)
  ltiming = LEN_TRIM(cldum) > 0
  CALL SYSTEM_CLOCK(iclk0, iclk_rate)
  IF ( lg_vvl ) nbatch = 1                                ! the metric changes with every record
  IF ( ldec   ) nbatch = 1                                ! cdft_moc_decomp works record by record

  CALL cdf_phase('start -> arguments, dimensions')
  CALL hip_shard(ngpu, igpus)                                ! CDFT_GPUS / CDFT_GPU_MAP: row blocks over the node's GPUs
  lfeed = feed_wanted(4_8 * npiglo * npjglo * npk) .AND. ngpu == 1 .AND. .NOT. ( ldec .OR. lfull )
  IF ( lfeed ) nbatch = 1                                 ! a record is computed on as soon as its last slab has left
  ! ... unless the run's records fit the device beside the weights: then every slab goes from the ring straight into the
  ! plan's pool (feed_moc_record: the H2D copy is the staging) and ONE zonal launch streams all the records of a batch past
  ! the weights (cdft_moc_compute_staged) -- 4*(1+1/npt) bytes per cell and record instead of 8 (DESIGN 4.1).  The batch is
  ! what HBM holds: the pool is (1 + nbatch) fields, beside the plan's weights and the mesh fields on their way through.
  ! CDFT_MOC_BATCH=n sets it (1: the record-by-record way).
  croute = 'stream'
  CALL GET_ENVIRONMENT_VARIABLE('CDFT_MOC_ROUTE', cldum)
  IF ( LEN_TRIM(cldum) > 0 ) croute = cldum(1:16)
  IF ( lg_vvl .AND. croute == 'pool' ) croute = 'stream'     ! (a pool holds ONE set of weights: with -vvl every record has its own)
  IF ( lfeed .AND. croute /= 'record' ) THEN
     CALL hip_check(cdft_mem_info(ifree, itotal), 'cdft_mem_info')
     dl_field = 4.d0 * ( ( npiglo + 3 ) / 4 * 4 ) * REAL(npjglo, 8) * npk
     IF ( croute == 'pool' ) THEN                          ! every record of a batch resident: (1 + nbatch) fields + weights + mesh
        nbatch = MIN(npt, INT(MIN(1.d6, ( 0.92d0 * REAL(ifree, 8) - 2.3d0 * dl_field ) / dl_field)) - 1)
     ELSE                                                  ! a window of nlevw levels of 1 + nbatch slabs in half of what is free
        nlevw  = MAX(1, MIN(npk - 1, feed_levels(npiglo, npjglo, npk)))
        nbatch = MIN(npt, INT(MIN(1.d6, 0.5d0 * REAL(ifree, 8) / ( dl_field / npk * nlevw ))) - 1)
     ENDIF
     CALL GET_ENVIRONMENT_VARIABLE('CDFT_MOC_BATCH', cldum)
     IF ( LEN_TRIM(cldum) > 0 ) THEN
        READ(cldum, *) nbatch ; nbatch = MIN(npt, nbatch)
     ENDIF
     nbatch = MAX(1, nbatch)
     IF ( lg_vvl ) nbatch = 1                             ! the metric changes with every record: a window holds e3v and v of ONE
     IF ( croute /= 'pool' ) THEN
        ! ... and as many levels per window as 4 GB hold: a launch on 2 levels of eORCA12 is 7 000 workgroups, under five rounds
        ! of what the chip holds at once (0.65 of the HBM peak, profiles/r05), on 5 levels twelve rounds; more HBM than that
        ! buys nothing -- an allocation costs up to 50 ms per GB on this pool, when the driver still has the pages of an earlier
        ! process to clear (profiles/r05/alloc_probe.json: 61 GB in 0.0002 .. 6.3 s), the file read 30 ms per GB.
        ! CDFT_MOC_WINDOW=n sets the levels per window.
        nlevw = MAX(nlevw, MIN(npk - 1, INT(MIN(4.d9, 0.125d0 * REAL(ifree, 8)) / ( dl_field / npk * ( 1 + nbatch ) ))))
        CALL GET_ENVIRONMENT_VARIABLE('CDFT_MOC_WINDOW', cldum)
        IF ( LEN_TRIM(cldum) > 0 ) THEN
           READ(cldum, *) nlevw ; nlevw = MAX(1, MIN(npk - 1, nlevw))
        ENDIF
     ENDIF
     lpool   = croute == 'pool' .AND. nbatch >= 2
     lstream = croute /= 'pool' .AND. nbatch >= 1
     nbatch = MAX(1, nbatch)
  ENDIF
  ! -decomp: V, T and S of a record go through the ring too (the mesh of the decomposition stays a host array: once per run)
  lfeed_dec = ldec .AND. .NOT. lfull .AND. ngpu == 1 .AND. feed_wanted(4_8 * npiglo * npjglo * npk)
  nij8 = INT(npiglo, 8) * npjglo ; n38 = nij8 * npk

  CALL cdf_phase('device count, memory info (the context comes up), route')
  ALLOCATE( ibmask(nbasins, npiglo, npjglo), zmsk(npiglo, npjglo) )
  ALLOCATE( e1v(npiglo, npjglo), gphiv(npiglo, npjglo), dmoc(nbasins, npjglo, npk, nbatch) )
  IF ( .NOT. ( lfeed .OR. lfeed_dec ) ) ALLOCATE( e3v(npiglo, npjglo, npk), zv(npiglo, npjglo, npk, nbatch) )
  ALLOCATE( rdumlon(1, npjglo), rdumlat(1, npjglo), gdepw(npk), e31d(npk), dtim(npt) )
  ALLOCATE( stypvar(nvarout), ipk(nvarout), id_varout(nvarout) )

  CALL getvar_to(cn_fhgr, cn_ve1v, 1, e1v(:,:))
  CALL getvar_to(cn_fhgr, cn_gphiv, 1, gphiv(:,:))
  gdepw(:)   = getvare3(cn_fzgr, cn_gdepw, npk)
  gdepw(:)   = -1. * gdepw(:)
  IF ( lfull ) e31d(:) = getvare3(cn_fzgr, cn_ve3t1d, npk)
  IF ( ldec ) THEN
     ALLOCATE( e1u(npiglo, npjglo), zcoef(npiglo, npjglo), gdept(npk) )
     IF ( .NOT. lfeed_dec ) ALLOCATE( itmask(npiglo, npjglo, npk), iumask(npiglo, npjglo, npk) )
     IF ( .NOT. lfeed_dec ) ALLOCATE( ztemp(npiglo, npjglo, npk), zsal(npiglo, npjglo, npk) )
     ALLOCATE( dmoc_sh(nbasins, npjglo, npk), dmoc_bt(nbasins, npjglo, npk), dmoc_ag(nbasins, npjglo, npk) )
     gdept(:) = getvare3(cn_fzgr, cn_gdept, npk)
     CALL getvar_to(cn_fhgr, cn_ve1u, 1, e1u(:,:))
     ! f at V points, then -g/rau0/f, in r4 as the reference writes it (src/cdfmoc.f90:425-433)
     rpi = ACOS( -1. )
     zcoef(:,:) = 2*2*rpi/( 24.0 * 3600. )* SIN ( rpi * gphiv(:,:) /180.0)
     WHERE ( zcoef /= 0 )
        zcoef(:,:) = -grav/ rau0 / zcoef(:,:)
     ELSEWHERE
        zcoef(:,:) = 0.
     END WHERE
     IF ( .NOT. lfeed_dec ) THEN                            ! (through the feed: file -> HBM, below)
        CALL readvar3d_i1(cn_fmsk, cn_tmask, itmask)        ! (bytes in the file, bytes here: a level at a time through getvar's r4
        CALL readvar3d_i1(cn_fmsk, cn_umask, iumask)        !  slab and back was 3 of the 3.7 s this program spent before its first record at eORCA12)
     ENDIF
     dmoc_sh(:,:,:) = 0.d0            ! the reference never clears it: a fresh allocation (zero pages) for
  ENDIF                               ! the first record, then each record accumulates onto the last (:471)

  CALL cdf_phase('allocations, e1v gphiv gdepw read (+ -decomp: masks, e1u)')
  iloc = MAXLOC(gphiv)
  rdumlat(1,:) = gphiv(iloc(1),:)
  rdumlon(:,:) = 0.

  CALL cdf_phase('MAXLOC(gphiv)')
  CALL CreateOutput

  CALL cdf_phase('CreateOutput')
  ! basin masks, basin index fastest (src/cdfmoc.f90:325-336)
  ! (the same values as the reference's five strided array assignments and its WHERE, in ONE pass over the points with the basin
  !  index innermost, as ibmask lies in memory: 0.13 s instead of 0.33 s of host time at eORCA12 on the build container's cores)
  IF ( lbas ) THEN
     ALLOCATE( zbas(npiglo, npjglo, 3) )
     CALL getvar_to(cn_fmsk, cn_vmask, 1, zmsk(:,:))
     CALL getvar_to(cn_fbasins, cn_tmaskatl, 1, zbas(:,:,1))
     CALL getvar_to(cn_fbasins, cn_tmaskind, 1, zbas(:,:,2))
     CALL getvar_to(cn_fbasins, cn_tmaskpac, 1, zbas(:,:,3))
     DO jj = 1, npjglo
        DO ji = 1, npiglo
           ibmask(npglo,ji,jj) = INT(zmsk(ji,jj), 2)
           ibmask(npatl,ji,jj) = INT(zbas(ji,jj,1), 2)
           ibmask(npind,ji,jj) = INT(zbas(ji,jj,2), 2)
           ibmask(nppac,ji,jj) = INT(zbas(ji,jj,3), 2)
           ibmask(npinp,ji,jj) = ibmask(nppac,ji,jj) + ibmask(npind,ji,jj)
           IF ( ibmask(npinp,ji,jj) > 0 ) ibmask(npinp,ji,jj) = 1
        END DO
     END DO
     DEALLOCATE( zbas )
  ELSE
     CALL getvar_to(cn_fmsk, cn_vmask, 1, zmsk(:,:)); ibmask(npglo,:,:) = INT(zmsk, 2)
  ENDIF
  ibmask(npglo, 1,      :) = 0                            ! E-W periodic overlap, global basin only
  ibmask(npglo, npiglo, :) = 0

  CALL cdf_phase('basin masks read and built')
  IF ( ngpu > 1 .AND. .NOT. ldec ) THEN                      ! (-decomp stays on one device: its stencil crosses rows)
     CALL hip_check(cdft_moc_create_sharded(plan, npiglo, npjglo, npk, nbasins, ngpu, igpus), 'cdft_moc_create_sharded')
  ELSE
     CALL hip_check(cdft_moc_create(plan, npiglo, npjglo, npk, nbasins), 'cdft_moc_create')
  ENDIF
  IF ( lpool ) THEN                                        ! the one large allocation first, while HBM is in one piece
     CALL feed_open()
     IF ( cdft_moc_reserve(plan, nbatch, feed_stream) /= CDFT_OK ) THEN
        WRITE(0, '(a,i0,a)') ' cdfmoc: no room in HBM for a pool of ', nbatch, ' records: working record by record'
        lpool = .FALSE. ; nbatch = 1
     ENDIF
  ENDIF

  CALL cdf_phase('plan')
  it0 = 1
  IF ( lstream ) THEN
     CALL run_stream
     it0 = npt + 1                                        ! (the record loop below has nothing left to do)
  ENDIF
  DO WHILE ( it0 <= npt )
     nrec = MIN(nbatch, npt - it0 + 1)
     IF ( it0 == 1 .OR. lg_vvl ) THEN                     ! e3v masked by vmask, every level (get_e3v)
        CALL tick(1)                                         ! what came before: coordinates, masks, output file
        IF ( lfeed ) THEN                                ! file -> HBM through the ring, get_e3v's product on the device
           IF ( it0 == 1 ) THEN
              pd_e1v = feed_device(4 * nij8) ; pd_ib = feed_device(2 * nbasins * nij8)
              pd_vm  = feed_device(n38)      ; pd_e3v = feed_device(4 * n38)
              CALL C_F_POINTER(pd_e1v, e1v_d, (/ nij8 /))   ; CALL C_F_POINTER(pd_ib, ib_d, (/ nbasins * nij8 /))
              CALL C_F_POINTER(pd_vm, ivm_d, (/ n38 /))     ; CALL C_F_POINTER(pd_e3v, e3v_d, (/ n38 /))
              IF ( .NOT. lpool ) THEN                       ! (the pool route has no dense record in HBM)
                 pd_zv = feed_device(4 * n38) ; CALL C_F_POINTER(pd_zv, zv_d, (/ n38 /))
              ENDIF
              CALL feed_host(C_LOC(e1v), 4 * nij8, pd_e1v)
              CALL feed_host(C_LOC(ibmask), 2 * nbasins * nij8, pd_ib)
              CALL feed_mask(cn_fmsk, cn_vmask, pd_vm, npiglo, npjglo, npk)
           ENDIF
           CALL feed_field(cn_fe3v, cn_ve3v, pd_e3v, npiglo, npjglo, npk, ktime=it0, ldiom=.NOT. lg_vvl)
           CALL tick(2)
           CALL hip_check(cdft_moc_set_mesh_vmask(plan, e1v_d, e3v_d, ivm_d, ib_d, CDFT_MEM_DEVICE, feed_stream), &
                &         'cdft_moc_set_mesh_vmask')
           CALL hip_check(cdft_stream_sync(feed_stream), 'cdft_stream_sync')
           IF ( .NOT. lg_vvl ) THEN                       ! the plan keeps its own weights and classes: these can go
              CALL feed_release(pd_e3v) ; CALL feed_release(pd_vm) ; CALL feed_release(pd_e1v) ; CALL feed_release(pd_ib)
           ENDIF
        ELSE IF ( lfeed_dec ) THEN                       ! -decomp on large fields: the mesh of both plans file -> HBM, nothing 3-D here
           CALL set_mesh_decomp_fed(it0)
        ELSE IF ( ldec .OR. lfull ) THEN                 ! the masked metric is needed on the host as well
           CALL get_e3v_all(it0)
           CALL tick(2)
           CALL hip_check(cdft_moc_set_mesh(plan, e1v, e3v, ibmask, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_moc_set_mesh')
        ELSE                                              ! raw metric + byte mask: get_e3v's product on the device
           IF ( .NOT. ALLOCATED(ivmask) ) ALLOCATE( ivmask(npiglo, npjglo, npk) )
           CALL readvar3d_i1(cn_fmsk, cn_vmask, ivmask)
           CALL readvar3d(cn_fe3v, cn_ve3v, e3v, ktime=it0, ldiom=.NOT. lg_vvl)
           CALL tick(2)
           CALL hip_check(cdft_moc_set_mesh_vmask(plan, e1v, e3v, ivmask, ibmask, CDFT_MEM_HOST, C_NULL_PTR), &
                &         'cdft_moc_set_mesh_vmask')
        ENDIF
        CALL tick(4)
        IF ( ldec .AND. .NOT. lfeed_dec ) CALL hip_check(cdft_moc_decomp_set_mesh(plan, e3v, e1u, zcoef, gdept, C_LOC(itmask), C_LOC(iumask), &
             &                     CDFT_MEM_HOST, C_NULL_PTR), 'cdft_moc_decomp_set_mesh')
     ENDIF
     IF ( ldec ) THEN                                     ! one record: total + the three components
        jt = it0
        IF ( lfeed_dec ) THEN                             ! the three fields leave for HBM slab by slab as they are read
           IF ( .NOT. C_ASSOCIATED(pd_t) ) THEN
              pd_zv = feed_device(4 * n38) ; pd_t = feed_device(4 * n38) ; pd_s = feed_device(4 * n38)
              CALL C_F_POINTER(pd_zv, zv_d, (/ n38 /)) ; CALL C_F_POINTER(pd_t, zt_d, (/ n38 /)) ; CALL C_F_POINTER(pd_s, zs_d, (/ n38 /))
           ENDIF
           CALL feed_field(cf_vfil, cn_vomecrty, pd_zv, npiglo, npjglo, npk, ktime=jt)
           CALL feed_field(cf_tfil, cn_votemper, pd_t,  npiglo, npjglo, npk, ktime=jt)
           CALL feed_field(cf_sfil, cn_vosaline, pd_s,  npiglo, npjglo, npk, ktime=jt)
           CALL hip_check(cdft_moc_decomp(plan, zv_d, zt_d, zs_d, MERGE(CDFT_EOS_TEOS10, CDFT_EOS_80, ll_teos10), CDFT_MEM_DEVICE, &
                &         dmoc, dmoc_sh, dmoc_bt, dmoc_ag, CDFT_MEM_HOST, feed_stream), 'cdft_moc_decomp')
        ELSE
        CALL readvar3d(cf_vfil, cn_vomecrty, zv(:,:,:,1), ktime=jt)
        CALL readvar3d(cf_tfil, cn_votemper, ztemp(:,:,:), ktime=jt)
        CALL readvar3d(cf_sfil, cn_vosaline, zsal(:,:,:), ktime=jt)
        CALL hip_check(cdft_moc_decomp(plan, zv, ztemp, zsal, MERGE(CDFT_EOS_TEOS10, CDFT_EOS_80, ll_teos10), CDFT_MEM_HOST, &
             &         dmoc, dmoc_sh, dmoc_bt, dmoc_ag, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_moc_decomp')
        ENDIF
        ijvar = 1                                         ! output order of src/cdfmoc.f90:520-569
        DO jbasin = 1, nbasins
           CALL put4(dmoc(jbasin,:,:,1), dmoc_sh(jbasin,:,:), dmoc_bt(jbasin,:,:), dmoc_ag(jbasin,:,:))
        END DO
        IF ( lbas ) CALL put4(dmoc(npglo,:,:,1) - dmoc(npatl,:,:,1), dmoc_sh(npglo,:,:) - dmoc_sh(npatl,:,:), &
             &                dmoc_bt(npglo,:,:) - dmoc_bt(npatl,:,:), dmoc_ag(npglo,:,:) - dmoc_ag(npatl,:,:))
        it0 = it0 + 1
        CYCLE
     ENDIF
     CALL tick(5)
     IF ( lpool ) THEN                                    ! the batch's slabs leave for their pool slots as they are read
        CALL hip_check(cdft_moc_stage_open(plan, nrec, feed_stream), 'cdft_moc_stage_open')
        DO jr = 1, nrec                                   ! levels 1 .. npk-1, as the reference reads them (:357)
           CALL feed_moc_record(cf_vfil, cn_vomecrty, plan, jr, npiglo, npjglo, npk - 1, it0 + jr - 1)
        END DO
        CALL hip_check(cdft_moc_stage_close(plan), 'cdft_moc_stage_close')
        CALL tick(3)
        CALL hip_check(cdft_moc_compute_staged(plan, dmoc, CDFT_MEM_HOST, feed_stream), 'cdft_moc_compute_staged')
     ELSE IF ( lfeed ) THEN                               ! the record's slabs leave for HBM as they are read
        CALL feed_field(cf_vfil, cn_vomecrty, pd_zv, npiglo, npjglo, npk, ktime=it0)
        CALL tick(3)
        CALL hip_check(cdft_moc_compute(plan, zv_d, INT(n38, C_INT64_T), 1, CDFT_MEM_DEVICE, dmoc, CDFT_MEM_HOST, feed_stream), &
             &         'cdft_moc_compute')
     ELSE
        DO jr = 1, nrec                                   ! velocities of levels 1..npk-1 are used
           CALL readvar3d(cf_vfil, cn_vomecrty, zv(:,:,:,jr), ktime=it0 + jr - 1)
        END DO
        CALL tick(3)
        CALL hip_check(cdft_moc_compute(plan, zv, INT(npiglo, C_INT64_T) * npjglo * npk, nrec, CDFT_MEM_HOST, dmoc, &
             &                          CDFT_MEM_HOST, C_NULL_PTR), 'cdft_moc_compute')
     ENDIF
     CALL tick(4)
     CALL put_batch
     it0 = it0 + nrec
  END DO

  IF ( ltiming ) THEN                                      ! which kernel read which layout, in how many launches (stderr)
     ierr = cdft_moc_describe(plan, cl_plan, 512_C_SIZE_T)
     ierr = cdft_moc_stats(plan, ilaunch, irecs)
     cplan = ' '
     DO ii = 1, 512
        IF ( cl_plan(ii) == C_NULL_CHAR ) EXIT
        cplan(ii:ii) = cl_plan(ii)
     END DO
     WRITE(0, '(a,a,a,i0,a,i0)') ' CDFT_PLAN cdfmoc: ', TRIM(cplan), ' zonal_launches=', ilaunch, ' records=', irecs
  ENDIF
  ierr = cdft_moc_destroy(plan)
  IF ( lfeed_dec .AND. C_ASSOCIATED(pd_t) ) THEN
     CALL feed_release(pd_zv) ; CALL feed_release(pd_t) ; CALL feed_release(pd_s) ; CALL feed_close()
  ENDIF
  IF ( lfeed .AND. .NOT. lstream ) THEN
     CALL feed_release(pd_zv) ; CALL feed_release(pd_e3v) ; CALL feed_release(pd_vm) ; CALL feed_release(pd_e1v) ; CALL feed_release(pd_ib)
     CALL feed_close()
  ENDIF
  ierr = closeout(ncout)
  CALL tick(5)
  IF ( ltiming ) PRINT '(a,5(a,f8.3))', ' CDFT_TIMING (s):', ' setup', dl_t(1), '  mesh reads', dl_t(2), '  data reads', dl_t(3), &
       &                                 '  GPU calls (H2D + kernels + D2H)', dl_t(4), '  output', dl_t(5)

CONTAINS

  SUBROUTINE put_batch
    !! the nrec records of dmoc, from record it0 on, into moc.nc (src/cdfmoc.f90:520-551)
    DO jr = 1, nrec
       jt = it0 + jr - 1
       ijvar = 1
       DO jbasin = 1, nbasins
          DO jk = 1, npk
             ierr = putvar(ncout, id_varout(ijvar), REAL(dmoc(jbasin,:,jk,jr)), jk, 1, npjglo, ktime=jt)
          END DO
          ijvar = ijvar + 1
       END DO
       IF ( lbas ) THEN                                  ! net Indo-Pacific = global - Atlantic
          DO jk = 1, npk
             ierr = putvar(ncout, id_varout(ijvar), REAL(dmoc(npglo,:,jk,jr) - dmoc(npatl,:,jk,jr)), jk, 1, npjglo, ktime=jt)
          END DO
       ENDIF
    END DO
  END SUBROUTINE put_batch

  SUBROUTINE run_stream
    !! The stream route.  The reference's loops are record { level { read zv(:,:) ; basin { jj { ji } } } } (:338-379); a sum
    !! over ji of one (jj,jk) row of one record depends on nothing else, so here the order is batch of records { window of
    !! levels { read e3v, vmask of the window ; record { read the window's levels } ; ONE launch } }, then the cumulated sums
    !! (:382-388) of every record of the batch and their output.  Each file byte is read once per batch (the metric once per
    !! batch: once per run when the records fit one batch), each HBM byte once.
    INTEGER(KIND=4) :: ik0, inl
    CALL tick(1)
    CALL hip_check(cdft_moc_set_basins(plan, e1v, ibmask, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_moc_set_basins')
    CALL feed_open()
    DO WHILE ( it0 <= npt )
       nrec = MIN(nbatch, npt - it0 + 1)
       CALL hip_check(cdft_moc_stream_begin(plan, nrec, nlevw, feed_stream), 'cdft_moc_stream_begin')
       DO ik0 = 1, npk - 1, nlevw                           ! levels 1 .. npk-1, as the reference reads them (:355)
          inl = MIN(nlevw, npk - ik0)
          CALL feed_moc_window_mask(cn_fmsk, cn_vmask, plan, npiglo, npjglo, ik0, inl)
          CALL feed_moc_window(cn_fe3v, cn_ve3v, plan, 0, npiglo, npjglo, ik0, inl, ktime=MERGE(it0, 1, lg_vvl), ldiom=.NOT. lg_vvl)
          DO jr = 1, nrec
             CALL feed_moc_window(cf_vfil, cn_vomecrty, plan, jr, npiglo, npjglo, ik0, inl, ktime=it0 + jr - 1)
          END DO
          CALL hip_check(cdft_moc_stream_compute(plan, INT(ik0 - 1, C_INT), INT(inl, C_INT), 1_C_INT, feed_stream), 'cdft_moc_stream_compute')
       END DO
       CALL tick(3)
       CALL hip_check(cdft_moc_stream_end(plan, dmoc, CDFT_MEM_HOST, feed_stream), 'cdft_moc_stream_end')
       CALL tick(4)
       CALL put_batch
       CALL tick(5)
       it0 = it0 + nrec
    END DO
    CALL feed_close()
  END SUBROUTINE run_stream

  SUBROUTINE tick(kphase)
    !! the time since the last call goes to phase kphase
    INTEGER(KIND=4), INTENT(in) :: kphase
    CALL SYSTEM_CLOCK(iclk1)
    dl_t(kphase) = dl_t(kphase) + REAL(iclk1 - iclk0, 8) / REAL(iclk_rate, 8)
    iclk0 = iclk1
  END SUBROUTINE tick

  SUBROUTINE put4(dd_tot, dd_sh, dd_bt, dd_ag)
    !! one basin: total, _sh, _bt, _ag, every level, as REAL() (src/cdfmoc.f90:522-545)
    REAL(KIND=8), DIMENSION(npjglo,npk), INTENT(in) :: dd_tot, dd_sh, dd_bt, dd_ag
    INTEGER(KIND=4) :: ik
    DO ik = 1, npk ; ierr = putvar(ncout, id_varout(ijvar), REAL(dd_tot(:,ik)), ik, 1, npjglo, ktime=jt) ; END DO
    ijvar = ijvar + 1
    DO ik = 1, npk ; ierr = putvar(ncout, id_varout(ijvar), REAL(dd_sh(:,ik)), ik, 1, npjglo, ktime=jt) ; END DO
    ijvar = ijvar + 1
    DO ik = 1, npk ; ierr = putvar(ncout, id_varout(ijvar), REAL(dd_bt(:,ik)), ik, 1, npjglo, ktime=jt) ; END DO
    ijvar = ijvar + 1
    DO ik = 1, npk ; ierr = putvar(ncout, id_varout(ijvar), REAL(dd_ag(:,ik)), ik, 1, npjglo, ktime=jt) ; END DO
    ijvar = ijvar + 1
  END SUBROUTINE put4

  SUBROUTINE set_mesh_decomp_fed(kt)
    !! cdft_moc_set_mesh + cdft_moc_decomp_set_mesh of record kt with every 3-D array going from its file to HBM through the
    !! ring: e3v and vmask (get_e3v's product formed there, cdft_moc_mask_e3v), tmask, umask; the plan keeps its own copies.
    !! At eORCA12 the host way (whole fields read, multiplied and handed over from pageable arrays) was 3 of the program's 4.6 s.
    INTEGER(KIND=4), INTENT(in) :: kt
    TYPE(C_PTR) :: pd_tm, pd_um, pd_e1u, pd_zc
    REAL(KIND=4), POINTER :: e1u_d(:), zc_d(:)
    pd_e1v = feed_device(4 * nij8) ; pd_ib = feed_device(2 * nbasins * nij8) ; pd_vm = feed_device(n38) ; pd_e3v = feed_device(4 * n38)
    pd_tm  = feed_device(n38) ; pd_um = feed_device(n38) ; pd_e1u = feed_device(4 * nij8) ; pd_zc = feed_device(4 * nij8)
    CALL C_F_POINTER(pd_e1v, e1v_d, (/ nij8 /)) ; CALL C_F_POINTER(pd_ib, ib_d, (/ nbasins * nij8 /))
    CALL C_F_POINTER(pd_e3v, e3v_d, (/ n38 /))  ; CALL C_F_POINTER(pd_e1u, e1u_d, (/ nij8 /)) ; CALL C_F_POINTER(pd_zc, zc_d, (/ nij8 /))
    CALL feed_host(C_LOC(e1v), 4 * nij8, pd_e1v) ; CALL feed_host(C_LOC(ibmask), 2 * nbasins * nij8, pd_ib)
    CALL feed_host(C_LOC(e1u), 4 * nij8, pd_e1u) ; CALL feed_host(C_LOC(zcoef), 4 * nij8, pd_zc)
    CALL feed_mask(cn_fmsk, cn_vmask, pd_vm, npiglo, npjglo, npk)
    CALL feed_field(cn_fe3v, cn_ve3v, pd_e3v, npiglo, npjglo, npk, ktime=kt, ldiom=.NOT. lg_vvl)
    CALL feed_mask(cn_fmsk, cn_tmask, pd_tm, npiglo, npjglo, npk)
    CALL feed_mask(cn_fmsk, cn_umask, pd_um, npiglo, npjglo, npk)
    CALL tick(2)
    CALL hip_check(cdft_moc_mask_e3v(pd_e3v, pd_vm, n38, feed_stream), 'cdft_moc_mask_e3v')
    CALL hip_check(cdft_moc_set_mesh(plan, e1v_d, e3v_d, ib_d, CDFT_MEM_DEVICE, feed_stream), 'cdft_moc_set_mesh')
    CALL hip_check(cdft_moc_decomp_set_mesh(plan, e3v_d, e1u_d, zc_d, gdept, pd_tm, pd_um, CDFT_MEM_DEVICE, feed_stream), &
         &         'cdft_moc_decomp_set_mesh')
    CALL hip_check(cdft_stream_sync(feed_stream), 'cdft_stream_sync')
    CALL feed_release(pd_e3v) ; CALL feed_release(pd_vm) ; CALL feed_release(pd_e1v) ; CALL feed_release(pd_ib)
    CALL feed_release(pd_tm)  ; CALL feed_release(pd_um) ; CALL feed_release(pd_e1u) ; CALL feed_release(pd_zc)
  END SUBROUTINE set_mesh_decomp_fed

  SUBROUTINE get_e3v_all(kt)
    !! e3v of every level (or e31d with -full) times vmask of that level (src/cdfmoc.f90:576-596, get_e3v), read as
    !! two whole fields: the mask goes through zv, which is free at that point
    INTEGER(KIND=4), INTENT(in) :: kt
    INTEGER(KIND=4) :: ik
    CALL readvar3d(cn_fmsk, cn_vmask, zv(:,:,:,1))
    IF ( lfull ) THEN
       DO ik = 1, npk ; e3v(:,:,ik) = e31d(ik) ; END DO
    ELSE
       CALL readvar3d(cn_fe3v, cn_ve3v, e3v, ktime=kt, ldiom=.NOT. lg_vvl)
    ENDIF
    DO ik = 1, npk
       e3v(:,:,ik) = e3v(:,:,ik) * INT(zv(:,:,ik,1), 2)        ! the reference's INTEGER*2 mask (:581-593)
    END DO
  END SUBROUTINE get_e3v_all

  SUBROUTINE CreateOutput
    !! moc.nc exactly as the reference builds it (src/cdfmoc.f90:1006-1188)
    CHARACTER(LEN=256) :: cl_names(6), cl_long(6), cl_tag(6), cl_ag(6)
    INTEGER(KIND=4) :: ib
    stypvar%cunits            = 'Sverdrup'
    stypvar%rmissing_value    = 99999.
    stypvar%valid_min         = -1000.
    stypvar%valid_max         =  1000.
    stypvar%scale_factor      = 1.
    stypvar%add_offset        = 0.
    stypvar%savelog10         = 0.
    stypvar%conline_operation = 'N/A'
    stypvar%caxis             = 'TZY'
    ipk(:) = npk
    cl_names = (/ cn_zomsfglo, cn_zomsfatl, cn_zomsfinp, cn_zomsfind, cn_zomsfpac, cn_zomsfinp0 /)
    cl_long(1) = 'Meridional_Overt.Cell_Global'
    cl_long(2) = 'Meridional_Overt.Cell_Atlantic'
    cl_long(3) = 'Meridional_Overt.Cell_IndoPacif'
    cl_long(4) = 'Meridional_Overt.Cell_Indian'
    cl_long(5) = 'Meridional_Overt.Cell_pacif'
    cl_long(6) = 'Meridional_Overt.Cell_IndPac0'
    ! long-name suffix of the decomposition variables, and the reference's spelling of the
    ! ageostrophic one (src/cdfmoc.f90:1041, :1061: 'Ageostoph' for the global basin only)
    cl_tag = (/ '          ', '_Atlantic ', '_IndoPacif', '_Indian   ', '_Pacif    ', '_IndPac0  ' /)
    cl_ag(:)  = 'Ageostroph_Merid_StreamFunction'
    cl_ag(1)  = 'Ageostoph_Merid_StreamFunction'
    iv = 1
    DO ib = 1, nbasinso
       CALL setvar(TRIM(cl_names(ib)), TRIM(cl_long(ib)))
       IF ( ldec ) THEN
          CALL setvar(TRIM(cl_names(ib))//'_sh', 'GeoShear_Merid_StreamFunction'//TRIM(cl_tag(ib)))
          CALL setvar(TRIM(cl_names(ib))//'_bt', 'Barotropic_Merid_StreamFunction'//TRIM(cl_tag(ib)))
          CALL setvar(TRIM(cl_names(ib))//'_ag', TRIM(cl_ag(ib))//TRIM(cl_tag(ib)))
       ENDIF
    END DO
    ncout = create      (cf_moc, 'none', 1, npjglo, npk, cdep=cn_vdepthw)
    ierr  = createvar   (ncout, stypvar, nvarout, ipk, id_varout, cdglobal=TRIM(cglobal))
    ierr  = putheadervar(ncout, cf_vfil, 1, npjglo, npk, pnavlon=rdumlon, pnavlat=rdumlat, pdep=gdepw)
    dtim  = getvar1d    (cf_vfil, cn_vtimec, npt)
    ierr  = putvar1d    (ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

  SUBROUTINE setvar(cd_name, cd_long)
    !! next output variable (iv is the program's variable counter)
    CHARACTER(LEN=*), INTENT(in) :: cd_name, cd_long
    PRINT *, 'Variable ', iv, ' is ', cd_name
    stypvar(iv)%cname       = cd_name
    stypvar(iv)%clong_name  = cd_long
    stypvar(iv)%cshort_name = cd_name
    iv = iv + 1
  END SUBROUTINE setvar

END PROGRAM cdfmoc
