PROGRAM cdfvT
  !!======================================================================
  !!                     ***  PROGRAM  cdfvT  ***
  !! Time-mean products U.T, U.S (at U points) and V.T, V.S (at V points) over a list of tags.
  !!
  !! Drop-in for the reference tool (src/cdfvT.f90): same options (-c CONFIG -l tags... [-o] [-nc4] [-vvl]
  !! [-namT/-namU/-namV]), same file naming through SetFileName, same vt.nc (vomevt, vomevs, vozout,
  !! vozous float (t,z,y,x), iweight = number of frames, :262-310).  Like the reference the work goes
  !! level by level (:160); all the time frames of one level are sent to the GPU together
  !! (cdft_vt_compute with npk = 1), which interpolates, multiplies and accumulates them in the
  !! reference's order (bit-identical).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE modutils
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(C_INT)  :: ngpu, igpus(64)                          ! row blocks and their HIP devices (CDFT_GPUS, CDFT_GPU_MAP)
  INTEGER(KIND=4) :: narg, ijarg, ierr, jk, jt, jtt, jv, npiglo, npjglo, npk, npt, ntags = 0, ntframe, nvaro, ncout, ifr
  INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:), nptag(:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: zu(:,:,:), zv(:,:,:), zt(:,:,:), zs(:,:,:), e3u(:,:,:), e3v(:,:,:)
  REAL(KIND=8), ALLOCATABLE, TARGET :: dvt(:,:), dvs(:,:), dut(:,:), dus(:,:), de3u(:,:), de3v(:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:)
  REAL(KIND=8) :: dtotal_time, dtimean(1)
  CHARACTER(LEN=256) :: cf_tfil, cf_sfil, cf_ufil, cf_vfil, cf_out = 'vt.nc', config = 'none', cldum
  CHARACTER(LEN=256), ALLOCATABLE :: ctag_lst(:)
  TYPE(variable), ALLOCATABLE :: stypvar(:)
  TYPE(C_PTR) :: pe3u, pe3v, pe3um, pe3vm
  TYPE(C_PTR) :: ph_out(6,2) = C_NULL_PTR                        ! the host feed's two sets of page-locked r4 results
  INTEGER(KIND=4) :: nqo
  ! the frames of one level in HBM when they are fed there through the ring (module cdffeed): never touched by the host
  TYPE(C_PTR) :: pd_u = C_NULL_PTR, pd_v = C_NULL_PTR, pd_t = C_NULL_PTR, pd_s = C_NULL_PTR, pd_e3u = C_NULL_PTR, pd_e3v = C_NULL_PTR
  REAL(KIND=4), POINTER :: zu_d(:), zv_d(:), zt_d(:), zs_d(:)
  INTEGER(KIND=8) :: nfr8
  LOGICAL :: lnc4 = .FALSE., lfeed = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfvT.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-c'    ) ; CALL GET_COMMAND_ARGUMENT(ijarg, config) ; ijarg = ijarg + 1
     CASE ( '-l'    ) ; CALL GetTagList
     CASE ( '-o'    ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out) ; ijarg = ijarg + 1
     CASE ( '-nc4'  ) ; lnc4   = .TRUE.
     CASE ( '-vvl'  ) ; lg_vvl = .TRUE.
     CASE ( '-namT' ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cn_votemper) ; ijarg = ijarg + 1
     CASE ( '-namU' ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cn_vozocrtx) ; ijarg = ijarg + 1
     CASE ( '-namV' ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cn_vomecrty) ; ijarg = ijarg + 1
     CASE DEFAULT     ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO
  IF ( ntags == 0 ) THEN
     PRINT *, ' ERROR : a list of tags ( -l ) is mandatory.' ; STOP 99
  ENDIF

  cf_tfil = SetFileName(config, ctag_lst(1), 'T')
  npiglo = getdim(cf_tfil, cn_x)
  npjglo = getdim(cf_tfil, cn_y)
  npk    = getdim(cf_tfil, cn_z)
  PRINT *, 'npiglo =', npiglo
  PRINT *, 'npjglo =', npjglo
  PRINT *, 'npk    =', npk

  nvaro = 4
  IF ( lg_vvl ) THEN
     nvaro = 6
     cn_ve3u = cn_ve3uvvl
     cn_ve3v = cn_ve3vvvl
  ENDIF

  ! number of frames of every tag, and the mean time
  ALLOCATE( nptag(ntags) )
  ntframe = 0 ; dtotal_time = 0.d0
  DO jt = 1, ntags
     cf_tfil = SetFileName(config, ctag_lst(jt), 'T', ld_stop=.TRUE.)
     npt = getdim(cf_tfil, cn_t)
     nptag(jt) = npt
     ALLOCATE( dtim(npt) )
     dtim = getvar1d(cf_tfil, cn_vtimec, npt)
     dtotal_time = dtotal_time + SUM(dtim(1:npt))
     DEALLOCATE( dtim )
     ntframe = ntframe + npt
  END DO

  CALL hip_shard(ngpu, igpus)                                ! CDFT_GPUS / CDFT_GPU_MAP
  ! one device: every slab file -> HBM through the ring (module cdffeed).  What decides is what the RUN moves -- a field's frames of
  ! every level -- not a level's frames: one frame of eORCA12 is 62 MB a level, and the level loop below it (pageable arrays
  ! in, r8 means out) took 8.0 s for what the ring does in ... (profiles/r05/e2e_all_eorca12.txt)
  lfeed = feed_wanted(4_8 * npiglo * npjglo * ntframe * npk) .AND. ngpu == 1
  nfr8 = 4_8 * npiglo * npjglo * ntframe
  IF ( lfeed ) THEN
     pd_u = feed_device(nfr8) ; pd_v = feed_device(nfr8) ; pd_t = feed_device(nfr8) ; pd_s = feed_device(nfr8)
     CALL C_F_POINTER(pd_u, zu_d, (/ nfr8 / 4 /)) ; CALL C_F_POINTER(pd_v, zv_d, (/ nfr8 / 4 /))
     CALL C_F_POINTER(pd_t, zt_d, (/ nfr8 / 4 /)) ; CALL C_F_POINTER(pd_s, zs_d, (/ nfr8 / 4 /))
  ELSE
     ALLOCATE( zu(npiglo,npjglo,ntframe), zv(npiglo,npjglo,ntframe), zt(npiglo,npjglo,ntframe), zs(npiglo,npjglo,ntframe) )
  ENDIF
  ALLOCATE( dvt(npiglo,npjglo), dvs(npiglo,npjglo), dut(npiglo,npjglo), dus(npiglo,npjglo) )
  ALLOCATE( stypvar(nvaro), ipk(nvaro), id_varout(nvaro) )
  pe3u = C_NULL_PTR ; pe3v = C_NULL_PTR ; pe3um = C_NULL_PTR ; pe3vm = C_NULL_PTR
  IF ( lg_vvl ) THEN
     ALLOCATE( de3u(npiglo,npjglo), de3v(npiglo,npjglo) )
     pe3um = C_LOC(de3u) ; pe3vm = C_LOC(de3v)
     IF ( lfeed ) THEN
        pd_e3u = feed_device(nfr8) ; pd_e3v = feed_device(nfr8)
        pe3u = pd_e3u ; pe3v = pd_e3v
     ELSE
        ALLOCATE( e3u(npiglo,npjglo,ntframe), e3v(npiglo,npjglo,ntframe) )
        pe3u = C_LOC(e3u) ; pe3v = C_LOC(e3v)
     ENDIF
  ENDIF
  cf_tfil = SetFileName(config, ctag_lst(1), 'T')
  CALL CreateOutput

  IF ( lfeed ) THEN
     ! Level k+1 is read (its frames leave for HBM slab by slab) and computed while level k is on its way to the host and to
     ! the file.  The means come back as r4 -- what NetCDF stores in the float variables when the reference hands it the r8
     ! arrays (the same rounding, done on the GPU: half the bytes over PCIe, no conversion in the writer) -- into two sets of
     ! page-locked arrays.
     nqo = 4 ; IF ( lg_vvl ) nqo = 6
     DO jt = 1, nqo
        CALL hip_check(cdft_host_alloc(ph_out(jt,1), INT(4_8 * npiglo * npjglo, C_SIZE_T)), 'cdft_host_alloc')
        CALL hip_check(cdft_host_alloc(ph_out(jt,2), INT(4_8 * npiglo * npjglo, C_SIZE_T)), 'cdft_host_alloc')
     END DO
     CALL load_level(1) ; CALL compute_level(1)
     DO jk = 1, npk
        IF ( jk < npk ) CALL load_level(jk + 1)            ! (level jk-1 is being written by the output file's thread meanwhile)
        CALL hip_check(cdft_stream_sync(feed_stream), 'cdft_stream_sync')
        CALL wait_writes(ncout)                            ! level jk-1 is in the file: its arrays take level jk+1
        IF ( jk < npk ) CALL compute_level(jk + 1)
        IF ( jk == 1 ) THEN
           dtimean(1) = dtotal_time / ntframe
           ierr = putvar1d(ncout, dtimean, 1, 'T')
        ENDIF
        DO jt = 1, nqo            ! (the reference rewrites iweight = ntframe at every level: written once it is the same file)
           IF ( jk == 1 ) THEN ; ierr = putvar_behind(ncout, id_varout(jt), ph_out(jt,1), .FALSE., jk, npiglo, npjglo, kwght=ntframe)
           ELSE ; ierr = putvar_behind(ncout, id_varout(jt), ph_out(jt, 1 + MOD(jk - 1, 2)), .FALSE., jk, npiglo, npjglo)
           ENDIF
        END DO
     END DO
     CALL wait_writes(ncout)
     DO jt = 1, nqo
        ierr = cdft_host_free(ph_out(jt,1)) ; ierr = cdft_host_free(ph_out(jt,2))
     END DO
  ENDIF

  DO jk = 1, MERGE(0, npk, lfeed)                            ! (the reference's level loop: read, compute, write in turn)
     PRINT *, 'level ', jk
     ifr = 0
     DO jt = 1, ntags
        cf_tfil = SetFileName(config, ctag_lst(jt), 'T', ld_stop=.TRUE.)
        cf_sfil = SetFileName(config, ctag_lst(jt), 'S', ld_stop=.FALSE.)
        IF ( chkfile(cf_sfil, ld_verbose=.FALSE.) ) cf_sfil = cf_tfil      ! salinity in the T file
        cf_ufil = SetFileName(config, ctag_lst(jt), 'U')
        cf_vfil = SetFileName(config, ctag_lst(jt), 'V')
        DO jtt = 1, nptag(jt)
           ifr = ifr + 1
           CALL getvar_to(cf_ufil, cn_vozocrtx, jk, zu(:,:,ifr), ktime=jtt)
           CALL getvar_to(cf_vfil, cn_vomecrty, jk, zv(:,:,ifr), ktime=jtt)
           CALL getvar_to(cf_tfil, cn_votemper, jk, zt(:,:,ifr), ktime=jtt)
           CALL getvar_to(cf_sfil, cn_vosaline, jk, zs(:,:,ifr), ktime=jtt)
           IF ( lg_vvl ) THEN
              CALL getvar_to(cf_ufil, cn_ve3u, jk, e3u(:,:,ifr), ktime=jtt)
              CALL getvar_to(cf_vfil, cn_ve3v, jk, e3v(:,:,ifr), ktime=jtt)
           ENDIF
        END DO
     END DO
     IF ( ngpu > 1 ) THEN                                    ! CDFT_GPUS / CDFT_GPU_MAP: row blocks over the node's GPUs
        CALL hip_check(cdft_vt_compute_sharded(zu, zv, zt, zs, pe3u, pe3v, INT(npiglo, C_INT64_T) * npjglo, ntframe, npiglo,  &
             &         npjglo, 1_C_INT, C_LOC(dvt), C_LOC(dvs), C_LOC(dut), C_LOC(dus), pe3um, pe3vm, 0_C_INT, ngpu, igpus),  &
             &         'cdft_vt_compute_sharded')
     ELSE
     CALL hip_check(cdft_vt_compute(zu, zv, zt, zs, pe3u, pe3v, INT(npiglo, C_INT64_T) * npjglo, ntframe, npiglo, npjglo,   &
          &         1_C_INT, CDFT_MEM_HOST, C_LOC(dvt), C_LOC(dvs), C_LOC(dut), C_LOC(dus), pe3um, pe3vm, 0_C_INT,         &
          &         CDFT_MEM_HOST, C_NULL_PTR), 'cdft_vt_compute')
     ENDIF
     ierr = putvar(ncout, id_varout(1), dvt, jk, npiglo, npjglo, kwght=ntframe)
     ierr = putvar(ncout, id_varout(2), dvs, jk, npiglo, npjglo, kwght=ntframe)
     ierr = putvar(ncout, id_varout(3), dut, jk, npiglo, npjglo, kwght=ntframe)
     ierr = putvar(ncout, id_varout(4), dus, jk, npiglo, npjglo, kwght=ntframe)
     IF ( lg_vvl ) THEN
        ierr = putvar(ncout, id_varout(5), de3u, jk, npiglo, npjglo, kwght=ntframe)
        ierr = putvar(ncout, id_varout(6), de3v, jk, npiglo, npjglo, kwght=ntframe)
     ENDIF
     IF ( jk == 1 ) THEN
        dtimean(1) = dtotal_time / ntframe
        ierr = putvar1d(ncout, dtimean, 1, 'T')
     ENDIF
  END DO
  IF ( lfeed ) THEN
     CALL feed_release(pd_u) ; CALL feed_release(pd_v) ; CALL feed_release(pd_t) ; CALL feed_release(pd_s)
     CALL feed_release(pd_e3u) ; CALL feed_release(pd_e3v)
     CALL feed_close()
  ENDIF
  ierr = closeout(ncout)

CONTAINS

  SUBROUTINE load_level(kk)
    !! every frame of level kk (all tags, src/cdfvT.f90:168-196) from the files to HBM through the ring
    INTEGER(KIND=4), INTENT(in) :: kk
    INTEGER(KIND=4) :: jtg, jfr, ifrm
    PRINT *, 'level ', kk
    ifrm = 0
    DO jtg = 1, ntags
       cf_tfil = SetFileName(config, ctag_lst(jtg), 'T', ld_stop=.TRUE.)
       cf_sfil = SetFileName(config, ctag_lst(jtg), 'S', ld_stop=.FALSE.)
       IF ( chkfile(cf_sfil, ld_verbose=.FALSE.) ) cf_sfil = cf_tfil      ! salinity in the T file
       cf_ufil = SetFileName(config, ctag_lst(jtg), 'U')
       cf_vfil = SetFileName(config, ctag_lst(jtg), 'V')
       DO jfr = 1, nptag(jtg)
          ifrm = ifrm + 1
          CALL feed_field(cf_ufil, cn_vozocrtx, pd_u, npiglo, npjglo, 1, ktime=jfr, kkmin=kk, kplane=ifrm)
          CALL feed_field(cf_vfil, cn_vomecrty, pd_v, npiglo, npjglo, 1, ktime=jfr, kkmin=kk, kplane=ifrm)
          CALL feed_field(cf_tfil, cn_votemper, pd_t, npiglo, npjglo, 1, ktime=jfr, kkmin=kk, kplane=ifrm)
          CALL feed_field(cf_sfil, cn_vosaline, pd_s, npiglo, npjglo, 1, ktime=jfr, kkmin=kk, kplane=ifrm)
          IF ( lg_vvl ) THEN
             CALL feed_field(cf_ufil, cn_ve3u, pd_e3u, npiglo, npjglo, 1, ktime=jfr, kkmin=kk, kplane=ifrm)
             CALL feed_field(cf_vfil, cn_ve3v, pd_e3v, npiglo, npjglo, 1, ktime=jfr, kkmin=kk, kplane=ifrm)
          ENDIF
       END DO
    END DO
  END SUBROUTINE load_level

  SUBROUTINE compute_level(kk)
    !! queued behind the copies of load_level(kk); the r4 means on their way to the page-locked arrays of kk's parity
    INTEGER(KIND=4), INTENT(in) :: kk
    INTEGER(KIND=4) :: ib
    TYPE(C_PTR) :: pm5, pm6
    ib = 1 + MOD(kk - 1, 2)
    pm5 = C_NULL_PTR ; pm6 = C_NULL_PTR
    IF ( lg_vvl ) THEN ; pm5 = ph_out(5,ib) ; pm6 = ph_out(6,ib) ; ENDIF
    CALL hip_check(cdft_vt_compute(zu_d, zv_d, zt_d, zs_d, pe3u, pe3v, INT(npiglo, C_INT64_T) * npjglo, ntframe, npiglo, npjglo, &
         &         1_C_INT, CDFT_MEM_DEVICE, ph_out(1,ib), ph_out(2,ib), ph_out(3,ib), ph_out(4,ib), pm5, pm6, 1_C_INT,           &
         &         CDFT_MEM_PINNED, feed_stream), 'cdft_vt_compute')
  END SUBROUTINE compute_level

  SUBROUTINE GetTagList
    !! every argument up to the next option is a tag (src/cdfvT.f90:312-339)
    INTEGER(KIND=4) :: ji, icur
    ntags = 0
    icur = ijarg
    DO ji = icur, narg
       CALL GET_COMMAND_ARGUMENT(ji, cldum)
       IF ( cldum(1:1) == '-' ) EXIT
       ntags = ntags + 1
    END DO
    ALLOCATE( ctag_lst(ntags) )
    DO ji = icur, icur + ntags - 1
       CALL GET_COMMAND_ARGUMENT(ji, ctag_lst(ji - icur + 1)) ; ijarg = ijarg + 1
    END DO
  END SUBROUTINE GetTagList

  SUBROUTINE CreateOutput
    !! src/cdfvT.f90:262-310
    ipk(:) = npk
    stypvar%rmissing_value    = 0.
    stypvar%valid_min         = -100.
    stypvar%valid_max         = 100.
    stypvar%conline_operation = 'N/A'
    stypvar%caxis             = 'TZYX'
    DO jv = 1, nvaro
       stypvar(jv)%ichunk = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
    END DO
    stypvar(1)%cname = cn_vomevt ; stypvar(1)%cunits = 'm.DegC.s-1' ; stypvar(1)%clong_name = 'Meridional_VT'
    stypvar(2)%cname = cn_vomevs ; stypvar(2)%cunits = 'm.PSU.s-1'  ; stypvar(2)%clong_name = 'Meridional_VS'
    stypvar(3)%cname = cn_vozout ; stypvar(3)%cunits = 'm.DegC.s-1' ; stypvar(3)%clong_name = 'Zonal_UT'
    stypvar(4)%cname = cn_vozous ; stypvar(4)%cunits = 'm.PSU.s-1'  ; stypvar(4)%clong_name = 'Zonal_US'
    IF ( lg_vvl ) THEN
       stypvar(5)%cname = cn_ve3u ; stypvar(5)%cunits = 'm' ; stypvar(5)%clong_name = 'Mean_e3u'
       stypvar(6)%cname = cn_ve3v ; stypvar(6)%cunits = 'm' ; stypvar(6)%clong_name = 'Mean_e3v'
    ENDIF
    DO jv = 1, nvaro
       stypvar(jv)%cshort_name = stypvar(jv)%cname
    END DO
    ncout = create      (cf_out, cf_tfil, npiglo, npjglo, npk, ld_xycoo=.TRUE., ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, nvaro, ipk, id_varout, ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_tfil, npiglo, npjglo, npk, ld_xycoo=.TRUE.)
  END SUBROUTINE CreateOutput

END PROGRAM cdfvT
