MODULE cdffeed
  !!======================================================================
  !! cdffeed : file -> HBM without the host waiting for PCIe (not in the reference: host side of this build).
  !! The reference reads a slab with getvar and works on it (e.g. src/cdfmoc.f90:355-379); a program of this build reads a
  !! whole record and hands it to the GPU, which made a record cost its read PLUS its transfer.  feed_field reads a field
  !! slab by slab (a few levels at a time) into the page-locked ring of cdft_feed_* and queues each slab's copy to its place
  !! in a device array as soon as it is read: the DMA of slab s runs while the file read of slab s+1 does, so a record
  !! costs the longer of the two.  The compute call on that device array is then given feed_stream, which orders it after
  !! the copies.  Unpacking (scale_factor ...), windows and the mesh_zgr names are readvar3d's, unchanged.
  !! The ring is made when the first field goes through it, sized for that field: page-locked memory costs 0.17 s per GB to
  !! get and 0.10 s per GB to give back on this pool (tools/pin_probe.hip), so the slots hold a sixteenth of the field, at
  !! least one plane and at most CDFT_FEED_MB (default 128) MB, CDFT_FEED_SLOTS (default 2) of them.  A program takes this
  !! route when its fields are large enough for it to pay (feed_wanted: CDFT_FEED=0 / 1 decide for it).
  !! Beyond feed_field: feed_mean_mesh (the mesh of cdfmean / cdfsum without a 3-D host array) and feed_sig_record (cdfsig0 /
  !! cdfsigi, whose result is as large as their input: slabs in through the ring, results out through two page-locked arrays,
  !! slab s queued for the output file's writing thread -- cdfio's putvar_behind -- while slab s+1 is read and computed).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames, ONLY : cn_x, cn_y
  USE cdftools_hip
  IMPLICIT NONE
  PRIVATE

  TYPE(C_PTR), PUBLIC :: feed_stream = C_NULL_PTR          ! the stream the copies are queued on
  TYPE(C_PTR)         :: hfeed = C_NULL_PTR
  INTEGER(KIND=8)     :: nslot_bytes = 0

  PUBLIC :: feed_moc_record, feed_moc_window, feed_moc_window_mask, feed_levels
  PUBLIC :: feed_wanted, feed_open, feed_close, feed_field, feed_mask, feed_host, feed_device, feed_release, feed_sig_record, feed_mean_mesh

  ! feed_sig_record's slab buffers (kept from record to record): T and S of a slab in HBM, two page-locked result slabs
  TYPE(C_PTR)     :: pd_slab(2) = C_NULL_PTR, ph_slab(2) = C_NULL_PTR
  INTEGER(KIND=8) :: nslab_bytes = 0

CONTAINS

  LOGICAL FUNCTION feed_wanted(kfield)
    !! whether a program whose 3-D fields hold kfield bytes goes through the ring: CDFT_FEED=1 always, CDFT_FEED=0 never;
    !! left alone, from CDFT_FEED_MIN_MB (default 1024) on -- below that a transfer takes a few milliseconds and the ring's
    !! page-locked memory costs as much as it hides (measured at eORCA025, 0.5 GB fields: +- 0.05 s either way on 0.5 s runs)
    INTEGER(KIND=8), INTENT(in) :: kfield
    CHARACTER(LEN=16) :: cl
    INTEGER(KIND=4) :: imin
    CALL GET_ENVIRONMENT_VARIABLE('CDFT_FEED', cl)
    IF ( cl(1:1) == '0' .OR. cl(1:1) == '1' ) THEN
       feed_wanted = ( cl(1:1) == '1' )
       RETURN
    ENDIF
    imin = 1024
    CALL GET_ENVIRONMENT_VARIABLE('CDFT_FEED_MIN_MB', cl) ; IF ( LEN_TRIM(cl) > 0 ) READ(cl, *) imin
    feed_wanted = ( kfield >= INT(imin, 8) * 1048576_8 )
  END FUNCTION feed_wanted

  LOGICAL FUNCTION feed_bytes_wanted()
    CHARACTER(LEN=16) :: cl
    CALL GET_ENVIRONMENT_VARIABLE('CDFT_FEED_BYTES', cl)
    feed_bytes_wanted = ( cl(1:1) /= '0' )
  END FUNCTION feed_bytes_wanted

  SUBROUTINE feed_open()
    !! the feed's stream, with a ring of token size: the slots proper come with the first field (feed_size)
    IF ( C_ASSOCIATED(hfeed) ) RETURN
    nslot_bytes = 0
    CALL hip_check(cdft_feed_create(hfeed, 4096_C_SIZE_T, 2), 'cdft_feed_create')
    CALL hip_check(cdft_feed_stream(hfeed, feed_stream), 'cdft_feed_stream')
  END SUBROUTINE feed_open

  SUBROUTINE feed_size(kplane, kfield)
    ! slots for fields of kfield bytes made of planes of kplane bytes: a sixteenth of the field, whole planes, at least one,
    ! at most CDFT_FEED_MB; a ring that already holds a plane is kept
    INTEGER(KIND=8), INTENT(in) :: kplane, kfield
    INTEGER(KIND=4) :: islots
    INTEGER(KIND=8) :: iwant
    REAL(KIND=8)    :: dl_mb
    CHARACTER(LEN=16) :: cl
    CALL feed_open()
    IF ( nslot_bytes >= kplane ) RETURN
    ! two slots: the copy of one leaves faster than the next is read (57 against 30-35 GB/s), and page-locking costs 0.12 s per GB
    ! to make and as much to undo -- four slots of 124 MB were 0.1 s of every program's 0.7-1.1 s at eORCA12 (profiles/r05 §10)
    dl_mb = 128.d0 ; islots = 2
    CALL GET_ENVIRONMENT_VARIABLE('CDFT_FEED_MB', cl)    ; IF ( LEN_TRIM(cl) > 0 ) READ(cl, *) dl_mb
    CALL GET_ENVIRONMENT_VARIABLE('CDFT_FEED_SLOTS', cl) ; IF ( LEN_TRIM(cl) > 0 ) READ(cl, *) islots
    iwant = MIN(INT(dl_mb * 1048576.d0, 8), MAX(kfield / 16, 4194304_8))
    iwant = MAX(( iwant / kplane ) * kplane, kplane)
    CALL hip_check(cdft_feed_resize(hfeed, INT(iwant, C_SIZE_T), MAX(2, islots)), 'cdft_feed_resize')
    nslot_bytes = iwant
  END SUBROUTINE feed_size

  SUBROUTINE feed_close()
    INTEGER(KIND=4) :: ierr, jb
    DO jb = 1, 2
       IF ( C_ASSOCIATED(pd_slab(jb)) ) ierr = cdft_free(pd_slab(jb))
       IF ( C_ASSOCIATED(ph_slab(jb)) ) ierr = cdft_host_free(ph_slab(jb))
       pd_slab(jb) = C_NULL_PTR ; ph_slab(jb) = C_NULL_PTR
    END DO
    nslab_bytes = 0
    IF ( C_ASSOCIATED(hfeed) ) CALL hip_check(cdft_feed_destroy(hfeed), 'cdft_feed_destroy')
    hfeed = C_NULL_PTR ; feed_stream = C_NULL_PTR
  END SUBROUTINE feed_close

  FUNCTION feed_device(kbytes) RESULT(pdev)
    !! kbytes of HBM (the destination of feed_field / feed_mask / feed_host)
    INTEGER(KIND=8), INTENT(in) :: kbytes
    TYPE(C_PTR) :: pdev
    CALL feed_open()
    CALL hip_check(cdft_malloc(pdev, INT(kbytes, C_SIZE_T)), 'cdft_malloc')
  END FUNCTION feed_device

  SUBROUTINE feed_release(pdev)
    TYPE(C_PTR), INTENT(inout) :: pdev
    INTEGER(KIND=4) :: ierr
    IF ( C_ASSOCIATED(pdev) ) ierr = cdft_free(pdev)
    pdev = C_NULL_PTR
  END SUBROUTINE feed_release

  SUBROUTINE feed_field(cdfile, cdvar, pdev, kpi, kpj, kpk, ktime, ldiom, kimin, kjmin, kkmin, kplane)
    !! the (kpi,kpj,kpk) r4 block of record ktime (window origin kimin, kjmin, kkmin; ldiom as in getvar) into the device
    !! array pdev -- from its plane number kplane on (default 1: the start) -- slab by slab through the ring.  Returns when
    !! the last slab is QUEUED.
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    TYPE(C_PTR),               INTENT(in) :: pdev
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj, kpk
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: ktime, kimin, kjmin, kkmin, kplane
    LOGICAL,         OPTIONAL, INTENT(in) :: ldiom
    INTEGER(KIND=4) :: ik0, inl, inlmax, ikfirst, iplane
    INTEGER(KIND=8) :: islab
    INTEGER(C_SIZE_T) :: ibytes
    TYPE(C_PTR) :: pslot
    REAL(KIND=4), POINTER :: zslab(:,:,:)
    INTEGER(KIND=1), POINTER :: islabs(:,:,:)
    LOGICAL :: ll_whole, ll_bytes
    ll_bytes = .FALSE.
    islab = 4_8 * kpi * kpj
    CALL feed_size(islab, islab * kpk)
    ikfirst = 1 ; IF ( PRESENT(kkmin) ) ikfirst = kkmin
    iplane = 1  ; IF ( PRESENT(kplane) ) iplane = kplane
    ! A field the file holds as bytes (the masks of mask.nc), read whole in i and j: the bytes go through the ring and are widened to
    ! r4 in HBM (cdft_feed_push_i1_as_r4) -- a quarter of the host conversion and of the PCIe transfer of a 4.7 GB r4 mask at eORCA12.
    ! CDFT_FEED_BYTES=0: the r4 way.
    ll_whole = .NOT. PRESENT(ldiom) .AND. kpk >= 1
    IF ( PRESENT(kimin) ) ll_whole = ll_whole .AND. kimin == 1
    IF ( PRESENT(kjmin) ) ll_whole = ll_whole .AND. kjmin == 1
    IF ( ll_whole ) THEN
       IF ( feed_bytes_wanted() ) THEN
          IF ( var_is_plain_bytes(cdfile, cdvar) ) ll_bytes = getdim(cdfile, cn_x, ldexact=.TRUE.) == kpi .AND. getdim(cdfile, cn_y) == kpj
          IF ( ll_bytes ) THEN
             inlmax = INT(MAX(1_8, nslot_bytes / ( islab / 4 )), 4)
             DO ik0 = 1, kpk, inlmax
                inl = MIN(inlmax, kpk - ik0 + 1)
                CALL hip_check(cdft_feed_slot(hfeed, pslot, ibytes), 'cdft_feed_slot')
                CALL C_F_POINTER(pslot, islabs, (/ kpi, kpj, inl /))
                CALL readvar3d_i1(cdfile, cdvar, islabs, ktime=ktime, kkmin=ikfirst + ik0 - 1)
                CALL hip_check(cdft_feed_push_i1_as_r4(hfeed, pdev, INT(( islab / 4 ) * (iplane + ik0 - 2), C_SIZE_T),  &
                     &         INT(( islab / 4 ) * inl, C_SIZE_T)), 'cdft_feed_push_i1_as_r4')
             END DO
             RETURN
          ENDIF
       ENDIF
    ENDIF
    inlmax = INT(MAX(1_8, nslot_bytes / islab), 4)
    DO ik0 = 1, kpk, inlmax
       inl = MIN(inlmax, kpk - ik0 + 1)
       CALL hip_check(cdft_feed_slot(hfeed, pslot, ibytes), 'cdft_feed_slot')
       CALL C_F_POINTER(pslot, zslab, (/ kpi, kpj, inl /))
       CALL readvar3d(cdfile, cdvar, zslab, ktime=ktime, ldiom=ldiom, kimin=kimin, kjmin=kjmin, kkmin=ikfirst + ik0 - 1)
       CALL hip_check(cdft_feed_push(hfeed, pdev, INT(islab * (iplane + ik0 - 2), C_SIZE_T), INT(islab * inl, C_SIZE_T)), &
            &         'cdft_feed_push')
    END DO
  END SUBROUTINE feed_field

  SUBROUTINE feed_moc_record(cdfile, cdvar, plan, krec, kpi, kpj, kpk, ktime)
    !! cdfmoc: levels 1 .. kpk of record ktime of the velocity file (the reference's getvar per level, src/cdfmoc.f90:357)
    !! straight into the slots of record krec (1-based) of the plan's pool (cdft_moc_stage_open before, cdft_moc_stage_close
    !! after): the H2D copy of a slab IS its staging, rows spread to the pool's pitch on the way (cdft_feed_push_rows).
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar
    TYPE(C_PTR),      INTENT(in) :: plan
    INTEGER(KIND=4),  INTENT(in) :: krec, kpi, kpj, kpk, ktime
    INTEGER(KIND=4) :: ik0, inl, inlmax
    INTEGER(KIND=8) :: islab
    INTEGER(C_SIZE_T) :: ibytes, ipitch
    TYPE(C_PTR) :: pslot, pdst
    REAL(KIND=4), POINTER :: zslab(:,:,:)
    islab = 4_8 * kpi * kpj
    CALL feed_size(islab, islab * kpk)
    inlmax = INT(MAX(1_8, nslot_bytes / islab), 4)
    DO ik0 = 1, kpk, inlmax
       inl = MIN(inlmax, kpk - ik0 + 1)
       CALL hip_check(cdft_feed_slot(hfeed, pslot, ibytes), 'cdft_feed_slot')
       CALL C_F_POINTER(pslot, zslab, (/ kpi, kpj, inl /))
       CALL readvar3d(cdfile, cdvar, zslab, ktime=ktime, kkmin=ik0)
       CALL hip_check(cdft_moc_stage_dest(plan, INT(krec - 1, C_INT), INT(ik0 - 1, C_INT), pdst, ipitch), 'cdft_moc_stage_dest')
       CALL hip_check(cdft_feed_push_rows(hfeed, pdst, 0_C_SIZE_T, ipitch, INT(4_8 * kpi, C_SIZE_T), INT(kpj, C_SIZE_T) * inl), &
            &         'cdft_feed_push_rows')
    END DO
  END SUBROUTINE feed_moc_record

  INTEGER(KIND=4) FUNCTION feed_levels(kpi, kpj, kpk)
    !! the levels of a (kpi,kpj,kpk) r4 field that one slot of the ring holds (the ring is made for such fields if it is not there)
    INTEGER(KIND=4), INTENT(in) :: kpi, kpj, kpk
    INTEGER(KIND=8) :: islab
    islab = 4_8 * kpi * kpj
    CALL feed_size(islab, islab * kpk)
    feed_levels = INT(MAX(1_8, nslot_bytes / islab), 4)
  END FUNCTION feed_levels

  SUBROUTINE feed_moc_window(cdfile, cdvar, plan, kslot, kpi, kpj, klev0, knlev, ktime, ldiom)
    !! cdfmoc, the stream route: levels klev0 .. klev0+knlev-1 of record ktime of cdvar (the reference's getvar per level,
    !! src/cdfmoc.f90:357; kslot = 0: the metric e3v, :343-348) into slot kslot of the plan's window of levels, a ring slot at a
    !! time: the H2D copy spreads the rows to the window's pitch (cdft_feed_push_rows)
    CHARACTER(LEN=*),  INTENT(in) :: cdfile, cdvar
    TYPE(C_PTR),       INTENT(in) :: plan
    INTEGER(KIND=4),   INTENT(in) :: kslot, kpi, kpj, klev0, knlev, ktime
    LOGICAL, OPTIONAL, INTENT(in) :: ldiom
    INTEGER(KIND=4) :: ik0, inl, inlmax
    INTEGER(KIND=8) :: islab
    INTEGER(C_SIZE_T) :: ibytes, ipitch
    TYPE(C_PTR) :: pslot, pdst
    REAL(KIND=4), POINTER :: zslab(:,:,:)
    islab = 4_8 * kpi * kpj
    CALL feed_size(islab, islab * knlev)
    inlmax = INT(MAX(1_8, nslot_bytes / islab), 4)
    DO ik0 = 1, knlev, inlmax
       inl = MIN(inlmax, knlev - ik0 + 1)
       CALL hip_check(cdft_feed_slot(hfeed, pslot, ibytes), 'cdft_feed_slot')
       CALL C_F_POINTER(pslot, zslab, (/ kpi, kpj, inl /))
       CALL readvar3d(cdfile, cdvar, zslab, ktime=ktime, ldiom=ldiom, kkmin=klev0 + ik0 - 1)
       CALL hip_check(cdft_moc_stream_dest(plan, INT(kslot, C_INT), INT(ik0 - 1, C_INT), pdst, ipitch), 'cdft_moc_stream_dest')
       CALL hip_check(cdft_feed_push_rows(hfeed, pdst, 0_C_SIZE_T, ipitch, INT(4_8 * kpi, C_SIZE_T), INT(kpj, C_SIZE_T) * inl), &
            &         'cdft_feed_push_rows')
    END DO
  END SUBROUTINE feed_moc_window

  SUBROUTINE feed_moc_window_mask(cdfile, cdvar, plan, kpi, kpj, klev0, knlev)
    !! the same for the window's mask, as bytes (dense in HBM: a plain copy)
    CHARACTER(LEN=*), INTENT(in) :: cdfile, cdvar
    TYPE(C_PTR),      INTENT(in) :: plan
    INTEGER(KIND=4),  INTENT(in) :: kpi, kpj, klev0, knlev
    INTEGER(KIND=4) :: ik0, inl, inlmax
    INTEGER(KIND=8) :: islab
    INTEGER(C_SIZE_T) :: ibytes
    TYPE(C_PTR) :: pslot, pdst
    INTEGER(KIND=1), POINTER :: islabs(:,:,:)
    islab = 1_8 * kpi * kpj
    CALL feed_size(4 * islab, 4 * islab * knlev)
    inlmax = INT(MAX(1_8, nslot_bytes / islab), 4)
    DO ik0 = 1, knlev, inlmax
       inl = MIN(inlmax, knlev - ik0 + 1)
       CALL hip_check(cdft_feed_slot(hfeed, pslot, ibytes), 'cdft_feed_slot')
       CALL C_F_POINTER(pslot, islabs, (/ kpi, kpj, inl /))
       CALL readvar3d_i1(cdfile, cdvar, islabs, kkmin=klev0 + ik0 - 1)
       CALL hip_check(cdft_moc_stream_vmask_dest(plan, INT(ik0 - 1, C_INT), pdst), 'cdft_moc_stream_vmask_dest')
       CALL hip_check(cdft_feed_push(hfeed, pdst, 0_C_SIZE_T, INT(islab * inl, C_SIZE_T)), 'cdft_feed_push')
    END DO
  END SUBROUTINE feed_moc_window_mask

  SUBROUTINE feed_mask(cdfile, cdvar, pdev, kpi, kpj, kpk, ktime)
    !! a (kpi,kpj,kpk) mask as bytes into the device array pdev, the same way
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    TYPE(C_PTR),               INTENT(in) :: pdev
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj, kpk
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: ktime
    INTEGER(KIND=4) :: ik0, inl, inlmax
    INTEGER(KIND=8) :: islab
    INTEGER(C_SIZE_T) :: ibytes
    TYPE(C_PTR) :: pslot
    INTEGER(KIND=1), POINTER :: islabs(:,:,:)
    islab = 1_8 * kpi * kpj
    CALL feed_size(islab, 4 * islab * kpk)
    inlmax = INT(MAX(1_8, nslot_bytes / islab), 4)
    DO ik0 = 1, kpk, inlmax
       inl = MIN(inlmax, kpk - ik0 + 1)
       CALL hip_check(cdft_feed_slot(hfeed, pslot, ibytes), 'cdft_feed_slot')
       CALL C_F_POINTER(pslot, islabs, (/ kpi, kpj, inl /))
       CALL readvar3d_i1(cdfile, cdvar, islabs, ktime=ktime, kkmin=ik0)
       CALL hip_check(cdft_feed_push(hfeed, pdev, INT(islab * (ik0 - 1), C_SIZE_T), INT(islab * inl, C_SIZE_T)), 'cdft_feed_push')
    END DO
  END SUBROUTINE feed_mask

  SUBROUTINE feed_host(psrc, kbytes, pdev)
    !! kbytes of a host array (C_LOC of it: a 2-D metric, a mask, a profile) into the device array pdev, there when the
    !! call returns.  Small arrays: a plain copy from pageable memory, no ring.
    TYPE(C_PTR),     INTENT(in) :: psrc, pdev
    INTEGER(KIND=8), INTENT(in) :: kbytes
    CALL hip_check(cdft_memcpy_h2d(pdev, psrc, INT(kbytes, C_SIZE_T), C_NULL_PTR), 'cdft_memcpy_h2d')
    CALL hip_check(cdft_stream_sync(C_NULL_PTR), 'cdft_stream_sync')
  END SUBROUTINE feed_host

  SUBROUTINE feed_mean_mesh(plan, pe1, pe2, pe31d, cde3fil, cde3var, cdmskfil, cdmskvar, pib, kpi, kpj, kpk, knb, kimin, kjmin, kkmin, &
       &                    ktime, ldiom)
    !! cdft_mean_set_mesh (cdfmean, cdfsum) with nothing 3-D in host memory: the mask of the working levels and e3 go from
    !! their files to HBM through the ring; e1, e2, the basin masks (and e31d with -full: pe31d associated, e3 not read) are
    !! the caller's host arrays (C_LOC).  The plan keeps its own copies: everything fed here is released on return.
    TYPE(C_PTR),      INTENT(in) :: plan, pe1, pe2, pe31d, pib
    CHARACTER(LEN=*), INTENT(in) :: cde3fil, cde3var, cdmskfil, cdmskvar
    INTEGER(KIND=4),  INTENT(in) :: kpi, kpj, kpk, knb, kimin, kjmin, kkmin, ktime
    LOGICAL,          INTENT(in) :: ldiom
    TYPE(C_PTR) :: pd1, pd2, pdib, pdm, pde3, pd31
    INTEGER(KIND=8) :: inij, in3
    REAL(KIND=4), POINTER :: z1(:), z2(:), zm(:)
    INTEGER(KIND=2), POINTER :: ib(:)
    inij = INT(kpi, 8) * kpj ; in3 = inij * kpk
    pd1 = feed_device(4 * inij) ; pd2 = feed_device(4 * inij) ; pdib = feed_device(2 * knb * inij) ; pdm = feed_device(4 * in3)
    pde3 = C_NULL_PTR ; pd31 = C_NULL_PTR
    CALL feed_host(pe1, 4 * inij, pd1) ; CALL feed_host(pe2, 4 * inij, pd2) ; CALL feed_host(pib, 2 * knb * inij, pdib)
    CALL feed_field(cdmskfil, cdmskvar, pdm, kpi, kpj, kpk, kimin=kimin, kjmin=kjmin, kkmin=kkmin)
    IF ( C_ASSOCIATED(pe31d) ) THEN
       pd31 = feed_device(4_8 * kpk) ; CALL feed_host(pe31d, 4_8 * kpk, pd31)
    ELSE
       pde3 = feed_device(4 * in3)
       CALL feed_field(cde3fil, cde3var, pde3, kpi, kpj, kpk, ktime=ktime, ldiom=ldiom, kimin=kimin, kjmin=kjmin, kkmin=kkmin)
    ENDIF
    CALL C_F_POINTER(pd1, z1, (/ inij /)) ; CALL C_F_POINTER(pd2, z2, (/ inij /)) ; CALL C_F_POINTER(pdm, zm, (/ in3 /))
    CALL C_F_POINTER(pdib, ib, (/ knb * inij /))
    CALL hip_check(cdft_mean_set_mesh(plan, z1, z2, pde3, pd31, zm, ib, CDFT_MEM_DEVICE, feed_stream), 'cdft_mean_set_mesh')
    CALL hip_check(cdft_stream_sync(feed_stream), 'cdft_stream_sync')
    CALL feed_release(pd1) ; CALL feed_release(pd2) ; CALL feed_release(pdib) ; CALL feed_release(pdm)
    CALL feed_release(pde3) ; CALL feed_release(pd31)
  END SUBROUTINE feed_mean_mesh

  SUBROUTINE feed_sig_record(cdtfil, cdtem, cdsfil, cdsal, kpi, kpj, kpk, ktime, pref, pspval, keos, kout, kidvar)
    !! One record of cdfsig0 / cdfsigi (the level loop of src/cdfsig0.f90:139-155, src/cdfsigi.f90:131-137) a slab of levels at
    !! a time: T and S of slab s+1 go from the file to HBM through the ring while the GPU works on slab s and sends its
    !! result to a page-locked array; slab s is written to the file kout while slab s+1 is computed.  Nothing of the size of
    !! a record is held in host memory, and the host waits for its file reads and writes only.
    CHARACTER(LEN=*), INTENT(in) :: cdtfil, cdtem, cdsfil, cdsal
    INTEGER(KIND=4),  INTENT(in) :: kpi, kpj, kpk, ktime, kout, kidvar
    REAL(KIND=4),     INTENT(in) :: pref, pspval
    INTEGER(C_INT),   INTENT(in) :: keos
    INTEGER(KIND=4) :: inl, ins, js, jk, ib, ierr
    INTEGER(KIND=8) :: islab
    REAL(KIND=4), POINTER :: zt_d(:), zs_d(:)
    islab = 4_8 * kpi * kpj
    CALL feed_size(islab, islab * kpk)
    inl = INT(MAX(1_8, MIN(INT(kpk, 8), nslot_bytes / islab)), 4)        ! levels per slab: what a slot of the ring holds
    IF ( nslab_bytes < islab * inl ) THEN
       DO ib = 1, 2
          IF ( C_ASSOCIATED(pd_slab(ib)) ) ierr = cdft_free(pd_slab(ib))
          IF ( C_ASSOCIATED(ph_slab(ib)) ) ierr = cdft_host_free(ph_slab(ib))
          CALL hip_check(cdft_malloc(pd_slab(ib), INT(islab * inl, C_SIZE_T)), 'cdft_malloc')
          CALL hip_check(cdft_host_alloc(ph_slab(ib), INT(islab * inl, C_SIZE_T)), 'cdft_host_alloc')
       END DO
       nslab_bytes = islab * inl
    ENDIF
    CALL C_F_POINTER(pd_slab(1), zt_d, (/ islab / 4 * inl /)) ; CALL C_F_POINTER(pd_slab(2), zs_d, (/ islab / 4 * inl /))
    ins = ( kpk + inl - 1 ) / inl
    CALL load_slab(1) ; CALL compute_slab(1)
    DO js = 1, ins
       IF ( js < ins ) CALL load_slab(js + 1)                  ! queued behind the kernel of slab js, read while it runs -- and
       !                                                         while slab js-1 is written by the file's own thread
       CALL hip_check(cdft_stream_sync(feed_stream), 'cdft_stream_sync')      ! the result of slab js is in its host array
       CALL wait_writes(kout)                                  ! slab js-1 is in the file: its host array is free again ...
       IF ( js < ins ) CALL compute_slab(js + 1)               ! ... for the result of slab js+1
       ib = 1 + MOD(js - 1, 2)
       DO jk = ( js - 1 ) * inl + 1, MIN(js * inl, kpk)        ! queued: written while slab js+2 is read
          ierr = putvar_behind(kout, kidvar, slab_level(ph_slab(ib), jk - ( js - 1 ) * inl), .FALSE., jk, kpi, kpj, ktime=ktime)
       END DO
    END DO
    CALL wait_writes(kout)

  CONTAINS

    SUBROUTINE load_slab(ks)
      INTEGER(KIND=4), INTENT(in) :: ks
      INTEGER(KIND=4) :: ik0, il
      ik0 = ( ks - 1 ) * inl + 1 ; il = MIN(inl, kpk - ik0 + 1)
      CALL feed_field(cdtfil, cdtem, pd_slab(1), kpi, kpj, il, ktime=ktime, kkmin=ik0)
      CALL feed_field(cdsfil, cdsal, pd_slab(2), kpi, kpj, il, ktime=ktime, kkmin=ik0)
    END SUBROUTINE load_slab

    FUNCTION slab_level(pbase, kl) RESULT(plev)
      ! the address of level kl of a page-locked slab
      TYPE(C_PTR), INTENT(in) :: pbase
      INTEGER(KIND=4), INTENT(in) :: kl
      TYPE(C_PTR) :: plev
      REAL(KIND=4), POINTER :: zall(:,:,:)
      CALL C_F_POINTER(pbase, zall, (/ kpi, kpj, inl /))
      plev = C_LOC(zall(1,1,kl))
    END FUNCTION slab_level

    SUBROUTINE compute_slab(ks)
      INTEGER(KIND=4), INTENT(in) :: ks
      INTEGER(KIND=4) :: ik0, il
      REAL(KIND=4), POINTER :: zres(:)
      ik0 = ( ks - 1 ) * inl + 1 ; il = MIN(inl, kpk - ik0 + 1)
      CALL C_F_POINTER(ph_slab(1 + MOD(ks - 1, 2)), zres, (/ islab / 4 * il /))
      CALL hip_check(cdft_eos_sig_tool(zt_d, zs_d, INT(islab / 4 * il, C_INT64_T), pref, pspval, keos, 0_C_INT, zres,  &
           &         CDFT_MEM_DEVICE, CDFT_MEM_PINNED, feed_stream), 'cdft_eos_sig_tool')
    END SUBROUTINE compute_slab

  END SUBROUTINE feed_sig_record

END MODULE cdffeed
