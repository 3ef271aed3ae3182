MODULE cdffeed
  !!======================================================================
  !! cdffeed : file -> HBM without the host waiting for PCIe (not in the reference: host side of this build).
  !! The reference reads a slab with getvar and works on it (e.g. src/cdfmoc.f90:355-379); a program of this build reads a
  !! whole record and hands it to the GPU, which made a record cost its read PLUS its transfer.  feed_field reads a field
  !! slab by slab (a few levels at a time) into the page-locked ring of cdft_feed_* and queues each slab's copy to its place
  !! in a device array as soon as it is read: the DMA of slab s runs while the file read of slab s+1 does, so a record
  !! costs the longer of the two.  The compute call on that device array is then given feed_stream, which orders it after
  !! the copies.  Unpacking (scale_factor ...), windows and the mesh_zgr names are readvar3d's, unchanged.
  !! CDFT_FEED_MB (slot size, default 128) and CDFT_FEED_SLOTS (default 4) size the ring; CDFT_FEED=0 switches the programs
  !! back to whole records in pageable memory.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE cdftools_hip
  IMPLICIT NONE
  PRIVATE

  TYPE(C_PTR), PUBLIC :: feed_stream = C_NULL_PTR          ! the stream the copies are queued on
  TYPE(C_PTR)         :: hfeed = C_NULL_PTR
  INTEGER(KIND=8)     :: nslot_bytes = 0

  PUBLIC :: feed_wanted, feed_open, feed_close, feed_field, feed_mask, feed_host, feed_device, feed_release

CONTAINS

  LOGICAL FUNCTION feed_wanted()
    !! .FALSE. with CDFT_FEED=0 in the environment
    CHARACTER(LEN=8) :: cl
    CALL GET_ENVIRONMENT_VARIABLE('CDFT_FEED', cl)
    feed_wanted = ( cl(1:1) /= '0' )
  END FUNCTION feed_wanted

  SUBROUTINE feed_open()
    INTEGER(KIND=4) :: imb, islots
    CHARACTER(LEN=16) :: cl
    IF ( C_ASSOCIATED(hfeed) ) RETURN
    imb = 128 ; islots = 4
    CALL GET_ENVIRONMENT_VARIABLE('CDFT_FEED_MB', cl)    ; IF ( LEN_TRIM(cl) > 0 ) READ(cl, *) imb
    CALL GET_ENVIRONMENT_VARIABLE('CDFT_FEED_SLOTS', cl) ; IF ( LEN_TRIM(cl) > 0 ) READ(cl, *) islots
    nslot_bytes = INT(MAX(imb, 1), 8) * 1048576_8
    CALL hip_check(cdft_feed_create(hfeed, INT(nslot_bytes, C_SIZE_T), MAX(2, islots)), 'cdft_feed_create')
    CALL hip_check(cdft_feed_stream(hfeed, feed_stream), 'cdft_feed_stream')
  END SUBROUTINE feed_open

  SUBROUTINE feed_close()
    IF ( C_ASSOCIATED(hfeed) ) CALL hip_check(cdft_feed_destroy(hfeed), 'cdft_feed_destroy')
    hfeed = C_NULL_PTR ; feed_stream = C_NULL_PTR
  END SUBROUTINE feed_close

  FUNCTION feed_device(kbytes) RESULT(pdev)
    !! kbytes of HBM (the destination of feed_field / feed_mask / feed_host)
    INTEGER(KIND=8), INTENT(in) :: kbytes
    TYPE(C_PTR) :: pdev
    CALL hip_check(cdft_malloc(pdev, INT(kbytes, C_SIZE_T)), 'cdft_malloc')
  END FUNCTION feed_device

  SUBROUTINE feed_release(pdev)
    TYPE(C_PTR), INTENT(inout) :: pdev
    INTEGER(KIND=4) :: ierr
    IF ( C_ASSOCIATED(pdev) ) ierr = cdft_free(pdev)
    pdev = C_NULL_PTR
  END SUBROUTINE feed_release

  SUBROUTINE feed_field(cdfile, cdvar, pdev, kpi, kpj, kpk, ktime, ldiom, kimin, kjmin, kkmin)
    !! the (kpi,kpj,kpk) r4 block of record ktime (window origin kimin, kjmin, kkmin; ldiom as in getvar) into the device
    !! array pdev, slab by slab through the ring.  Returns when the last slab is QUEUED.
    CHARACTER(LEN=*),          INTENT(in) :: cdfile, cdvar
    TYPE(C_PTR),               INTENT(in) :: pdev
    INTEGER(KIND=4),           INTENT(in) :: kpi, kpj, kpk
    INTEGER(KIND=4), OPTIONAL, INTENT(in) :: ktime, kimin, kjmin, kkmin
    LOGICAL,         OPTIONAL, INTENT(in) :: ldiom
    INTEGER(KIND=4) :: ik0, inl, inlmax, ikfirst
    INTEGER(KIND=8) :: islab
    INTEGER(C_SIZE_T) :: ibytes
    TYPE(C_PTR) :: pslot
    REAL(KIND=4), POINTER :: zslab(:,:,:)
    CALL feed_open()
    islab = 4_8 * kpi * kpj
    IF ( islab > nslot_bytes ) CALL feed_resize(islab)
    inlmax = INT(MAX(1_8, nslot_bytes / islab), 4)
    ikfirst = 1 ; IF ( PRESENT(kkmin) ) ikfirst = kkmin
    DO ik0 = 1, kpk, inlmax
       inl = MIN(inlmax, kpk - ik0 + 1)
       CALL hip_check(cdft_feed_slot(hfeed, pslot, ibytes), 'cdft_feed_slot')
       CALL C_F_POINTER(pslot, zslab, (/ kpi, kpj, inl /))
       CALL readvar3d(cdfile, cdvar, zslab, ktime=ktime, ldiom=ldiom, kimin=kimin, kjmin=kjmin, kkmin=ikfirst + ik0 - 1)
       CALL hip_check(cdft_feed_push(hfeed, pdev, INT(islab * (ik0 - 1), C_SIZE_T), INT(islab * inl, C_SIZE_T)), 'cdft_feed_push')
    END DO
  END SUBROUTINE feed_field

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
    CALL feed_open()
    islab = 1_8 * kpi * kpj
    IF ( islab > nslot_bytes ) CALL feed_resize(islab)
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
    !! kbytes of a host array (C_LOC of it) into the device array pdev through the ring
    TYPE(C_PTR),     INTENT(in) :: psrc, pdev
    INTEGER(KIND=8), INTENT(in) :: kbytes
    INTEGER(KIND=8) :: ioff, in
    INTEGER(C_SIZE_T) :: ibytes
    TYPE(C_PTR) :: pslot
    INTEGER(KIND=1), POINTER :: isrc(:), idst(:)
    CALL feed_open()
    CALL C_F_POINTER(psrc, isrc, (/ kbytes /))
    ioff = 0
    DO WHILE ( ioff < kbytes )
       in = MIN(nslot_bytes, kbytes - ioff)
       CALL hip_check(cdft_feed_slot(hfeed, pslot, ibytes), 'cdft_feed_slot')
       CALL C_F_POINTER(pslot, idst, (/ in /))
       idst(1:in) = isrc(ioff+1:ioff+in)
       CALL hip_check(cdft_feed_push(hfeed, pdev, INT(ioff, C_SIZE_T), INT(in, C_SIZE_T)), 'cdft_feed_push')
       ioff = ioff + in
    END DO
  END SUBROUTINE feed_host

  SUBROUTINE feed_resize(kbytes)
    ! one level does not fit a slot (a very large plane): a new ring whose slots hold one
    INTEGER(KIND=8), INTENT(in) :: kbytes
    CALL hip_check(cdft_feed_destroy(hfeed), 'cdft_feed_destroy')
    nslot_bytes = kbytes
    CALL hip_check(cdft_feed_create(hfeed, INT(nslot_bytes, C_SIZE_T), 3), 'cdft_feed_create')
    CALL hip_check(cdft_feed_stream(hfeed, feed_stream), 'cdft_feed_stream')
  END SUBROUTINE feed_resize

END MODULE cdffeed
