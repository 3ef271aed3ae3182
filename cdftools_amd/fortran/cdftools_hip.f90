MODULE cdftools_hip
  !!======================================================================
  !! cdftools_hip : the thin ISO_C_BINDING shim between the Fortran host programs and the hand-written
  !! HIP kernels in libcdftools_hip.so.  One interface per entry point of include/cdftools_hip.h (same
  !! names, same argument order); arrays are passed as they sit in the Fortran program (column-major,
  !! assumed-size), plans are opaque TYPE(C_PTR).  hip_check turns a non-zero status into the
  !! reference's fatal-error convention (message + STOP 98, src/cdfio.F90:2924-2940).
  !!======================================================================
  USE ISO_C_BINDING
  IMPLICIT NONE
  PUBLIC

  INTEGER(C_INT), PARAMETER :: CDFT_OK = 0, CDFT_MEM_HOST = 0, CDFT_MEM_DEVICE = 1, CDFT_MEM_PINNED = 2
  INTEGER(C_INT), PARAMETER :: CDFT_COLLECTIVE_PEER = 0, CDFT_COLLECTIVE_RCCL = 1
  INTEGER(C_INT), PARAMETER :: CDFT_HEATC_SUM_R8 = 0, CDFT_HEATC_SUM_R4SEQ = 1
  INTEGER(C_INT), PARAMETER :: CDFT_MEAN_FORM_MEAN = 0, CDFT_MEAN_FORM_SUM = 1, CDFT_MEAN_FORM_SUM_FORCING = 2, &
       &                       CDFT_MEAN_FORM_MEAN_NOVAR = 3
  INTEGER(C_INT), PARAMETER :: CDFT_EOS_80 = 0, CDFT_EOS_TEOS10 = 1, CDFT_EOS_JMD94 = 2, CDFT_EOS_NTR = 3, CDFT_EOS_FAST_FMA = 1
  INTEGER(C_INT), PARAMETER :: CDFT_EOS_MCD87 = 4
  INTEGER(C_INT), PARAMETER :: CDFT_EOSQ_ALPHA = 0, CDFT_EOSQ_BETA = 1, CDFT_EOSQ_ALBET_MCD87 = 2, CDFT_EOSQ_BETA_MCD87 = 3
  INTEGER(C_INT), PARAMETER :: CDFT_EOSQ_SPICE_POL = 4, CDFT_EOSQ_SPICE_F02 = 5

  INTERFACE
     FUNCTION cdft_last_error() BIND(C, name='cdft_last_error')
       IMPORT ; TYPE(C_PTR) :: cdft_last_error
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_device_count(n) BIND(C, name='cdft_device_count')
       IMPORT ; INTEGER(C_INT), INTENT(out) :: n
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_set_device(idev) BIND(C, name='cdft_set_device')
       IMPORT ; INTEGER(C_INT), VALUE :: idev
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_malloc(ptr, bytes) BIND(C, name='cdft_malloc')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: ptr ; INTEGER(C_SIZE_T), VALUE :: bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_free(ptr) BIND(C, name='cdft_free')
       IMPORT ; TYPE(C_PTR), VALUE :: ptr
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_memcpy_h2d(dst, src, bytes, stream) BIND(C, name='cdft_memcpy_h2d')
       IMPORT ; TYPE(C_PTR), VALUE :: dst, src, stream ; INTEGER(C_SIZE_T), VALUE :: bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_memcpy_d2h(dst, src, bytes, stream) BIND(C, name='cdft_memcpy_d2h')     ! (synchronises)
       IMPORT ; TYPE(C_PTR), VALUE :: dst, src, stream ; INTEGER(C_SIZE_T), VALUE :: bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_stream_sync(stream) BIND(C, name='cdft_stream_sync')
       IMPORT ; TYPE(C_PTR), VALUE :: stream
     END FUNCTION
     ! ---- sharded plans: one process, several GPUs, row blocks (include/cdftools_hip.h "sharded plans") ------
     INTEGER(C_INT) FUNCTION cdft_shard_from_env(ndev, devices, capacity) BIND(C, name='cdft_shard_from_env')
       IMPORT ; INTEGER(C_INT), INTENT(out) :: ndev, devices(*) ; INTEGER(C_INT), VALUE :: capacity
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_shard_set_collective(mode) BIND(C, name='cdft_shard_set_collective')
       IMPORT ; INTEGER(C_INT), VALUE :: mode
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_shard_get_collective(mode) BIND(C, name='cdft_shard_get_collective')
       IMPORT ; INTEGER(C_INT), INTENT(out) :: mode
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mem_info(free_bytes, total_bytes) BIND(C, name='cdft_mem_info')
       IMPORT ; INTEGER(C_SIZE_T), INTENT(out) :: free_bytes, total_bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_host_alloc(ptr, bytes) BIND(C, name='cdft_host_alloc')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: ptr ; INTEGER(C_SIZE_T), VALUE :: bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_host_free(ptr) BIND(C, name='cdft_host_free')
       IMPORT ; TYPE(C_PTR), VALUE :: ptr
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_create(feed, slot_bytes, nslots) BIND(C, name='cdft_feed_create')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: feed ; INTEGER(C_SIZE_T), VALUE :: slot_bytes ; INTEGER(C_INT), VALUE :: nslots
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_resize(feed, slot_bytes, nslots) BIND(C, name='cdft_feed_resize')
       IMPORT ; TYPE(C_PTR), VALUE :: feed ; INTEGER(C_SIZE_T), VALUE :: slot_bytes ; INTEGER(C_INT), VALUE :: nslots
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_slot(feed, host, slot_bytes) BIND(C, name='cdft_feed_slot')
       IMPORT ; TYPE(C_PTR), VALUE :: feed ; TYPE(C_PTR), INTENT(out) :: host ; INTEGER(C_SIZE_T), INTENT(out) :: slot_bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_push(feed, dst, dst_offset, bytes) BIND(C, name='cdft_feed_push')
       IMPORT ; TYPE(C_PTR), VALUE :: feed, dst ; INTEGER(C_SIZE_T), VALUE :: dst_offset, bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_push_rows(feed, dst, dst_offset, dst_pitch, row_bytes, nrows) BIND(C, name='cdft_feed_push_rows')
       IMPORT ; TYPE(C_PTR), VALUE :: feed, dst ; INTEGER(C_SIZE_T), VALUE :: dst_offset, dst_pitch, row_bytes, nrows
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_push_i1_as_r4(feed, dst, dst_offset, n) BIND(C, name='cdft_feed_push_i1_as_r4')
       IMPORT ; TYPE(C_PTR), VALUE :: feed, dst ; INTEGER(C_SIZE_T), VALUE :: dst_offset, n
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_stream(feed, stream) BIND(C, name='cdft_feed_stream')
       IMPORT ; TYPE(C_PTR), VALUE :: feed ; TYPE(C_PTR), INTENT(out) :: stream
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_wait(feed) BIND(C, name='cdft_feed_wait')
       IMPORT ; TYPE(C_PTR), VALUE :: feed
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_feed_destroy(feed) BIND(C, name='cdft_feed_destroy')
       IMPORT ; TYPE(C_PTR), VALUE :: feed
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_create_sharded(plan, npiglo, npjglo, npk, nbasins, ndev, devices) &
          &         BIND(C, name='cdft_moc_create_sharded')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, nbasins, ndev
       INTEGER(C_INT), INTENT(in) :: devices(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_heatc_create_sharded(plan, npiglo, npjglo, npk, ndev, devices) &
          &         BIND(C, name='cdft_heatc_create_sharded')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, ndev
       INTEGER(C_INT), INTENT(in) :: devices(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mhst_create_sharded(plan, npiglo, npjglo, npk, nbasins, ndev, devices) &
          &         BIND(C, name='cdft_mhst_create_sharded')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, nbasins, ndev
       INTEGER(C_INT), INTENT(in) :: devices(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_psi_create_sharded(plan, npiglo, npjglo, npk, ndev, devices) &
          &         BIND(C, name='cdft_psi_create_sharded')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, ndev
       INTEGER(C_INT), INTENT(in) :: devices(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_vt_compute_sharded(u, v, t, s, e3u, e3v, fstride, nframes, npiglo, npjglo, npk, &
          &                                  vt, vs, ut, us, e3um, e3vm, out_is_r4, ndev, devices)               &
          &         BIND(C, name='cdft_vt_compute_sharded')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: u(*), v(*), t(*), s(*) ; TYPE(C_PTR), VALUE :: e3u, e3v, e3um, e3vm
       INTEGER(C_INT64_T), VALUE :: fstride ; INTEGER(C_INT), VALUE :: nframes, npiglo, npjglo, npk
       TYPE(C_PTR), VALUE :: vt, vs, ut, us ; INTEGER(C_INT), VALUE :: out_is_r4, ndev ; INTEGER(C_INT), INTENT(in) :: devices(*)
     END FUNCTION
     ! ---- cdfmoc ---------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_moc_create(plan, npiglo, npjglo, npk, nbasins) BIND(C, name='cdft_moc_create')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, nbasins
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_set_mesh(plan, e1v, e3v, ibmask, memspace, stream) BIND(C, name='cdft_moc_set_mesh')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: e1v(*), e3v(*)
       INTEGER(C_INT16_T), INTENT(in) :: ibmask(*) ; INTEGER(C_INT), VALUE :: memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_set_mesh_vmask(plan, e1v, e3v, vmask, ibmask, memspace, stream) &
          &         BIND(C, name='cdft_moc_set_mesh_vmask')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: e1v(*), e3v(*)
       INTEGER(C_INT8_T), INTENT(in) :: vmask(*) ; INTEGER(C_INT16_T), INTENT(in) :: ibmask(*) ; INTEGER(C_INT), VALUE :: memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_compute(plan, zv, rec_stride, npt, zv_space, dmoc, dmoc_space, stream) &
          &         BIND(C, name='cdft_moc_compute')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: zv(*)
       INTEGER(C_INT64_T), VALUE :: rec_stride ; INTEGER(C_INT), VALUE :: npt, zv_space, dmoc_space
       REAL(C_DOUBLE), INTENT(out) :: dmoc(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stage(plan, zv, rec_stride, npt, zv_space, stream) BIND(C, name='cdft_moc_stage')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: zv(*)
       INTEGER(C_INT64_T), VALUE :: rec_stride ; INTEGER(C_INT), VALUE :: npt, zv_space
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_compute_staged(plan, dmoc, dmoc_space, stream) BIND(C, name='cdft_moc_compute_staged')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: dmoc_space ; REAL(C_DOUBLE), INTENT(out) :: dmoc(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_describe(plan, buf, buflen) BIND(C, name='cdft_moc_describe')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; CHARACTER(KIND=C_CHAR), INTENT(out) :: buf(*) ; INTEGER(C_SIZE_T), VALUE :: buflen
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_set_basins(plan, e1v, ibmask, memspace, stream) BIND(C, name='cdft_moc_set_basins')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: e1v(*)
       INTEGER(C_INT16_T), INTENT(in) :: ibmask(*) ; INTEGER(C_INT), VALUE :: memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stream_begin(plan, npt, nlev, stream) BIND(C, name='cdft_moc_stream_begin')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: npt, nlev
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stream_dest(plan, slot, level, dst, row_pitch_bytes) BIND(C, name='cdft_moc_stream_dest')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; INTEGER(C_INT), VALUE :: slot, level
       TYPE(C_PTR), INTENT(out) :: dst ; INTEGER(C_SIZE_T), INTENT(out) :: row_pitch_bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stream_vmask_dest(plan, level, dst) BIND(C, name='cdft_moc_stream_vmask_dest')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; INTEGER(C_INT), VALUE :: level ; TYPE(C_PTR), INTENT(out) :: dst
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stream_compute(plan, k0, nlev, with_vmask, stream) BIND(C, name='cdft_moc_stream_compute')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: k0, nlev, with_vmask
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stream_end(plan, dmoc, dmoc_space, stream) BIND(C, name='cdft_moc_stream_end')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: dmoc_space ; REAL(C_DOUBLE), INTENT(out) :: dmoc(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stage_open(plan, npt, stream) BIND(C, name='cdft_moc_stage_open')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: npt
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stage_dest(plan, record, level, dst, row_pitch_bytes) BIND(C, name='cdft_moc_stage_dest')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; INTEGER(C_INT), VALUE :: record, level
       TYPE(C_PTR), INTENT(out) :: dst ; INTEGER(C_SIZE_T), INTENT(out) :: row_pitch_bytes
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stage_close(plan) BIND(C, name='cdft_moc_stage_close')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_release_staged(plan) BIND(C, name='cdft_moc_release_staged')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stats(plan, zonal_launches, records) BIND(C, name='cdft_moc_stats')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; INTEGER(C_INT64_T), INTENT(out) :: zonal_launches, records
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_compute_staged_windows(plan, nlev, dmoc, dmoc_space, stream) &
          &         BIND(C, name='cdft_moc_compute_staged_windows')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: nlev, dmoc_space ; REAL(C_DOUBLE), INTENT(out) :: dmoc(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_reserve(plan, npt, stream) BIND(C, name='cdft_moc_reserve')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: npt
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_stage_rows(plan, zv, rec_stride, lev_stride, npt, zv_space, stream) &
          &         BIND(C, name='cdft_moc_stage_rows')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: zv(*)
       INTEGER(C_INT64_T), VALUE :: rec_stride, lev_stride ; INTEGER(C_INT), VALUE :: npt, zv_space
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_compute_pitched(plan, zv, rec_stride, row_stride, npt, dmoc, dmoc_space, stream) &
          &         BIND(C, name='cdft_moc_compute_pitched')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: zv(*)
       INTEGER(C_INT64_T), VALUE :: rec_stride, row_stride ; INTEGER(C_INT), VALUE :: npt, dmoc_space
       REAL(C_DOUBLE), INTENT(out) :: dmoc(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_profile_read_block(plan, block, ms_each, capacity, launches) &
          &         BIND(C, name='cdft_moc_profile_read_block')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; INTEGER(C_INT), VALUE :: block, capacity
       REAL(C_FLOAT), INTENT(out) :: ms_each(*) ; INTEGER(C_INT), INTENT(out) :: launches
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_block_info(plan, block, device, j0, j1) BIND(C, name='cdft_moc_block_info')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; INTEGER(C_INT), VALUE :: block ; INTEGER(C_INT), INTENT(out) :: device, j0, j1
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_decomp_set_mesh(plan, e3v, e1u, zcoef, gdept, tmask, umask, memspace, stream) &
          &         BIND(C, name='cdft_moc_decomp_set_mesh')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, tmask, umask
       REAL(C_FLOAT), INTENT(in) :: e3v(*), e1u(*), zcoef(*), gdept(*) ; INTEGER(C_INT), VALUE :: memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_mask_e3v(e3v, vmask, n, stream) BIND(C, name='cdft_moc_mask_e3v')
       IMPORT ; TYPE(C_PTR), VALUE :: e3v, vmask, stream ; INTEGER(C_INT64_T), VALUE :: n
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_decomp(plan, zv, ztemp, zsal, eos, in_space, dmoc, dmoc_sh, dmoc_bt, dmoc_ag, &
          &                                  out_space, stream) BIND(C, name='cdft_moc_decomp')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: zv(*), ztemp(*), zsal(*)
       INTEGER(C_INT), VALUE :: eos, in_space, out_space
       REAL(C_DOUBLE), INTENT(inout) :: dmoc(*), dmoc_sh(*), dmoc_bt(*), dmoc_ag(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_moc_destroy(plan) BIND(C, name='cdft_moc_destroy')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     ! ---- eos ------------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_eos_sigmai(ptem, psal, n, pref, keos, kflags, sig, in_space, out_space, stream) &
          &         BIND(C, name='cdft_eos_sigmai')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: ptem(*), psal(*) ; INTEGER(C_INT64_T), VALUE :: n
       REAL(C_FLOAT), VALUE :: pref ; INTEGER(C_INT), VALUE :: keos, kflags, in_space, out_space
       REAL(C_DOUBLE), INTENT(out) :: sig(*) ; TYPE(C_PTR), VALUE :: stream
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_eos_sigmai2d(ptem, psal, pref2d, n, keos, kflags, sig, in_space, out_space, stream) &
          &         BIND(C, name='cdft_eos_sigmai2d')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: ptem(*), psal(*), pref2d(*) ; INTEGER(C_INT64_T), VALUE :: n
       INTEGER(C_INT), VALUE :: keos, kflags, in_space, out_space
       REAL(C_DOUBLE), INTENT(out) :: sig(*) ; TYPE(C_PTR), VALUE :: stream
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_eos_sig_tool(ptem, psal, n, pref, spval, keos, kflags, sig4, in_space, out_space, stream) &
          &         BIND(C, name='cdft_eos_sig_tool')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: ptem(*), psal(*) ; INTEGER(C_INT64_T), VALUE :: n
       REAL(C_FLOAT), VALUE :: pref, spval ; INTEGER(C_INT), VALUE :: keos, kflags, in_space, out_space
       REAL(C_FLOAT), INTENT(out) :: sig4(*) ; TYPE(C_PTR), VALUE :: stream
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_eos_field(kquantity, ptem, psal, pdep_n, pdep, n, keos, dout, in_space, out_space, stream) &
          &         BIND(C, name='cdft_eos_field')
       IMPORT ; INTEGER(C_INT), VALUE :: kquantity, keos, in_space, out_space ; REAL(C_FLOAT), INTENT(in) :: ptem(*), psal(*)
       TYPE(C_PTR), VALUE :: pdep_n, stream ; REAL(C_FLOAT), VALUE :: pdep ; INTEGER(C_INT64_T), VALUE :: n
       REAL(C_DOUBLE), INTENT(out) :: dout(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_eos_bn2(ptem_up, ptem_down, psal_up, psal_down, pe3w, pdep, n, keos, pbn2, in_space,     &
          &         out_space, stream) BIND(C, name='cdft_eos_bn2')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: ptem_up(*), ptem_down(*), psal_up(*), psal_down(*), pe3w(*)
       REAL(C_FLOAT), VALUE :: pdep ; INTEGER(C_INT64_T), VALUE :: n ; INTEGER(C_INT), VALUE :: keos, in_space, out_space
       REAL(C_FLOAT), INTENT(out) :: pbn2(*) ; TYPE(C_PTR), VALUE :: stream
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_eos_point(kquantity, dtem, dsal, ddep, keos, dout) BIND(C, name='cdft_eos_point')
       IMPORT ; INTEGER(C_INT), VALUE :: kquantity, keos ; REAL(C_DOUBLE), VALUE :: dtem, dsal, ddep
       REAL(C_DOUBLE), INTENT(out) :: dout
     END FUNCTION
     ! ---- cdfheatc -------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_heatc_create(plan, npiglo, npjglo, npk) BIND(C, name='cdft_heatc_create')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_heatc_set_sum_mode(plan, mode) BIND(C, name='cdft_heatc_set_sum_mode')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; INTEGER(C_INT), VALUE :: mode
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_heatc_set_mesh(plan, e1t, e2t, e3t, e31d, tmask, mask_is_r4, memspace, stream) &
          &         BIND(C, name='cdft_heatc_set_mesh')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, e3t, e31d, tmask ; REAL(C_FLOAT), INTENT(in) :: e1t(*), e2t(*)
       INTEGER(C_INT), VALUE :: mask_is_r4, memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_heatc_compute(plan, temp, temp_space, rmxldep, gdepw, mxloption, sums, out_space, stream) &
          &         BIND(C, name='cdft_heatc_compute')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, rmxldep, gdepw ; REAL(C_FLOAT), INTENT(in) :: temp(*)
       INTEGER(C_INT), VALUE :: temp_space, mxloption, out_space ; REAL(C_DOUBLE), INTENT(out) :: sums(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_heatc_destroy(plan) BIND(C, name='cdft_heatc_destroy')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     ! ---- cdfmhst --------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_mhst_create(plan, npiglo, npjglo, npk, nbasins) BIND(C, name='cdft_mhst_create')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, nbasins
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mhst_set_mesh(plan, e1v, e3v, e31d, masks, memspace, stream) BIND(C, name='cdft_mhst_set_mesh')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, e3v, e31d ; REAL(C_FLOAT), INTENT(in) :: e1v(*), masks(*)
       INTEGER(C_INT), VALUE :: memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mhst_compute(plan, vt, vs, zv, zt, zs, in_space, zdim, heat, salt, out_space, stream) &
          &         BIND(C, name='cdft_mhst_compute')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, vt, vs, zv, zt, zs ; INTEGER(C_INT), VALUE :: in_space, zdim, out_space
       REAL(C_DOUBLE), INTENT(out) :: heat(*), salt(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mhst_destroy(plan) BIND(C, name='cdft_mhst_destroy')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     ! ---- cdfvtrp, cdfvhst -----------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_vint_transport(vel, m2d, e3, e31d, npiglo, npjglo, npk, in_space, dtrp, out_space, stream) &
          &         BIND(C, name='cdft_vint_transport')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: vel(*), m2d(*) ; TYPE(C_PTR), VALUE :: e3, e31d, stream
       INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, in_space, out_space ; REAL(C_DOUBLE), INTENT(out) :: dtrp(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_vint_slope_rotate(dtrpu, dtrpv, hdepw, e1u, e2v, tmask, npiglo, npjglo, in_space, dalong, &
          &                                         dcross, out_space, stream) BIND(C, name='cdft_vint_slope_rotate')
       IMPORT ; REAL(C_DOUBLE), INTENT(in) :: dtrpu(*), dtrpv(*) ; REAL(C_FLOAT), INTENT(in) :: hdepw(*), e1u(*), e2v(*), tmask(*)
       INTEGER(C_INT), VALUE :: npiglo, npjglo, in_space, out_space ; TYPE(C_PTR), VALUE :: stream
       REAL(C_DOUBLE), INTENT(out) :: dalong(*), dcross(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_vhst_transport(xt, xs, m2d, e3, e31d, npiglo, npjglo, npk, in_space, dheat, dsalt, &
          &                                      out_space, stream) BIND(C, name='cdft_vhst_transport')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: xt(*), xs(*), m2d(*) ; TYPE(C_PTR), VALUE :: e3, e31d, stream
       INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, in_space, out_space ; REAL(C_DOUBLE), INTENT(out) :: dheat(*), dsalt(*)
     END FUNCTION
     ! ---- cdfmean --------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_mean_create(plan, npiglo, npjglo, npk, nbasins) BIND(C, name='cdft_mean_create')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, nbasins
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mean_set_mesh(plan, e1, e2, e3, e31d, zmask, ibmask, memspace, stream) &
          &         BIND(C, name='cdft_mean_set_mesh')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, e3, e31d ; REAL(C_FLOAT), INTENT(in) :: e1(*), e2(*), zmask(*)
       INTEGER(C_INT16_T), INTENT(in) :: ibmask(*) ; INTEGER(C_INT), VALUE :: memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mean_compute(plan, zv, in_space, sums, out_space, stream) BIND(C, name='cdft_mean_compute')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: zv(*)
       INTEGER(C_INT), VALUE :: in_space, out_space ; REAL(C_DOUBLE), INTENT(out) :: sums(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mean_set_form(plan, form) BIND(C, name='cdft_mean_set_form')
       IMPORT ; TYPE(C_PTR), VALUE :: plan ; INTEGER(C_INT), VALUE :: form
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_mean_destroy(plan) BIND(C, name='cdft_mean_destroy')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     ! ---- cdfzonalsum, cdfzonalmean --------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_zonal_create(plan, npiglo, npjglo, npk, nbasins) BIND(C, name='cdft_zonal_create')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, nbasins
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_zonal_set_mesh(plan, e1, e2, zmaskvar, zmask, memspace, stream) BIND(C, name='cdft_zonal_set_mesh')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; REAL(C_FLOAT), INTENT(in) :: e1(*), e2(*), zmaskvar(*), zmask(*)
       INTEGER(C_INT), VALUE :: memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_zonal_compute(plan, zv, nlev, in_space, mean_form, dzo, darea, rzomax, rzomin, out_space, &
          &                                     stream) BIND(C, name='cdft_zonal_compute')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, darea, rzomax, rzomin ; REAL(C_FLOAT), INTENT(in) :: zv(*)
       INTEGER(C_INT), VALUE :: nlev, in_space, mean_form, out_space ; REAL(C_DOUBLE), INTENT(out) :: dzo(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_zonal_area(plan, nlev, darea, out_space, stream) BIND(C, name='cdft_zonal_area')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: nlev, out_space ; REAL(C_DOUBLE), INTENT(out) :: darea(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_zonal_destroy(plan) BIND(C, name='cdft_zonal_destroy')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     ! ---- cdfvint, cdfvertmean -------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_vint_cumul(zt, e3t, e31d, tmask, npiglo, npjglo, npk, kout, khit, clip, rdep2, npko, sclf, &
          &                                  mean, in_space, zout, out_space, stream) BIND(C, name='cdft_vint_cumul')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: zt(*), tmask(*), clip(*), rdep2(*) ; TYPE(C_PTR), VALUE :: e3t, e31d, stream
       INTEGER(C_INT), INTENT(in) :: kout(*), khit(*) ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, npko, mean, in_space, out_space
       REAL(C_FLOAT), VALUE :: sclf ; REAL(C_FLOAT), INTENT(out) :: zout(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_vertmean(zv, e3, e31d, zmask, npiglo, npjglo, nlev, gdep_first, gdep_last, rdep_up, rdep_down, &
          &                                in_space, dvertmean, out_space, stream) BIND(C, name='cdft_vertmean')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: zv(*), zmask(*) ; TYPE(C_PTR), VALUE :: e3, e31d, stream
       INTEGER(C_INT), VALUE :: npiglo, npjglo, nlev, in_space, out_space
       REAL(C_FLOAT), VALUE :: gdep_first, gdep_last, rdep_up, rdep_down ; REAL(C_DOUBLE), INTENT(out) :: dvertmean(*)
     END FUNCTION
     ! ---- cdfmocsig ------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_mocsig_compute(zv, zveiv, zt, zs, e1v, e3v, e31d, ibmask, gdep, npiglo, npjglo, npk, nbasins,  &
          &         zspv, zspt, zsps, pref, keos, nbins, sigmin, sigstp, in_space, dmoc, depi, out_space, stream)              &
          &         BIND(C, name='cdft_mocsig_compute')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: zv(*), zt(*), zs(*), e1v(*) ; TYPE(C_PTR), VALUE :: zveiv, e3v, e31d, gdep, depi, stream
       INTEGER(C_INT16_T), INTENT(in) :: ibmask(*) ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk, nbasins, keos, nbins, in_space, out_space
       REAL(C_FLOAT), VALUE :: zspv, zspt, zsps, pref, sigmin, sigstp ; REAL(C_DOUBLE), INTENT(out) :: dmoc(*)
     END FUNCTION
     ! ---- cdfpsi ---------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_psi_create(plan, npiglo, npjglo, npk) BIND(C, name='cdft_psi_create')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo, npk
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_psi_compute(plan, zu, e2u, e3u, zv, e1v, e3v, e31d, iiref, ijref, fmask, in_space, &
          &                                   psi4, dpsi8, out_space, stream) BIND(C, name='cdft_psi_compute')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, zu, e2u, e3u, zv, e1v, e3v, e31d, fmask, dpsi8
       INTEGER(C_INT), VALUE :: iiref, ijref, in_space, out_space ; REAL(C_FLOAT), INTENT(out) :: psi4(*)
     END FUNCTION
     ! staged form for a row-block partitioned domain (one plan per block, see include/cdftools_hip.h)
     INTEGER(C_INT) FUNCTION cdft_psi_integrate(plan, zu, e2u, e3u, zv, e1v, e3v, e31d, in_space, stream) &
          &         BIND(C, name='cdft_psi_integrate')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, zu, e2u, e3u, zv, e1v, e3v, e31d ; INTEGER(C_INT), VALUE :: in_space
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_psi_scan_u(plan, carry, carry_space, first_block, last_row, last_space, stream) &
          &         BIND(C, name='cdft_psi_scan_u')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, carry, last_row ; INTEGER(C_INT), VALUE :: carry_space, first_block, last_space
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_psi_scan_open(plan, lmean, stream) BIND(C, name='cdft_psi_scan_open')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: lmean
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_psi_value(plan, ii, ij, dvalue, stream) BIND(C, name='cdft_psi_value')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream ; INTEGER(C_INT), VALUE :: ii, ij ; REAL(C_DOUBLE), INTENT(out) :: dvalue
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_psi_finish(plan, use_ref, ref_value, iiref, ijref, fmask, mask_space, psi4, dpsi8, &
          &                                  out_space, stream) BIND(C, name='cdft_psi_finish')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, fmask, dpsi8 ; REAL(C_DOUBLE), VALUE :: ref_value
       INTEGER(C_INT), VALUE :: use_ref, iiref, ijref, mask_space, out_space ; REAL(C_FLOAT), INTENT(out) :: psi4(*)
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_psi_destroy(plan) BIND(C, name='cdft_psi_destroy')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     ! ---- cdfvT ----------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_vt_compute(u, v, t, s, e3u, e3v, fstride, nframes, npiglo, npjglo, npk, in_space, &
          &                                  vt, vs, ut, us, e3um, e3vm, out_is_r4, out_space, stream)                &
          &         BIND(C, name='cdft_vt_compute')
       IMPORT ; REAL(C_FLOAT), INTENT(in) :: u(*), v(*), t(*), s(*) ; TYPE(C_PTR), VALUE :: e3u, e3v, e3um, e3vm, stream
       INTEGER(C_INT64_T), VALUE :: fstride ; INTEGER(C_INT), VALUE :: nframes, npiglo, npjglo, npk, in_space
       TYPE(C_PTR), VALUE :: vt, vs, ut, us ; INTEGER(C_INT), VALUE :: out_is_r4, out_space
     END FUNCTION
     ! ---- cdfcurl --------------------------------------------------------------------------
     INTEGER(C_INT) FUNCTION cdft_curl_create(plan, npiglo, npjglo) BIND(C, name='cdft_curl_create')
       IMPORT ; TYPE(C_PTR), INTENT(out) :: plan ; INTEGER(C_INT), VALUE :: npiglo, npjglo
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_curl_set_mesh(plan, e1u, e2v, e1f, e2f, ff, lperio, memspace, stream) &
          &         BIND(C, name='cdft_curl_set_mesh')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, ff ; REAL(C_FLOAT), INTENT(in) :: e1u(*), e2v(*), e1f(*), e2f(*)
       INTEGER(C_INT), VALUE :: lperio, memspace
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_curl_compute(plan, un, vn, nlev, in_space, ltpoint, rotn, out_is_r8, out_space, stream) &
          &         BIND(C, name='cdft_curl_compute')
       IMPORT ; TYPE(C_PTR), VALUE :: plan, stream, rotn ; REAL(C_FLOAT), INTENT(in) :: un(*), vn(*)
       INTEGER(C_INT), VALUE :: nlev, in_space, ltpoint, out_is_r8, out_space
     END FUNCTION
     INTEGER(C_INT) FUNCTION cdft_curl_destroy(plan) BIND(C, name='cdft_curl_destroy')
       IMPORT ; TYPE(C_PTR), VALUE :: plan
     END FUNCTION
     INTEGER(C_SIZE_T) FUNCTION hip_strlen(s) BIND(C, name='strlen')
       IMPORT ; TYPE(C_PTR), VALUE :: s
     END FUNCTION
  END INTERFACE

CONTAINS

  SUBROUTINE hip_shard(kgpu, kdev)
    !! The row blocks the run is split into: CDFT_GPU_MAP="a,b,.." (block b on HIP device map(b)) or CDFT_GPUS=n
    !! (devices 0..n-1); kgpu = 1 without either.  With one block on a named device that device becomes current.
    INTEGER(C_INT), INTENT(out) :: kgpu, kdev(64)
    CALL hip_check(cdft_shard_from_env(kgpu, kdev, 64_C_INT), 'cdft_shard_from_env')
    IF ( kgpu == 1 .AND. kdev(1) >= 0 ) CALL hip_check(cdft_set_device(kdev(1)), 'cdft_set_device')
  END SUBROUTINE hip_shard

  SUBROUTINE hip_check(kstatus, cdwhere)
    !! non-zero status of a libcdftools_hip call: print its message and stop like a NetCDF failure
    INTEGER(C_INT),   INTENT(in) :: kstatus
    CHARACTER(LEN=*), INTENT(in) :: cdwhere
    TYPE(C_PTR) :: p
    CHARACTER(KIND=C_CHAR), POINTER :: c(:)
    CHARACTER(LEN=512) :: cl
    INTEGER :: i, n
    IF ( kstatus == CDFT_OK ) RETURN
    cl = ' '
    p = cdft_last_error()
    IF ( C_ASSOCIATED(p) ) THEN
       n = MIN(INT(hip_strlen(p)), LEN(cl))
       CALL C_F_POINTER(p, c, (/ n /))
       DO i = 1, n ; cl(i:i) = c(i) ; END DO
    ENDIF
    PRINT *, ' ERROR in ', TRIM(cdwhere), ' : libcdftools_hip status ', kstatus
    PRINT *, ' ', TRIM(cl)
    STOP 98
  END SUBROUTINE hip_check

END MODULE cdftools_hip
