/*
 * cdftools_hip.h -- C ABI of the MI355X (gfx950) hot-path library libcdftools_hip.so
 *
 * This is the drop-in boundary between the Fortran host programs (cdfmoc, cdfmhst, cdfvT,
 * cdfheatc, cdfpsi, cdfcurl, cdfsig0, cdfsigi) and the hand-written HIP kernels.  The reference
 * (meom-group/CDFTOOLS) has no FFI: its numerical loops are written inline in each PROGRAM unit.
 * Every entry point below therefore cites the reference loop (file:line under src/) that it
 * replaces; the Fortran ISO_C_BINDING interface blocks for these symbols live in
 * cdftools_amd/fortran/cdftools_hip.f90 and are reproduced in INTEGRATION.md.
 *
 * Conventions
 *   - plain C symbols, plain pointers and sizes; no C++/torch types.
 *   - every function returns an int status: CDFT_OK (0) or a CDFT_ERR_* code; the message is
 *     available from cdft_last_error().  The Fortran wrappers turn a non-zero status into the
 *     reference's fatal-error convention (message + STOP 98, cf. src/cdfio.F90:2924-2940).
 *   - arrays are Fortran column-major, first index fastest: a "slab" is (npiglo,npjglo) r4,
 *     a 3-D field is (npiglo,npjglo,npk), records are appended as the slowest index.
 *   - every array argument carries a memory-space tag: CDFT_MEM_HOST (the library stages it
 *     through HBM itself) or CDFT_MEM_DEVICE (already resident in HBM; nothing is copied).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls whose output
 *     space is CDFT_MEM_DEVICE are asynchronous on that stream; calls that return results to
 *     host memory synchronise the stream before returning.
 *   - a plan owns its device memory; the caller owns every array it passes in.
 *   - one plan must not be used from two host threads at once (the reference is single-threaded,
 *     src/cdfio.F90:72-73 keeps module-level state).
 *   - there is NO CPU fallback inside this library: without a usable HIP device every compute
 *     entry point returns CDFT_ERR_NODEVICE.
 */
#ifndef CDFTOOLS_HIP_H
#define CDFTOOLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDFT_OK            0
#define CDFT_ERR_ARG       1   /* invalid argument (NULL pointer, bad size, bad mask value ...) */
#define CDFT_ERR_HIP       2   /* a HIP runtime call or kernel launch failed                     */
#define CDFT_ERR_ALLOC     3   /* device or host allocation failed                               */
#define CDFT_ERR_STATE     4   /* call sequence error (e.g. compute before set_mesh)             */
#define CDFT_ERR_NODEVICE  5   /* no HIP device available                                        */

#define CDFT_MEM_HOST      0
#define CDFT_MEM_DEVICE    1
#define CDFT_MEM_PINNED    2   /* host memory from cdft_host_alloc: transfers are asynchronous (see "host feed" below)   */

/* ------------------------------------------------------------------------------------------ */
/* runtime                                                                                    */
/* ------------------------------------------------------------------------------------------ */
int         cdft_version(void);                       /* 100*major + minor */
const char *cdft_last_error(void);                    /* message of the last failing call (thread-local) */
int         cdft_device_count(int *count);            /* never fails on a box without GPU: *count = 0 */
int         cdft_set_device(int device);
int         cdft_malloc(void **ptr, size_t bytes);    /* HBM allocation */
int         cdft_free(void *ptr);
int         cdft_mem_info(size_t *free_bytes, size_t *total_bytes);   /* HBM of the current device (hipMemGetInfo) */
/* Host feed without stalls.  The reference reads a slab and consumes it (src/cdfmoc.f90:355-379: getvar, then the loops);
 * a GPU is fed through PCIe, and a program that reads record t+1 while record t travels needs (a) page-locked memory, so
 * that the copy is a DMA the host does not take part in, and (b) calls that return once their work is QUEUED:
 *   - cdft_host_alloc gives such memory (hipHostMalloc); arrays in it may be passed under CDFT_MEM_HOST like any host array;
 *   - an OUTPUT array tagged CDFT_MEM_PINNED (it must come from cdft_host_alloc) is filled asynchronously: the call returns
 *     without synchronising the stream, and the caller reads the result after cdft_stream_sync(stream).  Inputs tagged
 *     CDFT_MEM_PINNED are host arrays whose copy is known to be asynchronous.  Entry points that take the tag say so. */
int         cdft_host_alloc(void **ptr, size_t bytes);
int         cdft_host_free(void *ptr);
/* The feed: a ring of page-locked slots with a stream of its own, for a host that produces its input piece by piece (a
 * file read per slab of levels).  cdft_feed_slot gives the slot to fill (waiting, if need be, for the copy that last used
 * it); cdft_feed_push queues the copy of its first `bytes` to dst + dst_offset in HBM and moves to the next slot, so the
 * next read runs while the DMA does; work given cdft_feed_stream's stream afterwards (a compute call on the record just
 * pushed) runs after the copies.  What this replaces is the time, not an interface: the reference reads a slab, then
 * computes on it (src/cdfmoc.f90:355-379).  Measured on this pool: a pageable first touch or hipHostMalloc cost 0.17-0.19 s
 * per GB and a copy runs at 57 GB/s either way (tools/pin_probe.hip) -- hence a small ring, not page-locked records. */
typedef struct cdft_feed cdft_feed;
int         cdft_feed_create(cdft_feed **feed, size_t slot_bytes, int nslots);
int         cdft_feed_resize(cdft_feed *feed, size_t slot_bytes, int nslots);   /* other slots, the same stream (waits for it) */
int         cdft_feed_slot(cdft_feed *feed, void **host, size_t *slot_bytes);
int         cdft_feed_push(cdft_feed *feed, void *dst, size_t dst_offset, size_t bytes);
/* cdft_feed_push for a destination whose rows are not dense: the slot holds nrows rows of row_bytes bytes back to back (the
 * (npiglo,npjglo) slabs getvar returns, src/cdfio.F90:1425-1609); they go to dst + dst_offset with dst_pitch bytes between row
 * starts -- e.g. straight into the slots of cdfmoc's pool (cdft_moc_stage_dest), so that the H2D copy IS the staging.  One
 * linear DMA into HBM and a row-spreading kernel behind it on the feed's stream; CDFT_FEED_ROWS=2d: one hipMemcpy2DAsync. */
int         cdft_feed_push_rows(cdft_feed *feed, void *dst, size_t dst_offset, size_t dst_pitch, size_t row_bytes, size_t nrows);
/* cdft_feed_push for a field the file holds as BYTES (the masks of mask.nc) and the kernels want as r4: the slot holds n signed bytes;
 * they cross PCIe as bytes and are widened to r4 at dst + dst_offset ELEMENTS by a kernel on the feed's stream -- the conversion
 * nf90_get_var makes on the host when the reference reads a byte mask into its r4 slab (src/cdfio.F90:1593), a quarter of the transfer. */
int         cdft_feed_push_i1_as_r4(cdft_feed *feed, float *dst, size_t dst_offset, size_t n);
int         cdft_feed_stream(cdft_feed *feed, void **stream);
int         cdft_feed_wait(cdft_feed *feed);
int         cdft_feed_destroy(cdft_feed *feed);
int         cdft_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int         cdft_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream); /* synchronises */
int         cdft_stream_sync(void *stream);
/* launch-duration probe: HIP events recorded on the stream the kernels are launched on */
int         cdft_timer_create(void **timer);
int         cdft_timer_start(void *timer, void *stream);
int         cdft_timer_stop(void *timer, void *stream);
int         cdft_timer_elapsed_ms(void *timer, float *ms); /* synchronises on the stop event */
int         cdft_timer_destroy(void *timer);

/* ------------------------------------------------------------------------------------------ */
/* sharded plans : ONE process, several GPUs of the node, row-block (j) partition              */
/*   The reference's only parallel split of these loops is OpenMP over jj                      */
/*   (src/cdfmoc.f90:369-377, :382-388); this is its analogue on a multi-GPU node, behind the  */
/*   same boundary.  A plan made by cdft_<tool>_create_sharded owns one single-device plan per */
/*   row block (block b on devices[b], rows split as evenly as possible, south to north) and   */
/*   is then used through the SAME entry points as a plan of cdft_<tool>_create:               */
/*     - array arguments are the caller's FULL-DOMAIN arrays, host or device memory alike;     */
/*       each block's rows are moved with strided copies (blocks in parallel, one host thread  */
/*       and one stream per block), results land in place in the caller's arrays;              */
/*     - the calls return when every block is done (the `stream` argument is not used);        */
/*     - cdfmoc, cdfmhst, cdfvT: rows are independent, results are bit-identical to one device;*/
/*       cdfpsi: the running difference along j (src/cdfpsi.f90:395-398) is chained through   */
/*       the blocks in order, 8*npiglo bytes from HBM to HBM per hop, bit-identical as well;   */
/*       cdfheatc: the blocks' (npk,3) r8 sums are added in block order (1e-15 relative to one */
/*       device: the r8 additions are regrouped, as between any two launch shapes).            */
/*   cdft_shard_from_env reads what the drop-in programs honour:                               */
/*     CDFT_GPU_MAP="a,b,.."  block b on HIP device map[b] (a device may appear twice: that is */
/*                            how the path is tested on a 1-GPU box), else                     */
/*     CDFT_GPUS=n            devices 0..n-1, else one block on the current device             */
/*   (*ndev = 1, devices[0] = -1).                                                             */
/* ------------------------------------------------------------------------------------------ */
int cdft_shard_from_env(int *ndev, int *devices, int capacity);
/* How the blocks of the sharded plans made FROM NOW ON exchange what crosses them (cdfheatc's level sums, cdfpsi's carry and
 * reference value -- src/cdfheatc.f90:225-229, src/cdfpsi.f90:395-398, :423):
 *   CDFT_COLLECTIVE_PEER (default)  device-to-device copies and a host sum, as described above;
 *   CDFT_COLLECTIVE_RCCL            RCCL over xGMI: one ncclAllGather of the blocks' (npk,3) r8 sums, added in block order
 *                                   (the same additions: bit-identical to the default), ncclSend / ncclRecv for the carry
 *                                   hop, ncclBroadcast of the reference value.  One communicator per block from
 *                                   ncclCommInitAll; librccl is opened at run time, only in this mode; two blocks on one
 *                                   device are refused (RCCL takes one rank per device).
 * The environment variable CDFT_SHARD_COLLECTIVE=rccl|peer sets the initial mode (what the drop-in programs honour). */
#define CDFT_COLLECTIVE_PEER 0
#define CDFT_COLLECTIVE_RCCL 1
int cdft_shard_set_collective(int mode);
int cdft_shard_get_collective(int *mode);

/* ------------------------------------------------------------------------------------------ */
/* cdfmoc : meridional overturning stream function                                            */
/*   replaces src/cdfmoc.f90:355-379 (zonal masked weighted sum, hot loop 1)                   */
/*        and src/cdfmoc.f90:382-388 (bottom-to-surface cumulated sum, hot loop 2)             */
/* ------------------------------------------------------------------------------------------ */
typedef struct cdft_moc_plan cdft_moc_plan;

/* npiglo,npjglo: horizontal size of the (local row block of the) domain; npk: number of levels;
 * nbasins: 1..8 (the reference uses 1 or 5, src/cdfmoc.f90:270-274). */
int cdft_moc_create(cdft_moc_plan **plan, int npiglo, int npjglo, int npk, int nbasins);

/* The same plan over ndev row blocks, block b on HIP device devices[b] (see "sharded plans" above).  -decomp
 * (cdft_moc_decomp_*) is not offered on it: its U-point stencil and depth means cross rows. */
int cdft_moc_create_sharded(cdft_moc_plan **plan, int npiglo, int npjglo, int npk, int nbasins, int ndev,
                            const int *devices);

/* Time-invariant inputs (once per run; once per record with -vvl, src/cdfmoc.f90:343-348):
 *   e1v    (npiglo,npjglo)         r4  src/cdfmoc.f90:306
 *   e3v    (npiglo,npjglo,npk)     r4  already multiplied by vmask(level) as get_e3v does,
 *                                      src/cdfmoc.f90:576-596
 *   ibmask (nbasins,npiglo,npjglo) i2  values 0/1, basin index fastest, src/cdfmoc.f90:281,325-336
 * The plan keeps one r4 weight e1v*e3v (rounded to r4 exactly as the reference expression's
 * first product) and one byte of basin bits per (i,j); the inputs may be released afterwards. */
int cdft_moc_set_mesh(cdft_moc_plan *plan, const float *e1v, const float *e3v,
                      const int16_t *ibmask, int memspace, void *stream);
/* The same with get_e3v's mask product (src/cdfmoc.f90:581-593: e3v = e3v * ivmask, ivmask the level's vmask as
 * INTEGER*2) done on the device: e3v is the raw metric as read from the file and vmask (npiglo,npjglo,npk) its 0/1
 * mask as bytes.  Saves the host a conversion of the mask to r4 and a pass over the metric (two passes over a
 * 4.68 GB field at eORCA12).  vmask = NULL is cdft_moc_set_mesh. */
int cdft_moc_set_mesh_vmask(cdft_moc_plan *plan, const float *e1v, const float *e3v, const int8_t *vmask,
                            const int16_t *ibmask, int memspace, void *stream);

/* zv   : npt records of (npiglo,npjglo,npk) r4 meridional velocity, record r starting at
 *        zv + r*rec_stride (rec_stride in elements, >= npiglo*npjglo*npk; level npk is never
 *        read, as in the reference loop jk=1,npk-1).
 * dmoc : (nbasins,npjglo,npk,npt) r8, in Sverdrup, exactly the reference's dmoc(:,:,:) after
 *        its loop 2 for each record; level npk is 0. */
int cdft_moc_compute(cdft_moc_plan *plan, const float *zv, int64_t rec_stride, int npt,
                     int zv_space, double *dmoc, int dmoc_space, void *stream);
/* (dmoc_space = CDFT_MEM_PINNED: the result is on its way when the call returns; cdft_stream_sync(stream) before reading.) */

/* The records handed over ONCE and used by any number of cdft_moc_compute_staged calls: the equivalent of the program
 * having read its velocity records (src/cdfmoc.f90:355-358) before the loops run.  The plan copies them (host or device
 * arrays, dense like cdft_moc_compute's) into its own layout in HBM, the POOL: row (jj,jk) of the domain owns one contiguous
 * run of 1+npt slots of P r4 (P = npiglo rounded up to a multiple of 4) -- the row's weights e1v*e3v (src/cdfmoc.f90:374),
 * then the row of record 1, of record 2, ... -- which is what the workgroup of that row reads, front to back, instead of
 * 1+npt runs a whole field (4.7 GB at eORCA12) apart.  The caller's array is free again when the stream has passed the call
 * (at once for pageable host memory).  A sharded plan moves every block's rows into that block's pool, so that the compute
 * calls that follow involve no transfer of inputs at all -- what a resident multi-GPU run is.  A new mesh
 * (cdft_moc_set_mesh) keeps the staged records and brings its weights into the pool at the next cdft_moc_compute_staged.
 * cdft_moc_stage_rows: the same for records whose LEVELS are lev_stride elements apart (>= npiglo*npjglo; rows dense): a
 * row block of a larger domain, taken where it lies (single-device plans).
 * cdft_moc_stage_open / _dest / _close: the caller writes the records into the pool ITSELF -- a program that reads its file slab
 * by slab (src/cdfmoc.f90:357, one getvar per level) and whose H2D copies then ARE the staging (cdft_feed_push_rows): _open
 * lays the pool out for npt records and brings the weights in (on `stream`), _dest gives the address of row jj = 1 of level
 * `level` (0-based) of record `record` (0-based) and the distance in bytes between consecutive rows (rows of a level, then
 * of the next level, follow at that pitch), _close declares the npt records staged once every copy has been QUEUED on the
 * stream the compute call is given.  Single-device plans.
 * Lifetime: staged records stay until the next cdft_moc_stage* call, cdft_moc_release_staged (which gives the pool's memory
 * back: 61 GB at eORCA12 for 12 records) or cdft_moc_destroy; cdft_moc_compute on other records does not touch them, on
 * single-device and sharded plans alike.
 * CDFT_MOC_POOL=0 (A/B hook of cdft_moc_stage): keep the records as they came instead (a device array is then borrowed, not
 * copied, and a later cdft_moc_compute from host memory forgets them).
 * cdft_moc_reserve: make the pool for npt records now (optional: cdft_moc_stage makes it when it is not there).  A program
 * that knows its record count asks for the pool before it allocates anything else in HBM: the one large allocation then gets
 * the device's memory while it is still in one piece. */
int cdft_moc_reserve(cdft_moc_plan *plan, int npt, void *stream);
int cdft_moc_stage(cdft_moc_plan *plan, const float *zv, int64_t rec_stride, int npt, int zv_space, void *stream);
int cdft_moc_stage_rows(cdft_moc_plan *plan, const float *zv, int64_t rec_stride, int64_t lev_stride, int npt, int zv_space,
                        void *stream);
int cdft_moc_stage_open(cdft_moc_plan *plan, int npt, void *stream);
int cdft_moc_stage_dest(cdft_moc_plan *plan, int record, int level, void **dst, size_t *row_pitch_bytes);
int cdft_moc_stage_close(cdft_moc_plan *plan);
int cdft_moc_release_staged(cdft_moc_plan *plan);
int cdft_moc_compute_staged(cdft_moc_plan *plan, double *dmoc, int dmoc_space, void *stream);
/* the same result from the same resident pool with one launch per window of nlev levels instead of one for all: the launches
 * the stream route (below) makes, on resident data -- what bench.py reports as roofline.level_windows */
int cdft_moc_compute_staged_windows(cdft_moc_plan *plan, int nlev, double *dmoc, int dmoc_space, void *stream);
/* The stream route: cdfmoc with NOTHING 3-D resident -- what the cdfmoc program does when its fields are large.  The reference
 * reads one (npiglo,npjglo) slab per level and record and works on it (src/cdfmoc.f90:355-379); here the domain goes through
 * HBM a WINDOW of nlev levels at a time, every record of the run beside the window's metric, laid out like the pool: row
 * (jj,jk) of the window owns one run of 1+npt slots -- slot 0 the raw e3v of the row (src/cdfmoc.f90:343-348), slots 1..npt
 * the row of records 1..npt -- so that one launch per window streams all the records past the weights and every byte that
 * came from the files is read from HBM once: the same 4*(1+1/npt) bytes per cell and record as the resident pool, with
 * (1+npt) * nlev slabs of HBM instead of (1+npt) * npk (1.6 GB instead of 61 GB at eORCA12 with 12 records and nlev = 2:
 * hipMalloc costs about as much per byte as reading the file does).  The caller reads its files level-window by
 * level-window (records inner) and writes each slab where _dest says (cdft_feed_push_rows).
 *   cdft_moc_set_basins   the 2-D part of cdft_moc_set_mesh: e1v (npiglo,npjglo) r4 and ibmask (nbasins,npiglo,npjglo) i2
 *   cdft_moc_stream_begin the window for npt records and nlev levels (allocated once, zeroed), the sums of the run zeroed
 *   cdft_moc_stream_dest  slot 0..npt, level 0..nlev-1 of the window -> address of its row jj = 1 and the pitch between rows
 *   cdft_moc_stream_vmask_dest  where the vmask bytes (npiglo,npjglo) of level `level` of the window go (dense)
 *   cdft_moc_stream_compute     the window holds levels k0 .. k0+nlev-1 (0-based, < npk-1): slot 0 becomes the weight
 *                         e1v*(e3v*vmask) in place (with_vmask = 0: e3v arrives masked), then the zonal kernel of the window;
 *                         queued on `stream` -- the stream the copies were queued on -- so the next window's copies follow it
 *   cdft_moc_stream_end   the cumulated sums (src/cdfmoc.f90:382-388) and dmoc (nbasins,npjglo,npk,npt) as cdft_moc_compute's */
int cdft_moc_set_basins(cdft_moc_plan *plan, const float *e1v, const int16_t *ibmask, int memspace, void *stream);
int cdft_moc_stream_begin(cdft_moc_plan *plan, int npt, int nlev, void *stream);
int cdft_moc_stream_dest(cdft_moc_plan *plan, int slot, int level, void **dst, size_t *row_pitch_bytes);
int cdft_moc_stream_vmask_dest(cdft_moc_plan *plan, int level, void **dst);
int cdft_moc_stream_compute(cdft_moc_plan *plan, int k0, int nlev, int with_vmask, void *stream);
int cdft_moc_stream_end(cdft_moc_plan *plan, double *dmoc, int dmoc_space, void *stream);
/* zonal-kernel launches this plan has made and the records they covered (12 records in 1 launch: every input byte read once;
 * 12 launches: the weights read 12 times) -- what a program prints under CDFT_TIMING and what the tests assert */
int cdft_moc_stats(cdft_moc_plan *plan, int64_t *zonal_launches, int64_t *records);

/* Records the CALLER keeps in a pitched layout in HBM.  The reference's zv(npiglo,npjglo) slab per (level, record) is
 * dense (src/cdfmoc.f90:357 getvar); in HBM the rows of a field need not be: element (ji,jj,jk) of record t lies at
 *     zv[t * rec_stride + ((jk-1) * npjglo + (jj-1)) * row_stride + (ji-1)]          (elements, r4)
 * with row_stride >= npiglo.  Either the records are the outer dimension (rec_stride >= the span of a record) or the rows
 * are (row_stride >= the span of the npt records of a row).  The weights stay the plan's dense array: 16-byte vector loads
 * need every record base 16-byte aligned and row_stride = npiglo modulo 4; anything else is read through the (slower)
 * unaligned path, with the same results up to the order of the r8 additions.  Device memory, single-device plans. */
int cdft_moc_compute_pitched(cdft_moc_plan *plan, const float *zv, int64_t rec_stride, int64_t row_stride, int npt,
                             double *dmoc, int dmoc_space, void *stream);

/* cdfmoc -decomp (src/cdfmoc.f90:353, :360-365, :390-517): the total overturning of ONE record and its
 * geostrophic-shear, barotropic and ageostrophic parts.  Needs cdft_moc_set_mesh (weights, basins)
 * and, once per run (per record with -vvl):
 *   e3v   (npiglo,npjglo,npk) r4, the same masked e3v handed to cdft_moc_set_mesh (:343-348)
 *   e1u   (npiglo,npjglo) r4 (:313)
 *   zcoef (npiglo,npjglo) r4: the reference's array after :428-433, -grav/rau0/f at V points and 0
 *         where f = 0.  Computed by the caller: it goes through the compiler's r4 SIN.
 *   gdept (npk) r4 on the HOST (:312)
 *   tmask, umask (npiglo,npjglo,npk) bytes 0/1 (the reference reads them into INTEGER(2), :439-440) */
int cdft_moc_decomp_set_mesh(cdft_moc_plan *plan, const float *e3v, const float *e1u, const float *zcoef,
                             const float *gdept, const uint8_t *tmask, const uint8_t *umask, int memspace,
                             void *stream);
/* get_e3v's product (src/cdfmoc.f90:576-596) in place on DEVICE arrays: e3v(i) = e3v(i) * vmask(i), n cells, the mask as bytes
 * (the reference multiplies by its INTEGER(2) mask: the same r4 product).  For a caller whose raw e3v and vmask went from their
 * files to HBM through the feed and who needs the masked metric for cdft_moc_set_mesh + cdft_moc_decomp_set_mesh. */
int cdft_moc_mask_e3v(float *e3v, const int8_t *vmask, int64_t n, void *stream);
/* zv, ztemp, zsal (npiglo,npjglo,npk) r4 of one record; eos = CDFT_EOS_80 | CDFT_EOS_TEOS10 (-teos10).
 * dmoc, dmoc_bt, dmoc_ag (nbasins,npjglo,npk) r8 out.
 * dmoc_sh (nbasins,npjglo,npk) r8 IN/OUT: the reference never clears it and accumulates each record
 * onto what the array holds (:471) -- pass zeros for the first record and the previous record's
 * result afterwards to reproduce a multi-record run of the reference, or zeros every time for the
 * per-record decomposition.  Another quirk kept: the border of the geostrophic velocity holds the
 * velocity of level npk-1 (zv(:,:) is only assigned on ji=2..npiglo-1, jj=2..npjglo-1, :447-462). */
int cdft_moc_decomp(cdft_moc_plan *plan, const float *zv, const float *ztemp, const float *zsal, int eos,
                    int in_space, double *dmoc, double *dmoc_sh, double *dmoc_bt, double *dmoc_ag,
                    int out_space, void *stream);

/* Tuning hook for the measurement harness: waves per row (0 = automatic) and float4 vectors
 * per lane (0 = automatic), accumulate mode (0 = automatic, 1 = select per basin, 2 = fma with
 * 0/1 masks, 3 = one LDS accumulator per basin-bit pattern ("class"; needs several basins, at most
 * 5 (nbasins <= 5) or 8 distinct non-empty patterns, 16-byte aligned records; automatic choice
 * when those hold).  All modes form identical r8 terms; they differ in the (fixed) order of the
 * r8 additions along i, i.e. within the stated 1e-13 * sum|terms| tolerance.
 * In class mode waves_per_row x vec_per_lane need not cover a row: they make a SEGMENT of it (one workgroup), the row is
 * cut into as many as it takes and the segments' sums are added in order (the automatic choice for rows longer than
 * 9.2k cells: 2 waves x 8..10 vectors).  The other modes refuse a shape that does not cover the row. */
int cdft_moc_set_tuning(cdft_moc_plan *plan, int waves_per_row, int vec_per_lane, int mode);
/* name of the zonal kernel variant the next compute will launch (for profiles) */
int cdft_moc_describe(cdft_moc_plan *plan, char *buf, size_t buflen);
/* Launch-duration probe for the dominant (zonal) kernel: while enabled, every cdft_moc_compute
 * brackets its zonal-kernel launch with HIP events on the launch stream.  cdft_moc_profile_read
 * synchronises on them, returns the summed duration and the number of launches, and resets. */
int cdft_moc_profile_enable(cdft_moc_plan *plan, int on);
int cdft_moc_profile_read(cdft_moc_plan *plan, double *zonal_ms_total, int *launches);
/* The same probe, one duration per launch (ms_each[0..min(launches,capacity)-1]) so that the harness can report the
 * minimum, median and maximum beside the mean; synchronises and resets like cdft_moc_profile_read. */
int cdft_moc_profile_read_each(cdft_moc_plan *plan, float *ms_each, int capacity, int *launches);
/* On a sharded plan the two readers above give, launch by launch, the duration of the SLOWEST block (what the call waited
 * for); this one gives the launches of one block (and resets that block only), cdft_moc_block_info its device and its
 * rows [j0,j1) of the domain (0-based, j1 exclusive).  Block 0 of an ordinary plan is the plan itself. */
int cdft_moc_profile_read_block(cdft_moc_plan *plan, int block, float *ms_each, int capacity, int *launches);
int cdft_moc_block_info(cdft_moc_plan *plan, int block, int *device, int *j0, int *j1);

int cdft_moc_destroy(cdft_moc_plan *plan);

/* ------------------------------------------------------------------------------------------ */
/* eos : potential density sigma0 / sigmai                                                    */
/*   replaces src/eos.f90:601-632 (sigma0), :802-887 (sigmai_dep), :1002-1087 (sigmai_dep2d),  */
/*   :889-1000 and :1089-1201 (Jackett-McDougall 1994 variants); the coefficient set is what   */
/*   eos_init (:207-598) selects.  Pointwise: n = number of points (a slab or a whole field).  */
/* ------------------------------------------------------------------------------------------ */
#define CDFT_EOS_80        0   /* NEMO polynomial EOS-80  (eos_init(.FALSE.), the tools' default) */
#define CDFT_EOS_TEOS10    1   /* NEMO polynomial TEOS-10 (eos_init(.TRUE.), -teos10)             */
#define CDFT_EOS_JMD94     2   /* sigmai(..., ld_eos80=.TRUE.) legacy rational form               */
#define CDFT_EOS_NTR       3   /* sigmantr: neutral density of McDougall & Jackett 2005 (src/eos.f90:634-687);
                                * pref is ignored and the result is NOT multiplied by (psal /= 0)        */
#define CDFT_EOS_FAST_FMA  1   /* flags: allow FMA contraction (not bit-identical: few ulp of r8)  */

/* sig(n) r8 = sigmai(ptem, psal, pref): density anomaly referenced to depth pref (sigma0: pref = 0),
 * already multiplied by (psal /= 0) as the reference does (src/eos.f90:852-856,:880). */
int cdft_eos_sigmai(const float *ptem, const float *psal, int64_t n, float pref, int eos, int flags,
                    double *sig, int in_space, int out_space, void *stream);
/* per-point reference depth pref2d(n) (sigmai_dep2d) */
int cdft_eos_sigmai2d(const float *ptem, const float *psal, const float *pref2d, int64_t n, int eos,
                      int flags, double *sig, int in_space, int out_space, void *stream);
/* The level loop of cdfsig0 / cdfsigi fused (src/cdfsig0.f90:146-152, src/cdfsigi.f90:131-137):
 * sig4(n) r4 = REAL( sigmai(ptem,psal,pref) * zmask ), zmask = 0 where psal == spval else 1.
 * The three entries take out_space = CDFT_MEM_PINNED (a result array from cdft_host_alloc): the result is on its way when the
 * call returns, cdft_stream_sync(stream) before reading it -- the programs compute slab s+1 while they write slab s. */
int cdft_eos_sig_tool(const float *ptem, const float *psal, int64_t n, float pref, float spval, int eos,
                      int flags, float *sig4, int in_space, int out_space, void *stream);
int cdft_eos_release(void);   /* frees the staging buffers used for host arrays */

/* ------------------------------------------------------------------------------------------ */
/* eos, the other members of the module's public list (src/eos.f90:24-32): alpha, beta, albet, */
/*   spice, eosbn2.  Pointwise like sigmai; the same evaluation order and r4/r8 mixing as the  */
/*   reference, bit-identical results.  (sigmantr is cdft_eos_sigmai with CDFT_EOS_NTR.)       */
/* ------------------------------------------------------------------------------------------ */
#define CDFT_EOSQ_ALPHA        0   /* alpha_2d / alpha_z  (src/eos.f90:1452-1503, :1545-1594), eos = 80 | TEOS10        */
#define CDFT_EOSQ_BETA         1   /* beta_2d / beta_z    (:1400-1450, :1643-1691),            eos = 80 | TEOS10        */
#define CDFT_EOSQ_ALBET_MCD87  2   /* albet_2d / albet_z  (:1315-1398): alpha/beta of McDougall (1987); eos ignored     */
#define CDFT_EOSQ_BETA_MCD87   3   /* beta_2d_80 / beta_z_80 (:1693-1775): beta(..., ld_eos80=.TRUE.); eos ignored      */
#define CDFT_EOSQ_SPICE_POL    4   /* spice_pol (:755-800): spiciness-0 fit of the GSW toolbox; depth and eos ignored   */
#define CDFT_EOSQ_SPICE_F02    5   /* spice_80  (:689-753): Flament (2002); depth and eos ignored                       */
#define CDFT_EOS_MCD87         4   /* cdft_eos_bn2 only: eosbn2(..., ld_eos80=.TRUE.), McDougall (1987) alpha/beta, beta */

/* out(n) r8 = the quantity at every point.  pdep_n == NULL: one depth pdep for all points (the 2-D members of the
 * generics); else pdep_n(n) holds a depth per point (the profile members, which take pdep(kpk)) and pdep is ignored.
 * alpha and beta are multiplied by (psal /= 0) as the reference does (:1419-1423); the other four are not masked. */
int cdft_eos_field(int quantity, const float *ptem, const float *psal, const float *pdep_n, float pdep, int64_t n, int eos,
                   double *out, int in_space, int out_space, void *stream);
/* bn2(n) r4 = eosbn2(ptem, psal, pdep, pe3w, kpi, kpj, kup, kdown): N^2 at the w level between the T levels "up" and
 * "down" (the reference's ptem(:,:,kup), ptem(:,:,kdown) planes are passed as separate arrays).  eos = CDFT_EOS_80 or
 * CDFT_EOS_TEOS10: eosbn2_pol (:1203-1244) with that coefficient set -- including its habit of handing alpha and beta
 * the salinity anomaly s-35 (:1231-1235); CDFT_EOS_MCD87: eosbn2_80 (:1246-1312).  Not masked. */
int cdft_eos_bn2(const float *ptem_up, const float *ptem_down, const float *psal_up, const float *psal_down, const float *pe3w,
                 float pdep, int64_t n, int eos, float *bn2, int in_space, int out_space, void *stream);
/* alpha_0d / beta_0d (:1505-1543, :1596-1641): one point, r8 arguments, r8 result in host memory (synchronous) */
int cdft_eos_point(int quantity, double ptem, double psal, double pdep, int eos, double *out);
int cdft_eos_more_release(void);   /* frees the staging buffers of the two entries above */

/* ------------------------------------------------------------------------------------------ */
/* cdfheatc : heat content (masked volume integrals per level)                                */
/*   replaces src/cdfheatc.f90:205-240 (the three SUMs of each level)                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct cdft_heatc_plan cdft_heatc_plan;

/* sizes of the (zoomed) working window, as the reference's npiglo/npjglo/nvpk after -zoom */
int cdft_heatc_create(cdft_heatc_plan **plan, int npiglo, int npjglo, int npk);
int cdft_heatc_create_sharded(cdft_heatc_plan **plan, int npiglo, int npjglo, int npk, int ndev, const int *devices);
/* e1t,e2t (npiglo,npjglo) r4 src/cdfheatc.f90:188-189; exactly one of
 *   e3t  (npiglo,npjglo,npk) r4  (:216, again per record with -vvl)  or
 *   e31d (npk) r4                (-full, :213)
 * tmask (npiglo,npjglo,npk): bytes with values 0/1 (mask_is_r4 = 0) or r4 values (mask_is_r4 = 1,
 * for -M masks that are not 0/1), :209.  The plan keeps its own copies. */
int cdft_heatc_set_mesh(cdft_heatc_plan *plan, const float *e1t, const float *e2t, const float *e3t,
                        const float *e31d, const void *tmask, int mask_is_r4, int memspace, void *stream);
/* temp (npiglo,npjglo,npk) r4, one record (:208).  mxloption 0 | 1 | -1 with rmxldep (npiglo,npjglo)
 * r4 in the same space as temp and gdepw (npk) r4 on the HOST (:218-223).
 * sums (3,npk) r8 = dsurf, dvol2d, dsum2d of every level (:225-229); the caller forms
 * heatc2d = pprho0*ppcp*dsum2d and the running dvol / dsum exactly as :227,:230,:236-247. */
int cdft_heatc_compute(cdft_heatc_plan *plan, const float *temp, int temp_space, const float *rmxldep,
                       const float *gdepw, int mxloption, double *sums, int out_space, void *stream);
/* How the three SUMs of a level are accumulated:
 *   CDFT_HEATC_SUM_R8     (default) r4 products as written, r8 sums in a fixed parallel order: the accurate value,
 *                         at the HBM roofline; 1e-13 from a sequential r8 sum.
 *   CDFT_HEATC_SUM_R4SEQ  the reference's own arithmetic: its SUM is REAL(KIND=4) (:225-229), compiled by gfortran -O /
 *                         flang -O1 into one sequential r4 loop in memory order.  Reproduced bit for bit, so that the
 *                         drop-in program returns what a reference build returns (9e-5 from the r8 value at eORCA025,
 *                         per cents at eORCA12): inside a binade of the running sum the chain is an integer sum, formed
 *                         in parallel; crossings of a power of two and ties are resolved cell by cell (heatc_scan_kernel:
 *                         one pass over the fields after a 1/64 sample that names the binades, 1.4-3 times the r8 mode's
 *                         time; CDFT_HEATC_SEQ=chain in the environment selects the plain chain of v_add_f32 it is checked
 *                         against, 25-40 times slower; a wait of the scan that does not end within 2 s makes the call fall
 *                         back to that chain, with a line on stderr).  The r4 results are widened exactly into `sums`.
 *                         Sharded plans: a sequential sum has no parallel form across row blocks -- block b starts where
 *                         block b-1 ended -- so the blocks' rows are staged side by side but their chains run ONE BLOCK
 *                         AFTER THE OTHER, south to north: N devices take as long as one in this mode (and hold 1/N of
 *                         the fields each); the r8 mode is the one that scales. */
#define CDFT_HEATC_SUM_R8     0
#define CDFT_HEATC_SUM_R4SEQ  1
int cdft_heatc_set_sum_mode(cdft_heatc_plan *plan, int mode);
/* CDFT_HEATC_SUM_R4SEQ over row blocks held by separate plans (what a sharded plan does internally, and what ranks
 * of cdftools_amd/shard.py do across processes): last_out (3*npk r4, may be NULL) receives the r4 sums the last
 * compute ended with; carry_in (3*npk r4, may be NULL) makes the NEXT compute start its chains there instead of 0. */
int cdft_heatc_chain(cdft_heatc_plan *plan, const float *carry_in, float *last_out);
int cdft_heatc_destroy(cdft_heatc_plan *plan);

/* ------------------------------------------------------------------------------------------ */
/* cdfmhst : meridional heat / salt transport                                                 */
/*   replaces src/cdfmhst.f90:308-367 (vertical integration of vt, vs and the zonal SUMs)      */
/* ------------------------------------------------------------------------------------------ */
typedef struct cdft_mhst_plan cdft_mhst_plan;

int cdft_mhst_create(cdft_mhst_plan **plan, int npiglo, int npjglo, int npk, int nbasins);
/* sharded: every block but the last also reads the row north of it (the V,T,S form looks at T, S of row j+1, :318) */
int cdft_mhst_create_sharded(cdft_mhst_plan **plan, int npiglo, int npjglo, int npk, int nbasins, int ndev, const int *devices);
/* e1v (npiglo,npjglo) r4 :276; exactly one of e3v (npiglo,npjglo,npk) r4 (:330, again per record
 * with -vvl) or e31d (npk) r4 (-full, :329); masks (npiglo,npjglo,nbasins) r4: slab 0 is the global
 * mask (vmask level 1, summed over ji = 2..npiglo-1 only, :340-344), the others are the basin masks
 * summed over the whole row (:349-367).  The plan keeps copies. */
int cdft_mhst_set_mesh(cdft_mhst_plan *plan, const float *e1v, const float *e3v, const float *e31d,
                       const float *masks, int memspace, void *stream);
/* One record.  Either vt, vs (npiglo,npjglo,npk) r4 from a VT file (:325-326), or vt = vs = NULL and
 * zv, zt, zs given: the kernel forms 0.5*(t(j)+t(j+1))*v itself (:312-322).
 * zdim = 0: heat, salt (nbasins,npjglo) r8 = the zonal sums after the last level;
 * zdim = 1: (nbasins,npjglo,npk) running totals after every level (-Zdim, :389-392).
 * Raw sums: the caller divides by 1.d15 / 1.d6, forms inp / inp0 and substitutes 9999.99 (:395-443). */
int cdft_mhst_compute(cdft_mhst_plan *plan, const float *vt, const float *vs, const float *zv,
                      const float *zt, const float *zs, int in_space, int zdim, double *heat, double *salt,
                      int out_space, void *stream);
int cdft_mhst_destroy(cdft_mhst_plan *plan);

/* ------------------------------------------------------------------------------------------ */
/* cdfvtrp, cdfvhst : vertically integrated transports per grid cell     */
/*   the same vertical integral as cdfpsi / cdfmhst, handed back as a 2-D r8 field               */
/* ------------------------------------------------------------------------------------------ */
/* src/cdfvtrp.f90:211-218: dtrp(ji,jj) = sum_k vel*m2d*e3*1.d0 (r4 product left to right, r8 sum in
 * level order).  vel (npiglo,npjglo,npk) r4, m2d (npiglo,npjglo) r4 (e2u or e1v), e3 (npiglo,npjglo,npk)
 * r4 or NULL with e31d (npk) r4 (-full).  dtrp (npiglo,npjglo) r8: the caller stores REAL(dtrp). */
int cdft_vint_transport(const float *vel, const float *m2d, const float *e3, const float *e31d, int npiglo,
                        int npjglo, int npk, int in_space, double *dtrp, int out_space, void *stream);
/* cdfvtrp -bathy, src/cdfvtrp.f90:210-252: along- and cross-slope components of the two vertically integrated
 * transports.  dtrpu, dtrpv (npiglo,npjglo) r8 as cdft_vint_transport returns them; hdepw (sea-floor depth), e1u, e2v,
 * tmask (level 1) (npiglo,npjglo) r4.  dwku = 0.5*(dtrpu(i)+dtrpu(i-1)) with dwku(1,:) = dwku(npiglo-1,:), dwkv likewise
 * in j with row 1 = 0; slopes zdhdx, zdhdy by centred r4 differences (0 on the border), zalpha = ATAN2(zdhdx,zdhdy)*tmask;
 *   dalong = ( dwku*COS(zalpha) + dwkv*SIN(zalpha)) * tmask      dcross = (-dwku*SIN(zalpha) + dwkv*COS(zalpha)) * tmask
 * both (npiglo,npjglo) r8; the caller stores REAL(.).  Pointwise r8 arithmetic on r4 ATAN2 / COS / SIN values: equal to the
 * reference up to the last bit of those three library results (|d| <= 3 ulp(r4) * (|dwku| + |dwkv|)). */
int cdft_vint_slope_rotate(const double *dtrpu, const double *dtrpv, const float *hdepw, const float *e1u,
                           const float *e2v, const float *tmask, int npiglo, int npjglo, int in_space, double *dalong,
                           double *dcross, int out_space, void *stream);
/* src/cdfvhst.f90:167-178 for one pair of VT-file variables (vomevt, vomevs with e1v, e3v; vozout,
 * vozous with e2u, e3u): e3m = e3*m2d in r4 first, then
 *   dheat += xt*e3m*pp_rau0*pp_rcp*1.d0 (1000., 4000., r4 chain), dsalt += xs*e3m*1.d0. */
int cdft_vhst_transport(const float *xt, const float *xs, const float *m2d, const float *e3, const float *e31d,
                        int npiglo, int npjglo, int npk, int in_space, double *dheat, double *dsalt, int out_space,
                        void *stream);
int cdft_vint_release(void);   /* frees the staging buffers of the two calls above */

/* ------------------------------------------------------------------------------------------ */
/* cdfmean, cdfsum : weighted mean / variance / sum per level and sub-basin */
/*   replaces src/cdfmean.f90:405-431 (the four SUMs of each level and basin)                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct cdft_mean_plan cdft_mean_plan;
int cdft_mean_create(cdft_mean_plan **plan, int npiglo, int npjglo, int npk, int nbasins /* 1..8 */);
/* e1, e2 (npiglo,npjglo) r4 (:363-364); exactly one of e3 (npiglo,npjglo,npk) r4 (:414, again per record
 * with -vvl) or e31d (npk) r4 (-full, :412); zmask (npiglo,npjglo,npk) r4 as getvar returns it (:409; any
 * values, e.g. a -M mask); ibmask (nbasins,npiglo,npjglo) i2, basin index fastest (:385-391; all 1
 * without -B).  (npiglo,npjglo,npk) is the -w window the caller read. */
int cdft_mean_set_mesh(cdft_mean_plan *plan, const float *e1, const float *e2, const float *e3, const float *e31d,
                       const float *zmask, const int16_t *ibmask, int memspace, void *stream);
/* zv (npiglo,npjglo,npk) r4, one record (:408; NOT yet multiplied by zmask: the kernel does :410).
 * sums (4,nbasins,npk) r8: dsurf, dvol2d, dsum2d, dvar2d of every level and basin (:419-424); the
 * caller forms the means, variances and 3-D totals (:425-499). */
int cdft_mean_compute(cdft_mean_plan *plan, const float *zv, int in_space, double *sums, int out_space, void *stream);
/* cdfsum (src/cdfsum.f90:303-327) forms the same dsurf and dvol2d but a different third sum, and no fourth:
 *   CDFT_MEAN_FORM_MEAN (default)  dsum2d = SUM(DBLE(zv*e1*e2*e3)),  dvar2d = SUM(DBLE(zv*zv*e1*e2*e3))   (cdfmean)
 *   CDFT_MEAN_FORM_SUM             dsum2d = SUM(DBLE(zv*e1*e2)),     dvar2d = 0          (cdfsum :309, model field)
 *   CDFT_MEAN_FORM_SUM_FORCING     dsum2d = SUM(DBLE(zv*e1*e2*zmask)), dvar2d = 0        (cdfsum :320, file without depth)
 * with zv = zv*zmask first in every form (:299).  Applies to the following cdft_mean_compute calls. */
#define CDFT_MEAN_FORM_MEAN        0
#define CDFT_MEAN_FORM_SUM         1
#define CDFT_MEAN_FORM_SUM_FORCING 2
#define CDFT_MEAN_FORM_MEAN_NOVAR  3   /* cdfmean without -var: the three sums of FORM_MEAN, dvar2d = 0 (its products are not formed) */
int cdft_mean_set_form(cdft_mean_plan *plan, int form);
int cdft_mean_destroy(cdft_mean_plan *plan);

/* ------------------------------------------------------------------------------------------ */
/* cdfzonalsum, cdfzonalmean : zonal integral / mean of any field        */
/*   replaces src/cdfzonalsum.f90:296-313 and src/cdfzonalmean.f90:313-330                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct cdft_zonal_plan cdft_zonal_plan;
int cdft_zonal_create(cdft_zonal_plan **plan, int npiglo, int npjglo, int npk, int nbasins /* 1..8 */);
/* e1, e2 (npiglo,npjglo) r4 of the variable's grid point (dl_surf = 1.d0*e1*e2); zmaskvar (npiglo,npjglo,npk)
 * r4, the mask of that grid point at every level; zmask (nbasins,npiglo,npjglo) r4, basin index fastest,
 * the reference's zmask array (global, then the sub-basins). */
int cdft_zonal_set_mesh(cdft_zonal_plan *plan, const float *e1, const float *e2, const float *zmaskvar,
                        const float *zmask, int memspace, void *stream);
/* zv (npiglo,npjglo,nlev) r4: the nlev levels (1 for a 2-D variable) of one variable and record.
 * mean_form = 0 (cdfzonalsum): dzo (npjglo,nlev,nbasins) r8 = the integral along i, BEFORE the optional
 *   division by alpha(jj) (-pdeg), which the caller applies.
 * mean_form = 1 (cdfzonalmean): dzo = sum of dl_surf*dtmp, darea = sum of dl_surf*zmask*zmaskvar, both
 *   (npjglo,nlev,nbasins) r8; the caller divides where darea /= 0.  rzomax, rzomin (npjglo,nlev,nbasins) r4
 *   or NULL: the -max option (extrema of dtmp, the minimum over dtmp /= 0, start values -1.e20 / 1.e20). */
int cdft_zonal_compute(cdft_zonal_plan *plan, const float *zv, int nlev, int in_space, int mean_form, double *dzo,
                       double *darea, float *rzomax, float *rzomin, int out_space, void *stream);
/* darea does not depend on the field: it is formed once by cdft_zonal_set_mesh.  A caller that takes it from here (once per
 * mesh) may pass darea = NULL to the mean form of cdft_zonal_compute, which then returns one array instead of two.
 * out_space = CDFT_MEM_PINNED in cdft_zonal_compute: results into page-locked arrays of the caller (cdft_host_alloc) without
 * a closing synchronisation -- cdft_stream_sync(stream) before reading them. */
int cdft_zonal_area(cdft_zonal_plan *plan, int nlev, double *darea, int out_space, void *stream);
int cdft_zonal_destroy(cdft_zonal_plan *plan);

/* ------------------------------------------------------------------------------------------ */
/* cdfvint, cdfvertmean : per-column vertical integrals                  */
/* ------------------------------------------------------------------------------------------ */
/* src/cdfvint.f90:266-311 for one record.  zt, tmask (npiglo,npjglo,npk) r4 (the mask as getvar returns
 * it), e3t (npiglo,npjglo,npk) r4 or NULL with e31d (npk) r4 (-full).  The level schedule is the
 * reference's scalar logic (:271-292), computed by the caller, npk HOST values each:
 *   kout(k)  0-based index of the output depth written after level k, -1 for none (lout, iko)
 *   hit(k)   1 when rdep2 >= gdepo(iko) - tol (:285): the level is clipped to clip(k) = gdepo(iko) - rdep1
 *   rdep2(k) the divisor of -tmean / -smean (mean = 1), unused otherwise
 * out (npiglo,npjglo,npko) r4 = REAL(dl_vint2*sclf) [/ rdep2]; depths never reached stay 0. */
int cdft_vint_cumul(const float *zt, const float *e3t, const float *e31d, const float *tmask, int npiglo, int npjglo,
                    int npk, const int *kout, const int *hit, const float *clip, const float *rdep2, int npko, float sclf,
                    int mean, int in_space, float *out, int out_space, void *stream);
/* src/cdfvertmean.f90:236-270 for one variable and record: the nlev levels ik1..ik2 of zv, zmask and e3
 * (or e31d(ik1:ik2), -full); gdep_first = gdep(ik1), gdep_last = gdep(ik2); dvertmean (npiglo,npjglo) r8,
 * divided by the integrated thickness where that is not 0.  (The reference leaves the level loop when a
 * level has no wet point in the window: such a level adds exact zeros, so it is not special-cased.) */
int cdft_vertmean(const float *zv, const float *e3, const float *e31d, const float *zmask, int npiglo, int npjglo, int nlev,
                  float gdep_first, float gdep_last, float rdep_up, float rdep_down, int in_space, double *dvertmean,
                  int out_space, void *stream);
int cdft_column_release(void);   /* frees the staging buffers of the two calls above */

/* ------------------------------------------------------------------------------------------ */
/* cdfmocsig : meridional overturning in density classes                 */
/*   replaces src/cdfmocsig.f90:368-470 (level loop, density binning, zonal sums, bin cumulation) */
/* ------------------------------------------------------------------------------------------ */
/* One record.  zv, zt, zs (npiglo,npjglo,npk) r4 as read from the files (fill values are replaced by 0
 * here, :374,:382-383, with the spvals given); zveiv the eddy-induced velocity or NULL (-eiv); e1v
 * (npiglo,npjglo); e3v (npiglo,npjglo,npk) or NULL with e31d (npk) (-full); ibmask (nbasins,npiglo,npjglo)
 * i2, basin index fastest, the global one already cleared on columns 1 and npiglo (:357-358);
 * pref, eos: the density is sigmai(zt,zs,pref) (-r, -teos10) or, with eos = CDFT_EOS_NTR, sigmantr(zt,zs) (-ntr, :395);
 * nbins, sigmin, sigstp: the density classes (:282-318).
 * dmoc (nbasins,nbins,npjglo) r8 in Sv, cumulated from the highest density class down (:463-467).
 * depi (nbasins,nbins,npjglo) r8 or NULL: -isodep, the transport-weighted mean depth of each class,
 *   99999. where the class is empty; needs gdep (npk) r4 on the HOST, the NEGATIVE depths of :336. */
int cdft_mocsig_compute(const float *zv, const float *zveiv, const float *zt, const float *zs, const float *e1v,
                        const float *e3v, const float *e31d, const int16_t *ibmask, const float *gdep, int npiglo,
                        int npjglo, int npk, int nbasins, float zspv, float zspt, float zsps, float pref, int eos,
                        int nbins, float sigmin, float sigstp, int in_space, double *dmoc, double *depi, int out_space,
                        void *stream);
int cdft_mocsig_release(void);   /* frees the staging buffers of cdft_mocsig_compute */

/* ------------------------------------------------------------------------------------------ */
/* cdfpsi : barotropic stream function                                                        */
/*   replaces src/cdfpsi.f90:271-298 (vertical integration), :377-423 (scans, mean, reference  */
/*   point) and the SNGL / mask at :426-433.  Not covered: -open, --ssh-file, -lev.            */
/* ------------------------------------------------------------------------------------------ */
typedef struct cdft_psi_plan cdft_psi_plan;

int cdft_psi_create(cdft_psi_plan **plan, int npiglo, int npjglo, int npk);
/* sharded ("sharded plans" above): cdft_psi_compute and the staged entry points below all work on it (cdft_psi_scan_u
 * then takes no carry and returns no row: the plan chains its own blocks); only cdft_psi_scan_open (-open, a line
 * integral along the row npjglo-2 of the whole domain) needs a plan of cdft_psi_create. */
int cdft_psi_create_sharded(cdft_psi_plan **plan, int npiglo, int npjglo, int npk, int ndev, const int *devices);
/* One record.  U set (default path): zu (npiglo,npjglo,npk), e2u (npiglo,npjglo), e3u (3-D) r4;
 * V set (-V): zv, e1v, e3v; both sets = -mean.  e31d (npk) replaces the 3-D e3 with -full.
 * (iiref,ijref): 1-based reference point, default (npiglo,npjglo) (:203-206); fmask (npiglo,npjglo)
 * r4 or NULL (-mask).  psi4 (npiglo,npjglo) r4 = what putvar writes; dpsi8 (optional, may be NULL)
 * = the r8 field before SNGL. */
int cdft_psi_compute(cdft_psi_plan *plan, const float *zu, const float *e2u, const float *e3u,
                     const float *zv, const float *e1v, const float *e3v, const float *e31d, int iiref,
                     int ijref, const float *fmask, int in_space, float *psi4, double *dpsi8, int out_space,
                     void *stream);
/* The same computation in stages, for a domain that is row-block partitioned over several GPUs
 * (cdftools_amd/shard.py): cdft_psi_compute == integrate + scan_u(no carry, first block) + finish.
 *   integrate : vertical integration of the block (and the i-scan of the V path, which stays inside rows).
 *   scan_u    : the j-scan of the U path.  A block that does not hold global row 1 passes
 *               first_block = 0 and carry(npiglo) r8 = the last row of the block south of it; last_row
 *               (optional) returns this block's last row for the next one.  Chained in rank order the
 *               result is bit-identical to the single-GPU one.
 *   value     : the (mean of the) stream function at a local point, before the reference shift
 *               (the owner of the reference point broadcasts it).
 *   finish    : reference shift (use_ref = 1: ref_value given; 0: local point (iiref,ijref)), SNGL, mask. */
int cdft_psi_integrate(cdft_psi_plan *plan, const float *zu, const float *e2u, const float *e3u,
                       const float *zv, const float *e1v, const float *e3v, const float *e31d, int in_space,
                       void *stream);
int cdft_psi_scan_u(cdft_psi_plan *plan, const double *carry, int carry_space, int first_block,
                    double *last_row, int last_space, void *stream);
/* -open (src/cdfpsi.f90:316-366): after cdft_psi_integrate with BOTH velocity sets, psi is built from
 * the row npjglo-2 (line integral of the V transport, then southwards with the U transport) instead
 * of cdft_psi_scan_u; lmean adds the estimate from the column npiglo-2 and averages (-open -mean).
 * Follow with cdft_psi_finish as usual.  (The whole domain: not for row-block partitioned runs.) */
int cdft_psi_scan_open(cdft_psi_plan *plan, int lmean, void *stream);
int cdft_psi_value(cdft_psi_plan *plan, int ii, int ij, double *value, void *stream);
int cdft_psi_finish(cdft_psi_plan *plan, int use_ref, double ref_value, int iiref, int ijref,
                    const float *fmask, int mask_space, float *psi4, double *dpsi8, int out_space, void *stream);
int cdft_psi_destroy(cdft_psi_plan *plan);

/* ------------------------------------------------------------------------------------------ */
/* cdfvT : time-mean of u.T, v.T, u.S, v.S at U / V points                                    */
/*   replaces src/cdfvT.f90:160-259 (all levels at once instead of the reference's level loop; */
/*   per cell the accumulation over frames keeps the reference's order)                       */
/* ------------------------------------------------------------------------------------------ */
/* u, v, t, s: nframes frames of (npiglo,npjglo,npk) r4, frame f at ptr + f*fstride elements (all
 * tags concatenated, :168-191).  e3u, e3v: same shape with -vvl, else NULL (:193-196).
 * Outputs (npiglo,npjglo,npk), r8 (out_is_r4 = 0: the array handed to putvar, :247-250) or r4
 * (out_is_r4 = 1: what NetCDF stores in the float variable): vt, vs, ut, us, and with -vvl the
 * mean e3u, e3v (e3um, e3vm; NULL otherwise).  Call with npk = 1 to work level by level.
 * out_space = CDFT_MEM_PINNED (outputs from cdft_host_alloc): on their way when the call returns, cdft_stream_sync(stream)
 * before reading them -- the program writes level k while level k+1 is read and computed. */
int cdft_vt_compute(const float *u, const float *v, const float *t, const float *s, const float *e3u,
                    const float *e3v, int64_t fstride, int nframes, int npiglo, int npjglo, int npk,
                    int in_space, void *vt, void *vs, void *ut, void *us, void *e3um, void *e3vm,
                    int out_is_r4, int out_space, void *stream);
/* The same over ndev row blocks, block b on HIP device devices[b] ("sharded plans" above): every block but the last also
 * reads the row north of it (T_v looks at jj+1, :205); arrays in any memory space, returns when all blocks are done. */
int cdft_vt_compute_sharded(const float *u, const float *v, const float *t, const float *s, const float *e3u,
                            const float *e3v, int64_t fstride, int nframes, int npiglo, int npjglo, int npk,
                            void *vt, void *vs, void *ut, void *us, void *e3um, void *e3vm, int out_is_r4, int ndev,
                            const int *devices);
int cdft_stencil_release(void);   /* frees the staging buffers cdft_vt_compute uses for host arrays */

/* ------------------------------------------------------------------------------------------ */
/* cdfcurl : relative vorticity at F points                                                   */
/*   replaces src/cdfcurl.f90:262-301 (fmask, curl, E-W wrap, -T, -overf).  Not covered: the   */
/*   A-grid averaging of forcing fields (:250-258).                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct cdft_curl_plan cdft_curl_plan;

int cdft_curl_create(cdft_curl_plan **plan, int npiglo, int npjglo);
/* e1u, e2v, e1f, e2f (npiglo,npjglo) r4 (:200-203; pass arrays of 1. for -nometric);
 * ff (npiglo,npjglo) r8 = 2*omega*sin(gphif*pi/180) for -overf (:225-227) or NULL;
 * lperio: glamf(1,1) == glamf(npiglo-1,1) (:238). */
int cdft_curl_set_mesh(cdft_curl_plan *plan, const float *e1u, const float *e2v, const float *e1f,
                       const float *e2f, const double *ff, int lperio, int memspace, void *stream);
/* un, vn (npiglo,npjglo,nlev) r4: the selected levels of one record (:247-248).
 * out (npiglo,npjglo,nlev): r8 (drotn as handed to putvar, -8) or r4 (float variable).
 * ltpoint: -T, 4-point mean to T points (:286-297).
 * out_space = CDFT_MEM_PINNED (out from cdft_host_alloc): the result is on its way when the call returns,
 * cdft_stream_sync(stream) before reading it -- the program computes level k+1 while it writes level k. */
int cdft_curl_compute(cdft_curl_plan *plan, const float *un, const float *vn, int nlev, int in_space,
                      int ltpoint, void *out, int out_is_r8, int out_space, void *stream);
int cdft_curl_destroy(cdft_curl_plan *plan);

#ifdef __cplusplus
}
#endif
#endif /* CDFTOOLS_HIP_H */
