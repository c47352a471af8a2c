/*
 * homonim_hk.h -- C ABI of the MI355X-native homonim kernel-model hot path (libhomonim_hk.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Every entry point names the
 * reference interface it replaces (paths relative to the reference repo leftfield-geospatial/homonim v0.4.3).
 * The reference has no FFI of its own (it is pure Python on top of OpenCV/GDAL wheels); the binding a maintainer
 * would add is a ctypes stub inside homonim/kernel_model.py -- shown in INTEGRATION.md and implemented in
 * homonim_amd/_hk.py.
 *
 * Conventions
 *   - All functions return 0 on success, a negative hk_status otherwise; hk_last_error() gives the text for the
 *     calling thread.  Nothing aborts the process.
 *   - Rasters are float32, row-major, one band per 2-D plane; `stride` is in ELEMENTS between rows.
 *   - nodata is passed as (mode, value): HK_NODATA_NONE = RasterArray.nodata is None (all pixels valid),
 *     HK_NODATA_NAN = nodata is NaN, HK_NODATA_VALUE = numeric nodata, compared with utils.nan_equals semantics
 *     (homonim/utils.py:54-56; homonim/raster_array.py:298-308).
 *   - kernel_shape is (kh, kw) = (rows, cols), both odd (homonim/utils.py:104-133).
 *   - Thread safety: any number of host threads may call into one hk_ctx concurrently (the reference calls
 *     fit/apply from a ThreadPoolExecutor on ONE shared model, homonim/fuse.py:396-401); each call checks a
 *     stream + staging slot out of the context's pool.
 */
#ifndef HOMONIM_HK_H
#define HOMONIM_HK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hk_ctx hk_ctx;
typedef struct hk_event hk_event;   /* a point in a context stream (hk_event_*, bottom of this header) */

typedef enum {
    HK_OK = 0,
    HK_ERR_ARG = -1,       /* bad argument (shape, kernel, null pointer ...) -> Python ValueError */
    HK_ERR_HIP = -2,       /* HIP runtime error -> Python RuntimeError */
    HK_ERR_NODEVICE = -3,  /* no usable GPU */
    HK_ERR_UNSUPPORTED = -4,
    HK_ERR_NOMEM = -5,
    HK_ERR_ALREADY = -6    /* hk_host_register: the range is page-locked already (nothing was changed; do not unregister it) */
} hk_status;

/* homonim/enums.py:22-41 (Model) */
typedef enum { HK_MODEL_GAIN = 0, HK_MODEL_GAIN_BLK_OFFSET = 1, HK_MODEL_GAIN_OFFSET = 2 } hk_model;

typedef enum { HK_NODATA_NONE = 0, HK_NODATA_NAN = 1, HK_NODATA_VALUE = 2 } hk_nodata_mode;

/*
 * Everything KernelModel.__init__ / create_config fix for the hot path (homonim/kernel_model.py:40-81,98-136)
 * plus the per-call nodata of the two RasterArrays.
 */
typedef struct {
    int32_t model;          /* hk_model */
    int32_t kh, kw;         /* kernel_shape (rows, cols), odd */
    int32_t find_r2;        /* KernelModel.find_r2: emit the R2 band (kernel_model.py:267-272,353-359) */
    int32_t has_r2_thresh;  /* r2_inpaint_thresh is not None (gain-offset only; kernel_model.py:325,361) */
    float r2_thresh;        /* r2_inpaint_thresh */
    int32_t src_nodata_mode;
    float src_nodata;
    int32_t ref_nodata_mode;
    float ref_nodata;
} hk_fit_desc;

/* ------------------------------------------------------------------------------------------------------------------
 * library / context
 */
/* Version of this header's struct layouts and prototypes.  A consumer built against one header and loading a library of
 * another must not call further: structs grow at their end between versions (hk_out_window: 32 -> 40 bytes in version 3) and
 * carry no size field; version 5 added entry points (hk_device_pci_bus_id; hk_debug_staging_counters in the devtools header); version 6: a raw
 * r2-mask failure counter is always a count (HK_COUNT_RETRY is never set any more), and a device-resident job that carries `scratch`
 * always gets the in-painting's inputs left there.  hk_abi_version() returns the library's HK_ABI_VERSION; compare it with the header's at load time. */
#define HK_ABI_VERSION 6
int hk_abi_version(void);
const char* hk_backend_name(void);            /* "hip-gfx950" */
const char* hk_last_error(void);              /* thread-local text of the last failure */
int hk_device_count(int* count);
/* PCI bus address of HIP device `device_id` as sysfs spells it ("0000:c1:00.0"; `len` >= 13), so that a rank can look up its
 * GPU's NUMA node (/sys/bus/pci/devices/<address>/numa_node) and run its host threads -- the staging copies, the pinned
 * allocations -- on that node's cores (homonim_amd/topology.py).  The reference's "devices" are the host's own cores
 * (homonim/fuse.py:396-401: a ThreadPoolExecutor); there is nothing to place. */
int hk_device_pci_bus_id(int device_id, char* out, int len);
/* One context per GPU: owns `n_streams` HIP streams, each with a pinned-host + device staging slot that grows on
 * demand.  Replaces nothing in the reference (its "device" is the host CPU); mirrors the thread pool of
 * homonim/fuse.py:396.
 * No process-wide side effects by default; the diagnostic switches (HK_FAULT_REPORT=1, HK_GUARD_ALLOC) are described in
 * homonim_hk_devtools.h.  Every entry point that selects the context's device also clears the calling thread's sticky HIP
 * "last error" (hipGetLastError), so that an error this library reported does not resurface in the caller's next launch. */
int hk_ctx_create(int device_id, int n_streams, hk_ctx** ctx);
int hk_ctx_destroy(hk_ctx* ctx);
int hk_ctx_sync(hk_ctx* ctx);                 /* hipDeviceSynchronize on the context's device */

/* ------------------------------------------------------------------------------------------------------------------
 * host-pointer entry points (numpy arrays in, numpy arrays out): H2D -> kernels -> D2H on one pooled stream.
 */

/* KernelModel._fit_block_norm (homonim/kernel_model.py:216-229): norm[0] = std(ref[mask]) / std(src[mask]),
 * norm[1] = percentile(ref[mask], 1) - percentile(src[mask], 1) * norm[0]; [0, 0] when no pixel is valid. */
int hk_block_norm(hk_ctx* ctx, const hk_fit_desc* desc, const float* src, int64_t src_stride, const float* ref,
                  int64_t ref_stride, int32_t height, int32_t width, double norm_out[2]);

/* RasterCompare.process / get_block_sums (homonim/compare.py:243-255) for one band of two rasters on the same grid:
 * sums_out = [ sum src, sum ref, sum src^2, sum ref^2, sum src*ref, sum (ref - src)^2, N ] over jointly valid pixels
 * (see hk_compare_sums_dev for the arithmetic). */
int hk_compare_sums(hk_ctx* ctx, const float* src, int64_t src_stride, int32_t src_nodata_mode, float src_nodata,
                    const float* ref, int64_t ref_stride, int32_t ref_nodata_mode, float ref_nodata, int32_t height,
                    int32_t width, double sums_out[7]);

/* KernelModel.fit (homonim/kernel_model.py:411-440 -> _fit_gain :231-274, _fit_gain_blk_offset :276-303,
 * _fit_gain_offset :305-373, _r2_array :142-214).
 *   params_out : n_param_bands x height x width float32, band 0 gain, 1 offset, 2 R2 (iff n_param_bands == 3;
 *                must equal 3 exactly when find_r2 || (model == gain-offset && has_r2_thresh)), NaN outside mask.
 *   norm_in    : gain-blk-offset only: float64[2] block normalisation to use, or NULL to compute it (hk_block_norm).
 *   norm_out   : optional float64[2], receives the normalisation actually used.
 *   r2_fail_count : optional; gain-offset with has_r2_thresh: number of valid pixels failing
 *                (r2 > thresh) & (gain > 0) (kernel_model.py:363).  When it is 0 the reference's in-paint branch is
 *                the identity; when > 0 the offsets of those pixels have been in-painted from the passing ones and
 *                their gains recomputed on the device (kernel_model.py:364-371; restated GDALFillNodata, hk_inpaint.hip)
 *                before params_out is written.
 * Unlike the reference, src/ref are NOT modified (no in-place zero-fill, kernel_model.py:246-247,320-321). */
int hk_fit(hk_ctx* ctx, const hk_fit_desc* desc, const float* src, int64_t src_stride, const float* ref,
           int64_t ref_stride, int32_t height, int32_t width, const double* norm_in, float* params_out,
           int32_t n_param_bands, double* norm_out, uint64_t* r2_fail_count);

/* KernelModel.apply (homonim/kernel_model.py:442-463): out = params[0] * src + params[1] (two float32 roundings). */
int hk_apply(hk_ctx* ctx, const float* src, int64_t src_stride, const float* params /* 2 x H x W */,
             int32_t height, int32_t width, float* out);

/* fit + apply of RasterFuse._process_block (homonim/fuse.py:305-307) fused in one pass: window sums, solve, R2 test
 * and correction in a single kernel; parameters are materialised only if params_out != NULL. */
int hk_fit_apply(hk_ctx* ctx, const hk_fit_desc* desc, const float* src, int64_t src_stride, const float* ref,
                 int64_t ref_stride, int32_t height, int32_t width, const double* norm_in,
                 float* params_out /* nullable */, int32_t n_param_bands, float* corr_out, double* norm_out,
                 uint64_t* r2_fail_count);

/* RasterArray.reproject (homonim/raster_array.py:526-578 -> GDAL warp) between two grids of the SAME CRS that are north-up
 * and axis-aligned, as RefSpaceModel / SrcSpaceModel call it (homonim/kernel_model.py:397,480,491,497,520).
 *   mapping   : src_col = kx * dst_col + ox, src_row = ky * dst_row + oy on continuous pixel coordinates whose integers
 *               are pixel edges (kx, ky > 0);
 *   resampling: rasterio.enums.Resampling value -- 0 nearest, 1 bilinear, 3 cubic_spline (both up-sampling only),
 *               5 average;
 *   src       : n_bands x src_h x src_w float32 with the given nodata; dst: n_bands x dst_h x dst_w float32, pixels that
 *               receive nothing are set to dst_fill (the destination nodata; 0 when that is None). */
int hk_reproject(hk_ctx* ctx, const float* src, int32_t n_bands, int32_t src_height, int32_t src_width,
                 int32_t src_nodata_mode, float src_nodata, double kx, double ox, double ky, double oy, int32_t resampling,
                 float* dst, int32_t dst_height, int32_t dst_width, float dst_fill);

/* `mask_partial` on a shared grid: KernelModel._full_coverage_mask (homonim/kernel_model.py:375-409) -- the mask of
 * pixels that are valid in `in` (nodata as given) and have parameters, eroded by a (kh+2) x (kw+2) rectangle with a
 * zero border -- followed by what the reference does with it:
 *   RefSpaceModel.apply (:493-503): in = the source block, corr_out = gain * src + offset with parameters outside the
 *                                   mask set to NaN (pass src = in);
 *   SrcSpaceModel.fit   (:526-531): in = the reference block, params_out = all n_param_bands masked.
 * params: n_param_bands x H x W float32 (band 0 gain, 1 offset); params_out / corr_out / mask_out (uint8) nullable.
 * in_nodata_mode 3 = `in` is a coverage fraction (the mask re-projected with `average`): valid where in >= 1 (:399). */
int hk_partial_mask(hk_ctx* ctx, const float* in, int64_t in_stride, int32_t in_nodata_mode, float in_nodata,
                    const float* params, int32_t n_param_bands, const float* src, int64_t src_stride, int32_t height,
                    int32_t width, int32_t kh, int32_t kw, float* params_out, float* corr_out, uint8_t* mask_out);

/* Typed rasters either side of the path: integer / float64 inputs are converted to float32 on the device exactly as
 * RasterArray.from_rio_dataset reads them (homonim/raster_array.py:178-188), the corrected block is converted to the
 * output dtype as RasterArray._convert_array_dtype does for to_rio_dataset (homonim/raster_array.py:353-387: round
 * half-to-even, clip, masked pixels -> out_nodata).  PCIe then carries 1-2 B per pixel instead of 4. */
typedef enum { HK_DTYPE_F32 = 0, HK_DTYPE_U8 = 1, HK_DTYPE_U16 = 2, HK_DTYPE_I16 = 3, HK_DTYPE_U32 = 4, HK_DTYPE_I32 = 5,
               HK_DTYPE_F64 = 6 } hk_dtype;
typedef struct {
    int32_t src_dtype, ref_dtype;  /* hk_dtype of the src / ref host arrays */
    int32_t out_dtype;             /* hk_dtype of corr_out */
    int32_t out_has_nodata;        /* 0: masked pixels keep NaN (float outputs) / become 0 (integer outputs) */
    double out_nodata;
} hk_io_desc;
/* hk_fit_apply with typed src / ref / corr_out (strides in ELEMENTS of the respective dtype); io == NULL means float32
 * everywhere.  params_out stays float32. */
int hk_fit_apply_io(hk_ctx* ctx, const hk_fit_desc* desc, const hk_io_desc* io, const void* src, int64_t src_stride,
                    const void* ref, int64_t ref_stride, int32_t height, int32_t width, const double* norm_in,
                    float* params_out /* nullable */, int32_t n_param_bands, void* corr_out /* nullable */,
                    double* norm_out, uint64_t* r2_fail_count);

/* Where the outputs of a block go when the caller keeps whole rasters on the host (homonim/fuse.py:310-319 writes the
 * out-block of every block into the corrected / parameter files, raster_array.py:478-491 crops the halo): rows
 * [row0, row0 + rows) x columns [col0, col0 + cols) of the block are copied to corr_out / params_out, whose rows are
 * `stride` elements apart and whose parameter planes `band_stride` elements -- i.e. corr_out points at the pixel of
 * the caller's raster where the window's first pixel belongs. */
typedef struct hk_out_window {
    int64_t stride;       /* elements between rows of corr_out (and of params_out unless param_stride is set); >= cols */
    int64_t band_stride;  /* elements between the planes of params_out */
    int32_t row0, col0;   /* first row / column of the block that is written */
    int32_t rows, cols;   /* size of the written window */
    int64_t param_stride; /* elements between rows of params_out when it differs from corr_out's; 0: `stride` serves both */
} hk_out_window;

/* hk_fit_apply_io with an output window: the block loop of RasterFuse.process (homonim/fuse.py:295-319) per call --
 * read-block in, out-block of the corrected raster (and of the parameters) out, straight into the caller's arrays.
 * When src / ref / corr_out / params_out are page-locked (hk_host_alloc / hk_host_register) the copies are asynchronous
 * and the call synchronises its stream once; calls from several threads overlap their transfers and kernels. */
int hk_fit_apply_block(hk_ctx* ctx, const hk_fit_desc* desc, const hk_io_desc* io, const void* src, int64_t src_stride,
                       const void* ref, int64_t ref_stride, int32_t height, int32_t width, const double* norm_in,
                       float* params_out, int32_t n_param_bands, void* corr_out, const hk_out_window* window,
                       double* norm_out, uint64_t* r2_fail_count);


/* RefSpaceModel.fit + RefSpaceModel.apply (homonim/kernel_model.py:476-503) of one block pair on DIFFERENT grids of one
 * CRS, entirely on the device -- source and reference blocks in, corrected block (source grid) out:
 *   source --down_resampling--> reference grid; KernelModel.fit there (incl. block statistics / in-painting);
 *   gain, offset --up_resampling--> source grid; re-masked with the source mask, or (mask_partial) with the full-coverage
 *   mask of kernel_model.py:375-409 brought back with `nearest`; KernelModel.apply.
 * down = mapping source <- reference grid (src_col = down[0] * ref_col + down[1], src_row = down[2] * ref_row + down[3]),
 * up = mapping reference <- source grid; both as hk_reproject defines them.  params_out (nullable): n_param_bands planes on
 * the REFERENCE grid.  io (nullable) types src / ref / corr_out as in hk_fit_apply_io. */
typedef struct {
    double down[4];
    double up[4];
    int32_t down_resampling;  /* rasterio.enums.Resampling value, KernelModel._get_resampling(src.res, ref.res) */
    int32_t up_resampling;    /* KernelModel._get_resampling(param.res, src.res) */
    int32_t mask_partial;
} hk_space_desc;
int hk_refspace_fit_apply(hk_ctx* ctx, const hk_fit_desc* desc, const hk_io_desc* io, const hk_space_desc* space,
                          const void* src, int64_t src_stride, int32_t src_height, int32_t src_width, const void* ref,
                          int64_t ref_stride, int32_t ref_height, int32_t ref_width, float* params_out,
                          int32_t n_param_bands, void* corr_out, uint64_t* r2_fail_count);

/* Page-lock caller memory so the host-pointer entry points above become truly asynchronous: with pinned src/ref/output
 * arrays the H2D copy, the kernel and the D2H copy of different calls (different host threads, different pooled streams)
 * overlap; with pageable memory HIP stages every copy synchronously.  The reference has no counterpart (its blocks are
 * numpy arrays read by rasterio, homonim/raster_pair.py:331-340). */
int hk_host_alloc(hk_ctx* ctx, size_t bytes, void** hptr);   /* pinned allocation */
int hk_host_free(hk_ctx* ctx, void* hptr);                   /* ctx may be NULL (the memory belongs to no device) */
int hk_host_register(hk_ctx* ctx, void* hptr, size_t bytes); /* pin existing memory in place (usable from every device) */
int hk_host_unregister(hk_ctx* ctx, void* hptr);             /* ctx may be NULL */

/* ------------------------------------------------------------------------------------------------------------------
 * device-resident entry points (inputs already in HBM): what bench.py times and what the streaming tile pipeline
 * is built from.  Buffers come from hk_dev_alloc; planes are height x stride float32 with stride % 4 == 0.
 * A job names its stream by index: jobs on one stream must be issued by one host thread at a time (stream order is the
 * only ordering).  Host-pointer calls (hk_fit, ...) lease the same pooled streams and their scratch buffers; the library
 * keeps the two apart at run time: a lease prefers a stream no device job has used, drains one that was used before taking
 * it, and a device-job call waits while its stream is leased.  Mixing both on one context is therefore safe call by call but
 * serialises them on that stream -- use a second context where they should overlap, and always when a gain-offset job with
 * an r2 threshold runs WITHOUT hk_dev_job.scratch: between its hk_fit_apply_dev and its hk_inpaint_dev* the in-painting's
 * inputs live in the stream's own scratch, which a host-pointer call leasing that stream would overwrite.
 */
int hk_dev_alloc(hk_ctx* ctx, size_t bytes, void** dptr);
int hk_dev_free(hk_ctx* ctx, void* dptr);
int hk_memcpy_h2d(hk_ctx* ctx, void* dst, const void* src, size_t bytes);
int hk_memcpy_d2h(hk_ctx* ctx, void* dst, const void* src, size_t bytes);
int hk_memset(hk_ctx* ctx, void* dst, int value, size_t bytes);

typedef struct {
    const float* src;      /* device, n_bands planes */
    const float* ref;      /* device, n_bands planes */
    float* gain;           /* device, nullable */
    float* offset;         /* device, nullable */
    float* r2;             /* device, nullable (written only when the R2 variant runs) */
    float* corr;           /* device, nullable */
    const double* norm;    /* device, n_bands x 2 float64 (gain-blk-offset), else NULL */
    uint64_t* fail_count;  /* device, n_bands counters (must be zeroed by the caller), nullable */
    int32_t n_bands;
    int32_t height, width;
    int64_t stride;        /* elements between rows, all planes */
    int64_t band_stride;   /* elements between band planes, all arrays */
    int32_t seg_rows;      /* rows per wave segment; 0 = library default */
    int32_t stream;        /* index into the context's stream pool */
    /* Store window of hk_fit_apply_dev, in job coordinates: only these rows / columns of the job are written (and counted).
     * A block of a larger device-resident raster is processed in place -- src / ref / corr point at the block's first
     * pixel INCLUDING its halo, height / width are the in-block's -- and only its out-block is stored: the halo crop of
     * homonim/raster_array.py:478-491 as homonim/fuse.py:310-312 applies it.  All zero = the whole job.  out_col0 and
     * out_col0 + out_cols must be multiples of 4 (or end at the job's last column).  Not with r2_inpaint_thresh. */
    int32_t out_row0, out_col0, out_rows, out_cols;
    /* Optional device scratch for gain-offset with an r2 threshold (else NULL / 0): hk_dev_job_scratch_bytes() bytes,
     * 16-byte aligned, owned by the caller and tied to this job until its hk_inpaint_dev / hk_inpaint_dev_counts.  A job that
     * carries it gets the in-painting's inputs left there by hk_fit_apply_dev -- offsets and the one-byte source flags
     * (r2 > thresh) & (gain > 0) & valid (kernel_model.py:363), 5 bytes per pixel of stores -- and the in-painting starts from
     * them instead of running the fit once more: provide it where pixels are expected to fail the r2 mask (real imagery:
     * block after block), leave it NULL where they are not (the fit then runs its lighter build and moves 12 bytes per pixel). */
    void* scratch;
    uint64_t scratch_bytes;
} hk_dev_job;
/* Size of hk_dev_job.scratch for a job of this shape (5 bytes per pixel of the job's planes incl. row / band padding). */
uint64_t hk_dev_job_scratch_bytes(int32_t n_bands, int32_t height, int64_t stride, int64_t band_stride);

/* Launch the fused kernel over all bands of a device-resident job (asynchronous on stream `job->stream`).  With an r2
 * threshold the pixels failing the mask are only COUNTED (job->fail_count); follow with hk_inpaint_dev. */
int hk_fit_apply_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job);
/* Second half of gain-offset with an r2 threshold on a device-resident job (kernel_model.py:361-371): waits for the
 * stream, reads job->fail_count and, for every band with failing pixels, in-paints their offsets and re-runs the fit
 * with them (recomputed gains, re-applied correction; parameter planes of the job are updated when present).  A no-op
 * for the other models.  *n_fail_out (nullable) receives the number of failing pixels over all bands; the counters
 * are then cleared (asynchronously), ready for the job's next hk_fit_apply_dev.  Returns after queueing the passes on
 * `job->stream`. */
int hk_inpaint_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, uint64_t* n_fail_out);
/* The same in two halves, for pipelines that must not drain the stream after every launch: hk_fail_counts_async queues the
 * copy of the job's counters into `host_counts` (n_bands values, pinned host memory: hk_host_alloc), their clearing and
 * the recording of `ready`; after hk_event_sync(ready) the caller passes the counts to hk_inpaint_dev_counts, which only
 * queues work.  Launch N + 1 (into a second counter buffer) may be queued before the counts of launch N are looked at.
 * Pass the counts on unchanged.  (Up to ABI version 5 a raw counter could carry HK_COUNT_RETRY instead of a count: the lighter
 * kernel build such jobs start with voided the whole band when it could not settle a wave-row.  Since version 6 that build marks
 * the wave-rows it leaves open and a second launch of the complete build does exactly those: every counter is a count.  The
 * constant stays defined for consumers written against the older header; the library never sets it.) */
#define HK_COUNT_RETRY (1ull << 63)
int hk_fail_counts_async(hk_ctx* ctx, const hk_dev_job* job, uint64_t* host_counts, hk_event* ready);
/* 1 when the counts of a launch call for its second half (hk_inpaint_dev_counts): some band has failing pixels; 0 when the
 * launch's outputs are final.  Callers need not interpret the raw counters (host-only, no device call). */
int hk_counts_pending(const uint64_t* counts, int32_t n_bands);
int hk_inpaint_dev_counts(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, const uint64_t* counts,
                          uint64_t* n_fail_out);
/* Per-band block normalisation on device planes -> norm (device, n_bands x 2 float64); asynchronous. */
int hk_block_norm_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, double* norm_dev);
/* BATCHED launches: `n_jobs` device-resident jobs -- the block positions of a resident mosaic (homonim/raster_pair.py:342-428
 * yields them one by one and homonim/fuse.py:395-404 hands each to a thread), the tiles of a tile list -- as ONE launch per
 * kernel stage on jobs[0].stream (all jobs name that stream) instead of one launch, and one launch tail, per job.  Jobs may
 * differ in shape, planes and store window; they share `desc` and ask for the same set of outputs (the same pointers are NULL in
 * every job).  Results are bit-identical to the per-job calls: a job's statistics and wave units do not depend on the launch it
 * travels in.  hk_block_norm_batch_dev writes the jobs' statistics one after the other (job 0's n_bands x 2 float64, then job
 * 1's, ...) into norm_dev; hk_fit_apply_batch_dev reads each job's own `norm` pointer (point it into that buffer).  With an r2
 * threshold every job keeps its own fail_count / scratch and is finished by its own hk_inpaint_dev* call, as after
 * hk_fit_apply_dev.  The job tables travel through a small ring of pinned staging buffers of the stream: the calls only queue
 * work.  One fused launch exists for gain-blk-offset without R2 (the block-partitioned mosaic, where it pays); for the other
 * models hk_fit_apply_batch_dev queues the jobs' launches one after the other -- the job-table look-up at the start of every
 * wave costs their short waves more than the launches it saves.  Limits: 65536 jobs, 32767 planes (jobs x bands) per statistics
 * batch; the statistics workspace of the stream grows to planes x (190 KB + 8 % of the largest plane). */
int hk_block_norm_batch_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* jobs, int32_t n_jobs, double* norm_dev);
int hk_fit_apply_batch_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* jobs, int32_t n_jobs);
/* hk_fail_counts_async for the jobs of a batch: their counters, job after job (sum of n_bands values), into the pinned
 * `host_counts`, cleared on the device, ONE event.  Pass each job its own slice to hk_inpaint_dev_counts afterwards. */
int hk_fail_counts_batch_async(hk_ctx* ctx, const hk_dev_job* jobs, int32_t n_jobs, uint64_t* host_counts, hk_event* ready);
/* The same statistics (KernelModel._fit_block_norm, homonim/kernel_model.py:216-229) for a block whose ROWS are spread over
 * `world_size` ranks / devices -- the optional collective of a gain-blk-offset block too large for one GPU.  `job` is this
 * rank's slab of the block (any number of rows, the block's width).  Phases 0..5 are queued one at a time on job->stream;
 * between two phases the caller waits for the stream and all-reduces (SUM, float64) the exchange buffer `xchg_dev` over the
 * ranks (RCCL: torch.distributed.all_reduce on a tensor that owns the buffer; homonim_amd/split_norm.py), all ranks in step.
 * The buffer holds hk_block_norm_split_exchange_doubles(n_bands) values: the slabs' shifts, then their moments, then the
 * three histogram levels of the exact radix select (integer counts, exact in float64).  After phase 5 `norm_dev`
 * (n_bands x 2 float64, device) holds the block's norm, identical on every rank: order statistics exactly those of the whole
 * block, std ratio equal to the single-device value up to the order of the float64 sums. */
uint64_t hk_block_norm_split_exchange_doubles(int32_t n_bands);
int hk_block_norm_split_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, int32_t phase, int32_t world_size,
                            double* xchg_dev, double* norm_dev);
/* The same with the all-reduces done by the library: RCCL over xGMI, queued on job->stream between the phases -- no host
 * synchronisation, no tensor library in the data path (BASELINE.json north_star: "RCCL over xGMI only for the optional
 * block-mean offset reduction").  One process per GPU; every rank creates its context, joins the communicator once
 * (hk_comm_init) and then calls hk_block_norm_split_comm_dev with its slab of each block, all ranks in the same order.
 *   hk_comm_unique_id : rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks through whatever
 *                       launched them (a file, the torch.distributed store, MPI ...: homonim_amd/dist.py init_comm).
 *   hk_comm_init      : collective over the ranks (ncclCommInitRank); one communicator per context.
 *   hk_block_norm_split_comm_dev : asynchronous; norm_dev (n_bands x 2 float64, device) is valid in stream order and
 *                       identical on every rank.  A slab may have no rows (height 0): the rank contributes zeros.
 *                       A block of 2^32 or more valid pixels gets NaN statistics on every rank (32-bit histogram bins).
 *   hk_comm_allreduce_f64_dev : in-place SUM of a device buffer over the communicator on a pooled stream (the primitive
 *                       above; exported for callers with their own phase loop).
 * librccl is opened at run time on first use (HK_ERR_UNSUPPORTED if it is absent). */
#define HK_COMM_ID_BYTES 128
int hk_comm_unique_id(uint8_t id[HK_COMM_ID_BYTES]);
int hk_comm_init(hk_ctx* ctx, const uint8_t id[HK_COMM_ID_BYTES], int32_t rank, int32_t world_size);
int hk_comm_destroy(hk_ctx* ctx);
int hk_comm_info(hk_ctx* ctx, int32_t* rank, int32_t* world_size);   /* rank -1 / world 0: no communicator */
int hk_comm_allreduce_f64_dev(hk_ctx* ctx, double* buf_dev, uint64_t count, int32_t stream);
int hk_block_norm_split_comm_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, double* norm_dev);
/* The masked sums of RasterCompare.process / get_block_sums (homonim/compare.py:243-255) on the job's src and ref planes
 * (job->corr etc. are not used), per band into sums_dev (device, n_bands x 7 float64; asynchronous):
 *   [ sum src, sum ref, sum src^2, sum ref^2, sum src*ref, sum (ref - src)^2, number of pixels ]
 * over the pixels valid in both rasters.  Per-pixel terms are formed in float32 as numpy does on float32 arrays; they are
 * accumulated in float64 in a fixed order (the reference's float32 pairwise sums differ from these by ~1e-7 relative).
 * r2 / RMSE / rRMSE follow from them as in compare.py:142-160. */
int hk_compare_sums_dev(hk_ctx* ctx, const hk_dev_job* job, int32_t src_nodata_mode, float src_nodata,
                        int32_t ref_nodata_mode, float ref_nodata, double* sums_dev);


/* HIP events on the pooled streams, so callers time exactly the stream the kernels run on. */
int hk_event_create(hk_ctx* ctx, hk_event** ev);
int hk_event_destroy(hk_ctx* ctx, hk_event* ev);
int hk_event_record(hk_ctx* ctx, hk_event* ev, int32_t stream);
int hk_stream_wait_event(hk_ctx* ctx, int32_t stream, hk_event* ev); /* work queued on `stream` after this call waits for the event (device side) */
int hk_event_sync(hk_ctx* ctx, hk_event* ev);  /* blocks the calling thread until the event has happened */
int hk_event_elapsed_ms(hk_ctx* ctx, hk_event* start, hk_event* stop, float* ms); /* syncs on `stop` */
int hk_stream_sync(hk_ctx* ctx, int32_t stream);

/* Self-test of the cross-lane primitives the kernels rely on (DPP wave shifts); 0 = pass. */
int hk_selftest(hk_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* HOMONIM_HK_H */
