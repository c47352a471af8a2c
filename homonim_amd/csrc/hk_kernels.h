// hk_kernels.h -- launch interface between the C-ABI host layer (hk_api.hip) and the gfx950 kernels (hk_fit_kernel.h: the fused kernel; hk_kernels.hip: its dispatch, apply, synthetic data, self-test).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hk {

constexpr int PX = 4;      // pixels per lane per row (one 16-byte load)
constexpr int WAVE = 64;   // gfx950 wavefront

// ---------------------------------------------------------------------------------------------------------------------
// Launch ledger (test aid; include/homonim_hk_devtools.h hk_debug_build_ledger).  Every kernel BUILD of the library -- each
// instantiation of the fused kernel's template, each kernel of the other translation units -- owns one BuildRecord that puts
// itself on a process-wide list when the library is loaded (so a build nothing ever launched is on the list too) and counts
// its launches.  The test-suite reads the list around every test and keeps a ledger of which builds were launched by a test
// that compared its results with the oracle (tests/conftest.py, tests/test_zz_build_ledger.py).  One relaxed atomic add per launch.
struct BuildRecord {
    char name[88];
    unsigned long long launches;  // (atomic adds; read by ledger_text)
    BuildRecord* next;
    explicit BuildRecord(const char* kernel_name);
    // the fused kernel: "fit_apply_kernel<MODEL,R2,RW,DENSE,RING,CERT_ONLY,WPB,BATCH>"
    BuildRecord(int model, bool r2, int rw, bool dense, int ring, bool cert_only, int wpb, bool batch);
    // its list-launch twin: "fit_list_kernel<MODEL,R2,RW,DENSE,RING>"
    BuildRecord(const char* kernel_name, int model, bool r2, int rw, bool dense, int ring);
    void hit() { __atomic_fetch_add(&launches, 1ull, __ATOMIC_RELAXED); }
};
// one record per launch SITE of a kernel outside the fused template: the site's local tag type names the kernel
template <typename Tag>
struct SiteRecord {
    static BuildRecord rec;
};
template <typename Tag>
BuildRecord SiteRecord<Tag>::rec{Tag::name()};
#define HK_LAUNCH(kern, ...)                                               \
    do {                                                                   \
        struct Tag_ {                                                      \
            static const char* name() { return #kern; }                    \
        };                                                                 \
        ::hk::SiteRecord<Tag_>::rec.hit();                                 \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                             \
    } while (0)
// "name\tlaunches\n" per record, '\0'-terminated; returns the bytes needed (incl. the terminator) whatever `len` is
size_t ledger_text(char* buf, size_t len, bool reset);

// Device-side argument block of the fused fit(+apply) kernel.  One wave = one (band, row-segment, column-strip) unit.
// Set in a band's r2-failure counter by the certificate-only build: the count is void, run the band again with
// cert_only = 0.  (Counts themselves are < 2^63.)
constexpr unsigned long long FIT_RETRY_BIT = 1ull << 63;

// One job of a BATCHED launch (FitArgs::jobs, device memory): many device-resident jobs -- the block positions of a mosaic, the
// tiles of a tile list -- run as ONE kernel launch instead of one launch (and one launch tail) each.  A workgroup finds its job
// by a binary search over `first_group` and takes the job's planes, shape and unit grid from here instead of from FitArgs.
struct FitJob {
    const float* src;
    const float* ref;
    float* gain;
    float* offset;
    float* r2;
    float* corr;
    const double* norm;
    unsigned long long* fail_count;
    unsigned char* flag;
    long long stride, band_stride;
    int height, width, n_bands;
    int seg_rows, n_strips, n_segs, seg_rows_tail, n_segs_big;
    int out_y0, out_y1, out_x0, out_x1;
    int first_group[2];  // first workgroup of the job in the launch: [0] one strip per workgroup, [1] HK_WPB_MEM strips (lock-step builds)
    int pad_;
};
static_assert(sizeof(FitJob) % 8 == 0, "FitJob entries are read with scalar loads");

struct FitArgs {
    const FitJob* jobs;     // batched launch: n_jobs entries (device), else NULL -- the fields below then describe the one job
    int n_jobs;
    int batch_groups[2];    // batched launch: workgroups of all jobs, [0] one strip per workgroup, [1] HK_WPB_MEM strips
    const float* src;
    const float* ref;
    float* gain;
    float* offset;
    float* r2;
    float* corr;
    const double* norm;              // n_bands x 2 (gain-blk-offset)
    const float* offset_in;          // closing (in-paint) pass of gain-offset: in-painted offsets, else NULL
    const unsigned char* flag_in;    // ... with the source flags of the in-painting (1 byte per pixel): the build WITHOUT R2 then
                                     // takes the r2-mask decision from them instead of evaluating R2 again; else NULL
    unsigned long long* fail_count;  // n_bands
    unsigned char* flag;             // gain-offset with a threshold: 1 byte per pixel = 1: (r2 > thresh) & (gain > 0) & valid, 0: valid and failing, 2: invalid
                                     // (kernel_model.py:363), the in-painting's source mask; same strides as the planes; or NULL
    int height, width;
    long long stride;       // elements between rows
    long long band_stride;  // elements between planes
    int n_bands;
    int seg_rows;           // output rows per unit
    int n_strips, n_segs;   // units per band = n_strips * n_segs
    int seg_rows_tail;      // rows per unit of the last segments (launched last: they level the end of the launch)
    int n_segs_big;         // segments of seg_rows rows; the remaining n_segs - n_segs_big have seg_rows_tail rows
    int total_units;
    int rh, rw;             // kernel radii (kh = 2*rh+1, kw = 2*rw+1)
    int overlap_lanes;      // lanes per side that only feed neighbours: ceil(rw / PX)
    int src_nd_mode, ref_nd_mode;
    float src_nodata, ref_nodata;
    int has_thresh;
    float r2_thresh;
    int cert_only;          // gain-offset + r2 mask, no R2 plane, fail_count and open_rows set: run the CERTIFICATE build (launch_one),
                            // which settles a wave-row by the two-sided float32 certificate or marks it in `open_rows`
    // Round 6: the wave-rows the certificate build cannot settle: one bit per (band, strip, row), 32 rows per word, word index
    // (band * n_strips + strip) * ceil(height / 32) + row / 32 (zeroed before the certificate launch).
    // list_mode != 0: THIS launch is the list launch -- the complete build on a persistent grid over the runs of marked rows.
    unsigned* open_rows;
    int list_mode;
    float r2_fail_scale;    // kappa of the division-free r2-mask certificate: 1 - r2_pass_scale(), rounded up (hk_api.hip)
    float r2_failcert_scale;  // kappa_f of its mirror image (certain FAILURE; complete build): 1 - r2_fail_above(), rounded down; -inf: none
    double r2_pass_below;   // exact evaluation without R2 output: ssres < r2_pass_below * sstot proves the r2 test true,
    double r2_fail_above;   // ssres > r2_fail_above * sstot proves it false (sstot > 0); in between the division decides
    float n_full;           // kh * kw: the window count of every pixel away from the raster's edges (dense kernels)
    double nd_full;         // the same as float64
    double inv_n_full;      // RN64(1 / (kh * kw)) -- the 1/N table entry (hk_fit_kernel.h)
    int force_general;      // 1: never take the dense (nodata None) specialisation (testing)
    int seg_rows_pref;      // > 0: the build's preferred uniform segment height (hk_api.hip fill_args / fill_grid), 0: the default policy
    int use_ring;           // ring mode of fit_apply_kernel: 1 full LDS ring, 2 centre ring + re-loaded leaving row, 0 re-load both
    int xcd_remap;          // G > 0: blockIdx -> unit remap handing each XCD runs of G consecutive units (0: round-robin)
    int lds_pad;            // unused dynamic LDS bytes per wave: fewer resident waves per CU (hk_api.hip fill_args)
    // store window: only rows [out_y0, out_y1) x columns [out_x0, out_x1) of the job are written / counted (the halo crop of
    // homonim/raster_array.py:478-491 when a block of a larger device raster is processed in place); 0,height,0,width = all.
    // out_x0 and out_x1 are multiples of PX (or out_x1 == width): whole quads are stored.
    int out_y0, out_y1, out_x0, out_x1;
};

// model: 0 gain, 1 gain-blk-offset, 2 gain-offset.  with_r2: compute the R2 quantity set.
hipError_t launch_fit_apply(const FitArgs& a, int model, bool with_r2, hipStream_t stream);
// ... which dispatches to one translation unit per (model, with_r2) (hk_fit_tu.hip)
hipError_t launch_fit_m0_r0(const FitArgs& a, hipStream_t stream);
hipError_t launch_fit_m0_r1(const FitArgs& a, hipStream_t stream);
hipError_t launch_fit_m1_r0(const FitArgs& a, hipStream_t stream);
hipError_t launch_fit_m1_r1(const FitArgs& a, hipStream_t stream);
hipError_t launch_fit_m2_r0(const FitArgs& a, hipStream_t stream);
hipError_t launch_fit_m2_r1(const FitArgs& a, hipStream_t stream);
hipError_t read_stamps_m0_r0(unsigned long long* acc16, bool reset);
hipError_t read_stamps_m0_r1(unsigned long long* acc16, bool reset);
hipError_t read_stamps_m1_r0(unsigned long long* acc16, bool reset);
hipError_t read_stamps_m1_r1(unsigned long long* acc16, bool reset);
hipError_t read_stamps_m2_r0(unsigned long long* acc16, bool reset);
hipError_t read_stamps_m2_r1(unsigned long long* acc16, bool reset);
// Which builds of the fused kernel exist with the job-table look-up of a batched launch (FitArgs::jobs): gain-blk-offset without
// R2 -- the fused RasterFuse path of a block-partitioned mosaic, where one launch for all blocks pays (profiles/r03_batch.txt).
// The batched entry points run every other model as one launch per job (bit-identical either way).
constexpr bool fit_batch_build(int model, bool with_r2) { return model == 1 && !with_r2; }
bool fit_batch_supported(int model, bool with_r2);
// stage stamps of a -DHK_STAMPS build of the fused kernel (all zero otherwise): 16 counters, optionally cleared after reading
hipError_t read_stamps(unsigned long long* out16, bool reset);
// strips per workgroup of the lock-step builds (HK_WPB_MEM): FitJob::first_group[1] / FitArgs::batch_groups[1] count those
int fit_lockstep_waves();
// LDS bytes one wave needs (its row ring; hk_fit_kernel.h)
size_t fit_lds_bytes(int kh, int ring_mode, bool ahead);
// lanes per side that overlap with the neighbouring strip for kernel half-width rw
inline int overlap_lanes_for(int rw) { return (rw + PX - 1) / PX; }

hipError_t launch_apply(const float* src, const float* gain, const float* offset, float* out, int height, int width,
                        long long stride, hipStream_t stream);

// Block statistics for gain-blk-offset (kernel_model.py:216-229), per band.
// One plane of a BATCHED statistics launch (NormArgs::planes, device memory): plane p of the launch = (job, band)
struct NormPlane {
    const float* src;
    const float* ref;
    long long stride;
    int height, width;
};
// Register rows of the split ring (ring mode 3 of the fused kernel) a build holds = the largest kernel half-height it serves.
#ifndef HK_SRING_MAX
#define HK_SRING_MAX 7
#endif
#ifndef HK_SRING_MAX_BLK
#define HK_SRING_MAX_BLK 5
#endif
__host__ __device__ constexpr int split_ring_rows(int model) { return model == 1 ? HK_SRING_MAX_BLK : HK_SRING_MAX; }

struct NormArgs {
    const NormPlane* planes = nullptr;  // batched launch: n_bands entries (device), else NULL -- the planes are then src/ref + band * band_stride
    const float* src;
    const float* ref;
    int height, width;
    long long stride, band_stride;
    int n_bands;
    int src_nd_mode, ref_nd_mode;
    float src_nodata, ref_nodata;
    // batched launch: x-extent of the streaming pass's grid = the largest norm_pass_waves() over the planes (a plane with fewer
    // PIXELS than the largest one can still have more 1 KB chunks: width 1025 has 5 per row, width 1024 has 4); 0 = this shape's
    int grid_waves = 0;
};
// waves per plane of the streaming pass (a function of the plane's shape only)
int norm_pass_waves(int height, int width);
// workspace: see norm_workspace_bytes(); norm_out: n_bands x 2 float64 on device.
size_t norm_workspace_bytes(int n_bands, int height, int width);
// batched: a.planes set, a.n_bands planes, a.height x a.width = the LARGEST plane (sizes the workspace and the grid)
hipError_t launch_block_norm(const NormArgs& a, void* workspace, double* norm_out, hipStream_t stream);
// The same statistics for a block whose rows are spread over several ranks: six phases on this rank's slab, the caller
// all-reduces (SUM) the float64 exchange buffer (norm_split_exchange_doubles() values, device) between them.
size_t norm_split_exchange_doubles(int n_bands);
hipError_t launch_block_norm_split(const NormArgs& a, void* workspace, double* xchg, double inv_world, int phase,
                                   double* norm_out, hipStream_t stream);

// Masked comparison sums of homonim/compare.py:243-255 (hk_compare.hip), per band:
// sums_out[band * 7 + k] = [sum s, sum r, sum s^2, sum r^2, sum s*r, sum (r - s)^2, count] over jointly valid pixels.
struct CompareArgs {
    const float* src;
    const float* ref;
    int height, width;
    long long src_stride, ref_stride;            // elements between rows
    long long src_band_stride, ref_band_stride;  // elements between planes
    int n_bands;
    int src_nd_mode, ref_nd_mode;
    float src_nodata, ref_nodata;
    int vec_ok;                                  // set by the launcher: 16-byte row loads are legal
};
size_t compare_workspace_bytes(int n_bands);
hipError_t launch_compare_sums(const CompareArgs& a, void* workspace, double* sums_out, hipStream_t stream);

hipError_t launch_synth_fill(float* src, float* ref, int n_bands, int height, int width, long long stride,
                             long long band_stride, unsigned long long seed, int nodata_variant, hipStream_t stream);

// Typed IO (hk_convert.hip).  dtype codes = hk_dtype of include/homonim_hk.h: 0 f32, 1 u8, 2 u16, 3 i16, 4 u32, 5 i32, 6 f64.
hipError_t launch_cast_in(int dtype, const void* in, long long in_stride, float* out, long long out_stride, int height,
                          int width, hipStream_t stream);
hipError_t launch_cast_out(int dtype, const float* in, long long in_stride, void* out, long long out_stride, int height,
                           int width, int has_nodata, double nodata, hipStream_t stream);
inline int dtype_size(int dtype) { const int sz[7] = {4, 1, 2, 2, 4, 4, 8}; return dtype >= 0 && dtype < 7 ? sz[dtype] : 0; }

// mask_partial on a shared grid (hk_mask.hip): full-coverage mask from valid(in) & params, eroded by (kh+2) x (kw+2);
// writes masked parameters and/or the corrected block and/or the uint8 mask.  rowcnt_ws: height x stride uint16.
hipError_t launch_partial_mask(const float* in, int nd_mode, float nodata, const float* params, int n_bands,
                               long long band_stride, const float* src, int height, int width, long long stride, int kh,
                               int kw, unsigned short* rowcnt_ws, float* params_out, float* corr_out,
                               unsigned char* mask_out, hipStream_t stream);

// In-painting of the offset band (hk_inpaint.hip; GDALFillNodata restated): sources are pixels with r2 > thresh and
// gain > 0 (flag 1), flag 0 pixels are targets and are filled IN PLACE (a filled pixel never acts as a source: sources are read
// where the flag is 1 only), any other flag value is neither.  workspace: inpaint_workspace_bytes().
size_t inpaint_workspace_bytes(int height, long long stride);
// `flag_ready` (nullable): source flags already written by the fit kernel (FitArgs::flag; 1 byte per pixel, row stride
// `stride`) -- gain / r2 are then not read.  inpaint_flag_plane(): the workspace's own flag plane, for a fit to write into.
// `n_targets`: the number of failing pixels if the caller knows it (0 = unknown): picks the order in which a tile's targets are searched.
unsigned char* inpaint_flag_plane(void* workspace, int height, long long stride);
hipError_t launch_inpaint_offsets(float* offset, const float* gain, const float* r2, float thresh, long long stride,
                                  int height, int width, void* workspace, hipStream_t stream,
                                  const unsigned char* flag_ready = nullptr, unsigned long long n_targets = 0);

// Re-sampling between axis-aligned grids (hk_resample.hip).  mode = rasterio.enums.Resampling value (0, 1, 3, 5).
hipError_t launch_resample(int mode, const float* src, long long src_stride, long long src_band_stride, int sh, int sw,
                           int n_bands, int nd_mode, float nodata, double kx, double ox, double ky, double oy, float* dst,
                           long long dst_stride, long long dst_band_stride, int dh, int dw, float dst_fill,
                           hipStream_t stream);

size_t upsample_apply_workspace_bytes(int height);
hipError_t launch_upsample_apply(int mode, const float* src, long long src_stride, int nd_mode, float nodata,
                                 const float* gain, const float* offset, long long par_stride, int ph, int pw,
                                 const float* keep, long long keep_stride, float* out, long long out_stride, int height,
                                 int width, double kx, double ox, double ky, double oy, void* workspace, hipStream_t stream);
hipError_t launch_valid_plane(const float* in, long long in_stride, int nd_mode, float nodata, float* out,
                              long long out_stride, int height, int width, hipStream_t stream);
hipError_t launch_apply_space(const float* src, long long src_stride, int nd_mode, float nodata, const float* gain,
                              const float* offset, long long par_stride, const float* keep, float* out,
                              long long out_stride, int height, int width, hipStream_t stream);

// flat 2-read 1-write float4 stream (measurement aid: what the HBM gives the fused kernel's byte mix without its stencil)
hipError_t launch_stream_probe(const void* a, const void* b, void* out, size_t n_bytes, hipStream_t stream);

// sum of the bit patterns of a height x width window of a float32 plane, modulo 2^64, ADDED to *out_dev (test / bench aid)
hipError_t launch_checksum(const float* plane, long long stride, int height, int width, unsigned long long* out_dev, hipStream_t stream);

// returns 0 on pass; writes a diagnostic code otherwise
hipError_t launch_selftest(int* result_dev, hipStream_t stream);

}  // namespace hk
