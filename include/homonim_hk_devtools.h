/*
 * homonim_hk_devtools.h -- measurement and test aids exported by libhomonim_hk.so beside the drop-in boundary.
 *
 * Nothing here replaces a reference interface: these entry points exist for bench.py, the test-suite and the tools/ scripts
 * (synthetic workload on the device, a flat copy stream to price the HBM, kernel stage counters, the constants of the R2
 * certificate for checking them against their derivation).  A consumer of the hot path needs homonim_hk.h only.
 *
 * Diagnostic switches of the library (environment, read once):
 *   HK_FAULT_REPORT=1   the first hk_ctx_create adds a system-event callback to the HSA runtime (process-wide, never removed)
 *                       that prints GPU memory faults with the place of the address relative to the library's device
 *                       allocations, memory errors and hardware exceptions to stderr; it does not claim the event, the
 *                       runtime's own handling (ending the process) follows.  Off by default.
 *   HK_GUARD_ALLOC=lo|hi|poison   every device allocation of the library between unmapped address ranges (lo / hi: flush with
 *                       the lower / upper end), or plain allocations filled with 0xAB: an out-of-range access of a kernel
 *                       faults at its launch (tests/test_gpu_guard_alloc.py).
 *   HK_STAGE_CHUNK_KB   size of the pinned staging chunks of the host-pointer calls (default 8192; tests of the chunked paths).
 *   HK_ASSERT_PINNED=1  a caller array is copied directly (not through the staging ring) only when ONE range page-locked through
 *                       hk_host_alloc / hk_host_register holds all of it; with this switch the library also asks the HIP runtime
 *                       about every page of every row of such an array before the copy is queued and fails the call
 *                       (HK_ERR_ARG) if one is not page-locked host memory.  The test-suite runs with it.
 */
#ifndef HOMONIM_HK_DEVTOOLS_H
#define HOMONIM_HK_DEVTOOLS_H

#include "homonim_hk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Fill device planes with the synthetic workload of SURVEY.md section 8(d) (src ~ U[0.05,1), ref = g*src+o+noise);
 * nodata_variant 0: none, 1: 3-px NaN frame + 0.1 % NaN holes, 2: frame only, 3 / 4: none, noisy reference (35 % / 85 % r2-mask failures), 5: low-entropy data (64 source levels, exactly affine reference:
 * the same instruction stream at a lower energy per launch), 6: 3-px NaN frame + ~1 % of the area in round NaN blobs 32 - 128 px across, source and reference
 * independently (cloud / shadow-mask-like).  Test/bench data only. */
int hk_synth_fill_dev(hk_ctx* ctx, float* src, float* ref, int32_t n_bands, int32_t height, int32_t width,
                      int64_t stride, int64_t band_stride, uint64_t seed, int32_t nodata_variant, int32_t stream);

/* Measurement aid (bench.py `roofline.copy_gbps_measured`): ONE launch of a flat float4 stream over three device buffers
 * of n_bytes each -- out[i] = a[i] + b[i], two reads + one write like the fused kernel's 12 bytes per pixel, no stencil,
 * persistent grid, four 16-byte non-temporal loads in flight per lane and array -- on pooled stream `stream`
 * (asynchronous; time it with hk_event_*).  n_bytes must be a multiple of 16.  What this box's HBM gives that byte mix. */
int hk_stream_probe_dev(hk_ctx* ctx, const void* a, const void* b, void* out, size_t n_bytes, int32_t stream);

/* The constants the kernels decide `(r2 > thresh) & (gain > 0)` (kernel_model.py:363) with, for checking them against their
 * derivation (host-only, no device call; PROOFS.md appendix A): ssres < pass_below * sstot proves the decision true,
 * ssres > fail_above * sstot proves it false (sstot > 0); kappa / kappa_fail are the float32 factors of the division-free
 * certificate and of its fail side (+inf / -inf: nothing can be certified).  Any pointer may be NULL. */
int hk_r2_certificate_constants(float thresh, double* pass_below, double* fail_above, float* kappa, float* kappa_fail);

/* Measurement aid: the stage counters a -DHK_STAMPS build of the fused kernel accumulates (shader-clock cycles per stage of a row
 * iteration, [14] waves, [15] iterations; all zero in the shipped build; tools/stage_stamps.py).  Synchronises the device. */
int hk_debug_stage_stamps(hk_ctx* ctx, uint64_t out[16], int32_t reset);

/* Test aid: how the host-pointer entry points moved caller memory since the process started (or the last reset), all contexts:
 * out[0] = copies queued straight from / to caller arrays (page-locked through this library), out[1] = chunks that went through
 * the pinned staging ring. */
int hk_debug_staging_counters(uint64_t out[2], int32_t reset);

/* Test / bench aid: the exact, order-free checksum of a height x width window of a device-resident float32 plane (rows `stride`
 * elements apart): the sum of the pixels' 32-bit patterns modulo 2^64.  The union of N ranks' shards of a raster is compared with
 * the single-rank result through it (bench.py `shard_checksum`) without moving the rasters to the host.  Queued on pooled stream
 * `stream`, which is synchronised before the call returns. */
int hk_debug_checksum_dev(hk_ctx* ctx, const float* plane, int64_t stride, int32_t height, int32_t width, int32_t stream,
                          uint64_t* sum_out);

/* Test aid, fault injection: while on != 0, hk_fit / hk_fit_apply* on host pointers fail right behind their queued result copies --
 * where a HIP error would leave them -- so that the abandonment of those copies can be observed (no later call may write the failed
 * call's output arrays; no copy may still be in flight on them when the call returns).  Process-wide; switch it off again. */
int hk_debug_fail_after_d2h(int32_t on);

/* Test aid: the launch ledger.  Every kernel BUILD of the library -- each instantiation of the fused kernel's template
 * ("fit_apply_kernel<MODEL,R2,RW,DENSE,RING,CERT_ONLY,WPB,BATCH>"), each kernel of the statistics / in-painting / re-sampling /
 * mask / conversion / comparison units (one record per launch site, named after the kernel) -- is on a process-wide list from the
 * moment the library is loaded and counts its launches.  Writes "name<TAB>launches<NEWLINE>" per record into buf (NUL-terminated);
 * *needed = the bytes that takes (call with buf == NULL to size the buffer); reset != 0 clears the counts behind the reading.  No
 * device call.  tests/conftest.py reads it around every GPU test; tests/test_zz_build_ledger.py fails on a build that no test
 * comparing with the oracle ever launched. */
int hk_debug_build_ledger(char* buf, size_t len, size_t* needed, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* HOMONIM_HK_DEVTOOLS_H */
