// hk_fit_kernel.h -- the fused fit(+apply) kernel template of the homonim kernel-model hot path (gfx950 / CDNA4) and its launch
// ladder.  Included by hk_fit_tu.hip (one translation unit per MODEL x R2, compiled in parallel by homonim_amd/build.py) and by
// hk_kernels.hip (dispatch, self-test; with -DHK_FIT_ONE_TU every instantiation -- the A/B tooling under tools/).
//
// Reference behaviour being reproduced: homonim/kernel_model.py (v0.4.3)
//   _fit_gain :231-274, _fit_gain_blk_offset :276-303, _fit_gain_offset :305-373, _r2_array :142-214, apply :442-463,
// with OpenCV's boxFilter/sqrBoxFilter (zero border, un-normalised, float64 accumulation; sqrBoxFilter RETURNS
// float64) folded in.  See DESIGN.md "Numerics contract" for the exact expression order this file mirrors.
//
// Design (one wave = one unit, no workgroup barriers, no MFMA -- this is an HBM/VALU-bound stencil):
//   * a wave owns a column strip of 64 lanes x 4 px (one 16-byte load per lane per row per input) and marches down
//     a row segment; overlap lanes at both strip edges only feed their neighbours, so waves never talk to each other;
//   * vertical window sums are running float64 column sums (add entering row, subtract leaving row -- OpenCV's own
//     ColumnSum order); the leaving row is re-read from a wave-private LDS ring of kh raw rows (36 B per lane-row);
//   * horizontal window sums combine per-lane prefix/suffix partial sums with neighbours' through DPP wave shifts
//     (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- no LDS traffic, no barriers;
//   * the 2x2 normal-equation solve, R2, the r2-mask test and gain*src+offset run in registers on the four pixels,
//     in the reference's float32/float64 operation order (compiled with -ffp-contract=off; *_rn intrinsics);
//   * each input byte is read from HBM once (+ (64/(64-2*ol)) x (seg+2rh)/seg halo), each output written once.
#pragma once
#include "hk_kernels.h"

#include <type_traits>

namespace hk {

// In-kernel stage stamps (build with -DHK_STAMPS; tools/stage_stamps.py): every wave accumulates the shader-clock cycles it spends
// between the marked points of a row iteration (s_memtime; waits for memory land in the stage that needs the data) and adds them
// to hk_stamps[] when it ends; [15] counts iterations, [14] waves.  Costs ~10 % of the kernel's time; never in the shipped build.
// (one counter set per translation unit: read_stamps() of hk_kernels.hip adds them up)
static __device__ unsigned long long hk_stamps[16];
#ifdef HK_STAMPS
#define HK_STAMP(k)                                                       \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        st_acc[k] += now_ - st_prev;                                      \
        st_prev = now_;                                                   \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#else
#define HK_STAMP(k) do { } while (0)
#endif
// adds this translation unit's counters to acc16[]
static inline hipError_t read_stamps_tu(unsigned long long* acc16, bool reset) {
    unsigned long long v[16];
    hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(hk_stamps), sizeof(v));
    if (e != hipSuccess) return e;
    for (int k = 0; k < 16; ++k) acc16[k] += v[k];
    if (reset) {
        const unsigned long long zero[16] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(hk_stamps), zero, sizeof(zero));
    }
    return e;
}

// Measurement hook (tools/README.md, FLOOR.md section 3): timing builds with one ingredient of the fused kernel taken out --
// WRONG results, never shipped.  1: the leaving row is re-loaded from the entering row's address (no far re-load),
// 2: gain-blk-offset without its float64 quotient, 4: no horizontal sums, 8: no corrected-plane stores.
#ifndef HK_ABLATE
#define HK_ABLATE 0
#endif
#ifndef HK_PACKED_NSUM
#define HK_PACKED_NSUM 1  // window counts of the narrow kernels summed as packed bytes (fit_apply_kernel)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// cross-lane primitives
// bound_ctrl:1 makes lanes without a source read 0, so no `old` operand has to be materialised per shift.
__device__ __forceinline__ int dpp_from_left(int v) {  // value of lane-1; lane 0 receives 0
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ int dpp_from_right(int v) {  // value of lane+1; lane 63 receives 0
    return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
// float32: the compiler folds the shift into the consuming v_add_f32 (DPP operand modifier), no separate move
__device__ __forceinline__ float dpp_from_left(float v) { return __int_as_float(dpp_from_left(__float_as_int(v))); }
__device__ __forceinline__ float dpp_from_right(float v) { return __int_as_float(dpp_from_right(__float_as_int(v))); }
__device__ __forceinline__ double dpp_from_left(double v) {
    int lo = dpp_from_left(__double2loint(v)), hi = dpp_from_left(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_right(double v) {
    int lo = dpp_from_right(__double2loint(v)), hi = dpp_from_right(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Which part of the neighbour lane `j` lanes to the left/right falls inside the window of output pixel i (0..3):
// 0 = nothing, 1..3 = suffix/prefix of that length, 4 = the whole lane.
__host__ __device__ constexpr int left_len(int rw, int i, int j) {
    int lo = i - rw, first = -PX * j;
    return lo <= first ? PX : (lo <= first + PX - 1 ? PX - (lo - first) : 0);
}
__host__ __device__ constexpr int right_len(int rw, int i, int j) {
    int hi = i + rw, first = PX * j;
    return hi >= first + PX - 1 ? PX : (hi >= first ? hi - first + 1 : 0);
}
__host__ __device__ constexpr bool need_left_from(int rw, int j0, int len, int ol) {
    for (int j = j0; j <= ol; ++j)
        for (int i = 0; i < PX; ++i)
            if (left_len(rw, i, j) == len) return true;
    return false;
}
__host__ __device__ constexpr bool need_right_from(int rw, int j0, int len, int ol) {
    for (int j = j0; j <= ol; ++j)
        for (int i = 0; i < PX; ++i)
            if (right_len(rw, i, j) == len) return true;
    return false;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// value held by the lane `dist` lanes away (dist > 0: to the left) through the LDS crossbar; lanes without a source read
// an unspecified lane -- only ever used where those lanes are overlap lanes whose results are discarded
template <typename T>
__device__ __forceinline__ T bperm_from(T v, int src_lane);
template <>
__device__ __forceinline__ int bperm_from<int>(int v, int src_lane) {
    return __builtin_amdgcn_ds_bpermute((src_lane & (WAVE - 1)) << 2, v);
}
template <>
__device__ __forceinline__ float bperm_from<float>(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((src_lane & (WAVE - 1)) << 2, __float_as_int(v)));
}
template <>
__device__ __forceinline__ double bperm_from<double>(double v, int src_lane) {
    const int a = (src_lane & (WAVE - 1)) << 2;
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(a, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(a, __double2loint(v)));
}

__host__ __device__ constexpr bool need_left_at(int rw, int j, int len) {
    for (int i = 0; i < PX; ++i)
        if (left_len(rw, i, j) == len) return true;
    return false;
}
__host__ __device__ constexpr bool need_right_at(int rw, int j, int len) {
    for (int i = 0; i < PX; ++i)
        if (right_len(rw, i, j) == len) return true;
    return false;
}
// true when all four outputs take the WHOLE of the lane j to the left and to the right
__host__ __device__ constexpr bool lane_full_for_all(int rw, int j) {
    for (int i = 0; i < PX; ++i)
        if (left_len(rw, i, j) != PX || right_len(rw, i, j) != PX) return false;
    return true;
}

// Has output i already received a non-common term before (j, side) in the order j = 1.., left then right?
__host__ __device__ constexpr bool specific_before(int rw, int i, int j, bool right_side, bool own_full) {
    if (!own_full) return true;  // H[i] starts from the lane's own partial sum
    for (int jj = 1; jj <= j; ++jj) {
        if (lane_full_for_all(rw, jj)) continue;
        const bool l = left_len(rw, i, jj) > 0, r = right_len(rw, i, jj) > 0;
        if (jj < j ? (l || r) : (right_side && l)) return true;
    }
    return false;
}

// Horizontal window sums of half-width RW over the wave's 256 columns: lane holds V[0..3] (its 4 columns), receives
// H[i] = sum of columns [i-RW, i+RW].  Everything below is resolved at compile time into straight-line code:
//   * per lane 5 adds for the prefix / suffix partial sums of its own columns;
//   * neighbours at distance 1 arrive through DPP wave shifts (VALU), neighbours farther away through ds_bpermute (LDS
//     crossbar, one instruction per dword whatever the distance);
//   * lanes that lie wholly inside all four windows are summed once into a common term.
// RW = 2: 4 shifted values (8 DPP moves) + 9 adds per 4 pixels; RW = 7: 2 DPP-shifted + 6 permuted values + 13 adds.
// XCH (float64 quantities of the kernels whose strips overlap by one lane): the distance-1 neighbours' partial sums travel
// through a wave-private LDS exchange line instead of DPP moves -- the kernel is bound by VALU issue (a 64-bit value costs two
// 4-cycle v_mov_b32_dpp), while the LDS pipe has room: per quantity and direction one ds_write_b128 + one ds_read_b128 of the
// neighbour's slot.  `xch` = this lane's 16-byte slot; slots -1 and 64 exist (never written: lanes 0 and 63 are overlap lanes
// whose sums are discarded).  LDS operations of one wave execute in order, so the line is re-used without waiting.
__device__ __forceinline__ void xch_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
typedef double hk_d2 __attribute__((ext_vector_type(2)));
// DPP2: the neighbours at distance 2 through two chained wave shifts (four v_mov_b32_dpp per float64 value) instead of the LDS
// crossbar (two ds_bpermute_b32).  One ds_bpermute costs the CU 2.6 ns whichever SIMD issued it, a DPP move 0.6 ns of one SIMD's
// issue (profiles/r05_ubench_xlane.txt): the builds whose five float64 quantities make them crossbar-bound gain, the VALU-bound ones lose.
#ifndef HK_DPP2
#define HK_DPP2 1
#endif
template <int RW, typename T, bool XCH = false, bool DPP2 = false>
__device__ __forceinline__ void hsum(const T (&V)[PX], T (&H)[PX], int lane, [[maybe_unused]] char* xch = nullptr) {
    if constexpr (RW == 0) {
#pragma unroll
        for (int i = 0; i < PX; ++i) H[i] = V[i];
    } else {
        constexpr int OL = (RW + PX - 1) / PX;
        T pre[PX + 1], suf[PX + 1];  // pre[k] = V[0..k-1], suf[k] = V[4-k..3]
        pre[1] = V[0];
        pre[2] = V[0] + V[1];
        pre[3] = pre[2] + V[2];
        suf[1] = V[3];
        suf[2] = V[2] + V[3];
        suf[3] = V[1] + suf[2];
        pre[4] = suf[4] = pre[2] + suf[2];
        constexpr bool OWN_FULL = RW >= PX - 1;  // every output's window holds the lane's own four columns
        T common = pre[PX];
        static_for<0, PX>([&](auto I) {
            constexpr int i = decltype(I)::value, lo = i - RW, hi = i + RW;
            if constexpr (!OWN_FULL) {
                if constexpr (lo <= 0 && hi >= PX - 1)
                    H[i] = pre[PX];
                else if constexpr (lo <= 0)
                    H[i] = pre[hi + 1];
                else
                    H[i] = suf[PX - lo];
            }
        });
        static_for<1, OL + 1>([&](auto J) {
            constexpr int j = decltype(J)::value;
            T ls[PX + 1], rp[PX + 1];
            if constexpr (XCH && j == 1 && std::is_same<T, double>::value && (RW == 1 || RW == 2)) {
                // needed from the left: suf[1..RW], from the right: pre[1..RW]
                if constexpr (RW == 2) {
                    *reinterpret_cast<hk_d2*>(xch) = hk_d2{(double)suf[1], (double)suf[2]};
                    xch_order();
                    const hk_d2 l = *reinterpret_cast<const hk_d2*>(xch - 16);
                    xch_order();
                    *reinterpret_cast<hk_d2*>(xch) = hk_d2{(double)pre[1], (double)pre[2]};
                    xch_order();
                    const hk_d2 r = *reinterpret_cast<const hk_d2*>(xch + 16);
                    xch_order();
                    ls[1] = (T)l.x, ls[2] = (T)l.y, rp[1] = (T)r.x, rp[2] = (T)r.y;
                } else {
                    *reinterpret_cast<hk_d2*>(xch) = hk_d2{(double)suf[1], (double)pre[1]};
                    xch_order();
                    ls[1] = (T)*reinterpret_cast<const double*>(xch - 16);
                    rp[1] = (T)*reinterpret_cast<const double*>(xch + 16 + 8);
                    xch_order();
                }
            } else
            static_for<1, PX + 1>([&](auto K) {
                constexpr int k = decltype(K)::value;
                if constexpr (need_left_at(RW, j, k)) {
                    if constexpr (j == 1) ls[k] = dpp_from_left(suf[k]);
                    else if constexpr (j == 2 && DPP2) ls[k] = dpp_from_left(dpp_from_left(suf[k]));
                    else ls[k] = bperm_from<T>(suf[k], lane - j);
                }
                if constexpr (need_right_at(RW, j, k)) {
                    if constexpr (j == 1) rp[k] = dpp_from_right(pre[k]);
                    else if constexpr (j == 2 && DPP2) rp[k] = dpp_from_right(dpp_from_right(pre[k]));
                    else rp[k] = bperm_from<T>(pre[k], lane + j);
                }
            });
            if constexpr (OWN_FULL && lane_full_for_all(RW, j)) {
                common = common + ls[PX];
                common = common + rp[PX];
            } else {
                static_for<0, PX>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    constexpr int ll = left_len(RW, i, j), rl = right_len(RW, i, j);
                    if constexpr (ll > 0) {
                        if constexpr (specific_before(RW, i, j, false, OWN_FULL)) H[i] = H[i] + ls[ll];
                        else H[i] = ls[ll];
                    }
                    if constexpr (rl > 0) {
                        if constexpr (specific_before(RW, i, j, true, OWN_FULL)) H[i] = H[i] + rp[rl];
                        else H[i] = rp[rl];
                    }
                });
            }
        });
        if constexpr (OWN_FULL) {
            static_for<0, PX>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (specific_before(RW, i, OL + 1, false, true)) H[i] = common + H[i];
                else H[i] = common;
            });
        }
    }
}

// Kernels wider than 15 (half-width rw = 4 F + E, F >= 2): the build knows E = rw mod 4 at compile time, F is a launch argument.
// Seen from output pixel i of lane L, the lanes L -+ 1 .. L -+ (F - 1) lie wholly inside the window whatever F is, and the lanes
// L -+ F and L -+ (F + 1) are cut exactly as a window of half-width 8 + E cuts the lanes at distance 2 and 3 -- so the
// compile-time tables of hsum<8 + E> (which partial sum of which lane goes to which output) serve every F, and only the lane
// distances are run-time: four ds_bpermute addresses computed once per wave (WideLanes) and a wave-uniform loop over the whole
// lanes beyond the DPP neighbours (no trip for kernels up to 23 wide, one up to 31).  Per float64 quantity at 31 wide: 4 DPP
// moves + 20 ds_bpermute + 21 adds, nothing decided per lane.  (Rounds 1-4 ran these widths through a loop over every
// (lane distance, partial length) pair with two shuffles and a select per output each: 5-10 x the time per pixel of 15 wide.)
struct WideLanes {
    int f;           // F = rw / 4 (wave-uniform)
    int lf1, rf;     // ds_bpermute byte addresses of the lane F + 1 to the left / F to the right of this one; the lanes F to the
                     // left / F + 1 to the right are these + 4 (the instruction's offset field: two address registers, not four;
                     // where the sum runs past lane 63 the reader is an overlap lane)
};
__device__ __forceinline__ WideLanes make_wide_lanes(int rw, int lane) {
    WideLanes w;
    w.f = rw / PX;
    w.lf1 = ((lane - w.f - 1) & (WAVE - 1)) << 2, w.rf = ((lane + w.f) & (WAVE - 1)) << 2;
    return w;
}
// value at a precomputed ds_bpermute address (lanes without a source read some other lane: overlap lanes only, see bperm_from)
__device__ __forceinline__ int bperm_at(int v, int addr) { return __builtin_amdgcn_ds_bpermute(addr, v); }
__device__ __forceinline__ float bperm_at(float v, int addr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
__device__ __forceinline__ double bperm_at(double v, int addr) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}
// The WHOLE lanes of a wide window: those at distance 1 .. F - 1, and the lanes F too where the window takes them whole for all four
// outputs (rw mod 4 = 3: 23, 31, 39 ... wide).  W = the farthest whole lane.  For ODD W >= 3 (31 wide; 33 - 37 wide; ...) they can be
// summed as PAIRS (round 6): A = T + T(right neighbour) is one DPP move per word away, and the window of 2 W + 1 lanes is W pairs --
// the two next to the lane a DPP move away from A -- and the single lane at its right end: W - 1 values through the crossbar
// instead of 2 W - 2 (31 wide: 16 ds_bpermute instead of 20 per float64 quantity).  31 wide 6.55 -> 6.28 ms, on NaN-nodata rasters
// 7.05 -> 6.77, 33 - 37 wide -7 ... -8 %; even W gains nothing (the pairs save little and A delays the first fetch), and with the
// decision taken inside the row loop the widths that do not pair ran 2 - 3 % slower (profiles/r06b_ab_pairs.txt): the paired
// form is a BUILD of its own (RW = -5 - E instead of -1 - E; the centre-ring builds of every model, launch_wide).  As builds of
// their own (profiles/r06b_ab_pair_builds.txt): gain-offset + r2 mask 31 wide 6.28 -> 5.88 ms, on NaN-nodata rasters 6.89 -> 6.41,
// 33 / 35 wide 6.92 / 7.72 -> 6.25 / 6.79; without the r2 mask 31 wide 5.65 -> 5.19; gain-blk-offset 31 wide -3 %, gain 35 wide -6 %.
__host__ __device__ constexpr int wide_e(int rw_code) { return rw_code >= -PX ? -1 - rw_code : -1 - PX - rw_code; }  // rw mod 4 of a wide build
__host__ __device__ constexpr bool wide_paired(int rw_code) { return rw_code < -PX; }
__host__ __device__ constexpr int wide_whole_lanes(int e, int f) { return lane_full_for_all(2 * PX + e, 2) ? f : f - 1; }
__host__ __device__ constexpr bool wide_pairs(int e, int f) { return wide_whole_lanes(e, f) >= 3 && (wide_whole_lanes(e, f) & 1); }
template <int E, typename T, bool PAIR = false>
__device__ __forceinline__ void hsum_wide(const T (&V)[PX], T (&H)[PX], const WideLanes& wl, int lane) {
    constexpr int RV = 2 * PX + E;  // the virtual half-width whose lanes 2 and 3 stand for the lanes F and F + 1
    T pre[PX + 1], suf[PX + 1];     // pre[k] = V[0..k-1], suf[k] = V[4-k..3]
    pre[1] = V[0];
    pre[2] = V[0] + V[1];
    pre[3] = pre[2] + V[2];
    suf[1] = V[3];
    suf[2] = V[2] + V[3];
    suf[3] = V[1] + suf[2];
    pre[4] = suf[4] = pre[2] + suf[2];
    // the lane's own four columns and the whole lanes (PAIR: see above; the launch made sure that W is odd and >= 3)
    constexpr bool FULL2 = lane_full_for_all(RV, 2);
    T common = pre[PX];
    if constexpr (PAIR) {
        const int W = FULL2 ? wl.f : wl.f - 1;  // wave-uniform
        // pairs (o, o + 1) for o = -W, ..., -3, [-1], [1], 3, ..., W - 2 (in brackets: a DPP move away from A)
        const T A = pre[PX] + dpp_from_right(pre[PX]);
        common = bperm_from<T>(pre[PX], lane + W);
        common = common + dpp_from_left(A);
        common = common + dpp_from_right(A);
        for (int o = 3; o <= W - 2; o += 2) common = common + bperm_from<T>(A, lane + o);  // (wave-uniform trip counts)
        for (int o = 3; o <= W; o += 2) common = common + bperm_from<T>(A, lane - o);
    } else {
        if (wl.f >= 2) {  // wave-uniform (F = 1: a tall kernel 9 - 15 wide on the everything-re-loaded path)
            common = common + dpp_from_left(pre[PX]);
            common = common + dpp_from_right(pre[PX]);
        }
        for (int j = 2; j < wl.f; ++j) {  // wave-uniform
            common = common + bperm_from<T>(pre[PX], lane - j);
            common = common + bperm_from<T>(pre[PX], lane + j);
        }
    }
    static_for<2, 4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        T ls[PX + 1], rp[PX + 1];
        static_for<1, PX + 1>([&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (PAIR && j == 2 && k == PX && FULL2) {
                // (the lanes F as a whole: part of the pairs above)
            } else {
                if constexpr (need_left_at(RV, j, k)) ls[k] = bperm_at(suf[k], j == 2 ? wl.lf1 + 4 : wl.lf1);
                if constexpr (need_right_at(RV, j, k)) rp[k] = bperm_at(pre[k], j == 2 ? wl.rf : wl.rf + 4);
            }
        });
        if constexpr (lane_full_for_all(RV, j)) {
            if constexpr (!PAIR) {
                common = common + ls[PX];
                common = common + rp[PX];
            }
        } else {
            static_for<0, PX>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int ll = left_len(RV, i, j), rl = right_len(RV, i, j);
                if constexpr (ll > 0) {
                    if constexpr (specific_before(RV, i, j, false, true)) H[i] = H[i] + ls[ll];
                    else H[i] = ls[ll];
                }
                if constexpr (rl > 0) {
                    if constexpr (specific_before(RV, i, j, true, true)) H[i] = H[i] + rp[rl];
                    else H[i] = rp[rl];
                }
            });
        }
    });
    static_for<0, PX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (specific_before(RV, i, 4, false, true)) H[i] = common + H[i];
        else H[i] = common;
    });
}

// The float64 quantities of the kernels wider than 15 exchange their partial sums through wave-private LDS LINES instead of
// ds_bpermute (round 6).  A ds_bpermute_b32 moves 4 bytes per lane in 4 LDS-array cycles (2.6 ns of the CU's crossbar in
// isolation, profiles/r05_ubench_xlane.txt); a ds_read_b64 moves 8 in 2.  A lane writes its seven partial sums -- T = its four
// columns, suf1..3, pre1..3 -- into seven LINES of one float64 per lane (structure of arrays: consecutive lanes touch consecutive
// 8-byte words, no bank conflict; a first version with one 64-byte slot per lane ran 16-way conflicts and lost 50 %,
// profiles/r06_wline_aos.txt) and reads its neighbours' entries with ds_read_b64: at 31 wide 7 writes + 10 reads of 8 bytes
// instead of 20 ds_bpermute_b32 per quantity, at 63 wide 7 + 18 instead of 36.  Measured on the headline workload (profiles/r06_ab_wline.txt): 41 wide 9.54 -> 8.25 ms, 63 wide 13.2 -> 10.6 -- but 17 /
// 21 / 31 wide 5.27 / 5.54 / 6.66 -> 5.88 / 6.03 / 7.43: there the wave also holds a centre ring of kh / 2 + 1 rows in LDS, the
// lines' 6.3 KB cost it three of ten resident waves per CU, and at three whole neighbour lanes 17 LDS operations replace 20.  So
// the lines serve the builds that re-load both rows (RING 0: kernels taller than 39 rows, no LDS ring), ds_bpermute the others.
// Same terms, same order of additions as hsum_wide: bit-identical results.  `line` = this lane's entry of line 0; every
// line has WLINE_G guard entries either side that are never written: their readers are overlap lanes whose sums are discarded.
constexpr int WLINE_G = 26;                                 // >= F + 1 for every admitted width (overlap lanes <= 24: F <= 24)
constexpr int WLINE_STRIDE = (WAVE + 2 * WLINE_G) * 8;      // bytes per line
constexpr size_t WLINE_BYTES = 7 * (size_t)WLINE_STRIDE;    // 6.3 KB per wave
#ifndef HK_WLINE
#define HK_WLINE 1
#endif
template <int RW, int RING>
constexpr bool use_wline() { return HK_WLINE && RW < 0 && RING == 0; }
template <int E, bool PAIR = false>
__device__ __forceinline__ void hsum_wide_line(const double (&V)[PX], double (&H)[PX], const WideLanes& wl, int lane, char* line) {
    constexpr int RV = 2 * PX + E;
    double pre[PX + 1], suf[PX + 1];
    pre[1] = V[0];
    pre[2] = V[0] + V[1];
    pre[3] = pre[2] + V[2];
    suf[1] = V[3];
    suf[2] = V[2] + V[3];
    suf[3] = V[1] + suf[2];
    pre[4] = suf[4] = pre[2] + suf[2];
    auto at = [&](const char* p, int array) -> double { return *reinterpret_cast<const double*>(p + array * WLINE_STRIDE); };
    *reinterpret_cast<double*>(line) = pre[4];
#pragma unroll
    for (int k = 1; k <= 3; ++k) {
        *reinterpret_cast<double*>(line + k * WLINE_STRIDE) = suf[k];
        *reinterpret_cast<double*>(line + (3 + k) * WLINE_STRIDE) = pre[k];
    }
    xch_order();
    // the whole lanes: the same terms in the same order as hsum_wide (PAIR: pairs of neighbouring lanes, see there)
    constexpr bool FULL2 = lane_full_for_all(RV, 2);
    double common = pre[PX];
    if constexpr (PAIR) {
        const int W = FULL2 ? wl.f : wl.f - 1;  // wave-uniform
        const double tl = dpp_from_left(pre[PX]), tr = dpp_from_right(pre[PX]);
        common = at(line + W * 8, 0);
        common = common + (tl + pre[PX]);               // the pairs (-1, 0) and (1, 2)
        common = common + (tr + at(line + 2 * 8, 0));
        for (int o = 3; o <= W - 2; o += 2) common = common + (at(line + o * 8, 0) + at(line + (o + 1) * 8, 0));
        for (int o = 3; o <= W; o += 2) common = common + (at(line - o * 8, 0) + at(line - (o - 1) * 8, 0));
    } else {
        if (wl.f >= 2) {  // wave-uniform
            common = common + dpp_from_left(pre[PX]);
            common = common + dpp_from_right(pre[PX]);
        }
        for (int j = 2; j < wl.f; ++j) {  // wave-uniform: the whole lanes beyond the DPP neighbours
            common = common + at(line - j * 8, 0);
            common = common + at(line + j * 8, 0);
        }
    }
    static_for<2, 4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        // the lanes F (j == 2) and F + 1 (j == 3) to the left / right
        const char* const lb = line - (wl.f + j - 2) * 8;
        const char* const rb = line + (wl.f + j - 2) * 8;
        double ls[PX + 1], rp[PX + 1];
        static_for<1, PX + 1>([&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (PAIR && j == 2 && k == PX && FULL2) {
            } else {
                if constexpr (need_left_at(RV, j, k)) ls[k] = at(lb, k == PX ? 0 : k);
                if constexpr (need_right_at(RV, j, k)) rp[k] = at(rb, k == PX ? 0 : 3 + k);
            }
        });
        if constexpr (lane_full_for_all(RV, j)) {
            if constexpr (!PAIR) {
                common = common + ls[PX];
                common = common + rp[PX];
            }
        } else {
            static_for<0, PX>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int ll = left_len(RV, i, j), rl = right_len(RV, i, j);
                if constexpr (ll > 0) {
                    if constexpr (specific_before(RV, i, j, false, true)) H[i] = H[i] + ls[ll];
                    else H[i] = ls[ll];
                }
                if constexpr (rl > 0) {
                    if constexpr (specific_before(RV, i, j, true, true)) H[i] = H[i] + rp[rl];
                    else H[i] = rp[rl];
                }
            });
        }
    });
    static_for<0, PX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (specific_before(RV, i, 4, false, true)) H[i] = common + H[i];
        else H[i] = common;
    });
    xch_order();  // the next quantity rewrites the lines: every read above has been issued (LDS operations of a wave run in order)
}

// RW >= 0: compile-time half-width; RW = -1 - E: wide kernel with rw mod 4 == E (hsum_wide), RW = -5 - E: its paired form.  DPP2: see hsum.
// `xch`: the lane's slot of the LDS exchange line (XCH builds of hsum; the float64 sums of the wide kernels: hsum_wide_line)
template <int RW, typename T, bool XCH = false, bool DPP2 = false, bool WLINE = false>
__device__ __forceinline__ void hsum_any(const T (&V)[PX], T (&H)[PX], const WideLanes& wl, int lane, char* xch = nullptr) {
    if constexpr ((HK_ABLATE & 4) != 0) {
#pragma unroll
        for (int i = 0; i < PX; ++i) H[i] = V[i];
    } else if constexpr (RW >= 0)
        hsum<RW, T, XCH, DPP2>(V, H, lane, xch);
    else if constexpr (WLINE && std::is_same<T, double>::value) {
        hsum_wide_line<wide_e(RW), wide_paired(RW)>(V, H, wl, lane, xch);
    } else
        hsum_wide<wide_e(RW), T, wide_paired(RW)>(V, H, wl, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// ~utils.nan_equals(v, nodata) (utils.py:54-56) without branches: `cmp` is the numeric nodata (or NaN, which never
// compares equal, for the None / NaN modes) and `nan_is_nodata` selects the isnan test (raster_array.py:298-308).
struct NodataTest {
    float cmp;
    bool nan_is_nodata;
};
__device__ __forceinline__ NodataTest make_nodata_test(int mode, float nodata) {
    NodataTest t;
    t.cmp = mode == 2 ? nodata : __int_as_float(0x7fc00000);
    t.nan_is_nodata = mode == 1;
    return t;
}
__device__ __forceinline__ bool px_valid(float v, const NodataTest& t) {
    return !((v == t.cmp) | (t.nan_is_nodata & (v != v)));
}

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fc00000); }

struct RowRaw {
    float4 s, r;
};

// Packed float32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two IEEE operations per instruction).  With
// -ffp-contract=off `a * b - c` stays a rounded product followed by a rounded subtraction, element-wise identical to
// __fmul_rn / __fsub_rn; pk_fma is the fused single-rounding form (used by the r2-mask certificate only).
typedef float f2 __attribute__((ext_vector_type(2)));
#define HK_P2(a, j) (f2{(a)[2 * (j)], (a)[2 * (j) + 1]})
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// float32(RN64(n / d)) without the IEEE float64 division (13 VALU instructions, one of them v_rcp_f64 at quarter rate):
//     y0 = v_rcp_f64(d)            relative error <= 2^-22 ASSUMED (2^-24.4 measured over 4e6 operands, profiles/r02_ubench_valu.txt)
//     y1 = y0 + y0 * (1 - d * y0)  one Newton step: <= 2^-44 + 2^-53
//     q  = n * y1                  <= 2^-44 + 2^-52 relative = 513 float64 ulps of q at most
// RN64(n / d) lies within 514 ulps of q, so both round to the same float32 unless one of its rounding boundaries (the
// midpoints of neighbouring float32 values: low 29 mantissa bits == 0x10000000) lies within 514 ulps of q.  quot_guard()
// maps q to a word that is < 2 * HK_DIV_GUARD + 1 exactly when q is that close (probability 2^-18 per pixel); such a
// pixel pair -- and one with a quotient outside the float32 normal range (every infinite / NaN / zero-denominator case;
// an exactly zero quotient is exempt, it is exact) -- is divided again the IEEE way, so results are identical by
// construction.
#ifndef HK_DIV_GUARD
#define HK_DIV_GUARD 1024u
#endif
__device__ __forceinline__ double fast_quot(double n, double d) {
    double y = __builtin_amdgcn_rcp(d);
    const double e = __fma_rn(-d, y, 1.0);
    y = __fma_rn(y, e, y);
    return __dmul_rn(n, y);
}
__device__ __forceinline__ unsigned quot_guard(double q) {
    return ((unsigned)__double2loint(q) & 0x1fffffffu) - (0x10000000u - HK_DIV_GUARD);
}
// biased float64 exponent of q relative to that of 2^-126: <= 0x0fd00000 exactly for |q| in [2^-126, 2^128), and 0 for a
// zero / float64-denormal q (num == 0 or an underflowing quotient: the IEEE quotient rounds to the same signed float32 zero)
__device__ __forceinline__ unsigned quot_range(double q) {
    const unsigned e = (unsigned)__double2hiint(q) & 0x7ff00000u;
    return e ? e - 0x38100000u : 0u;
}

// A WAVE-UNIFORM global address as a scalar register pair that the optimiser cannot see through (SB builds, fit_scalar_bases()).
// Row addresses are uniform (plane + row * stride) and a lane adds its 32-bit byte offset.  Left to itself the compiler
// re-associates `(plane + lane offset) + row offset` and keeps `plane + lane offset` as a loop-invariant 64-bit VECTOR pair per
// plane: two registers each (the headline's certificate build holds eight for its three planes).  Behind the opaque value the
// lane's offset is added to a scalar row address at every access instead -- every build loses 2 to 10 registers that way
// (profiles/r06b_scalar_base.txt), but most run the same or up to 5 % slower (`gain` 15 wide), so only the builds that gain an
// occupancy step take it: the NaN-aware certificate builds wider than 15 with kw / 2 mod 4 = 1 .. 3, which were two registers short
// of a third wave per SIMD (HISTORY.md 70).  The address space is kept: an opaque GENERIC pointer turns every access into a
// flat_ instruction.
typedef __attribute__((address_space(1))) char hk_gchar;
__device__ __forceinline__ hk_gchar* row_address(const void* uniform_ptr) {
    unsigned long long v = reinterpret_cast<unsigned long long>(uniform_ptr);
    asm("" : "+s"(v));  // (not volatile: free to move with the access it serves)
    return (hk_gchar*)v;
}
// A plane pointer READ FROM A TABLE in device memory (the job table of a batched launch) carries no address space the compiler
// could infer: every access through it became a flat_ instruction -- which counts on the LDS counter too, so that a wait for the
// row ring also waited for the global stream (the batched builds of configs[3] ran that way from round 3 to round 6).  The
// planes live in device memory: the pointer is rebuilt as a global one behind an opaque scalar (a plain cast pair is folded away).
template <typename T>
__device__ __forceinline__ T* table_pointer(T* uniform_ptr) {
    unsigned long long v = reinterpret_cast<unsigned long long>(uniform_ptr);
    asm("" : "+s"(v));
    return (T*)(__attribute__((address_space(1))) T*)v;
}
template <int MODEL, bool R2, int RW, bool DENSE, int RING, bool CERT_ONLY>
constexpr bool fit_scalar_bases() {
    // gain-offset + r2 mask, NaN-aware, wider than 15: the builds that are short of registers for a third wave without it (kw / 2 mod 4
    // != 0 of the plain form, != 3 of the paired form -- that one needs 20 registers fewer anyway)
    return (MODEL == 2 && R2 && !DENSE && RW < 0 && (wide_paired(RW) ? wide_e(RW) != 3 : wide_e(RW) != 0) && CERT_ONLY) ||
           (MODEL == 0 && !R2 && !DENSE && RING == 3 && (RW == 5 || RW == 6));  // gain, NaN-aware split ring, 11 / 13 wide
}

// Streaming stores: the output planes are written once and never read by this launch -- non-temporal stores keep them
// from displacing the rows the neighbouring strips still share in L2 (strip-march pattern: -2 %, tools/ubench_strips.hip)
typedef float hk_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_nt(float4* p, float4 v) {
    __builtin_nontemporal_store(hk_v4{v.x, v.y, v.z, v.w}, reinterpret_cast<hk_v4*>(p));
}
typedef __attribute__((address_space(1))) hk_v4 hk_gv4;
__device__ __forceinline__ void store4_nt(hk_gchar* p, float4 v) {
    __builtin_nontemporal_store(hk_v4{v.x, v.y, v.z, v.w}, (hk_gv4*)p);
}

// RN64(1 / n) for window counts 0 .. HK_INV_N_MAX (entry 0 is never used for a stored pixel).  For a float32 t = M * 2^a (M a
// 24-bit integer) and an integer n < 2^16 the quotient t / n is never an exact float32 rounding midpoint (that would need the
// power of two in n to exceed n) and never closer to one than 1 / (2 n) units of the quotient's last place, i.e. 2^-41 relative,
// while f64(t) * RN64(1/n) is within 2^-52 of it -- so rounding that product to float32 IS the IEEE float32 division
// (kernel_model.py:351: float32 `t / mask_sum`; PROOFS.md appendix C, tests/test_inv_n_cpu.py).  Only wave-rows with a hole or a raster edge in their windows read the table
// (elsewhere 1/N is a kernel argument, valid for every window the launcher admits: kh * kw < 2^16), so it lives in global memory
// (8 KB, cache-resident) and costs no LDS.  1023 covers kernels up to 31 x 33; larger windows divide (those rows only).
constexpr int HK_INV_N_MAX = 1023;
struct InvTable {
    double v[HK_INV_N_MAX + 1];
    constexpr InvTable() : v() {
        v[0] = 0.0;
        for (int n = 1; n <= HK_INV_N_MAX; ++n) v[n] = 1.0 / (double)n;
    }
};
static __device__ const InvTable HK_INV_N = InvTable();

// Invalid pixels are kept in the LDS row ring as this NaN payload in the SOURCE plane (the reference plane holds their
// zero fill): no separate mask plane, i.e. the general kernels need exactly the LDS of the nodata=None ones.
constexpr unsigned RING_SENTINEL = 0x7fc0deadu;

// Rows are padded to a multiple of PX elements (stride % 4 == 0, checked on the host), so every lane moves a full
// 16 bytes.  The load is unconditional: the row is clamped into the raster (wave-uniform, so the row address is a
// scalar base and the lane offset a 32-bit VGPR -> no per-row vector address arithmetic) and `xq` is a safe in-raster
// quad for lanes outside it; whatever such a load returns is discarded by process_row (row_ok / colbits).
// Tall kernels re-load the leaving row from global memory (ring modes 2 / 0): that is the row's LAST use, so the re-load is
// non-temporal -- it no longer displaces the rows still waiting for theirs (gain-blk-offset 15x15 x 8 bands at 16384^2:
// 11.09 -> 10.37 ms; gain-offset 15x15 -0.6 %).  Entering rows stay cached: the neighbouring strips and the re-load need them.
#ifndef HK_NT_LEAVE
#define HK_NT_LEAVE true
#endif
template <bool NT = false, bool SB = false>
__device__ __forceinline__ RowRaw load_row(const float* __restrict__ sp, const float* __restrict__ rp, long long stride,
                                           int row, int height, unsigned xq) {
    const int rc = min(max(row, 0), height - 1);
    RowRaw o;
    if constexpr (SB) {
        const hk_gchar* const ps = row_address(sp + (long long)rc * stride);
        const hk_gchar* const pr = row_address(rp + (long long)rc * stride);
        typedef __attribute__((address_space(1))) const hk_v4 gv4c;
        hk_v4 sv, rv;
        if constexpr (NT) {
            sv = __builtin_nontemporal_load((gv4c*)(ps + xq));
            rv = __builtin_nontemporal_load((gv4c*)(pr + xq));
        } else {
            sv = *(gv4c*)(ps + xq);
            rv = *(gv4c*)(pr + xq);
        }
        o.s = make_float4(sv.x, sv.y, sv.z, sv.w), o.r = make_float4(rv.x, rv.y, rv.z, rv.w);
        return o;
    }
    const char* __restrict__ ps = reinterpret_cast<const char*>(sp + (long long)rc * stride);
    const char* __restrict__ pr = reinterpret_cast<const char*>(rp + (long long)rc * stride);
    if constexpr (NT) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v sv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(ps + xq));
        const f4v rv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(pr + xq));
        o.s = make_float4(sv.x, sv.y, sv.z, sv.w), o.r = make_float4(rv.x, rv.y, rv.z, rv.w);
    } else {
        o.s = *reinterpret_cast<const float4*>(ps + xq);  // xq: this lane's byte offset in the row (32-bit)
        o.r = *reinterpret_cast<const float4*>(pr + xq);
    }
    return o;
}

// A processed row as it sits in the LDS ring: zero-filled source/reference + one validity byte per pixel.
struct RowZ {
    float s[PX], r[PX];
    float e[PX];  // general kernels: source values as the LDS ring keeps them (RING_SENTINEL in place of invalid pixels)
    unsigned m;  // byte i = mask of pixel i (0/1)
    bool clean;  // wave-uniform: the row is inside the raster and every pixel of every lane of the strip is valid
};

// DENSE: both rasters have nodata None (raster_array.py:302-303: every pixel valid), so validity is purely geometric:
// only rows outside the raster / not yet added and the columns of the last strip beyond the raster need zeroing.
template <int MODEL, bool DENSE, bool PERPX = false>
__device__ __forceinline__ RowZ process_row(const RowRaw& raw, bool row_ok, unsigned colbits, bool full_wave,
                                            const NodataTest& ts, const NodataTest& tr, double n0, double n1) {
    const float s[PX] = {raw.s.x, raw.s.y, raw.s.z, raw.s.w};
    const float r[PX] = {raw.r.x, raw.r.y, raw.r.z, raw.r.w};
    RowZ z;
    z.m = 0;
    z.clean = false;
    if constexpr (DENSE) {
        if (row_ok && full_wave) {  // wave-uniform: every column of every lane is inside the raster -> nothing to zero
#pragma unroll
            for (int i = 0; i < PX; ++i) z.s[i] = s[i], z.r[i] = r[i];
        } else {
            const unsigned bits = row_ok ? colbits : 0u;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const bool in = (bits >> i) & 1u;
                z.s[i] = in ? s[i] : 0.f;
                z.r[i] = in ? r[i] : 0.f;
            }
        }
        return z;
    }
    bool ok[PX];
    if (ts.nan_is_nodata & tr.nan_is_nodata) {
        // wave-uniform: the RasterArray default (nodata = nan on both rasters): valid <=> neither value is NaN
        // <=> the pair compares ordered -- one v_cmp_o_f32 instead of four compares
#pragma unroll
        for (int i = 0; i < PX; ++i) ok[i] = !__builtin_isunordered(s[i], r[i]);
    } else {
#pragma unroll
        for (int i = 0; i < PX; ++i) ok[i] = px_valid(s[i], ts) & px_valid(r[i], tr);
        if constexpr (MODEL == 1 && !PERPX) {
            // gain-blk-offset re-derives the source mask from the NORMALISED source, whose nodata is NaN
            // (kernel_model.py:292-298): s * n0 + n1 is NaN exactly where s is (finite block statistics), whatever the
            // source's own nodata value
#pragma unroll
            for (int i = 0; i < PX; ++i) ok[i] = ok[i] & (s[i] == s[i]);
        }
    }
    if constexpr (!PERPX) {
        // wave-uniform short cut: nothing to zero, nothing to pack (the usual state away from the edges of real mosaics)
        if (row_ok && full_wave && __all((int)(ok[0] & ok[1] & ok[2] & ok[3]))) {
#pragma unroll
            for (int i = 0; i < PX; ++i) z.s[i] = z.e[i] = s[i], z.r[i] = r[i];
            z.m = 0x01010101u;
            z.clean = true;
            return z;
        }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        bool m = row_ok & (bool)((colbits >> i) & 1u) & ok[i];
        if constexpr (PERPX) {
            // gain-blk-offset with R2: the mask is re-derived from the NORMALISED float64 source (kernel_model.py:292-298)
            const double sd = __dadd_rn(__dmul_rn((double)s[i], n0), n1);
            m = m & !(sd != sd);
        }
        z.s[i] = m ? s[i] : 0.f;
        z.r[i] = m ? r[i] : 0.f;
        z.e[i] = m ? s[i] : __uint_as_float(RING_SENTINEL);
        z.m |= (m ? 1u : 0u) << (8 * i);
    }
    return z;
}

// Running float64 column sums of one wave.
template <int MODEL, bool R2, bool DENSE>
struct ColSums {
    // gain-blk-offset: with R2 the normalised source s' = s * n0 + n1 is formed per pixel in float64 (BLK); without it
    // the kernel sums the raw source and normalises the window sum (BLKA, see fit_apply_kernel)
    static constexpr bool GO = MODEL == 2, BLK = MODEL == 1 && R2, BLKA = MODEL == 1 && !R2;
    static constexpr bool NEED_N = ((GO || R2) && !DENSE) || BLKA, NEED_P = GO || R2, NEED_S2 = GO || R2, NEED_R2S = R2;
    double S[PX], R[PX], P[PX], S2[PX], R2s[PX];
    unsigned N;  // packed bytes

    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < PX; ++i) S[i] = R[i] = P[i] = S2[i] = R2s[i] = 0.0;
        N = 0;
    }

    template <bool ADD>
    __device__ __forceinline__ void update(const RowZ& z, double n0, double n1) {
        [[maybe_unused]] float sr[PX];  // src * ref rounded to float32 first (:175,:334), two pixels per instruction
        if constexpr (!BLK && NEED_P) {
#pragma unroll
            for (int j = 0; j < PX / 2; ++j) {
                const f2 pr = HK_P2(z.s, j) * HK_P2(z.r, j);
                sr[2 * j] = pr.x, sr[2 * j + 1] = pr.y;
            }
        }
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const bool m = DENSE ? true : (bool)((z.m >> (8 * i)) & 1u);
            const double dr = (double)z.r[i];
            if constexpr (BLK) {
                // normalised source in float64 (NumPy>=2 promotion of `src * np.float64`, kernel_model.py:295)
                const double sd = m ? __dadd_rn(__dmul_rn((double)z.s[i], n0), n1) : 0.0;
                S[i] = ADD ? __dadd_rn(S[i], sd) : __dsub_rn(S[i], sd);
                if constexpr (NEED_P) {
                    const double p = __dmul_rn(sd, dr);
                    P[i] = ADD ? __dadd_rn(P[i], p) : __dsub_rn(P[i], p);
                }
                if constexpr (NEED_S2) {
                    const double q = __dmul_rn(sd, sd);
                    S2[i] = ADD ? __dadd_rn(S2[i], q) : __dsub_rn(S2[i], q);
                }
            } else {
                const double ds = (double)z.s[i];
                S[i] = ADD ? __dadd_rn(S[i], ds) : __dsub_rn(S[i], ds);
                if constexpr (NEED_P) {
                    const double p = (double)sr[i];
                    P[i] = ADD ? __dadd_rn(P[i], p) : __dsub_rn(P[i], p);
                }
                if constexpr (NEED_S2) S2[i] = __fma_rn(ADD ? ds : -ds, ds, S2[i]);  // ds*ds exact in f64
            }
            R[i] = ADD ? __dadd_rn(R[i], dr) : __dsub_rn(R[i], dr);
            if constexpr (NEED_R2S) R2s[i] = __fma_rn(ADD ? dr : -dr, dr, R2s[i]);
        }
        if constexpr (NEED_N) N = ADD ? N + z.m : N - z.m;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// The fused kernel.  MODEL: 0 gain, 1 gain-blk-offset, 2 gain-offset.  R2: compute the R2 quantity set.
// RW: compile-time kernel half-width, or -1 - E for the kernels wider than 15 whose half-width is 4 F + E (F run-time, hsum_wide).  DENSE: both inputs have nodata None.
#ifndef HK_CERT_SKIP
#define HK_CERT_SKIP 3  // rows for which the r2-mask certificate is not attempted after it failed (measured)
#endif
#ifndef HK_FIT_MIN_WAVES_WIDE
#define HK_FIT_MIN_WAVES_WIDE 2  // the general gain-offset + R2 kernels of width >= 9 spill at 3 waves per SIMD
#endif
#ifndef HK_FIT_MIN_WAVES
#define HK_FIT_MIN_WAVES 3  // waves per SIMD the register allocator must leave room for (tuned on MI355X)
#endif
// RING: where the leaving row (t - kh) and the window's centre row (t - rh) come from:
//   1  both from a wave-private LDS ring of kh processed rows (short kernels, kh <= 5);
//   2  centre row from an LDS ring of rh + 1 rows holding only `s` + mask (20 B per lane-row), leaving row re-loaded from
//      global memory -- tall kernels: a full ring would cut occupancy to one wave per SIMD, re-loading BOTH rows makes
//      three streams that all miss L2 and the kernel fabric-bound;
//   0  both re-loaded (very tall kernels whose centre ring would not fit either);
//   3  SPLIT ring (light builds -- gain, gain-blk-offset without R2 -- with 7 <= kh <= 17): the rh newest rows stay in
//      REGISTERS (8 VGPRs per row), the rh + 1 older ones in an LDS ring (32 B per lane-row): no global re-load of the leaving
//      row (mode 2 moves 20 instead of 12 bytes per pixel through the fabric at 15x15) with 16 KB instead of 30 KB of LDS per
//      wave.  The row that leaves the registers IS the window's centre row, so the centre needs no LDS read either.
// WPB = waves per workgroup.  The memory-bound builds with a full LDS ring (gain, gain-blk-offset without R2, short kernels)
// put HK_WPB_MEM ADJACENT STRIPS of one segment into a workgroup and keep them in lock-step with a barrier per row: the
// workgroup then reads and writes 4 KB of every row together instead of 1 KB per wave at unrelated times, which the HBM
// pays back -- strip-march pattern without arithmetic 4 720 -> 4 910 GB/s (tools/ubench_strips.hip); gain 5x5 at 16384^2
// 2.70 -> 2.55 ms, gain-blk-offset 5x5 4.57 -> 4.34 ms (8 waves: 2.51 / 4.38; configs[1]'s smaller raster prefers 4).
// Not for the VALU-bound gain-offset builds with the R2 work (their waves would only wait for each other: 0 to +2 %) and not for the tall
// kernels that re-load their leaving rows (15x15: +14 %, the re-loads of a whole workgroup then collide).
#ifndef HK_WPB_MEM
#define HK_WPB_MEM 4
#endif
// Which float64 horizontal sums exchange their partial sums through LDS (hsum's XCH) instead of DPP: bit 0 S, 1 R, 2 P,
// 3 S2, 4 R2.  One 16-byte slot per lane + one at each end = XCH_BYTES per wave behind the row rings.
#ifndef HK_XCH
#define HK_XCH 0
#endif
// (HK_SRING_MAX / HK_SRING_MAX_BLK, hk_kernels.h: register rows of the split ring -- 7 = kernels up to 15 rows tall; the
// gain-blk-offset builds use the mode up to 11 rows only (hk_api.hip fill_args) and hold 5: 16 VGPRs less, no spill)
#ifndef HK_CERT_R2_F32
#define HK_CERT_R2_F32 1
#endif
constexpr size_t XCH_BYTES = (WAVE + 2) * 16;
template <int MODEL, int RW, int RING, int WPB>
constexpr int xch_mask() {
    return (MODEL == 2 && (RW == 1 || RW == 2) && RING == 1 && WPB == 1) ? HK_XCH : 0;
}

// Waves per SIMD the register allocator has to leave room for.  Four for the certificate-only build of the narrow kernels (128
// VGPRs), three (168 VGPRs) by default -- and two (256 VGPRs) for the builds that do not fit into 168 without spilling to
// scratch memory (HK_NOSPILL; tools/kernel_regs.py --spills lists none with it): the wide NaN-aware builds with the R2 work
// and the NaN-aware split-ring builds of `gain` -- measured equal or 10-15 % FASTER at two waves, profiles/r04_nospill.txt.  Where
// two waves were slower the registers were found elsewhere: gain-blk-offset's split ring holds 5 instead of 7 rows (it serves
// kernels up to 11 rows), the certificate-only builds of the kernels wider than 15 give up their leaving row in flight (PF_OLD).
#ifndef HK_NOSPILL
#define HK_NOSPILL 1
#endif
#ifndef HK_CERT_WIDE_3WAVES
#define HK_CERT_WIDE_3WAVES 1
#endif
// rows in flight of the kernels wider than 15: HK_PF_WIDE entering, HK_PO_WIDE leaving (see fit_unit; 1 / 1 measured best)
#ifndef HK_PF_WIDE
#define HK_PF_WIDE 1
#endif
#ifndef HK_PO_WIDE
#define HK_PO_WIDE 1
#endif
template <int MODEL, bool R2, int RW, bool DENSE, int RING, bool CERT_ONLY>
constexpr int fit_min_waves() {
    if (CERT_ONLY && RW >= 0 && RW <= 3) return 4;
    // (their certificate-only builds fit into 168 registers -- round 5: 17 wide on NaN-nodata rasters 6.46 -> 5.56 ms, 9 - 15 wide
    // equal.  Beyond 15 wide with kw / 2 mod 4 = 1 .. 3 they were two registers short -- a loop-invariant 64-bit vector address pair
    // per plane -- and take their row addresses as opaque scalars since round 6 (row_address, fit_scalar_bases): 19 / 21 / 23 wide
    // on NaN-nodata rasters 6.45 -> 5.78 ms, 31 wide 7.1 -> 6.9, profiles/r06b_ab_scalar_base.txt)
    if (MODEL == 2 && R2 && !DENSE && (RW < 0 || RW >= 4) && !(CERT_ONLY && HK_CERT_WIDE_3WAVES && RW < 0)) return HK_FIT_MIN_WAVES_WIDE;
    if (HK_NOSPILL) {
        // (round 6: with the LDS exchange lines the compiler issues a quantity's neighbour reads together -- 8 to 25 registers more
        // at the peak -- in the builds that use them: kernels taller than 39 rows)
        if (use_wline<RW, RING>() && MODEL == 2) return 2;
        if (RW < 0 && (HK_PF_WIDE > 1 || HK_PO_WIDE > 1)) return 2;  // (rows in flight instead of a third wave: see fit_unit)
        if (RW < 0 && R2 && !CERT_ONLY) return 2;                          // wider than 15 with the R2 work (10 - 28 spilled registers at three)
        if (MODEL == 2 && R2 && !DENSE && RW == 3) return 2;               // gain-offset + R2, 7 wide, NaN-aware
        if (MODEL != 2 && R2 && !DENSE && RW >= 4 && RING == 2) return 2;  // gain / gain-blk-offset + R2, 9-15 wide, NaN-aware
        // gain, NaN-aware split ring: 15 wide stays at two waves (175 registers); 11 / 13 wide take their row addresses as opaque
        // scalars (fit_scalar_bases) and fit three: 3.33 -> 3.05 / 3.52 -> 3.40 ms on NaN-nodata rasters -- at 15 rows the ring's
        // 16 KB allow ten waves per CU whatever the registers say and the tighter allocation ran 2 % slower
        // (profiles/r06b_scalar_base.txt)
        if (RING == 3 && !DENSE && MODEL == 0 && RW == 7) return 2;
    }
    return HK_FIT_MIN_WAVES;
}

// One unit of the fused kernel: the strip `strip` of band `band`, output rows [y0, y1) (priming rows included, the wave marches
// from y0 - rh).  LIST: the unit is a run of a list launch -- rows that are not marked in FitArgs::open_rows are neither stored
// nor counted (the certificate build settled them).
template <int MODEL, bool R2, int RW, bool DENSE, int RING, bool CERT_ONLY, int WPB, bool LIST>
__device__ __forceinline__ void fit_unit(const FitArgs& a, const int band, const int strip, const int y0, const int y1, const int lane,
                                         const int wave_in_wg, float4* const lds4) {
    using CS = ColSums<MODEL, R2, DENSE>;
    // gain-blk-offset (kernel_model.py:276-303) normalises the source with the block's statistics, s' = s * n0 + n1 in
    // float64 (NumPy >= 2 promotion), and fits `gain` to it.  With R2 (BLK) s' is formed per pixel.  Without (BLKA, the
    // fused RasterFuse path) the kernel keeps the exact sums of the RAW source and the window count and forms
    //     sum(s') = RN(RN(n0 * sum(s)) + RN(n1 * N))
    // per output pixel: algebraically the same number, rounded twice instead of once per pixel and addition -- like the
    // order of the float64 window summation itself a last-bit freedom of a float64 quantity whose
    // float32 quotient it moves with probability ~1e-8 per pixel; it halves the kernel's float64 work.
    constexpr bool GO = MODEL == 2, BLK = MODEL == 1 && R2, BLKA = MODEL == 1 && !R2;
    constexpr bool SB = fit_scalar_bases<MODEL, R2, RW, DENSE, RING, CERT_ONLY>();  // row addresses as opaque scalars (row_address)
    constexpr bool USE_N = GO || R2 || BLKA;
    constexpr bool UNIFORM_N = GO || BLKA;  // builds that track wave-rows whose every window is complete and all-valid
    static_assert(!(DENSE && MODEL == 1), "gain-blk-offset re-derives its mask from the normalised source");
    const int rh = a.rh, kh = 2 * rh + 1;
    const int rw = RW >= 0 ? RW : a.rw;
    const int ol = RW >= 0 ? (RW + PX - 1) / PX : a.overlap_lanes;
    const int out_lanes = WAVE - 2 * ol;
    [[maybe_unused]] const WideLanes wl = make_wide_lanes(rw, lane);  // kernels wider than 15 (RW < 0) only
    const int x = (strip * out_lanes + lane - ol) * PX;
    const int W = a.width, H = a.height;

    const float* __restrict__ sp = a.src + (long long)band * a.band_stride;
    const float* __restrict__ rp = a.ref + (long long)band * a.band_stride;
    const long long out_base = (long long)band * a.band_stride;

    const bool lane_in = x >= 0 && x < W;
    const unsigned xq = lane_in ? (unsigned)x * 4u : 0u;               // load byte offset: a safe quad for lanes outside
    const unsigned xbytes = (unsigned)(x > 0 ? x : 0) * 4u;            // store byte offset (only lanes inside ever store)

    // The wave's first rows are requested HERE, before the rest of the set-up (column masks, LDS ring, tables), so that their
    // latency runs beside it: 200 instead of 410 instructions before the first load (measured neutral on every configuration,
    // profiles/r03_early_loads.txt: the resident waves of a CU cover each other's start).
    const int t_first = y0 - rh, t_last = y1 - 1 + rh;
    // One row in flight: the next row's load is issued as soon as the current one has been consumed, so it lands in the
    // same registers (no queue rotation).  A two-row queue was measured equal or slower (8 more VGPRs + 8 moves per row).
    // The light `gain` kernel without R2 is HBM-bound (VALU 37 % busy, ~3 waves per SIMD because of the LDS ring): it keeps
    // HK_PF_GAIN rows in flight in a small register queue (moves are free there).
#ifndef HK_PF_GAIN
#define HK_PF_GAIN 2  // 4 and 6 rows measured the same (2.91-2.97 ms): the wait is on the LDS ring, not on HBM latency
#endif
#ifndef HK_PF_BLKA
#define HK_PF_BLKA 1  // gain-blk-offset: 2 / 3 / 4 rows in flight measured the same at 15x15 (profiles/r03_blk15_ablation.txt)
#endif
    // Kernels wider than 15 (round 6, profiles/r06_pmcl_k31.txt): at 31 x 31 a wave spends 37 % of its life in s_waitcnt and 19 % in
    // issue stalls with the VALU 22 % and the LDS array 45 % busy, and 82 % of its L2 requests go on to the fabric.  More rows in
    // flight per wave (register queues of HK_PF_WIDE entering / HK_PO_WIDE leaving rows, two waves per SIMD for the registers) do
    // NOT buy that time back: 2 / 2, 3 / 3, 4 / 4 and 3 / 1 rows all measured 5 - 15 % slower than one row each at three waves
    // (profiles/r06_ab_prefetch.txt) -- the third wave covers more latency than the queues.  What did pay: the certificate builds of
    // these widths now fetch their leaving row one iteration ahead like the others (it fits since the build lost its gain / R2
    // stores): 17 / 21 / 31 wide 5.27 / 5.54 / 6.66 -> 4.94 / 5.23 / 6.22 ms.
    constexpr int PFD = (MODEL == 0 && !R2) ? HK_PF_GAIN : ((MODEL == 1 && !R2) ? HK_PF_BLKA : (RW < 0 ? HK_PF_WIDE : 1));
    RowRaw q0 = load_row<false, SB>(sp, rp, a.stride, t_first, H, xq);
    [[maybe_unused]] RowRaw qq[PFD > 1 ? PFD - 1 : 1];
    if constexpr (PFD > 1) {
#pragma unroll
        for (int d = 1; d < PFD; ++d) qq[d - 1] = load_row<false, SB>(sp, rp, a.stride, min(t_first + d, t_last), H, xq);
    }

    const bool full_wave = __all((int)(x >= 0 && x + PX <= W));        // no column of this strip needs zeroing
    const NodataTest ts = make_nodata_test(a.src_nd_mode, a.src_nodata);
    const NodataTest tr = make_nodata_test(a.ref_nd_mode, a.ref_nodata);
    unsigned colbits = 0;
#pragma unroll
    for (int i = 0; i < PX; ++i) colbits |= (x + i >= 0 && x + i < W) ? (1u << i) : 0u;
    const bool out_lane = lane >= ol && lane < WAVE - ol && lane_in && x >= a.out_x0 && x < a.out_x1;

    // DENSE: the window count is geometric -- (rows of the window inside the raster) x (columns inside the raster)
    [[maybe_unused]] float ncolf[PX];
    if constexpr (DENSE) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int c = x + i;
            const int n = min(c + rw, W - 1) - max(c - rw, 0) + 1;
            ncolf[i] = (float)(n > 0 ? n : 1);
        }
    }

    double n0 = 0.0, n1 = 0.0;
    if constexpr (MODEL == 1) {
        n0 = a.norm[2 * band];
        n1 = a.norm[2 * band + 1];
    }
    [[maybe_unused]] const double n1_n_full = __dmul_rn(n1, a.nd_full);  // BLKA: RN(n1 * N) of a complete window

    constexpr bool ring = RING == 1;         // full ring: leaving + centre rows
    constexpr bool cring = RING == 2;        // centre-only ring
    constexpr bool sring = RING == 3;        // split ring: rh rows in registers + rh + 1 rows in LDS
    // The HBM-bound kernels without R2 (gain, gain-blk-offset) read the leaving row from LDS one iteration AHEAD (its
    // latency leaves the loop-carried path), which also frees its slot before the entering row is written: their ring has
    // kh - 1 rows (8 KB instead of 10 KB per wave at 5x5 = 20 instead of 16 waves per CU).
    constexpr bool RING_AHEAD = ring && MODEL != 2 && !R2;
    const int ring_rows = ring ? (RING_AHEAD ? (kh > 1 ? kh - 1 : 1) : kh) : ((cring || sring) ? rh + 1 : 0);
    constexpr bool ring2p = ring || sring;   // the LDS ring holds both planes (source + reference) of its rows
    // RING 1: [slot][s|r][lane]; RING 2: [slot][lane] (s only); one ring per wave of the workgroup
    float4* ring_v = lds4 + (size_t)wave_in_wg * (size_t)(ring_rows * (ring2p ? 2 : 1) * WAVE);
    // slots start as rows that were never added: zero contribution, no valid pixel
    constexpr int XCH = xch_mask<MODEL, RW, RING, WPB>();
    // The 15-wide builds of `gain` and gain-offset fetch their distance-2 neighbours through two chained DPP shifts instead of
    // ds_bpermute (hsum DPP2): 15 x 15 gain-offset + r2 mask 4.51 -> 4.44 ms dense, 4.96 -> 4.61 on NaN-nodata rasters, 5.91 -> 5.51
    // with scattered holes, `gain` 3.08 -> 2.99; 9 - 13 wide equal or 2 % slower, gain-blk-offset (VALU-bound) 1 % slower: not those
    // (profiles/r05_ab_dpp2_15wide.txt).  Kernels wider than 15 gain nothing from the same exchange (profiles/r05_ab_dpp_chain_wide.txt:
    // at 31 wide the launch moves 2.16 x its algorithmic bytes through the HBM -- the re-loaded leaving rows -- and is bound there).
    constexpr bool DPPX = HK_DPP2 && RW == 7 && MODEL != 1;
    // (kernels wider than 15: the lane's entry of the wave's first exchange line, hsum_wide_line; one wave per workgroup there)
    [[maybe_unused]] char* const xch = reinterpret_cast<char*>(lds4 + (size_t)WPB * (size_t)(ring_rows * (ring2p ? 2 : 1) * WAVE)) +
                                       (RW < 0 ? (size_t)(lane + WLINE_G) * 8 : (size_t)wave_in_wg * XCH_BYTES + 16 + (size_t)lane * 16);
    const float ring_init = DENSE ? 0.f : __uint_as_float(RING_SENTINEL);
    {
        // four registers the compiler must treat as unrelated: one ds_write_b128 per slot and plane (the vectorised form of this
        // loop scattered four ds_write_b32 per slot and plane -- 64 LDS instructions for a 5x5 ring at the start of every wave)
        float4 fill_s = make_float4(ring_init, ring_init, ring_init, ring_init), fill_r = make_float4(0.f, 0.f, 0.f, 0.f);
        asm volatile("" : "+v"(fill_s.x), "+v"(fill_s.y), "+v"(fill_s.z), "+v"(fill_s.w));
        asm volatile("" : "+v"(fill_r.x), "+v"(fill_r.y), "+v"(fill_r.z), "+v"(fill_r.w));
#pragma clang loop vectorize(disable) unroll(disable)
        for (int sl = 0; sl < ring_rows; ++sl) {
            if constexpr (ring2p) {
                ring_v[(sl * 2 + 0) * WAVE + lane] = fill_s;
                ring_v[(sl * 2 + 1) * WAVE + lane] = fill_r;
            } else {
                ring_v[sl * WAVE + lane] = fill_s;
            }
        }
    }
    // wave-uniform: bit k = the row in ring slot k is `clean` (inside the raster, every pixel of the strip valid) and is
    // stored as it was loaded; the other rows carry RING_SENTINEL in place of their invalid source pixels
    // (64 slots: a centre ring of more rows -- kernels 129 to 255 rows tall -- marks no row clean and decodes every one the long way)
    [[maybe_unused]] unsigned long long ring_clean = 0ull;
    [[maybe_unused]] const bool clean_bits_ok = ring_rows <= 64;
    auto ring_encode = [&](const RowZ& z) {
        if constexpr (DENSE) return make_float4(z.s[0], z.s[1], z.s[2], z.s[3]);
        else return make_float4(z.e[0], z.e[1], z.e[2], z.e[3]);
    };
    // source values + validity bytes of a ring row (zero fill restored)
    auto ring_decode = [&](const float4& v, bool clean, float (&sv)[PX], unsigned& m) {
        sv[0] = v.x, sv[1] = v.y, sv[2] = v.z, sv[3] = v.w;
        m = 0x01010101u;
        if constexpr (DENSE) {
            m = 0u;
        } else {
            if (!clean) {
                // integer form (2-cycle VALU ops): k = min(bits ^ sentinel, 1) is the validity, 0 - k the keep mask
                m = 0u;
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    const unsigned b = __float_as_uint(sv[i]);
                    const unsigned k = min(b ^ RING_SENTINEL, 1u);
                    sv[i] = __uint_as_float(b & (0u - k));
                    m |= k << (8 * i);
                }
            }
        }
    };

    // RING 3: the rh newest rows live in registers.  The slot (wave-uniform, t mod rh) is run-time, register indices are
    // not: a switch over the (at most SRING_MAX) slots exchanges the leaving row for the entering one with 16 moves.
    constexpr int SRING_MAX = split_ring_rows(MODEL);
    [[maybe_unused]] float4 rg_s[SRING_MAX], rg_r[SRING_MAX];
    [[maybe_unused]] unsigned rg_clean = 0u;  // wave-uniform: bit k = the row in register slot k is `clean`
    if constexpr (sring) {
#pragma unroll
        for (int k = 0; k < SRING_MAX; ++k) {
            rg_s[k] = make_float4(ring_init, ring_init, ring_init, ring_init);
            rg_r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    int slot_r = 0;  // RING 3: register slot of this iteration

    // 1/N for the offset division (see HK_INV_N)
    const bool lut_ok = GO && kh * (2 * rw + 1) <= HK_INV_N_MAX;
    const bool use_lut = lut_ok;
    const double* __restrict__ inv_lut = HK_INV_N.v;
    // General gain-offset builds of the narrow kernels: window counts <= 63 are looked up in a LANE-resident copy of the table
    // (lane n holds RN64(1/n); two ds_bpermute per pixel) instead of a global-memory gather whose latency sits in every
    // wave-row that has a hole in reach (HK_LANE_LUT, profiles/r03_lane_lut.txt).
#ifndef HK_LANE_LUT
#define HK_LANE_LUT 1
#endif
    constexpr bool LANE_LUT = HK_LANE_LUT && GO && !DENSE && RW >= 0 && RW <= 3;
    [[maybe_unused]] const bool lane_lut = LANE_LUT && lut_ok && kh * (2 * rw + 1) <= WAVE - 1;  // wave-uniform
    [[maybe_unused]] double lut_lane = 0.0;
    if constexpr (LANE_LUT) lut_lane = inv_lut[lane];

    CS cs;
    cs.clear();

    // r2-mask bookkeeping (gain-offset with a threshold, kernel_model.py:363): R2 values are only materialised when
    // asked for; otherwise pixels are first put through a division-free CERTIFIED test
    //     sstot > 0  &&  ssres < pass_scale * sstot  &&  gain > 0     ==>   (r2 > thresh) & (gain > 0)
    // (pass_scale sits 2^-40 below the rounding boundary of the reference's `1 - f32(ssres/sstot) > thresh`, hk_api.hip)
    // and the exact IEEE evaluation runs for the whole wave-row as soon as any pixel is not certified.
    // Second pass after in-painting (kernel_model.py:366-371): pixels failing the r2 mask take their offset from the
    // in-painted plane and get their gain recomputed as (ref_sum - mask_sum * offset) / src_sum (float32).
    const bool inpaint_pass = GO && R2 && a.offset_in != nullptr;
    const bool want_r2_values = R2 && (a.r2 != nullptr || inpaint_pass);
    const bool count_fails = GO && R2 && a.has_thresh;
    const bool cert_ok = kh * (2 * rw + 1) <= 65535;  // window-count bound assumed by the certificate's constants

    // DENSE gain-offset: away from the raster's edges every stored pixel of a wave-row has N = kh * kw -- then N, its
    // float64 image and RN64(1/N) are kernel arguments (SGPRs) instead of per-pixel conversions and table look-ups.
    // General gain-offset kernels reach the same state through data: `last_dirty` is the latest entering row that was
    // outside the raster or held an invalid pixel anywhere in the strip; while the whole window is newer than that,
    // every window count is kh * kw and the packed-count horizontal sum is skipped as well.
    [[maybe_unused]] bool n_uniform = false;
    [[maybe_unused]] bool n_uniform_cols = false;
    [[maybe_unused]] int last_dirty = t_first - 1;
    if constexpr (UNIFORM_N) {
        bool full = true;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int c = x + i;
            full &= (c - rw >= 0) && (c + rw < W);
        }
        n_uniform_cols = (GO ? cert_ok : true) && __all((int)(!out_lane || full));  // (1/N as a scalar: window counts < 2^16)
    }

    // RING 0 / 2: the re-loaded leaving row runs one iteration ahead where the registers allow it (not in the general
    // gain-offset kernels, which would spill)
    // (the builds of the kernels wider than 15 with the R2 work run at two waves per SIMD -- fit_min_waves -- and have the registers)
    // ... and so do, since round 6, their certificate builds at three waves)
    constexpr bool PF_OLD = !ring && !sring && (DENSE || MODEL != 2 || (RW < 0 && R2)) && !(RW < 0 && CERT_ONLY && HK_PO_WIDE < 1);
    constexpr int POD = PF_OLD ? (RW < 0 ? (HK_PO_WIDE > 1 ? HK_PO_WIDE : 1) : 1) : 0;  // leaving rows in flight
    [[maybe_unused]] RowRaw qo_next;
    [[maybe_unused]] RowRaw qoq[POD > 1 ? POD - 1 : 1];
    if constexpr (PF_OLD) {
        qo_next = load_row<HK_NT_LEAVE, SB>(sp, rp, a.stride, t_first - kh, H, xq);
        if constexpr (POD > 1) {
#pragma unroll
            for (int d = 1; d < POD; ++d) qoq[d - 1] = load_row<HK_NT_LEAVE, SB>(sp, rp, a.stride, t_first - kh + d, H, xq);
        }
    }
    // RING 1: the first leaving row is the zero row the ring was initialised with
    [[maybe_unused]] RowZ zold_next;
#pragma unroll
    for (int i = 0; i < PX; ++i) zold_next.s[i] = zold_next.r[i] = 0.f;
    zold_next.m = 0u, zold_next.clean = false;
    unsigned nfail = 0;
    [[maybe_unused]] int cert_skip = 0;  // wave-uniform: rows for which the r2-mask certificate is not attempted
    int slot = 0;
    int slot2 = 0;  // RING 2: write slot of the centre ring
    const int ring_mod = ring2p ? ring_rows : kh;
    int slot_c = ring_mod - rh;  // slot of the centre row of the output produced at this iteration: (slot - rh) mod ring_mod
    if (slot_c >= ring_mod) slot_c -= ring_mod;
#ifdef HK_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime(), st_iters = 0;
#endif
    for (int t = t_first; t <= t_last; ++t) {
#ifdef HK_STAMPS
        ++st_iters;
#endif
        HK_STAMP(5);  // loop bookkeeping + (first iteration) the set-up
        if constexpr (WPB > 1) {
            // lock-step: the workgroup's strips move down the rows together (same segment: same trip count in every wave)
            __syncthreads();
        }

        // rows that do not come from LDS: issue their loads now, consume them after the entering row has been folded in
        // (the leaving row t - kh is a zero row if it was never added; the centre row is t - rh)
        RowRaw qo, qc;
        const int t_old = t - kh, y_c = t - rh;
        if constexpr (PF_OLD) {
            // the leaving row is fetched one iteration ahead (it comes from L2 / the Infinity Cache): qo_next holds row t_old
            qo = qo_next;
            if constexpr (POD > 1) {
                qo_next = qoq[0];
#pragma unroll
                for (int d = 1; d < POD - 1; ++d) qoq[d - 1] = qoq[d];
                qoq[POD - 2] = load_row<HK_NT_LEAVE, SB>(sp, rp, a.stride, t_old + POD, H, xq);
            } else {
                qo_next = load_row<HK_NT_LEAVE, SB>(sp, rp, a.stride, (HK_ABLATE & 1) ? t + 1 : t_old + 1, H, xq);
            }
        } else if constexpr (!ring && !sring) {
            qo = load_row<HK_NT_LEAVE, SB>(sp, rp, a.stride, t_old, H, xq);
        }
        if constexpr (RING == 0) qc = load_row<false, SB>(sp, rp, a.stride, y_c, H, xq);

        const RowZ znew = process_row<MODEL, DENSE, MODEL == 1 && R2>(q0, t >= 0 && t < H, colbits, full_wave, ts, tr, n0, n1);
        HK_STAMP(0);  // requests of this iteration issued, entering row arrived and classified
        if constexpr (PFD > 1) {
            q0 = qq[0];
#pragma unroll
            for (int d = 1; d < PFD - 1; ++d) qq[d - 1] = qq[d];
            qq[PFD - 2] = load_row<false, SB>(sp, rp, a.stride, min(t + PFD, t_last), H, xq);
        } else {
            q0 = load_row<false, SB>(sp, rp, a.stride, min(t + 1, t_last), H, xq);  // next row (see above)
        }
        if constexpr (UNIFORM_N && !DENSE) {
            if (!znew.clean) last_dirty = t;  // wave-uniform
        }
        RowZ zold;
        [[maybe_unused]] float4 mid_s, mid_r;      // RING 3: the row t - rh as it leaves the registers
        [[maybe_unused]] bool mid_clean = false;
        if constexpr (ring) {
            // leaving row (t - kh) = the slot the entering row overwrites.  The HBM-bound kernels fetch it from LDS one
            // iteration ahead (RING_AHEAD above; 3.06 -> 2.99 ms for `gain` by the latency alone); the VALU-bound kernels do
            // not gain from that and keep the registers.
            const bool slot_clean = (ring_clean >> slot) & 1ull;
            if constexpr (RING_AHEAD) {
                // zold_next was read an iteration ago; now fetch the row that leaves at the NEXT iteration from the slot the
                // entering row is about to take (LDS operations of a wave execute in order)
                zold = zold_next;
                const float4 os = ring_v[(slot * 2 + 0) * WAVE + lane];
                const float4 orr = ring_v[(slot * 2 + 1) * WAVE + lane];
                ring_decode(os, slot_clean, zold_next.s, zold_next.m);
                zold_next.r[0] = orr.x, zold_next.r[1] = orr.y, zold_next.r[2] = orr.z, zold_next.r[3] = orr.w;
            } else {
                const float4 os = ring_v[(slot * 2 + 0) * WAVE + lane];
                const float4 orr = ring_v[(slot * 2 + 1) * WAVE + lane];
                ring_decode(os, slot_clean, zold.s, zold.m);
                zold.r[0] = orr.x, zold.r[1] = orr.y, zold.r[2] = orr.z, zold.r[3] = orr.w;
            }
            ring_v[(slot * 2 + 0) * WAVE + lane] = ring_encode(znew);
            ring_v[(slot * 2 + 1) * WAVE + lane] = make_float4(znew.r[0], znew.r[1], znew.r[2], znew.r[3]);
            if constexpr (!DENSE) ring_clean = (ring_clean & ~(1ull << slot)) | ((unsigned long long)znew.clean << slot);
        } else if constexpr (sring) {
            // registers: the row that entered rh iterations ago comes out -- it is the centre row of this iteration's output
            // and moves on to the LDS ring --, the entering row takes its slot
            const float4 ns = ring_encode(znew), nr = make_float4(znew.r[0], znew.r[1], znew.r[2], znew.r[3]);
            mid_clean = (rg_clean >> slot_r) & 1u;
            rg_clean = (rg_clean & ~(1u << slot_r)) | ((unsigned)znew.clean << slot_r);
#define HK_RG_SWAP(k) mid_s = rg_s[k], mid_r = rg_r[k], rg_s[k] = ns, rg_r[k] = nr
#define HK_RG_CASE(k) case k: HK_RG_SWAP(k); break;
            static_assert(SRING_MAX >= 5 && SRING_MAX <= 8, "register slots of the split ring");
            switch (slot_r) {   // slot_r < rh <= SRING_MAX (launch_rw); the last slot is the default
                HK_RG_CASE(0) HK_RG_CASE(1) HK_RG_CASE(2) HK_RG_CASE(3)
                case 4: if constexpr (SRING_MAX > 5) { HK_RG_SWAP(4); break; }
                case 5: if constexpr (SRING_MAX > 6) { HK_RG_SWAP(5 < SRING_MAX ? 5 : 0); break; }
                case 6: if constexpr (SRING_MAX > 7) { HK_RG_SWAP(6 < SRING_MAX ? 6 : 0); break; }
                default: HK_RG_SWAP(SRING_MAX - 1);
            }
#undef HK_RG_CASE
#undef HK_RG_SWAP
            // LDS: the row t - kh leaves, the row from the registers takes its slot
            const bool slot_clean = (ring_clean >> slot) & 1ull;
            const float4 os = ring_v[(slot * 2 + 0) * WAVE + lane];
            const float4 orr = ring_v[(slot * 2 + 1) * WAVE + lane];
            ring_decode(os, slot_clean, zold.s, zold.m);
            zold.r[0] = orr.x, zold.r[1] = orr.y, zold.r[2] = orr.z, zold.r[3] = orr.w;
            ring_v[(slot * 2 + 0) * WAVE + lane] = mid_s;
            ring_v[(slot * 2 + 1) * WAVE + lane] = mid_r;
            if constexpr (!DENSE) ring_clean = (ring_clean & ~(1ull << slot)) | ((unsigned long long)mid_clean << slot);
        } else {
            zold = process_row<MODEL, DENSE, MODEL == 1 && R2>(qo, t_old >= t_first && t_old >= 0 && t_old < H, colbits, full_wave, ts, tr, n0, n1);
            if constexpr (cring) {  // slot2 cycles over rh + 1 rows: the entering row replaces the centre row of rh + 1 ago
                ring_v[slot2 * WAVE + lane] = ring_encode(znew);
                if constexpr (!DENSE) {
                    if (clean_bits_ok) ring_clean = (ring_clean & ~(1ull << slot2)) | ((unsigned long long)znew.clean << slot2);
                }
            }
        }

        HK_STAMP(1);  // next row requested, ring traffic, leaving row arrived and classified
        if (kh == 1) {  // wave-uniform: a 1-row window IS the entering row -- no running sum, exact by construction
            cs.clear();
            cs.template update<true>(znew, n0, n1);
        } else {
            cs.template update<true>(znew, n0, n1);
            cs.template update<false>(zold, n0, n1);
        }

        HK_STAMP(2);  // column sums updated
        const int y = t - rh;
        if (y >= y0 && y >= a.out_y0 && y < a.out_y1) {  // wave-uniform: the first 2*rh iterations only prime the running sums
            // centre row of the window
            float sc[PX];
            unsigned mc;
            if constexpr (sring) {
                ring_decode(mid_s, mid_clean, sc, mc);
            } else if constexpr (ring || cring) {
                // RING 1: slot_c = (slot - rh) mod kh; RING 2: the slot after the one just written = (slot2 + 1) mod (rh + 1)
                int cs_slot = slot_c;
                if constexpr (cring) cs_slot = slot2 + 1 == rh + 1 ? 0 : slot2 + 1;
                const float4 cs4 = ring ? ring_v[(cs_slot * 2 + 0) * WAVE + lane] : ring_v[cs_slot * WAVE + lane];
                ring_decode(cs4, clean_bits_ok && ((ring_clean >> (cs_slot & 63)) & 1ull), sc, mc);
            } else {
                const RowZ zc = process_row<MODEL, DENSE, MODEL == 1 && R2>(qc, true, colbits, full_wave, ts, tr, n0, n1);
#pragma unroll
                for (int i = 0; i < PX; ++i) sc[i] = zc.s[i];
                mc = zc.m;
            }
            if constexpr (DENSE) mc = (colbits * 0x00204081u) & 0x01010101u;  // bit i -> byte i

            // gain-offset only ever uses the float32 images of S, R, P (boxFilter output depth = input depth): convert each
            // as soon as its horizontal sum exists and keep the scheduler from interleaving the five sums, so that at most
            // one of those float64 quadruples is live beside S2 / R2 (12-16 VGPRs less at the pressure peak)
            double HS[PX], HR[PX];
            [[maybe_unused]] float Sf0[PX], Rf0[PX], Pf0[PX];
            hsum_any<RW, double, (XCH & 1) != 0, DPPX, use_wline<RW, RING>()>(cs.S, HS, wl, lane, xch);
            if constexpr (GO) {
#pragma unroll
                for (int i = 0; i < PX; ++i) Sf0[i] = (float)HS[i];
                __builtin_amdgcn_sched_barrier(0);
            }
            hsum_any<RW, double, (XCH & 2) != 0, DPPX, use_wline<RW, RING>()>(cs.R, HR, wl, lane, xch);
            if constexpr (GO) {
#pragma unroll
                for (int i = 0; i < PX; ++i) Rf0[i] = (float)HR[i];
                __builtin_amdgcn_sched_barrier(0);
            }
            double HP[PX], HS2[PX], HR2[PX];
            float Nf[PX];
            if constexpr (CS::NEED_P) hsum_any<RW, double, (XCH & 4) != 0, DPPX, use_wline<RW, RING>()>(cs.P, HP, wl, lane, xch);
            if constexpr (GO) {
#pragma unroll
                for (int i = 0; i < PX; ++i) Pf0[i] = (float)HP[i];
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (CS::NEED_S2) hsum_any<RW, double, (XCH & 8) != 0, DPPX, use_wline<RW, RING>()>(cs.S2, HS2, wl, lane, xch);
            // Certificate-only build: the window sum of ref^2 feeds nothing but the float32 r2-mask certificate, so its
            // horizontal stage runs in float32 on the rounded column sums (non-negative terms: <= 5 roundings, relative
            // error <= 4.03 * 2^-24 instead of 2^-24 -- DESIGN.md appendix A budgets it): four converts + nine float32 adds,
            // the neighbours' values as DPP operands, instead of nine float64 adds + eight DPP moves + four converts
            [[maybe_unused]] float HR2f[PX];
            if constexpr (CS::NEED_R2S && CERT_ONLY && HK_CERT_R2_F32) {
                float V2[PX];
#pragma unroll
                for (int i = 0; i < PX; ++i) V2[i] = (float)cs.R2s[i];
                hsum_any<RW, float, false, DPPX>(V2, HR2f, wl, lane);
            } else if constexpr (CS::NEED_R2S) {
                hsum_any<RW, double, (XCH & 16) != 0, DPPX, use_wline<RW, RING>()>(cs.R2s, HR2, wl, lane, xch);
                if constexpr (CERT_ONLY) {
#pragma unroll
                    for (int i = 0; i < PX; ++i) HR2f[i] = (float)HR2[i];
                }
            }
            if constexpr (USE_N) {
                if constexpr (DENSE) {
                    const int nrows_i = min(y + rh, H - 1) - max(y - rh, 0) + 1;
                    n_uniform = n_uniform_cols && nrows_i == kh;  // wave-uniform: every stored pixel has the full window
                    const float nrows = (float)nrows_i;
#pragma unroll
                    for (int i = 0; i < PX; ++i) Nf[i] = n_uniform ? a.n_full : nrows * ncolf[i];  // exact small integers
                } else {
                    if constexpr (UNIFORM_N) n_uniform = n_uniform_cols && last_dirty < t - kh + 1;  // wave-uniform
                    if (UNIFORM_N && n_uniform) {
#pragma unroll
                        for (int i = 0; i < PX; ++i) Nf[i] = a.n_full;
                    } else {
                        bool packed = false;
                        if constexpr (HK_PACKED_NSUM && RW >= 1 && RW <= 3) {
                            if (kh * (2 * rw + 1) <= 255) {  // wave-uniform: every window count fits a byte
                                // the four column counts travel and add as the bytes of one word: the 2 * RW + 1 shifted views
                                // of the 12 columns (left lane | own | right lane) are v_alignbyte_b32 of neighbouring words,
                                // their byte-wise sum is a plain 32-bit add (no byte can carry), v_cvt_f32_ubyteN unpacks
                                const unsigned own = cs.N;
                                const unsigned lw = (unsigned)dpp_from_left((int)own), rw_ = (unsigned)dpp_from_right((int)own);
                                unsigned hn = own;
#pragma unroll
                                for (int k = 1; k <= RW; ++k)
                                    hn += __builtin_amdgcn_alignbyte(rw_, own, k) + __builtin_amdgcn_alignbyte(own, lw, 4 - k);
#pragma unroll
                                for (int i = 0; i < PX; ++i) Nf[i] = (float)((hn >> (8 * i)) & 0xffu);
                                packed = true;
                            }
                        }
                        if (!packed) {
                            const int VN[PX] = {(int)(cs.N & 0xffu), (int)((cs.N >> 8) & 0xffu), (int)((cs.N >> 16) & 0xffu),
                                                (int)(cs.N >> 24)};
                            int HN[PX];
                            hsum_any<RW, int, false, DPPX>(VN, HN, wl, lane);
#pragma unroll
                            for (int i = 0; i < PX; ++i) Nf[i] = (float)HN[i];
                        }
                    }
                }
            }

            // The pointwise stages exist in two versions: UN = every stored pixel of this wave-row has the full, all-valid
            // window (n_uniform): the window count, its float64 image and 1/N are scalars and the centre-row mask is all ones,
            // so every mask select and bit test folds away.
            HK_STAMP(3);  // centre row, horizontal sums, window counts
            auto pointwise = [&](auto uniform_n) {
                constexpr bool UN = decltype(uniform_n)::value;
                // the halo lanes only feed their neighbours' horizontal sums: masked out of the pointwise stages, they draw
                // no power there (the kernel runs at the package's power cap; -0.6 %)
                // (not where 1/N comes from the lane-resident table: ds_bpermute returns 0 for a source lane that is switched off,
                // so that version keeps every lane alive -- stores and counters are guarded by out_lane anyway)
                if (!(LANE_LUT && !UN && lane_lut) && !out_lane) return;
                const unsigned mcu = UN ? 0x01010101u : mc;
                // ---- stage A: gains and offsets -------------------------------------------------------------------------
                float g[PX], o[PX], r2v[PX], c[PX];
                [[maybe_unused]] float Rf[PX], Sf[PX], Pf[PX], gp[PX];
                // r2-mask decision of this wave-row (gain-offset with a threshold and no R2 plane to write): first through the
                // float32 CERTIFICATE below, evaluated right behind each pixel pair's gain while its operands are in
                // registers; the reference's own R2 expression (stage B) runs only if a pixel stays uncertain
                [[maybe_unused]] bool try_cert = false, uncertain = false;
                [[maybe_unused]] unsigned cert_failed = 0u;  // complete build: byte i = 0xff where pixel i is certified FAILING
                if constexpr (GO && R2) {
                    if (count_fails && !want_r2_values) {  // wave-uniform
                        if (!CERT_ONLY && cert_skip > 0) --cert_skip;  // the rows just above needed the exact evaluation: go straight to it
                        else try_cert = true;
                    }
                }
                if constexpr (GO) {
                    // kernel_model.py:338-351; src2_sum is float64 (sqrBoxFilter) so m_den and the division are f64.
                    // float32 steps run two pixels per instruction (packed), float64 steps per pixel.
                    {
                        [[maybe_unused]] unsigned gwin = 0u, twin = 0u;  // largest distance of a pixel's g / N*T' from its window's low end
                        uncertain = !cert_ok;
#pragma unroll
                        for (int j = 0; j < PX / 2; ++j) {
                            const f2 Rf2 = HK_P2(Rf0, j), Sf2 = HK_P2(Sf0, j), Pf2 = HK_P2(Pf0, j);
                            const f2 Nf2 = UN ? f2{a.n_full, a.n_full} : HK_P2(Nf, j);
                            const double Ndx = UN ? a.nd_full : (double)Nf2.x, Ndy = UN ? a.nd_full : (double)Nf2.y;
                            const f2 num2 = Nf2 * Pf2 - Sf2 * Rf2;
                            const f2 SS2 = Sf2 * Sf2;
                            const double denx = __dsub_rn(__dmul_rn(Ndx, HS2[2 * j]), (double)SS2.x);
                            const double deny = __dsub_rn(__dmul_rn(Ndy, HS2[2 * j + 1]), (double)SS2.y);
                            const double qx = fast_quot((double)num2.x, denx), qy = fast_quot((double)num2.y, deny);
                            f2 g2 = {(float)qx, (float)qy};
                            // a lane with a quotient too close to a float32 rounding boundary, or outside the float32 normal range
                            // (zero / infinite / NaN / denormal quotients), divides this pixel pair again, the IEEE way, while the
                            // operands are still in registers.  The certificate build leaves the range test to its certificate,
                            // which passes gains inside (2^-20, 2^20) only and has no other consumer of the rest (their rows are
                            // marked for the list launch).
                            bool again = min(quot_guard(qx), quot_guard(qy)) < 2u * HK_DIV_GUARD + 1u;
                            if constexpr (!CERT_ONLY) again |= max(quot_range(qx), quot_range(qy)) > 0x0fd00000u;
                            if (again) {
                                g2.x = (float)__ddiv_rn((double)num2.x, denx);
                                g2.y = (float)__ddiv_rn((double)num2.y, deny);
                            }
                            const f2 t2 = g2 * Sf2;
                            const f2 tn2 = Rf2 - t2;
                            f2 o2;
                            if constexpr (UN) {
                                o2.x = (float)__dmul_rn((double)tn2.x, a.inv_n_full);
                                o2.y = (float)__dmul_rn((double)tn2.y, a.inv_n_full);
                            } else if (LANE_LUT && lane_lut) {  // wave-uniform
                                o2.x = (float)__dmul_rn((double)tn2.x, bperm_from<double>(lut_lane, (int)Nf2.x));
                                o2.y = (float)__dmul_rn((double)tn2.y, bperm_from<double>(lut_lane, (int)Nf2.y));
                            } else if (use_lut) {  // wave-uniform
                                o2.x = (float)__dmul_rn((double)tn2.x, inv_lut[(int)Nf2.x]);
                                o2.y = (float)__dmul_rn((double)tn2.y, inv_lut[(int)Nf2.y]);
                            } else {
                                o2.x = __fdiv_rn(tn2.x, Nf2.x);
                                o2.y = __fdiv_rn(tn2.y, Nf2.y);
                            }
                            g[2 * j] = g2.x, g[2 * j + 1] = g2.y;
                            o[2 * j] = o2.x, o[2 * j + 1] = o2.y;
                            if constexpr (R2) {
                                if (try_cert) {  // wave-uniform
                                    // Division-free CERTIFICATE of `(r2 > thresh) & (gain > 0)` from float32 quantities (proof:
                                    // DESIGN.md appendix A).  With T = g^2*S2 + R2 + N*o^2 the reference's arithmetic obeys
                                    //   ssres_ref <= sstot_ref - g^2*den + 23.5*2^-24*N*T,  |sstot_ref - sst| <= 4.1*2^-24*N*T,
                                    // and g = fl32(fl64(num / den)) makes g^2*den = g*num*(1 + eps), |eps| <= 1.01*2^-24, so
                                    //   fl32(g*num) > kappa*sst + 2^-17*N*T'   (kappa = 1 - r2_pass_scale, rounded up)
                                    // proves ssres_ref < r2_pass_scale * sstot_ref, i.e. the reference's decision.  N*T' is
                                    // N*T = g^2*den_x + (g*S)^2 + N*R2 + (N*o)^2 rebuilt from this pixel pair's float32
                                    // operands: g*num + t^2 + N*R2 + tn^2 (t = g*S, tn = R - t), within 4*2^-24 of it.
                                    f2 R2f;
                                    if constexpr (CERT_ONLY) R2f = HK_P2(HR2f, j);
                                    else R2f = f2{(float)HR2[2 * j], (float)HR2[2 * j + 1]};
                                    const f2 lhs = g2 * num2;
                                    const f2 sst = pk_fma(Nf2, R2f, -(Rf2 * Rf2));
                                    const f2 NT = pk_fma(t2, t2, pk_fma(tn2, tn2, pk_fma(Nf2, R2f, lhs)));
                                    const f2 slack = NT * 0x1p-17f;
                                    const f2 rhs = pk_fma(f2{a.r2_fail_scale, a.r2_fail_scale}, sst, slack);
                                    // The mirror image (complete build; appendix A, "the fail side"): a pixel certainly FAILS if
                                    // its gain -- bit-exact -- is not positive, or if 0 < lhs < kappa_f * sst - slack.  A wave-row
                                    // whose every valid pixel is certain one way or the other needs no exact evaluation either:
                                    // on rasters with failing pixels (real imagery has them block after block) that is nearly
                                    // every row.
                                    [[maybe_unused]] f2 rhs_f;
                                    if constexpr (!CERT_ONLY) rhs_f = pk_fma(f2{a.r2_failcert_scale, a.r2_failcert_scale}, sst, -slack);
#pragma unroll
                                    for (int e = 0; e < 2; ++e) {
                                        const bool m = (mcu >> (8 * (2 * j + e))) & 1u;
                                        const bool sure = (lhs[e] > rhs[e]) & (sst[e] > slack[e]);
                                        // magnitude windows (no underflow / overflow anywhere in the reference's expression):
                                        // 2^-20 < g < 2^20 (also the `gain > 0` half of the decision), 2^-40 < N*T' < 2^60;
                                        // a masked pixel's quantities are arbitrary and must not count
                                        const unsigned gd = __float_as_uint(g2[e]) - 0x35800000u, td = __float_as_uint(NT[e]) - 0x2b800000u;
                                        if constexpr (CERT_ONLY) {
                                            // the certificate build certifies PASSING rows only (the fail side and the source flags
                                            // of the in-painting it feeds do not fit its 128 registers: 3 to 10 spilled with them)
                                            gwin = max(gwin, UN ? gd : (m ? gd : 0u));
                                            twin = max(twin, UN ? td : (m ? td : 0u));
                                            uncertain |= m & !sure;
                                        } else {
                                            // a pixel whose gain is not positive (or NaN) fails whatever its R2: no error model,
                                            // no window needed for it
                                            const bool gpos = g2[e] > 0.f;
                                            const bool sure_f = !gpos | ((lhs[e] < rhs_f[e]) & (lhs[e] > 0.f) & (sst[e] > slack[e]));
                                            const bool mg = UN ? gpos : (m & gpos);
                                            gwin = max(gwin, mg ? gd : 0u);
                                            twin = max(twin, mg ? td : 0u);
                                            uncertain |= m & !((sure & gpos) | sure_f);
                                            cert_failed |= ((m & sure_f) ? 0xffu : 0u) << (8 * (2 * j + e));
                                        }
                                    }
                                }
                            }
                        }
                        if constexpr (R2) {
                            if (try_cert) uncertain |= (gwin >= 0x49800000u - 0x35800000u) | (twin >= 0x5d800000u - 0x2b800000u);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < PX; ++i) Rf[i] = Rf0[i], Sf[i] = Sf0[i], Pf[i] = Pf0[i];
                }
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    if constexpr (!GO) Rf[i] = (float)HR[i];  // boxFilter output depth = input depth (float32)
                    if constexpr (GO) {
                    } else if constexpr (MODEL == 1) {
                        // kernel_model.py:265 with a float64 src_sum: np.divide(f32, f64, out=f32); then :301-302
                        double ssum = HS[i];
                        if constexpr (BLKA)
                            ssum = __dadd_rn(__dmul_rn(n0, HS[i]), UN ? n1_n_full : __dmul_rn(n1, (double)Nf[i]));
                        const double q = (HK_ABLATE & 2) ? __dadd_rn((double)Rf[i], ssum) : fast_quot((double)Rf[i], ssum);
                        gp[i] = (float)q;
                        if ((quot_guard(q) < 2u * HK_DIV_GUARD + 1u) | (quot_range(q) > 0x0fd00000u))
                            gp[i] = (float)__ddiv_rn((double)Rf[i], ssum);
                        o[i] = (float)__dmul_rn((double)gp[i], n1);
                        g[i] = (float)__dmul_rn((double)gp[i], n0);
                    } else {
                        // kernel_model.py:262-265
                        Sf[i] = (float)HS[i];
                        if constexpr (R2) Pf[i] = (float)HP[i];
                        g[i] = __fdiv_rn(Rf[i], Sf[i]);
                        o[i] = 0.f;
                    }
                    r2v[i] = qnan();
                }

                // source mask of the in-painting (:363): a certified wave-row passes wherever it is valid
                [[maybe_unused]] unsigned passed = mcu;
                // ---- stage B: R2 (kernel_model.py:179,189-195|201,203,212-213) ---------------------------------------------
                if constexpr (R2) {
                    if (want_r2_values || count_fails) {  // wave-uniform
                        bool exact = true;
                        if constexpr (GO) {
                            if (try_cert) {
                                exact = __any(uncertain & out_lane);
                                if (!CERT_ONLY && exact) cert_skip = HK_CERT_SKIP;  // failing regions are coherent: skip the certificate for a few rows
                                if (!exact) {  // every valid pixel of the wave-row is certified: the failing ones are known
                                    passed &= ~cert_failed;
                                    if (out_lane) nfail += (unsigned)__popc(cert_failed & 0x01010101u);
                                }
                            }
                        }
                        if constexpr (CERT_ONLY) {
                            // this build holds no exact evaluation (which costs the whole kernel a wave per SIMD): a wave-row the
                            // two-sided certificate cannot settle is marked for the LIST launch that follows (the complete build
                            // over the runs of marked rows, FitArgs::open_rows) -- nothing of it is stored or counted here
                            if (exact) {
                                if (lane == ol) {  // the first output lane (always active here: a strip has output columns)
                                    // The word's address is formed HERE, from a value the optimiser cannot see through: hoisted out of the
                                    // row loop its loop-invariant part sat in a register pair for the sake of this rare path (the two
                                    // registers the NaN-aware builds wider than 15 lacked for a third wave per SIMD, HISTORY.md 70).
                                    int unit_u = band * a.n_strips + strip;   // (wave-uniform: a scalar register)
                                    if constexpr (SB) asm volatile("" : "+s"(unit_u));
                                    atomicOr(a.open_rows + ((size_t)unit_u * (size_t)((H + 31) >> 5) + (size_t)(y >> 5)), 1u << (y & 31));
                                }
                                return;  // (skipping stage C also keeps the allocation spill-free)
                            }
                        } else if (exact) {
                            double sstot[PX], ssres[PX];
#pragma unroll
                            for (int i = 0; i < PX; ++i) {
                                sstot[i] = __dsub_rn(__dmul_rn((double)Nf[i], HR2[i]), (double)__fmul_rn(Rf[i], Rf[i]));
                                double q;
                                if constexpr (GO) {
                                    const double A = __dmul_rn((double)__fmul_rn(g[i], g[i]), HS2[i]);
                                    const float B = __fmul_rn(__fmul_rn(2.f, __fmul_rn(g[i], o[i])), Sf[i]);
                                    const float C = __fmul_rn(__fmul_rn(2.f, g[i]), Pf[i]);
                                    const float D = __fmul_rn(__fmul_rn(2.f, o[i]), Rf[i]);
                                    const float F = __fmul_rn(Nf[i], __fmul_rn(o[i], o[i]));
                                    q = __dadd_rn(A, (double)B);
                                    q = __dsub_rn(q, (double)C);
                                    q = __dsub_rn(q, (double)D);
                                    q = __dadd_rn(q, HR2[i]);
                                    q = __dadd_rn(q, (double)F);
                                } else if constexpr (BLK) {
                                    // float64 src2_sum / src_ref_sum (the normalised source is float64)
                                    q = __dmul_rn((double)__fmul_rn(gp[i], gp[i]), HS2[i]);
                                    q = __dsub_rn(q, __dmul_rn((double)__fmul_rn(2.f, gp[i]), HP[i]));
                                    q = __dadd_rn(q, HR2[i]);
                                } else {
                                    q = __dmul_rn((double)__fmul_rn(g[i], g[i]), HS2[i]);
                                    q = __dsub_rn(q, (double)__fmul_rn(__fmul_rn(2.f, g[i]), Pf[i]));
                                    q = __dadd_rn(q, HR2[i]);
                                }
                                ssres[i] = __dmul_rn(q, (double)Nf[i]);
                            }
                            // Without an R2 plane to write only the DECISION r2 > thresh is needed: ssres against the two
                            // bounds that bracket the float32 rounding boundary of the reference's quotient (hk_api.hip) settles
                            // it without the division; a pixel in the 2^-39-wide gap, or with sstot <= 0 / NaN, sends the
                            // wave-row through the division.
                            bool r2_ok[PX];
                            bool divide = !GO || a.r2 != nullptr;
                            if constexpr (GO) {
                                if (!divide) {
                                    bool unsure = false;
#pragma unroll
                                    for (int i = 0; i < PX; ++i) {
                                        const bool pos = sstot[i] > 0.0;
                                        const bool yes = pos & (ssres[i] < __dmul_rn(a.r2_pass_below, sstot[i]));
                                        const bool no = pos & (ssres[i] > __dmul_rn(a.r2_fail_above, sstot[i]));
                                        r2_ok[i] = yes;
                                        unsure |= out_lane & (bool)((mcu >> (8 * i)) & 1u) & !(yes | no);
                                    }
                                    divide = __any(unsure);
                                }
                            }
#pragma unroll
                            for (int i = 0; i < PX; ++i) {
                                if (divide) {
                                    r2v[i] = __fsub_rn(1.f, (float)__ddiv_rn(ssres[i], sstot[i]));
                                    r2_ok[i] = r2v[i] > a.r2_thresh;
                                }
                                if constexpr (GO) {
                                    const bool m = (mcu >> (8 * i)) & 1u;
                                    if (!(r2_ok[i] && (g[i] > 0.f))) passed &= ~(0xffu << (8 * i));
                                    // valid pixels failing (r2 > thresh) & (gain > 0) need in-painting (:363,:370)
                                    const bool failing = count_fails && m && !(r2_ok[i] && (g[i] > 0.f));
                                    if (failing && out_lane) ++nfail;
                                    if (failing && inpaint_pass && out_lane) {
                                        const float oin = a.offset_in[out_base + (long long)y * a.stride + x + i];
                                        o[i] = oin;
                                        g[i] = __fdiv_rn(__fsub_rn(Rf[i], __fmul_rn(Nf[i], oin)), Sf[i]);
                                    }
                                }
                            }
                        }
                    }
                }

                if constexpr (GO && !R2) {
                    // Closing pass of the in-painting branch (kernel_model.py:366-371) without the R2 work: which valid pixels
                    // failed the r2 mask is read from the in-painting's source flags (a.flag_in, written by the pass that
                    // counted them); they take the in-painted offset and gain = (ref_sum - mask_sum * offset) / src_sum.
                    if (a.offset_in != nullptr) {  // wave-uniform
                        const long long q = out_base + (long long)y * a.stride + (lane_in ? x : 0);  // a safe quad for lanes outside
                        const float4 oin4 = *reinterpret_cast<const float4*>(a.offset_in + q);
                        const unsigned fl = *reinterpret_cast<const unsigned*>(a.flag_in + q);
                        const float oin[PX] = {oin4.x, oin4.y, oin4.z, oin4.w};
#pragma unroll
                        for (int i = 0; i < PX; ++i) {
                            const bool failing = ((mcu >> (8 * i)) & 1u) && !((fl >> (8 * i)) & 0xffu);
                            if (failing) {
                                o[i] = oin[i];
                                g[i] = __fdiv_rn(__fsub_rn(Rf[i], __fmul_rn(Nf[i], oin[i])), Sf[i]);
                            }
                        }
                    }
                }

                // ---- stage C: apply (:461) and where=mask (every parameter write goes into a NaN-filled array, :261,:345) ----
                // A masked pixel has NaN parameters, hence a NaN corrected value: select once per stored plane.
#pragma unroll
                for (int j = 0; j < PX / 2; ++j) {
                    const f2 c2 = HK_P2(g, j) * HK_P2(sc, j) + HK_P2(o, j);  // two float32 roundings
                    c[2 * j] = c2.x, c[2 * j + 1] = c2.y;
                }
                if constexpr (!DENSE) {
#pragma unroll
                    for (int i = 0; i < PX; ++i) c[i] = ((mcu >> (8 * i)) & 1u) ? c[i] : qnan();
                }
                auto masked4 = [&](const float (&v)[PX]) {
                    float4 r4 = make_float4(v[0], v[1], v[2], v[3]);
                    if constexpr (!DENSE) {
                        r4.x = (mcu & 0x00000001u) ? r4.x : qnan();
                        r4.y = (mcu & 0x00000100u) ? r4.y : qnan();
                        r4.z = (mcu & 0x00010000u) ? r4.z : qnan();
                        r4.w = (mcu & 0x01000000u) ? r4.w : qnan();
                    }
                    return r4;
                };

                if (out_lane) {
                    // stride % 4 == 0: a quad never crosses the row end, columns >= W land in the row padding
                    // wave-uniform row offset (scalar) + this lane's 32-bit byte offset: no per-plane address registers
                    const long long row_off = out_base + (long long)y * a.stride;
                    auto at = [&](float* plane) {
                        if constexpr (SB) return row_address(plane + row_off) + xbytes;
                        else return reinterpret_cast<float4*>(reinterpret_cast<char*>(plane + row_off) + xbytes);
                    };
                    if (a.corr && !((HK_ABLATE & 8) && c[0] != 123.456f)) store4_nt(at(a.corr), make_float4(c[0], c[1], c[2], c[3]));
                    if constexpr (!CERT_ONLY) {  // (the certificate build serves launches without a gain / R2 plane: launch_one)
                        if (a.gain) store4_nt(at(a.gain), masked4(g));
                    }
                    if (a.offset) store4_nt(at(a.offset), masked4(o));
                    if constexpr (!CERT_ONLY) {
                        if (R2 && a.r2) store4_nt(at(a.r2), masked4(r2v));
                    }
                    if constexpr (GO && R2 && !CERT_ONLY) {
                        // 1: source of the in-painting, 0: target, 2: invalid -- neither (what a fill would put there is reset to NaN anyway, kernel_model.py:367)
                        if (a.flag) *reinterpret_cast<unsigned*>(a.flag + row_off + (xbytes >> 2)) = passed | ((~mcu & 0x01010101u) << 1);
                    }
                }
            };
            // list launch: a row of the run that is not marked was settled, stored and counted by the certificate build
            bool skip_row = false;
            if constexpr (LIST)  // wave-uniform (scalar load)
                skip_row = ((a.open_rows[(size_t)(band * a.n_strips + strip) * (size_t)((H + 31) >> 5) + (size_t)(y >> 5)] >> (y & 31)) & 1u) == 0u;
            if (skip_row) {
            } else if constexpr (UNIFORM_N) {
                if (n_uniform) pointwise(std::true_type{});
                else pointwise(std::false_type{});
            } else {
                pointwise(std::false_type{});
            }
            HK_STAMP(4);  // pointwise stages and stores
        }

        if (++slot == ring_mod) slot = 0;
        if (++slot_c == ring_mod) slot_c = 0;
        if (++slot2 == rh + 1) slot2 = 0;
        if (++slot_r >= rh) slot_r = 0;
    }
#ifdef HK_STAMPS
    if (lane == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&hk_stamps[k], st_acc[k]);
        atomicAdd(&hk_stamps[15], st_iters);
        atomicAdd(&hk_stamps[14], 1ull);
    }
#endif

    if constexpr (GO && R2) {
        if (a.fail_count != nullptr && a.has_thresh) {
#pragma unroll
            for (int d = WAVE / 2; d > 0; d >>= 1) nfail += __shfl_xor(nfail, d);
            if (lane == 0 && nfail) atomicAdd(a.fail_count + band, (unsigned long long)nfail);

        }
    }
}

template <int MODEL, bool R2, int RW, bool DENSE, int RING, bool CERT_ONLY, int WPB, bool BATCH = false>
__global__ void __launch_bounds__(WAVE * WPB, (fit_min_waves<MODEL, R2, RW, DENSE, RING, CERT_ONLY>()))
fit_apply_kernel(const FitArgs a_in) {
    extern __shared__ float4 lds4[];

    const int lane = threadIdx.x & (WAVE - 1), wave_in_wg = (WPB == 1 && fit_scalar_bases<MODEL, R2, RW, DENSE, RING, CERT_ONLY>()) ? 0 : threadIdx.x >> 6;  // (a constant there: `strip` is scalar)
    int group = blockIdx.x;
    if (a_in.xcd_remap) {
        // workgroups go round-robin to the 8 XCDs (each with its own L2): hand every XCD runs of `xcd_remap` consecutive
        // units, i.e. neighbouring strips of one segment, whose shared cache lines (strips start 16-byte-, not 128-byte-
        // aligned, and overlap by two lanes) are then fetched from HBM once instead of once per strip
        const int g = a_in.xcd_remap / WPB > 0 ? a_in.xcd_remap / WPB : 1, slot = blockIdx.x >> 3;
        group = ((slot / g) * 8 + (blockIdx.x & 7)) * g + slot % g;
    }
    // Batched launch (FitArgs::jobs; the BATCH builds): the workgroup's job is the last one whose first workgroup is not beyond
    // it -- a binary search over the job table with scalar loads (everything here is uniform over the workgroup) --, and the job's
    // planes, shape and unit grid replace the launch's.  A build of its own: with the look-up compiled into every kernel (argument
    // block copied and patched) the memory-bound builds lost up to 8 % (gain 5x5 at 8192^2 x 4; profiles/r03_batch.txt).
    [[maybe_unused]] FitArgs a_job;
    if constexpr (BATCH) {
        constexpr int FG = WPB > 1 ? 1 : 0;
        if (group >= a_in.batch_groups[FG]) return;  // the whole workgroup (grid padding)
        int lo = 0, hi = a_in.n_jobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a_in.jobs[mid].first_group[FG] <= group) lo = mid;
            else hi = mid - 1;
        }
        const FitJob& e = a_in.jobs[lo];
        group -= e.first_group[FG];
        a_job = a_in;
        a_job.src = table_pointer(e.src), a_job.ref = table_pointer(e.ref), a_job.gain = table_pointer(e.gain);
        a_job.offset = table_pointer(e.offset), a_job.r2 = table_pointer(e.r2), a_job.corr = table_pointer(e.corr);
        a_job.norm = table_pointer(e.norm), a_job.fail_count = table_pointer(e.fail_count), a_job.flag = table_pointer(e.flag);
        a_job.stride = e.stride, a_job.band_stride = e.band_stride, a_job.height = e.height, a_job.width = e.width;
        a_job.n_bands = e.n_bands, a_job.seg_rows = e.seg_rows, a_job.n_strips = e.n_strips, a_job.n_segs = e.n_segs;
        a_job.seg_rows_tail = e.seg_rows_tail, a_job.n_segs_big = e.n_segs_big;
        a_job.out_y0 = e.out_y0, a_job.out_y1 = e.out_y1, a_job.out_x0 = e.out_x0, a_job.out_x1 = e.out_x1;
    }
    const FitArgs& a = BATCH ? a_job : a_in;
    // a workgroup = WPB adjacent strips of one (segment, band); the strips of a row are padded to a multiple of WPB (a padded
    // strip lies outside the raster: every lane loads a clamped quad and stores nothing -- it only keeps the barriers whole)
    const int groups_per_row = (a.n_strips + WPB - 1) / WPB;
    if (group >= groups_per_row * a.n_segs * a.n_bands) return;  // the whole workgroup
    // segment-major order: the short tail segments (hk_api.hip fill_grid) are dispatched last
    const int strip = (group % groups_per_row) * WPB + wave_in_wg;
    const int t0 = group / groups_per_row;
    const int band = t0 % a.n_bands;
    const int seg = t0 / a.n_bands;
    const bool big = seg < a.n_segs_big;
    const int y0 = big ? seg * a.seg_rows : a.n_segs_big * a.seg_rows + (seg - a.n_segs_big) * a.seg_rows_tail;
    const int y1 = min(y0 + (big ? a.seg_rows : a.seg_rows_tail), a.height);
    fit_unit<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB, false>(a, band, strip, y0, y1, lane, wave_in_wg, lds4);
}

// LIST launch (round 6; the complete build of gain-offset with the r2 mask): the units are the runs of wave-rows the certificate
// build marked in FitArgs::open_rows -- one bit per (band, strip, row), 32 rows per word.  A persistent grid of single waves: a
// wave scans the words 64 at a time, takes every word that STARTS a run (non-zero, its predecessor in the strip zero), follows the
// run over the next non-zero words and runs it like a row segment of its own -- 2 rh priming rows, then the rows from the run's
// first to its last marked row, storing and counting the marked ones only.  A kernel of its own, so that the scan's state costs
// the full-grid builds nothing.
// (two waves per SIMD: with the scan's state a few of these builds are 1 - 3 registers short of the complete build's 168, and the
// rows of a list launch are few where it matters -- where most rows are open, callers let the complete build run at once: job
// scratch / the host path's expectation)
template <int MODEL, bool R2, int RW, bool DENSE, int RING>
__global__ void __launch_bounds__(WAVE, 2)
fit_list_kernel(const FitArgs a) {
    extern __shared__ float4 lds4[];
    const int lane = threadIdx.x;
    const int wps = (a.height + 31) >> 5;                     // words per (band, strip)
    const int n_words = a.n_bands * a.n_strips * wps;
    for (int chunk = (int)blockIdx.x * WAVE; chunk < n_words; chunk += (int)gridDim.x * WAVE) {
        const int idx = chunk + lane;
        const unsigned w = idx < n_words ? a.open_rows[idx] : 0u;
        const unsigned before = (idx < n_words && idx % wps != 0) ? a.open_rows[idx - 1] : 0u;
        unsigned long long starts = __ballot(w != 0u && before == 0u);
        while (starts) {
            const int pos = __ffsll((long long)starts) - 1;
            starts &= starts - 1ull;
            const int g = __builtin_amdgcn_readfirstlane(chunk + pos);
            const int unit = g / wps, w0 = g - unit * wps;
            const int band = unit / a.n_strips, strip = unit - band * a.n_strips;
            unsigned last = a.open_rows[g];
            const int y0 = w0 * 32 + (__ffs((int)last) - 1);
            int e = w0;
            while (e + 1 < wps) {
                const unsigned nx = a.open_rows[g + (e + 1 - w0)];
                if (nx == 0u) break;
                last = nx, ++e;
            }
            const int y1 = min(e * 32 + 32 - __clz((int)last), a.height);
            fit_unit<MODEL, R2, RW, DENSE, RING, false, 1, true>(a, band, strip, y0, y1, lane, 0, lds4);
        }
    }
}


// LDS bytes of one wave: the row ring of the mode (see fit_apply_kernel).  Validity travels inside the source plane
// (RING_SENTINEL) and the 1/N table sits in global memory, so every build of a kernel shape needs the same amount: 10 KB at
// 5x5 = 16 waves per CU (8 KB = 20 waves for the kernels that read the leaving row one iteration ahead).
inline size_t fit_lds_bytes_of(int kh, int ring_mode, bool ahead) {
    if (ring_mode == 1) return (size_t)(ahead && kh > 1 ? kh - 1 : kh) * 2 * WAVE * sizeof(float4);
    if (ring_mode == 2) return (size_t)(kh / 2 + 1) * WAVE * sizeof(float4);
    if (ring_mode == 3) return (size_t)(kh / 2 + 1) * 2 * WAVE * sizeof(float4);
    return 0;
}

// launch ledger (hk_kernels.h): one record per build of the fused kernel, on the list from the moment the library is loaded
template <int MODEL, bool R2, int RW, bool DENSE, int RING, bool CERT_ONLY, int WPB, bool BATCH>
struct FitBuildRecord {
    static BuildRecord rec;
};
template <int MODEL, bool R2, int RW, bool DENSE, int RING, bool CERT_ONLY, int WPB, bool BATCH>
BuildRecord FitBuildRecord<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB, BATCH>::rec{MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB, BATCH};

template <int MODEL, bool R2, int RW, bool DENSE, int RING>
struct ListBuildRecord {
    static BuildRecord rec;
};
template <int MODEL, bool R2, int RW, bool DENSE, int RING>
BuildRecord ListBuildRecord<MODEL, R2, RW, DENSE, RING>::rec{"fit_list_kernel", MODEL, R2, RW, DENSE, RING};

template <int MODEL, bool R2, int RW, bool DENSE, int RING, bool CERT_ONLY, int WPB>
static hipError_t launch_wpb(const FitArgs& a, size_t lds, hipStream_t stream) {
    if (lds * WPB > 64 * 1024) {  // forced LDS ring on a tall kernel (testing): raise the 64 KiB dynamic-LDS default
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fit_apply_kernel<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    if constexpr (MODEL == 2 && R2 && !CERT_ONLY && WPB == 1 && (RING == 1 || RING == 2)) {
        // list launch (the rows the certificate build marked): a persistent grid of single waves; an empty bit plane costs microseconds
        if (a.list_mode) {
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fit_list_kernel<MODEL, R2, RW, DENSE, RING>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
            ListBuildRecord<MODEL, R2, RW, DENSE, RING>::rec.hit();
            hipLaunchKernelGGL((fit_list_kernel<MODEL, R2, RW, DENSE, RING>), dim3(256 * 8), dim3(WAVE), lds, stream, a);
            return hipGetLastError();
        }
    }
    if (a.list_mode) return hipErrorInvalidValue;  // (hk_api.hip asks for a list launch only where the certificate build exists)
    int grid = (a.n_strips + WPB - 1) / WPB * a.n_segs * a.n_bands;  // workgroups of WPB adjacent strips
    constexpr bool CAN_BATCH = fit_batch_build(MODEL, R2) && !CERT_ONLY;
    if (a.jobs) {
        if (!CAN_BATCH) return hipErrorInvalidValue;  // hk_api.hip asks fit_batch_build() first
        grid = a.batch_groups[WPB > 1 ? 1 : 0];
    }
    if (a.xcd_remap) {
        const int g = a.xcd_remap / WPB > 0 ? a.xcd_remap / WPB : 1;
        grid = (grid + 8 * g - 1) / (8 * g) * (8 * g);
    }
    // a.lds_pad: extra dynamic LDS per wave that nothing uses -- it only lowers the number of resident waves per CU
    if constexpr (CAN_BATCH) {
        if (a.jobs) {
            if (lds * WPB > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fit_apply_kernel<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
            FitBuildRecord<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB, true>::rec.hit();
            hipLaunchKernelGGL((fit_apply_kernel<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB, true>), dim3(grid), dim3(WAVE * WPB),
                               (lds + (size_t)a.lds_pad) * WPB, stream, a);
            return hipGetLastError();
        }
    }
    FitBuildRecord<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB, false>::rec.hit();
    hipLaunchKernelGGL((fit_apply_kernel<MODEL, R2, RW, DENSE, RING, CERT_ONLY, WPB>), dim3(grid), dim3(WAVE * WPB),
                       (lds + (size_t)a.lds_pad) * WPB, stream, a);
    return hipGetLastError();
}

template <int MODEL, bool R2, int RW, bool DENSE, int RING, bool CERT_ONLY>
static hipError_t launch_build(const FitArgs& a, hipStream_t stream) {
    const size_t lds = fit_lds_bytes_of(2 * a.rh + 1, RING, RING == 1 && MODEL != 2 && !R2);
    // every build without the R2 work (gain-offset without a threshold gains 3 % as well: 2.77 -> 2.68 ms)
    constexpr bool LOCKSTEP = !R2 && RING == 1 && HK_WPB_MEM > 1;
    if constexpr (LOCKSTEP) {
        if (lds * HK_WPB_MEM <= 64 * 1024) return launch_wpb<MODEL, R2, RW, DENSE, RING, CERT_ONLY, HK_WPB_MEM>(a, lds, stream);
    }
    if constexpr (RW < 0) return launch_wpb<MODEL, R2, RW, DENSE, RING, CERT_ONLY, 1>(a, lds + (use_wline<RW, RING>() ? WLINE_BYTES : 0), stream);
    return launch_wpb<MODEL, R2, RW, DENSE, RING, CERT_ONLY, 1>(a, lds + (xch_mask<MODEL, RW, RING, 1>() ? XCH_BYTES : 0), stream);
}

// gain-offset with the r2 mask exists in two builds.  The FULL one carries the reference's R2 expression inline for the
// wave-rows the float32 certificate cannot settle (and for R2 output): 154 VGPRs, 3 waves per SIMD.  The CERTIFICATE-ONLY
// one (a.cert_only, chosen by the host when no R2 plane is written and the previous launch had no failures) has nothing
// but the certificate: 128 VGPRs and no LDS table = 4 waves per SIMD (-5 % on clean rasters); a wave-row it cannot settle
// sets FIT_RETRY_BIT in the band's fail counter and the host re-runs the band with the full build.
template <int MODEL, bool R2, int RW, bool DENSE, int RING>
static hipError_t launch_one(const FitArgs& a, hipStream_t stream) {
    if constexpr (MODEL == 2 && R2 && (RING == 1 || RING == 2)) {
        if (a.cert_only && a.has_thresh && a.fail_count && a.open_rows && !a.r2 && !a.gain && !a.flag && !a.offset_in && !a.list_mode)
            return launch_build<MODEL, R2, RW, DENSE, RING, true>(a, stream);
    }
    return launch_build<MODEL, R2, RW, DENSE, RING, false>(a, stream);
}

// kernels wider than 15 (and the everything-re-loaded path from 9 wide): rw mod 4 picks the build, rw / 4 is a launch argument
template <int MODEL, bool R2, bool DENSE, int RING>
static hipError_t launch_wide(const FitArgs& a, hipStream_t stream) {
    // the paired form of the whole-lane sums (hsum_wide): builds of their own on the centre ring (kernels up to 39 rows tall; the
    // LDS-line variant of the taller ones reads the same number of entries either way)
    if constexpr (RING == 2) {
        if (wide_pairs(a.rw & 3, a.rw / PX)) {
            switch (a.rw & 3) {
                case 0: return launch_one<MODEL, R2, -5, DENSE, RING>(a, stream);
                case 1: return launch_one<MODEL, R2, -6, DENSE, RING>(a, stream);
                case 2: return launch_one<MODEL, R2, -7, DENSE, RING>(a, stream);
                default: return launch_one<MODEL, R2, -8, DENSE, RING>(a, stream);
            }
        }
    }
    switch (a.rw & 3) {
        case 0: return launch_one<MODEL, R2, -1, DENSE, RING>(a, stream);
        case 1: return launch_one<MODEL, R2, -2, DENSE, RING>(a, stream);
        case 2: return launch_one<MODEL, R2, -3, DENSE, RING>(a, stream);
        default: return launch_one<MODEL, R2, -4, DENSE, RING>(a, stream);
    }
}

template <int MODEL, bool R2, bool DENSE>
static hipError_t launch_rw(const FitArgs& a, hipStream_t stream) {
#ifdef HK_DEV_SUBSET  // development builds: only the 5x5 kernels with the full LDS ring (seconds instead of a minute)
#ifdef HK_DEV_SUBSET15   // ... or only the 15-wide kernels with the centre / split ring
    if constexpr (MODEL != 2 && !R2) {
        if (a.use_ring == 3) return launch_one<MODEL, R2, 7, DENSE, 3>(a, stream);
    }
    return launch_one<MODEL, R2, 7, DENSE, 2>(a, stream);
#else
    return launch_one<MODEL, R2, 2, DENSE, 1>(a, stream);
#endif
#else
    // a.use_ring (hk_api.hip): 1 full LDS ring (short, narrow kernels only), 2 centre ring + re-loaded leaving row,
    // 0 everything re-loaded (very tall kernels; wide-kernel builds only, to bound the number of instantiations)
    if (a.use_ring == 1) {
        switch (a.rw) {
            case 0: return launch_one<MODEL, R2, 0, DENSE, 1>(a, stream);
            case 1: return launch_one<MODEL, R2, 1, DENSE, 1>(a, stream);
            case 2: return launch_one<MODEL, R2, 2, DENSE, 1>(a, stream);
            case 3: return launch_one<MODEL, R2, 3, DENSE, 1>(a, stream);
            default: break;
        }
        if constexpr (MODEL != 2 && !R2) {  // the memory-bound builds keep the full ring for wider kernels too (hk_api.hip)
            switch (a.rw) {
                case 4: return launch_one<MODEL, R2, 4, DENSE, 1>(a, stream);
                case 5: return launch_one<MODEL, R2, 5, DENSE, 1>(a, stream);
                case 6: return launch_one<MODEL, R2, 6, DENSE, 1>(a, stream);
                case 7: return launch_one<MODEL, R2, 7, DENSE, 1>(a, stream);
                default: break;
            }
        }
    }
    if (a.use_ring == 0 && a.rw >= PX) return launch_wide<MODEL, R2, DENSE, 0>(a, stream);  // (narrower: centre ring, below)
    if constexpr (MODEL != 2 && !R2) {  // split ring (hk_api.hip fill_args): rh rows in registers
        if (a.use_ring == 3 && a.rh <= split_ring_rows(MODEL)) {
            switch (a.rw) {
                case 2: return launch_one<MODEL, R2, 2, DENSE, 3>(a, stream);
                case 3: return launch_one<MODEL, R2, 3, DENSE, 3>(a, stream);
                case 4: return launch_one<MODEL, R2, 4, DENSE, 3>(a, stream);
                case 5: return launch_one<MODEL, R2, 5, DENSE, 3>(a, stream);
                case 6: return launch_one<MODEL, R2, 6, DENSE, 3>(a, stream);
                case 7: return launch_one<MODEL, R2, 7, DENSE, 3>(a, stream);
                default: break;   // other widths: centre ring
            }
        }
    }
    switch (a.rw) {
        case 0: return launch_one<MODEL, R2, 0, DENSE, 2>(a, stream);
        case 1: return launch_one<MODEL, R2, 1, DENSE, 2>(a, stream);
        case 2: return launch_one<MODEL, R2, 2, DENSE, 2>(a, stream);
        case 3: return launch_one<MODEL, R2, 3, DENSE, 2>(a, stream);
        case 4: return launch_one<MODEL, R2, 4, DENSE, 2>(a, stream);
        case 5: return launch_one<MODEL, R2, 5, DENSE, 2>(a, stream);
        case 6: return launch_one<MODEL, R2, 6, DENSE, 2>(a, stream);
        case 7: return launch_one<MODEL, R2, 7, DENSE, 2>(a, stream);
        default: return launch_wide<MODEL, R2, DENSE, 2>(a, stream);
    }
#endif
}

template <int MODEL, bool R2>
static hipError_t launch_dense(const FitArgs& a, hipStream_t stream) {
    // nodata None on both rasters (and not gain-blk-offset, whose mask depends on the normalised values)
    if constexpr (MODEL != 1) {
        if (a.src_nd_mode == 0 && a.ref_nd_mode == 0 && !a.force_general) return launch_rw<MODEL, R2, true>(a, stream);
    }
    return launch_rw<MODEL, R2, false>(a, stream);
}

}  // namespace hk
