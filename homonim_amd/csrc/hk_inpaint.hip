// hk_inpaint.hip -- in-painting of the offset band where the kernel models are poor (homonim/kernel_model.py:361-371).
//
// The reference calls rasterio.fill.fillnodata(offset, r2_mask) == GDALFillNodata(max_search_distance = 100,
// smoothing_iterations = 0).  GDAL is not part of /root/reference (rasterio>=1.1, un-pinned) and not installed here,
// so its published algorithm (gdal/alg/rasterfill.cpp) is RESTATED -- parity with GDAL itself is unpinned:
//   * two column scans give, for every pixel, the nearest source pixel (mask != 0) at-or-above it and strictly below
//     it in its own column, carried at most `max_dist` rows (kept here as uint16 row distances);
//   * a target pixel (mask == 0) steps left and right one column at a time (0..max_dist) and keeps, per quadrant
//     (top-left and bottom-left include the pixel's own column, the right quadrants do not), the closest source found
//     through those column tables -- strict `<` on the squared distance, so the first one met wins ties;
//   * value = sum(v_q / d_q) / sum(1 / d_q) over the quadrants with d_q <= max_dist, in float64, cast to float32;
//     targets without any source keep their value.  Filled pixels never act as sources.
// GDAL runs this sequentially line by line; every target is independent given the column tables.  Here:
//   inpaint_bits_kernel   the flags of a column as 64-row bit words (sources; targets)
//   inpaint_table_kernel  the column table: squared row distances up / down per pixel, from the bit words
//   inpaint_fill_fast_kernel     the PACKED search (16-bit keys on a table tile staged in LDS) for the targets with sources nearby
//                                -- the usual case; what it cannot settle it marks in a third bit plane
//   inpaint_fill_general_kernel  GDAL's search as it stands (32-bit keys, up to max_dist columns) for the marked targets
// The filled values are written in place (sources are read where the flag is 1 only).
#include "hk_kernels.h"

#include <stdlib.h>

#define HK_SQ(v) __mul24((v), (v))  // squares of column distances <= 100: the full-rate 24-bit multiply (v_mul_lo_u32 runs at a quarter)

#include <type_traits>

namespace hk {


// source = (r2 > thresh) & (gain > 0) & valid (kernel_model.py:363): one byte per pixel.  NaN parameters (masked pixels,
// degenerate windows) compare false.
__global__ void __launch_bounds__(256) inpaint_flag_kernel(const float* __restrict__ gain, const float* __restrict__ r2,
                                                           float thresh, long long stride, int height, int width,
                                                           unsigned char* __restrict__ flag) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const long long i = (long long)y * stride + x;
        flag[i] = ((r2[i] > thresh) && (gain[i] > 0.f)) ? 1 : 0;
    }
}

// Column table: per pixel two SQUARED row distances in one 32-bit word (round 5; rounds 1-4: the distances as two bytes) -- low
// half: rows up to the nearest source at-or-above (0..max_dist), squared; high half: rows down to the nearest source strictly
// below (1..max_dist + 1), squared; NONE_SQ when there is none in reach (max_dist = 100, rasterio's default).  A candidate test
// of the search then needs no multiplication and no separate row-distance field: its key is (entry half << 15) + a per-step
// constant (fill_one), five instructions instead of seven; the row distance comes back as an exact square root at the finish.
// The search reads the source's value from the offset plane itself.  GDAL carries both as the state of two sequential column scans; "the nearest source within reach" is
// the same thing without the sequence: the flags of a column are packed into 64-row bit words (inpaint_bits_kernel: 64
// independent byte loads per thread), and every pixel finds its two distances with clz / ctz on at most three words per
// direction (inpaint_table_kernel: one thread per column and word, no memory access inside its 64-row loop).  The
// sequential form (one thread per column and 64-row chunk, 100 rows of run-in per direction) took 1.15 ms per 16384^2
// band whatever its chunk height, load width or unrolling: its loads sit behind data-dependent state updates.
constexpr unsigned NONE_B = 0xffu;      // (row-distance byte of the table kernel's sweeps: none in reach)
constexpr unsigned NONE_SQ = 0x7fffu;  // squared-distance half of a table entry: none in reach (beats nothing: > 2 * 101^2)
constexpr int WORD_ROWS = 64;

__global__ void __launch_bounds__(256) inpaint_bits_kernel(const unsigned char* __restrict__ flag, long long stride, int height,
                                                           int width, unsigned long long* __restrict__ bits,
                                                           unsigned long long* __restrict__ tbits) {
    // a thread packs FOUR adjacent columns: 64 independent 4-byte loads (the flag plane's rows are 4-byte aligned: stride % 4 == 0)
    // instead of 64 single bytes per column -- a quarter of the load instructions for the same bytes (0.158 -> 0.064 ms per 16384^2 band)
    // `tbits` (round 5): the TARGETS (flag 0) in the same layout -- the search reads a tile's targets as one word per column
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x >= width) return;
    const int y0 = blockIdx.y * WORD_ROWS;
    unsigned lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0}, tlo[4] = {0, 0, 0, 0}, thi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int ya = y0 + r, yb = y0 + 32 + r;
        // rows past the raster are clamped into it and masked out (the loads stay unconditional and independent)
        const unsigned fa = *reinterpret_cast<const unsigned*>(flag + (long long)min(ya, height - 1) * stride + x);
        const unsigned fb = *reinterpret_cast<const unsigned*>(flag + (long long)min(yb, height - 1) * stride + x);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lo[c] |= (unsigned)(((fa >> (8 * c)) & 1u) != 0 && ya < height) << r;  // (flag 2: neither source nor target)
            hi[c] |= (unsigned)(((fb >> (8 * c)) & 1u) != 0 && yb < height) << r;
            tlo[c] |= (unsigned)(((fa >> (8 * c)) & 0xffu) == 0 && ya < height) << r;
            thi[c] |= (unsigned)(((fb >> (8 * c)) & 0xffu) == 0 && yb < height) << r;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (x + c < width) {
            bits[(long long)blockIdx.y * stride + x + c] = ((unsigned long long)hi[c] << 32) | lo[c];
            tbits[(long long)blockIdx.y * stride + x + c] = ((unsigned long long)thi[c] << 32) | tlo[c];
        }
}

__global__ void __launch_bounds__(256) inpaint_table_kernel(const unsigned long long* __restrict__ bits, long long stride,
                                                            int height, int width, int max_dist,
                                                            unsigned* __restrict__ tb) {
    // a thread makes the entries of TWO adjacent columns and stores them as one 8-byte pair per row (rows are padded: stride % 4 == 0;
    // the table starts 256-byte aligned)
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (x >= width) return;
    const int wb = blockIdx.y, n_words = (height + WORD_ROWS - 1) / WORD_ROWS;
    const int y0 = wb * WORD_ROWS, rows = min(WORD_ROWS, height - y0);
    unsigned long long wm2[2], wm1[2], w0[2], wp1[2], wp2[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        auto word = [&](int w) { return (w >= 0 && w < n_words) ? bits[(long long)w * stride + x + c] : 0ull; };  // x + 1 < stride
        wm2[c] = word(wb - 2), wm1[c] = word(wb - 1), w0[c] = word(wb), wp1[c] = word(wb + 1), wp2[c] = word(wb + 2);
    }
    // Both distances are running counters along the column: U(b) = 0 where bit b is set, else U(b - 1) + 1 (rows up to the nearest
    // source at or above), D(b) = 1 where bit b + 1 is set, else D(b + 1) + 1 (rows down to the nearest source strictly below); the
    // words above / below only seed them.  An ascending sweep leaves the 64 U bytes packed in registers, a descending sweep adds D
    // and stores -- two or three integer operations per row and column instead of 64-bit clz / ffs with three-way selects
    // (0.247 -> 0.157 ms per 16384^2 band).
    constexpr int BIG = 1 << 20;
    unsigned upk[2][WORD_ROWS / 4];
    int dseed[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int u = wm1[c] ? __clzll((long long)wm1[c]) : (wm2[c] ? 64 + __clzll((long long)wm2[c]) : BIG);  // U(-1)
#pragma unroll
        for (int q = 0; q < WORD_ROWS / 4; ++q) {
            unsigned pk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int bb = 4 * q + k;
                u = ((w0[c] >> bb) & 1ull) ? 0 : min(u + 1, BIG);
                pk |= (unsigned)(u <= max_dist ? u : (int)NONE_B) << (8 * k);
            }
            upk[c][q] = pk;
        }
        dseed[c] = wp1[c] ? 1 + (__ffsll((long long)wp1[c]) - 1) : (wp2[c] ? 65 + (__ffsll((long long)wp2[c]) - 1) : BIG);  // D(63)
    }
    int d[2] = {dseed[0], dseed[1]};
#pragma unroll
    for (int bb = WORD_ROWS - 1; bb >= 0; --bb) {
        unsigned e[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (bb < WORD_ROWS - 1) d[c] = ((w0[c] >> (bb + 1)) & 1ull) ? 1 : min(d[c] + 1, BIG);
            const unsigned ub = (upk[c][bb >> 2] >> (8 * (bb & 3))) & 0xffu;
            const unsigned usq = ub == NONE_B ? NONE_SQ : (unsigned)HK_SQ((int)ub);
            const unsigned dsq = d[c] <= max_dist + 1 ? (unsigned)HK_SQ(d[c]) : NONE_SQ;
            e[c] = (dsq << 16) | usq;
        }
        // the second column may lie in the row padding (odd width): its word of `bits` was never written, its entry is never read
        if (bb < rows) *reinterpret_cast<uint2*>(tb + (long long)(y0 + bb) * stride + x) = make_uint2(e[0], e[1]);
    }
}

// GDAL's QUAD_CHECK compares the squared distance of a candidate with the ROUNDED square of the current distance,
//     if (d2 < qd * qd) { qd = sqrt(d2); ... }          (float64; d2 is an exact integer < 2^15)
// so a candidate at the SAME squared distance n replaces the current source exactly when fl(fl(sqrt(n))^2) > n.  That is a
// property of n alone: tie_kernel tabulates it as a bitmap (TIE_N bits), and the search itself runs on integers -- no
// sqrt and no float64 in the loop, same decisions.
constexpr int TIE_N = 2 * 102 * 102;  // > the largest squared distance the search can meet (100^2 + 101^2)
constexpr int FILL_MAX_DIST = 100;  // rasterio.fill.fillnodata default max_search_distance (kernel_model.py:366)
constexpr int WTAB_N = FILL_MAX_DIST * FILL_MAX_DIST + 1;  // weights 1 / qd of the accepted distances (qd <= max_dist = 100)
__global__ void __launch_bounds__(256) tie_kernel(unsigned* __restrict__ tie, double* __restrict__ wtab) {
    const int word = blockIdx.x * blockDim.x + threadIdx.x;
    // the inverse-distance weights GDAL forms as 1.0 / qd with qd = sqrt(n): a table instead of a sqrt and a division
    // per quadrant and target
    for (int n = word; n < WTAB_N; n += gridDim.x * blockDim.x) wtab[n] = n ? 1.0 / sqrt((double)n) : 0.0;
    if (word * 32 >= TIE_N) return;
    unsigned bits = 0;
    for (int b = 0; b < 32; ++b) {
        const double n = (double)(word * 32 + b);
        const double q = sqrt(n);
        if (__dmul_rn(q, q) > n) bits |= 1u << b;
    }
    tie[word] = bits;
}

// floor(sqrt(n)) for 0 <= n < 2^23 (float32 sqrt is correctly rounded and n is exact in float32; the two corrections cost
// nothing and make the result independent of that argument)
__device__ __forceinline__ int isqrt_floor(int n) {
    int r = (int)__fsqrt_rn((float)n);
    if (r * r > n) --r;
    if ((r + 1) * (r + 1) <= n) ++r;
    return r;
}

template <bool INTERIOR = false>
__device__ __forceinline__ void fill_load_d0(const unsigned* __restrict__ trow, int x, int width, unsigned (&d0)[10]) {
    struct __attribute__((packed, aligned(4))) W10 { unsigned w[10]; };
    if (INTERIOR || (x - 4 >= 0 && x + 5 < width)) {
        const W10 v = *reinterpret_cast<const W10*>(trow + x - 4);
#pragma unroll
        for (int j = 0; j < 10; ++j) d0[j] = v.w[j];
    } else {  // GDAL's clamp: it re-checks the edge column
#pragma unroll
        for (int j = 0; j < 10; ++j) d0[j] = trow[j < 4 ? max(0, x - 4 + j) : min(width - 1, x - 4 + j)];
    }
}

// The search of ONE target pixel (x, y): GDAL's quadrant search through the column tables, then the inverse-distance mean of the
// quadrants' sources.  Returns the filled value (the pixel's own value if no source is in reach).
// INTERIOR (wave-uniform, tiled kernel): no step of any search of the wave can reach the raster's edge columns -- the clamps,
// their per-lane squares and the entry-by-entry table reads fall away.
template <bool INTERIOR = false>
__device__ __forceinline__ float fill_one(int x, int y, long long row, const float* offset, long long stride, int width,
                                          int max_dist, const unsigned* __restrict__ tb, const unsigned* __restrict__ tie,
                                          const double* __restrict__ wtab, bool no_below = false) {
        const long long i = row + x;
        float out = offset[i];  // a target without any source in reach keeps its value
        const int none2 = (max_dist + 1) * (max_dist + 1);  // qd = max_dist + 1: "nothing found yet" (a perfect square: no tie)
        // Per quadrant the search state is two KEYS (round 4): (squared distance << 15) | (column distance << 8).  The smallest key
        // is the FIRST candidate met at the best distance (steps ascend, so among equal distances the smallest column distance came
        // first), the smallest key with the low 15 bits inverted is the LAST one: two unsigned minima per candidate instead of two
        // compares and three selects.  A candidate that does not beat "nothing found yet" leaves the initial keys in place; quadrants
        // whose best distance exceeds max_dist are dropped at the end as before.  (Round 5: the table holds SQUARED row distances, so a
        // candidate's key is one shift-add of its table half and the step's constant; the row distance of the winner is the exact
        // square root of what is left of its squared distance at the finish.)
        constexpr unsigned SRC_BITS = 15u, SRC_MASK = (1u << SRC_BITS) - 1u;
        unsigned kf[4], kl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kf[q] = kl[q] = ((unsigned)none2 << SRC_BITS) | SRC_MASK;
        const unsigned* __restrict__ trow = tb + row;
        // GDAL's QUAD_CHECK on squared integer distances.  `sq` is the column table's half word: NONE_SQ (no source in reach) is more
        // than any squared distance the search accepts and more than the initial (max_dist + 1)^2, so it never wins.
        // A candidate at the SAME squared distance replaces the holder iff GDAL's float comparison says so for that distance (the
        // `tie` bit of c) -- a property of c alone: with the bit set the last candidate met at the best distance wins, without it
        // the first.  So the search keeps both, branch-free, and the bit is looked up once per quadrant at the end instead of behind
        // a divergent branch in every one of the 34 candidate tests.
        // dxk = (dx^2 << 15) | (dx << 8): the step's share of the key (wave-uniform away from the raster's edge columns)
        auto consider = [&](int q, unsigned sq, unsigned dxk) {
            const unsigned key = (sq << SRC_BITS) + dxk;
            kf[q] = min(kf[q], key);
            kl[q] = min(kl[q], key ^ SRC_MASK);
        };
        // `no_below` (the raster's last row): no source lies strictly below, the bottom quadrants stay empty whatever is searched --
        // GDAL walks all max_dist columns for them; the top quadrants' bound ends the search with the same result (25 dependent
        // round trips per pass at the very end of the launch otherwise: 0.2 ms per 16384^2 band)
        auto worst_qd2 = [&]() {
            const unsigned top = max(kf[0], kf[2]);
            return (int)((no_below ? top : max(top, max(kf[1], kf[3]))) >> SRC_BITS);
        };
        // Steps are taken in groups that end where GDAL re-derives its search bound (after steps 4, 8, 12, ...): the
        // bound is constant inside a group, so all of the group's table look-ups are issued before the checks, which then
        // run in the original order (ascending step; left quadrants before right ones).  The look-ups of the NEXT group are
        // issued before the checks of the current one (their latency hides behind the checks; a group that turns out not to be
        // needed costs two cached loads and nothing else).
        // Round 3: a group's table words are consecutive 16-bit entries of one row, so they are fetched by two wide loads
        // (2-byte aligned; steps 0..4: the ten entries around x as 16 + 4 bytes; later groups: 8 bytes per side) instead of
        // ten / eight 2-byte gathers -- the search was bound by the look-ups' issue and latency, not by its arithmetic.
        // Lanes whose group reaches past the raster's edge columns assemble the same words entry by entry with GDAL's clamp
        // (it re-checks the edge column).
        {
            struct __attribute__((packed, aligned(4))) W4 { unsigned w[4]; };
            // entry j of a run of table words: low half = squared distance up, high half = squared distance down
            auto up = [](const unsigned* d, int j) { return d[j] & 0xffffu; };
            auto dn = [](const unsigned* d, int j) { return d[j] >> 16; };
            unsigned d0[10];                 // entries 0..9 <-> columns x - 4 .. x + 5 (steps 0..4: left step k = entry 4 - k, right = 4 + k)
            fill_load_d0<INTERIOR>(trow, x, width, d0);
            // later groups (steps first .. first + 3): left entries 0..3 <-> columns x - first - 3 .. x - first (step first + k = entry
            // 3 - k), right entries 0..3 <-> columns x + first .. x + first + 3 (step first + k = entry k)
            auto fetch4 = [&](int first, unsigned (&l)[4], unsigned (&r)[4]) {
                if (INTERIOR || (x - first - 3 >= 0 && x + first + 3 < width)) {
                    const W4 lv = *reinterpret_cast<const W4*>(trow + x - first - 3);
                    const W4 rv = *reinterpret_cast<const W4*>(trow + x + first);
#pragma unroll
                    for (int j = 0; j < 4; ++j) l[j] = lv.w[j], r[j] = rv.w[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        l[j] = trow[max(0, x - first - 3 + j)];
                        r[j] = trow[min(width - 1, x + first + j)];
                    }
                }
            };
            unsigned nl[4], nr[4];
            fetch4(5, nl, nr);
            int this_max = max_dist;
            {   // steps 0 .. 4
                const int last = min(this_max, 4);
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    if (k <= last) {
                        const int dl = INTERIOR ? k : min(k, x), dr = INTERIOR ? k : min(k, width - 1 - x);  // clamped columns: the distance to the edge column
                        const unsigned dlk = ((unsigned)HK_SQ(dl) << SRC_BITS) | ((unsigned)dl << 8);
                        const unsigned drk = ((unsigned)HK_SQ(dr) << SRC_BITS) | ((unsigned)dr << 8);
                        consider(0, up(d0, 4 - k), dlk);  // top left
                        consider(1, dn(d0, 4 - k), dlk);  // bottom left
                        if (k != 0) {
                            consider(2, up(d0, 4 + k), drk);  // top right
                            consider(3, dn(d0, 4 + k), drk);  // bottom right
                        }
                    }
                }
                // no farther column can beat every quadrant's current distance: floor(max qd) = floor(sqrt(max qd2))
                if (last == 4) this_max = isqrt_floor(worst_qd2());
            }
            int first = 5;
            while (first <= this_max) {
                // The whole group is tested even when the bound falls inside it: a column farther than floor(sqrt(worst
                // quadrant distance^2)) cannot beat or tie any quadrant (its squared column distance alone exceeds every best),
                // so GDAL's bound only ever decides when to STOP, and a per-step predicate (an exec-mask round trip per step) buys
                // nothing.  Groups start at 5, 9, ..., 97: no step beyond max_dist = 100.
                const int last = first + 3;
                const unsigned cl[4] = {nl[0], nl[1], nl[2], nl[3]}, cr[4] = {nr[0], nr[1], nr[2], nr[3]};
                fetch4(first + 4, nl, nr);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int step = first + k;
                    {
                        const int dl = INTERIOR ? step : min(step, x), dr = INTERIOR ? step : min(step, width - 1 - x);
                        const unsigned dlk = ((unsigned)HK_SQ(dl) << SRC_BITS) | ((unsigned)dl << 8);
                        const unsigned drk = ((unsigned)HK_SQ(dr) << SRC_BITS) | ((unsigned)dr << 8);
                        consider(0, up(cl, 3 - k), dlk);
                        consider(1, dn(cl, 3 - k), dlk);
                        consider(2, up(cr, k), drk);
                        consider(3, dn(cr, k), drk);
                    }
                }
                this_max = min(this_max, isqrt_floor(worst_qd2()));
                first = last + 1;
            }
        }
        // The four quadrants' weights and source values are fetched WITHOUT branches -- a quadrant without a source in reach reads
        // entry 0 of the weight table (0.0) and the pixel's own value, and its terms are selected away -- so that the eight look-ups
        // go out together: as `if (found) { load; load; accumulate }` per quadrant they made four dependent round trips at the end of
        // every search (profiles/r03_fill_tile.txt).  The sums run in the same order over the same terms.
        double w4[4];
        float v4[4];
        bool ok4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int qd2 = (int)(kf[q] >> SRC_BITS);
            ok4[q] = qd2 <= max_dist * max_dist;  // qd <= max_dist
            const int cq = ok4[q] ? qd2 : 0;
            const unsigned src = ((tie[cq >> 5] >> (cq & 31)) & 1u) ? ((kl[q] & SRC_MASK) ^ SRC_MASK) : (kf[q] & SRC_MASK);  // (the 1.3 KB bitmap stays in cache)
            const int dx = ok4[q] ? (int)(src >> 8) : 0;
            const int dy = (int)__fsqrt_rn((float)(cq - dx * dx));  // exact: a perfect square <= 101^2 (0 for a quadrant that is dropped)
            const int sx = q < 2 ? x - dx : x + dx, sy = (q & 1) ? y + dy : y - dy;
            w4[q] = wtab[cq];
            v4[q] = offset[(long long)sy * stride + sx];
        }
        double wsum = 0.0, vsum = 0.0;
        bool has = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            has = ok4[q] ? (w4[q] != 0.0) : has;
            wsum = ok4[q] ? wsum + w4[q] : wsum;
            vsum = ok4[q] ? vsum + (double)v4[q] * w4[q] : vsum;
        }
        if (has) out = (float)(vsum / wsum);
        return out;
}

// ---- The PACKED search (round 5) ------------------------------------------------------------------------------------------
// Where the sources are dense -- the usual case: a failing pixel has passing ones a few pixels away in every quadrant -- the search
// state fits 16 bits per quadrant: key = (squared distance << 5) | column distance, for squared distances below FAST_CLIP and
// columns up to FAST_LAST away.  A table word holds the two row distances of a column (up | down << 16), so ONE packed
// instruction serves two quadrants: per column visited the search costs four VALU instructions (two additions of the step's
// constants, two minima; the words are clipped and scaled when they are staged) where the 32-bit keys of fill_one() cost ten.  The decisions are the same ones:
//   * first / last candidate at the best squared distance = smallest key with the column distance / its complement in the low bits;
//   * a quadrant is SETTLED after step S when its best squared distance is below (S + 1)^2: any candidate in a farther column,
//     and any candidate whose row distance was clipped (>= FAST_CLIP > (FAST_LAST + 1)^2), is strictly farther -- it can neither
//     beat nor tie the holder, which is all GDAL's search could still do with it (its own bound only decides when to stop);
//   * a target with a quadrant that does not settle by FAST_LAST (no source nearby, or none at all) is handed to fill_one().
// The table words come from LDS: the per-lane gathers of fill_one() (every lane its own row piece) kept the texture path busy
// 32 cycles per wave instruction and the waves waiting 71-82 % of their lives (PMC, profiles/r05b_inpaint_packed.txt); a
// workgroup now stages the table of its tile plus FAST_HALO columns either side once, coalesced, and the search reads it from
// there.  The finish reads the weight 1 / sqrt(n) (n's tie bit is its sign) and a square root from small LDS tables, and the
// source's value from the plane; all four quadrants of a settled target hold a source within max_dist, so GDAL's acceptance tests are true by
// construction.
#ifndef HK_FAST_LAST
#define HK_FAST_LAST 24
#endif
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ unsigned pk_add_u16(unsigned a, unsigned b) {  // (no carry between the halves; the callers keep every half below 2^16)
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(us2_t, a) + __builtin_bit_cast(us2_t, b));
}
constexpr int FAST_LAST = HK_FAST_LAST;        // last column distance of the packed search (groups end at 4, 8, ..., 24)
constexpr unsigned FAST_CLIP = 1263; // row distances squared are clipped here: (1263 + 24^2) * 32 + 31 < 65536, and 1263 > 25^2
constexpr int FAST_HALO = HK_FAST_LAST;        // table columns staged either side of a workgroup's 256 (>= FAST_LAST, a multiple of 4)
#ifndef HK_FAST_PITCH_PAD
#define HK_FAST_PITCH_PAD 4
#endif
constexpr int FAST_COLS = 256 + 2 * FAST_HALO, FAST_PITCH = FAST_COLS + HK_FAST_PITCH_PAD;  // (the rows of a pass spread over the banks)
static_assert((FAST_CLIP + FAST_LAST * FAST_LAST) * 32 + 31 <= 0xffffu, "a packed key overflows its half");
static_assert(FAST_CLIP > (FAST_LAST + 1) * (FAST_LAST + 1), "a clipped candidate could settle a quadrant");
static_assert(FAST_HALO >= FAST_LAST && FAST_HALO % 4 == 0, "the staged halo");
// What the finish looks up per quadrant, kept in LDS: for every squared distance n a settled quadrant can have, the weight
// 1 / sqrt(n) with n's TIE BIT AS ITS SIGN (weights are positive; the sums take the magnitude through the operand modifier),
// and the square roots of the perfect squares (the winner's row distance from what is left of n beside its column distance).
constexpr int FTAB_N = (FAST_LAST + 1) * (FAST_LAST + 1);  // a settled quadrant's squared distance is below this
// a word of the distance table as the packed search wants it: both halves clipped to FAST_CLIP and shifted into the key's place
__device__ __forceinline__ unsigned fast_stage_word(unsigned e) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(us2_t, pk_min_u16(e, FAST_CLIP * 0x10001u)) << (us2_t)(5));
}
struct FastTables {
    double w[FTAB_N];
    unsigned char root[FTAB_N + 7];
};
static_assert(sizeof(FastTables) % 8 == 0, "copied in 8-byte pieces");

__global__ void __launch_bounds__(256) fast_table_kernel(FastTables* __restrict__ ft) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= FTAB_N + 7) return;
    if (n < FTAB_N) {
        const double q = sqrt((double)n);
        const double w = n ? 1.0 / q : 0.0;
        ft->w[n] = __dmul_rn(q, q) > (double)n ? -w : w;
    }
    ft->root[n] = (unsigned char)isqrt_floor(n);
}

// The packed search of ONE target of a wave whose searches stay inside the staged tile.  `c0`: the target's own entry of the
// staged table; `base` + `rel0`: the target's pixel of the plane as a 32-bit offset from a point FAST_HALO rows and columns before
// the tile (so that every source's offset is non-negative).  Returns true and the filled value when all four quadrants settled;
// false when the target has to go through fill_one().  `cont_min`: a group of columns is searched only while at least that many
// lanes of the wave still need it.  Called by every lane of a pass that has a target (the votes count those lanes only).
__device__ __forceinline__ bool fill_fast(const unsigned* c0, const float* __restrict__ base, unsigned rel0, int stride,
                                          const FastTables& ft, int cont_min, float& out) {
    unsigned kfL = ~0u, klL = ~0u, kfR = ~0u, klR = ~0u;  // (down | up) halves: left quadrants 1 | 0, right quadrants 3 | 2
    // `e`: a staged table word -- both halves clipped and scaled already (fast_stage_word)
    auto cons = [](unsigned e, int k, unsigned& kf, unsigned& kl) {
        kf = pk_min_u16(kf, pk_add_u16(e, (unsigned)((k * k << 5) + k) * 0x10001u));
        kl = pk_min_u16(kl, pk_add_u16(e, (unsigned)((k * k << 5) + 31 - k) * 0x10001u));
    };
    auto worst = [&]() {
        const unsigned m = pk_max_u16(kfL, kfR);
        return max(m & 0xffffu, m >> 16);
    };
    cons(c0[0], 0, kfL, klL);  // the own column belongs to the left quadrants
#pragma unroll
    for (int k = 1; k <= 4; ++k) {
        cons(c0[-k], k, kfL, klL);
        cons(c0[k], k, kfR, klR);
    }
    bool ok = worst() < (25u << 5);
#pragma unroll
    for (int first = 5; first <= FAST_LAST - 3; first += 4) {
        const unsigned long long open = __ballot(!ok);
        if (open == 0ull || (int)__popcll(open) < cont_min) break;
#pragma unroll
        for (int k = first; k < first + 4; ++k) {
            cons(c0[-k], k, kfL, klL);
            cons(c0[k], k, kfR, klR);
        }
        ok = worst() < ((unsigned)((first + 4) * (first + 4)) << 5);
    }
    if (!ok) return false;
    // (in stages, so that the four quadrants' look-ups travel together)
    double ws[4];
    float v4[4];
    unsigned n4[4], dx4[4], dy4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        n4[q] = ((q < 2 ? kfL : kfR) >> ((q & 1) ? 21 : 5)) & 0x7ffu;
        ws[q] = ft.w[n4[q]];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned kf = q < 2 ? kfL : kfR, kl = q < 2 ? klL : klR;
        const int sh = (q & 1) ? 16 : 0;
        dx4[q] = __double2hiint(ws[q]) < 0 ? 31u - ((kl >> sh) & 31u) : (kf >> sh) & 31u;  // tie bit set: the last candidate at the best distance
        dy4[q] = ft.root[n4[q] - dx4[q] * dx4[q]];  // (a perfect square)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // source = (y -/+ dy, x -/+ dx): rows and columns count from FAST_HALO before the tile, so nothing is negative
        const unsigned rel = (q & 1) ? __umul24(dy4[q], (unsigned)stride) + rel0 : rel0 - __umul24(dy4[q], (unsigned)stride);
        v4[q] = base[q < 2 ? rel - dx4[q] : rel + dx4[q]];
    }
    double wsum = 0.0, vsum = 0.0;  // the same sums in the same order as fill_one()
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wsum = wsum + fabs(ws[q]);
        vsum = vsum + (double)v4[q] * fabs(ws[q]);
    }
    out = (float)(vsum / wsum);
    return true;
}

// TILED form of the search (round 3): a wave owns a tile of 64 columns x ROWS rows.  The per-row forms of rounds 1-2 started a
// workgroup per 256-pixel row piece, and each lived for three dependent memory round trips (flag -> compaction behind two barriers
// -> table words -> source values): PMC showed its waves WAITING 76 % of their 4.7 us life with the VALU 58 % busy.  Here a wave
// takes the targets of its whole tile at once (one word of a column bit plane per lane), compacts them into a wave-private LDS
// list (ballot + popcount) and then searches 64 targets per pass -- full waves whatever the failure rate, one wave start per ROWS
// rows.  Round 5: TWO kernels.  inpaint_fill_fast_kernel stages its tile of the distance table in LDS and runs the packed search
// (fill_fast) on 8-row tiles; the targets it cannot settle are marked in a second bit plane (`sbits`: a tile owns one byte of its
// columns' 64-row words, so plain stores do), and inpaint_fill_general_kernel takes fill_one() over that plane, 64 at a time from
// 32-row tiles.  (As one kernel the general search cost the packed one 7 % of the step at 35 % failing pixels by its mere presence
// -- 64 registers with spills against 34, four times the code -- although it ran for 0.01 % of the targets;
// profiles/r05b_inpaint_packed.txt.)
// The filled values are written IN PLACE: a filled pixel never acts as a source (sources are read where flag == 1, targets are
// written where flag == 0), and the closing pass reads the plane only at the failing pixels.
template <bool BY_COLUMN>
__global__ void __launch_bounds__(256, 8)
inpaint_fill_fast_kernel(float* plane, const unsigned long long* __restrict__ tbits, unsigned long long* __restrict__ sbits,
                         long long stride, int height, int width, const unsigned* __restrict__ tb,
                         const FastTables* __restrict__ ftab, int cont_min) {
    constexpr int ROWS = 8;  // a tile = one byte of the bit planes' words; list + staged table + finish tables: eight workgroups per CU
    __shared__ unsigned short lst[256 / WAVE][ROWS * WAVE];
    __shared__ unsigned tile[ROWS * FAST_PITCH];
    __shared__ FastTables ftl;
    __shared__ unsigned slow_rows[256 / WAVE][WAVE / 4];  // per column of the wave's tile one byte: its rows that did not settle
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int xb = blockIdx.x * blockDim.x, x0 = xb + wv * WAVE, x_own = x0 + lane;
    const unsigned long long lt = (1ull << lane) - 1ull;
    (void)lt;
    // Columns outside the raster are staged as "no source in reach": GDAL re-checks the edge column there, which changes nothing (the
    // edge column was the farthest candidate of its side already) -- except for the right quadrants of the LAST column, whose first
    // and only candidate is the re-checked own column: they find nothing here, do not settle and go to the general search.
    for (int j = tid; j < (int)(sizeof(FastTables) / 8); j += 256)
        reinterpret_cast<unsigned long long*>(&ftl)[j] = reinterpret_cast<const unsigned long long*>(ftab)[j];
    const int n_tiles = (height + ROWS - 1) / ROWS;
    for (int tile_i = blockIdx.y; tile_i < n_tiles; tile_i += gridDim.y) {
        const int y0 = tile_i * ROWS;
        // (never dereferenced before the raster's first pixel: sources and targets lie inside it)
        float* const origin = plane + ((long long)(y0 - FAST_HALO) * stride + (xb - FAST_HALO));
        // the tile's piece of the distance table, FAST_HALO columns either side (16-byte pieces: stride % 4 == 0)
        for (int j = tid; j < ROWS * (FAST_COLS / 4); j += 256) {
            const int r = j / (FAST_COLS / 4), c4 = j - r * (FAST_COLS / 4), xs = xb - FAST_HALO + 4 * c4;
            uint4 v = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);  // (outside the raster: no source in reach)
            if (y0 + r < height && xs >= 0 && xs < width) {  // (xs + 3 < stride: rows are padded to quads)
                v = *reinterpret_cast<const uint4*>(tb + (long long)(y0 + r) * stride + xs);
                if (xs + 3 >= width) {  // the quad straddles the raster's last column: the padding holds no entries
                    v.y = xs + 1 < width ? v.y : 0x7fff7fffu, v.z = xs + 2 < width ? v.z : 0x7fff7fffu, v.w = 0x7fff7fffu;
                }
            }
            *reinterpret_cast<uint4*>(&tile[r * FAST_PITCH + 4 * c4]) =
                make_uint4(fast_stage_word(v.x), fast_stage_word(v.y), fast_stage_word(v.z), fast_stage_word(v.w));
        }
        if (lane < WAVE / 4) slow_rows[wv][lane] = 0u;
        // the tile's targets: bit r of the lane's byte = row y0 + r of its column (rows past the raster are clear)
        unsigned tgt_rows = 0u;
        if (x_own < width) tgt_rows = (unsigned)(tbits[(long long)(y0 / WORD_ROWS) * stride + x_own] >> (y0 % WORD_ROWS)) & 0xffu;
        int n = 0;  // wave-uniform: targets of the tile
        if constexpr (BY_COLUMN) {
            // COLUMN by column: a pass of 64 targets then covers a compact block of the tile (8 rows x a few columns) instead of a
            // piece of one row -- neighbours need searches of similar length, and a pass lasts as long as its longest search
            // (at 94 % failing pixels the step takes 9 % less; at 35 % the row order is 3 % faster: the launcher picks by the share of
            // failing pixels; both orders in one kernel cost the sparse case 2.5 % whichever way a tile decides)
            const int cnt = __popc(tgt_rows);
            int pre = cnt;  // inclusive prefix over the lanes
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const int up = __shfl_up(pre, d);
                pre += lane >= d ? up : 0;
            }
            n = __shfl(pre, WAVE - 1);
            int pos = pre - cnt;
            for (unsigned mm = tgt_rows; mm; mm &= mm - 1u) lst[wv][pos++] = (unsigned short)(((__ffs((int)mm) - 1) << 6) | lane);
        } else {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const bool target = (tgt_rows >> r) & 1u;
                const unsigned long long bal = __ballot(target);
                if (target) lst[wv][n + (int)__popcll(bal & lt)] = (unsigned short)((r << 6) | lane);
                n += (int)__popcll(bal);
            }
        }
        __syncthreads();  // the staged table (and the first tile's ftl) for everyone; the list and slow_rows are the wave's own
        for (int p = 0; p < n; p += WAVE) {
            if (p + lane < n) {
                const unsigned e = lst[wv][p + lane];
                const int r = (int)(e >> 6), c = (int)(e & 63u), xl = wv * WAVE + c;
                float v;
                const unsigned rel0 = (unsigned)(r + FAST_HALO) * (unsigned)stride + (unsigned)(xl + FAST_HALO);
                if (fill_fast(&tile[r * FAST_PITCH + FAST_HALO + xl], origin, rel0, (int)stride, ftl, cont_min, v)) origin[rel0] = v;
                else atomicOr(&slow_rows[wv][c >> 2], 1u << (8 * (c & 3) + r));
            }
        }
        __syncthreads();  // (orders the wave's LDS atomics before the read below; the next tile re-uses list and staged table)
        if (x_own < width)  // this tile's byte of the column's word in the plane of unsettled targets (little-endian)
            reinterpret_cast<unsigned char*>(sbits + (long long)(y0 / WORD_ROWS) * stride + x_own)[(y0 % WORD_ROWS) / 8] =
                (unsigned char)(slow_rows[wv][lane >> 2] >> (8 * (lane & 3)));
    }
}

// The general search (fill_one) of the targets marked in `bits` -- what the packed search left over, or every target when the
// packed search is off: 64 columns x ROWS rows per wave, its targets compacted into an LDS list, 64 per pass.
#ifndef HK_FILL_TILE_WAVES
#define HK_FILL_TILE_WAVES 8
#endif
template <int ROWS>
__global__ void __launch_bounds__(256, (ROWS <= 32 ? HK_FILL_TILE_WAVES : 5))
inpaint_fill_general_kernel(float* plane, const unsigned long long* __restrict__ bits, long long stride, int height, int width,
                            const unsigned* __restrict__ tb, const unsigned* __restrict__ tie, const double* __restrict__ wtab) {
    static_assert(ROWS == 16 || ROWS == 32 || ROWS == 64, "tiles divide the 64-row words of the bit planes");
    __shared__ unsigned short lst[256 / WAVE][ROWS * WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int x0 = blockIdx.x * blockDim.x + wv * WAVE, x_own = x0 + lane;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // the search distance as a compile-time constant: the first five steps become straight-line code, the "nothing found yet" and
    // acceptance bounds immediates
    const int max_dist = FILL_MAX_DIST;
    // wave-uniform: the widest reach of a search stays inside the row -- the group that starts at step max_dist requests the one
    // after it (steps max_dist + 4 .. + 7) ahead, and a group's wide load covers 4 entries: 7 + 4 columns beyond max_dist
    const bool interior = x0 - max_dist - 12 >= 0 && x0 + WAVE - 1 + max_dist + 12 < width;
    const int n_tiles = (height + ROWS - 1) / ROWS;
    // (from the raster's bottom upwards: the last rows, whose targets have no sources below and search longest, start first)
    for (int tile_k = blockIdx.y; tile_k < n_tiles; tile_k += gridDim.y) {
        const int y0 = (n_tiles - 1 - tile_k) * ROWS;
        unsigned long long tgt_rows = 0ull;
        if (x_own < width) {
            tgt_rows = bits[(long long)(y0 / WORD_ROWS) * stride + x_own] >> (y0 % WORD_ROWS);
            const int rows_in = min(ROWS, height - y0);  // (the packed search writes whole bytes only where it has a tile)
            tgt_rows &= rows_in >= 64 ? ~0ull : (1ull << rows_in) - 1ull;
        }
        if (__ballot(tgt_rows != 0ull) == 0ull) continue;  // (wave-uniform; nothing of this wave's tile is left over: the usual case)
        int n = 0;  // wave-uniform: targets of the tile so far
#pragma unroll 4
        for (int r = 0; r < ROWS; ++r) {
            const bool target = (tgt_rows >> r) & 1ull;
            const unsigned long long bal = __ballot(target);
            if (target) lst[wv][n + (int)__popcll(bal & lt)] = (unsigned short)((r << 6) | lane);
            n += (int)__popcll(bal);
        }
        // (the list is the wave's own: no workgroup barrier -- the LDS operations of a wave complete in order; the fence keeps the
        // compiler from moving the reads of other lanes' entries above the writes)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int p = 0; p < n; p += WAVE) {
            if (p + lane < n) {
                const unsigned e = lst[wv][p + lane];
                const int x = x0 + (int)(e & 63u), y = y0 + (int)(e >> 6);
                const long long row = (long long)y * stride;
                const bool last_row = y == height - 1;
                const float v = interior ? fill_one<true>(x, y, row, plane, stride, width, max_dist, tb, tie, wtab, last_row)
                                         : fill_one<false>(x, y, row, plane, stride, width, max_dist, tb, tie, wtab, last_row);
                plane[row + x] = v;
            }
        }
    }
}

// workspace: the distance table + source flags (1 byte per pixel) + the tie bitmap + the weight table
// + the column bit words of sources, targets and the targets the packed search left over (one 64-bit word each per column and
// 64 rows; a plane of a few rows has more of those than spare table bytes) + the packed search's tables
static size_t bit_words(int height, long long stride) { return (size_t)((height + WORD_ROWS - 1) / WORD_ROWS) * (size_t)stride; }
size_t inpaint_workspace_bytes(int height, long long stride) {
    return (size_t)height * stride * 5 + 1024 + TIE_N / 8 + 256 + WTAB_N * 8 + 256 + 3 * (bit_words(height, stride) * 8 + 256)
           + sizeof(FastTables) + 256;
}

// the workspace's source-flag plane: the fit kernel can write it itself (FitArgs::flag), then gain / r2 are not needed here
unsigned char* inpaint_flag_plane(void* workspace, int height, long long stride) {
    return reinterpret_cast<unsigned char*>(static_cast<unsigned short*>(workspace) + 2 * (size_t)height * stride);
}

hipError_t launch_inpaint_offsets(float* offset, const float* gain, const float* r2, float thresh, long long stride,
                                  int height, int width, void* workspace, hipStream_t stream,
                                  const unsigned char* flag_ready, unsigned long long n_targets) {
    const size_t plane = (size_t)height * stride;
    unsigned* tb = static_cast<unsigned*>(workspace);  // (down^2 << 16) | up^2 row distances, 4 bytes per pixel
    unsigned char* ws_flag = inpaint_flag_plane(workspace, height, stride);
    const unsigned char* flag = flag_ready ? flag_ready : ws_flag;
    unsigned* tie = reinterpret_cast<unsigned*>(ws_flag + (plane + 255) / 256 * 256);
    double* wtab = reinterpret_cast<double*>(tie + (TIE_N / 32 + 64) / 64 * 64);
    HK_LAUNCH(tie_kernel, dim3((TIE_N / 32 + 255) / 256), dim3(256), 0, stream, tie, wtab);
    const int max_dist = FILL_MAX_DIST;
    static_assert(100 + 1 < (int)NONE_B, "the table's distance bytes");
    if (!flag_ready)  // else: the flag plane was written by the fit kernel (FitArgs::flag)
        HK_LAUNCH(inpaint_flag_kernel, dim3((width + 255) / 256, height < 1024 ? height : 1024), dim3(256), 0, stream,
                           gain, r2, thresh, stride, height, width, ws_flag);
    // column bit words behind the weight table (256-byte aligned): sources, then targets; the packed search's table behind them
    auto align256 = [](void* p) { return reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(p) + 255) / 256 * 256); };
    unsigned long long* bits = static_cast<unsigned long long*>(align256(wtab + WTAB_N));
    unsigned long long* tbits = static_cast<unsigned long long*>(align256(bits + bit_words(height, stride)));
    unsigned long long* sbits = static_cast<unsigned long long*>(align256(tbits + bit_words(height, stride)));
    FastTables* ftab = static_cast<FastTables*>(align256(sbits + bit_words(height, stride)));
    HK_LAUNCH(fast_table_kernel, dim3((FTAB_N + 7 + 255) / 256), dim3(256), 0, stream, ftab);
    const dim3 gbits((width + 1023) / 1024, (height + WORD_ROWS - 1) / WORD_ROWS);  // four columns per thread
    HK_LAUNCH(inpaint_bits_kernel, gbits, dim3(256), 0, stream, flag, stride, height, width, bits, tbits);
    const dim3 gtable((width + 511) / 512, (height + WORD_ROWS - 1) / WORD_ROWS);  // two columns per thread
    HK_LAUNCH(inpaint_table_kernel, gtable, dim3(256), 0, stream, bits, stride, height, width, max_dist, tb);
    // The packed search on 8-row tiles, then the general one (32-row tiles) over what it left.  The targets of a tile are listed
    // column by column where more than 60 % of the pixels fail, row by row otherwise (HISTORY.md 53).
    const bool by_column = (double)n_targets > 0.6 * (double)height * (double)width;
    // the packed search addresses the plane through 32-bit offsets from its tile and 24-bit multiplies by the row stride
    const bool fast = stride < (1ll << 23);
    if (fast) {
        const int n_tiles8 = (height + 7) / 8, wgs_y = (n_tiles8 + 3) / 4;  // a workgroup takes four tiles (it copies the finish tables into LDS once)
        const dim3 gf((width + 255) / 256, wgs_y < 65535 ? wgs_y : 65535);
        if (by_column)
            HK_LAUNCH(inpaint_fill_fast_kernel<true>, gf, dim3(256), 0, stream, offset, tbits, sbits, stride, height, width, tb, ftab, 1);
        else
            HK_LAUNCH(inpaint_fill_fast_kernel<false>, gf, dim3(256), 0, stream, offset, tbits, sbits, stride, height, width, tb, ftab, 1);
    }
    const unsigned long long* todo = fast ? sbits : tbits;
    const int n_tiles = (height + 31) / 32;
    const dim3 gt((width + 255) / 256, n_tiles < 65535 ? n_tiles : 65535);
    HK_LAUNCH(inpaint_fill_general_kernel<32>, gt, dim3(256), 0, stream, offset, todo, stride, height, width, tb, tie, wtab);
    return hipGetLastError();
}

}  // namespace hk
