// orb_kernels.hip -- gfx950 kernels of the ORB extractor (wave64, LDS-staged; no MFMA: there is no dense contraction).
//
// Pipeline per batch of B same-sized frames (all launches cover every frame of the batch):
//   k_resize_level      x (nlevels-1)  bilinear pyramid level l from level l-1     (ORBextractor.cc:1107-1132)
//   k_fast_cells        x 1            per-cell FAST-9/16 score + 3x3 NMS + threshold fallback (:789-829)
//   k_distribute        x 1            DistributeOctTree, one wave per (frame, level)           (:539-763)
//   k_level_offsets     x 1            per-frame level prefix sums / keypoint totals            (:1059-1062)
//   k_blur7             x 1            7x7 sigma-2 integer Gaussian of every level              (:1085-1086)
//   k_orient_describe2  x 1            IC_Angle + steered BRIEF, two keypoints per wave         (:77-147)
//
// Data layout in HBM (per extractor handle, frame-major): see DESIGN.md "ORB extractor: layout".
#include "orb_kernels.hpp"
#include "wave_dpp.hpp"

namespace orbfe {
// row * pitch as a 32-bit byte offset, for rows and pitches below 2^23 (any frame here).  Written as (size_t)row * pitch the product
// is a v_mad_u64_u32 into a register PAIR followed by a 64-bit pointer addition; as a 32-bit offset off a uniform base it is one
// v_mul_i32_i24 (or v_mad_u32_u24 with the column) and the load takes the base from scalar registers: fewer instructions and
// registers (k_blur7 68 -> 64, k_orient_describe2 64 -> 58), not a faster multiply -- 32-bit multiplies are NOT quarter-rate on
// gfx950 (tools/valu_rate.hip: 5.3 against 4.9 cycles per wave).
__device__ __forceinline__ uint32_t off24(int row, int pitch) { return (uint32_t)__mul24(row, pitch); }


// ------------------------------------------------------------------------------------------------ helpers --
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int lane_prefix(unsigned long long mask)
{
    // number of set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Diagnosis build only (-DORBFE_WAVE_TIMING, tools/wave_timing.sh): where a wave's life goes, summed over all waves of a kernel.
// WT_MARK(kernel, slot) adds the shader clocks since the previous mark of this wave to g_wt[kernel][slot] (slot 15 counts waves).
#ifdef ORBFE_WAVE_TIMING
// (2048 copies of every counter, picked by workgroup: one word takes ~90 atomics a microsecond, the kernels retire thousands of waves in that time)
__device__ unsigned long long g_wt[4][16][2048];
#define WT_BEGIN() long long wt_prev_ = clock64()
#define WT_MARK(kid, slot) do { const long long t_ = clock64(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_wt[kid][slot][(blockIdx.x * 4 + (threadIdx.x >> 6)) & 2047], (unsigned long long)(t_ - wt_prev_)); wt_prev_ = clock64(); } while (0)
#define WT_COUNT(kid) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_wt[kid][15][(blockIdx.x * 4 + (threadIdx.x >> 6)) & 2047], 1ull); } while (0)
#define WT_RETRY(kid) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_wt[kid][14][(blockIdx.x * 4 + (threadIdx.x >> 6)) & 2047], 1ull); } while (0)   // (FAST: cells that went to minThFAST)
} // namespace orbfe
extern "C" __attribute__((visibility("default"))) int orbfe_timing_read(unsigned long long* out, int reset)
{
    static unsigned long long h[4][16][2048];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(orbfe::g_wt), sizeof(h)) != hipSuccess) return -1;
    for (int k = 0; k < 4; k++) for (int s = 0; s < 16; s++) { unsigned long long t = 0; for (int i = 0; i < 2048; i++) t += h[k][s][i]; out[k * 16 + s] = t; }
    if (reset) { for (auto& a : h) for (auto& b : a) for (auto& c : b) c = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(orbfe::g_wt), h, sizeof(h)); }
    return 0;
}
namespace orbfe {
#else
#define WT_BEGIN()
#define WT_MARK(kid, slot)
#define WT_COUNT(kid)
#define WT_RETRY(kid)
#endif

// ------------------------------------------------------------------------------------------------ resize --
// One thread = 4 horizontally adjacent output pixels x 2 rows (two aligned u32 stores).  The cv::resize coefficient
// arithmetic (double/float) is evaluated per thread with exactly the host formulas -- contraction is
// off, so the values equal the host-built tables the first version of this kernel loaded -- which removes the
// dependent table-load -> pixel-load chain.
__device__ __forceinline__ void resize_coef(int d, double scale, int smax, int& s0, int& a0, int& a1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = orbfe_floor_d((double)f);
    f -= (float)si;
    if (si < 0) { f = 0.f; si = 0; }
    if (si >= smax - 1) { f = 0.f; si = smax - 1; }
    s0 = si;
    a0 = (short)orbfe_round_f((1.f - f) * 2048.f);
    a1 = (short)orbfe_round_f(f * 2048.f);
}

__global__ __launch_bounds__(256) void k_resize_level(ImgView src, ImgView dst, int sw, int sh, int dw4 /*ceil(dw/4)*/,
                                                      int dh, double scale_x, double scale_y, int dw)
{
    const int x4 = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
    const int f = blockIdx.z;
    if (x4 >= dw4 || dy0 >= dh) return;
    const uint8_t* S = src.base + (size_t)f * src.fstride;
    int sx[4], ax0[4], ax1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) resize_coef(min(x4 * 4 + k, dw - 1), scale_x, sw, sx[k], ax0[k], ax1[k]);
    uint8_t px[2][2][4][2];
    int b0[2], b1[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int dy = min(dy0 + r, dh - 1);
        // rows are NOT clamped like columns: cv::resize keeps the fractional weight and clips the row index
        float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
        int sy = orbfe_floor_d((double)fy);
        fy -= (float)sy;
        b0[r] = (short)orbfe_round_f((1.f - fy) * 2048.f);
        b1[r] = (short)orbfe_round_f(fy * 2048.f);
        const uint8_t* S0 = S + (size_t)min(max(sy, 0), sh - 1) * src.pitch;
        const uint8_t* S1 = S + (size_t)min(max(sy + 1, 0), sh - 1) * src.pitch;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s1 = min(sx[k] + 1, sw - 1);
            px[r][0][k][0] = S0[sx[k]]; px[r][0][k][1] = S0[s1];
            px[r][1][k][0] = S1[sx[k]]; px[r][1][k][1] = S1[s1];
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (dy0 + r >= dh) break;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int h0 = px[r][0][k][0] * ax0[k] + px[r][0][k][1] * ax1[k];
            const int h1 = px[r][1][k][0] * ax0[k] + px[r][1][k][1] * ax1[k];
            const int v = (((b0[r] * (h0 >> 4)) >> 16) + ((b1[r] * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xff) << (8 * k);
        }
        uint8_t* D = dst.base_w + (size_t)f * dst.fstride + (size_t)(dy0 + r) * dst.pitch;
        *reinterpret_cast<uint32_t*>(D + x4 * 4) = packed;
    }
}

// Table-driven variant for the ORB pyramid (scale 1.2 between levels).  One thread = 4 adjacent output pixels x
// RS_ROWS consecutive rows, threads numbered flat over (row group, x) so that no lane idles at the right edge.
//   * the 8 source bytes a thread needs per source row (columns sx[0] .. sx[3]+1, at most 6 apart at this scale) come
//     from ONE unaligned 8-byte load instead of 8 byte loads;
//   * v_perm_b32 picks each (left, right) pixel pair out of the window, v_dot2_u32_u16 does left*a0 + right*a1;
//   * the x coefficients (host tables, the cv::resize arithmetic of SURVEY App. B.2) are loaded once per thread and
//     reused for all rows; the y coefficients are two table words per row.
// The host checks that every window fits (resize_tab_ok) and otherwise launches k_resize_level.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t hdot(uint32_t pair, uint32_t al)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair), __builtin_bit_cast(u16x2, al), 0u, false);
}

// (b * (h >> 4)) >> 16 for a weight b <= 2048 and a row sum h <= 2048 * 255 as ONE full-rate instruction: with bs = b << 12 and
// hm = h & ~15 (both below 2^24), bs * hm = b * (h >> 4) * 2^16, and v_mul_hi_u32_u24 returns bits 32.. of that 48-bit product.
// (Written as (b * (h >> 4)) >> 16 it is three: shift, v_mul_lo_u32, shift -- 192 of the thread's ~570 instructions.)
__device__ __forceinline__ uint32_t rs_wmul(uint32_t bs, uint32_t hm)
{
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(bs), "v"(hm));
    return r;
}

__global__ __launch_bounds__(256) void k_resize_tab(ImgView src, ImgView dst, int sw, int sh, int dw4, int dh,
                                                    int nthreads, const int* __restrict__ xofs,
                                                    const int* __restrict__ xal, const int4* __restrict__ ytab, int nx, int total)
{
#ifndef ORBFE_PRIO_RESIZE
#define ORBFE_PRIO_RESIZE 2
#endif
    // Under the phase lock the step is the extractor chain resize -> FAST -> quadtree, and the resize chain's seven small launches are
    // stretched 2.4x by whatever shares their CUs (469 us in the pipeline, 194 alone): their waves go first.  Twelve interleaved runs
    // each: 1.3450 against 1.3607 ms per C2 step at priority 2 (medians 1.340 / 1.361); priority 3 loses (1.42: it starves the
    // detector's latency-bound kernels, which run at 2), FAST at priority 1 loses more (1.53).
    __builtin_amdgcn_s_setprio(ORBFE_PRIO_RESIZE);
    int bx, f;
    if (!xcd_remap(nx, total, bx, f)) return;
    const int t = bx * 256 + threadIdx.x;
    if (t >= nthreads) return;
    const int rg = t / dw4, x4 = t - rg * dw4;
    // uniform bases + 32-bit lane offsets (a per-lane 64-bit pointer costs a v_lshl_add_u64 per row and load)
    const uint8_t* S = src.base + (size_t)f * src.fstride;
    uint8_t* D = dst.base_w + (size_t)f * dst.fstride;
    const uint32_t xo = (uint32_t)x4 * 4u;
    const int4 sx = *reinterpret_cast<const int4*>(xofs + x4 * 4);
    const int4 al = *reinterpret_cast<const int4*>(xal + x4 * 4);
    const int w0 = min(sx.x, sw - 8);
    uint32_t sel[4];
    {
        const int s[4] = {sx.x, sx.y, sx.z, sx.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
            sel[k] = (uint32_t)(s[k] - w0) | ((uint32_t)(min(s[k] + 1, sw - 1) - w0) << 16) | 0x0c000c00u;
    }
    const uint32_t a[4] = {(uint32_t)al.x, (uint32_t)al.y, (uint32_t)al.z, (uint32_t)al.w};
    typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
    const int dy0 = rg * RS_ROWS;
#pragma unroll
    for (int r0 = 0; r0 < RS_ROWS; r0 += 4) {
        unsigned long long wt[4], wb[4];
        uint32_t b0s[4], b1s[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { // all loads of four rows in flight before the first use
            const int dy = min(dy0 + r0 + r, dh - 1);
            // a row's entry: the two source rows (clipped -- cv::resize keeps the fractional weight and clips the row index -- by the
            // host) and the two weights shifted for rs_wmul: one 16-byte load; the clamps and the unpacking were 8 of 64 instructions a row
            const int4 ye = ytab[dy];
            b0s[r] = (uint32_t)ye.z; b1s[r] = (uint32_t)ye.w;
            wt[r] = *reinterpret_cast<const u64_unaligned*>(S + (off24(ye.x, src.pitch) + (uint32_t)w0));
            wb[r] = *reinterpret_cast<const u64_unaligned*>(S + (off24(ye.y, src.pitch) + (uint32_t)w0));
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int dy = dy0 + r0 + r;
            if (dy >= dh) break;
            const uint32_t tl = (uint32_t)wt[r], th = (uint32_t)(wt[r] >> 32), bl = (uint32_t)wb[r], bh = (uint32_t)(wb[r] >> 32);
            uint32_t packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t h0 = hdot(__builtin_amdgcn_perm(th, tl, sel[k]), a[k]);
                const uint32_t h1 = hdot(__builtin_amdgcn_perm(bh, bl, sel[k]), a[k]);
                // (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2: at most 255 (a pixel's weights add up to 2048 both ways)
                const uint32_t v = (rs_wmul(b0s[r], h0 & ~15u) + rs_wmul(b1s[r], h1 & ~15u) + 2u) >> 2;
                packed |= v << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(D + (off24(dy, dst.pitch) + xo)) = packed;
        }
    }
}

// ------------------------------------------------------------------------------------------------ FAST ----
// Ring of the 9-16 segment test (radius 3), position k = 0..15:
//   dx = 0 1 2 3 3 3 2 1 0 -1 -2 -3 -3 -3 -2 -1,  dy = 3 3 2 1 0 -1 -2 -3 -3 -3 -2 -1 0 1 2 3
// (position k + 8 is the point opposite to k; the kernel below loads the ring as eight (k, k + 8) pairs).

// One WAVE per (active cell, frame), four cells per workgroup, no workgroup barriers: every step below is a
// wave-level operation (ballot / DPP-scan compaction, in-order LDS).  The wave stages the cell's ROI in LDS and runs the
// reference's two-threshold sequence (cv::FAST at iniThFAST; only if the cell yields nothing, again at minThFAST --
// :809-816).  The kernel is bound by VALU issue, so both stages run on packed 16-bit math (two values per op):
//   1. compass prefilter on every pixel, EIGHT ADJACENT PIXELS PER LANE from eight aligned LDS dwords: a 9-arc of the
//      16-ring always contains two adjacent compass points (ring positions 0, 4, 8, 12), so two adjacent compass pixels
//      must both differ from the centre by more than t with the same sign.  Survivors are compacted in raster order.
//   2. exact cornerScore of every survivor on (d[k], d[k+8]) pairs with three-input packed minimum / maximum (fast_score16_h);
//      "corner at t" <=> score >= t (the score is the largest threshold at which the pixel is still a corner), so no
//      separate segment test is needed.
//   3. strict 3x3 maximum inside the cell (cv::FAST runs on the ROI, so NMS never looks across cells), ordered write.
// LDS per wave (dynamic, sized by the host from the largest cell of the pyramid): ROI bytes, score map, one list.
// ROI pixel (x, y) lives at byte y * SP + x + 1: the +1 makes every 4-pixel group of step 1 one aligned dword.
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ i16x2 as_i16x2(uint32_t v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t as_u32(i16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2 pk_minu(u16x2 a, u16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ u16x2 pk_maxu(u16x2 a, u16x2 b) { return __builtin_elementwise_max(a, b); }

// min(x, 1) per 16-bit lane: non-zero lanes -> 1
__device__ __forceinline__ uint32_t pk_min1(uint32_t x)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "s"(0x00010001u));
    return r;
}
// compass test of two pixels at once (16-bit lanes); result lanes are non-zero where the pixel passes
__device__ __forceinline__ uint32_t compass_pass2(u16x2 v, u16x2 r0, u16x2 r4, u16x2 r8, u16x2 r12, u16x2 t2)
{
    // two adjacent compass pixels both > v + t  <=>  max over the 4 adjacent pairs of min(pair) > v + t.  The four adjacent pairs of
    // the cycle 0 - 4 - 8 - 12 are exactly {r0, r8} x {r4, r12}, so that maximum is min(max(r0, r8), max(r4, r12)): 3 operations, not 7
    const u16x2 hi = pk_minu(pk_maxu(r0, r8), pk_maxu(r4, r12));
    const u16x2 lo = pk_maxu(pk_minu(r0, r8), pk_minu(r4, r12));
    const u16x2 ph = __builtin_elementwise_sub_sat(hi, (u16x2)(v + t2));                           // hi > v + t
    const u16x2 pl = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(v, t2), lo);      // lo < v - t
    return as_u32(ph) | as_u32(pl);
}

// cornerScore<16> on pairs P[k] = (r[k], r[k+8]) of RING VALUES: the score is max over the 16 arcs of 9 of min(v - r) (and of
// min(r - v)), minus 1 = max(v - (smallest arc maximum), (largest arc minimum) - v) - 1, so the network runs on the pixels as loaded
// and the centre comes in once at the end (the eight subtractions d = v - r in front of it were 8 of ~86 instructions per trip).
// gfx950 has three-input packed f16 minimum / maximum (v_pk_minimum3_f16, v_pk_maximum3_f16) but no three-input packed integer ones.
// An integer 0 .. 255 read as f16 BITS is the denormal r * 2^-24: the order of such values is the order of the integers, their
// differences are exact, and the bits of a positive result are the integer again -- so the whole network runs on packed f16 without
// a single conversion (the kernel's f16 denormal mode is the default "preserve").  min over 9 consecutive = min3 of (min3 of 3) at
// offsets 0, 3, 6: 16 + 16 instructions for all 16 arcs of both kinds, and the half swaps a wrapped index needs are operand selects.
//   SW bit i: operand i is taken with its halves swapped
template <int SW> __device__ __forceinline__ uint32_t hmin3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    if constexpr (SW == 0) asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if constexpr (SW == 4) asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int SW> __device__ __forceinline__ uint32_t hmax3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    if constexpr (SW == 0) asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if constexpr (SW == 4) asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// v2 = the centre pixel in both halves
__device__ __forceinline__ int fast_score16_h(const uint32_t P[8], uint32_t v2)
{
    uint32_t n3[8], x3[8]; // (m3[k], m3[k+8]),  m3[k] = min / max of r[k], r[k+1], r[k+2]
#pragma unroll
    for (int k = 0; k < 6; k++) { n3[k] = hmin3<0>(P[k], P[k + 1], P[k + 2]); x3[k] = hmax3<0>(P[k], P[k + 1], P[k + 2]); }
    n3[6] = hmin3<4>(P[6], P[7], P[0]); x3[6] = hmax3<4>(P[6], P[7], P[0]);
    n3[7] = hmin3<6>(P[7], P[0], P[1]); x3[7] = hmax3<6>(P[7], P[0], P[1]);
    uint32_t n9[8], x9[8]; // m9[k] = min / max of m3[k], m3[k+3], m3[k+6] = of r[k] .. r[k+8]
    n9[0] = hmin3<0>(n3[0], n3[3], n3[6]); x9[0] = hmax3<0>(x3[0], x3[3], x3[6]);
    n9[1] = hmin3<0>(n3[1], n3[4], n3[7]); x9[1] = hmax3<0>(x3[1], x3[4], x3[7]);
    n9[2] = hmin3<4>(n3[2], n3[5], n3[0]); x9[2] = hmax3<4>(x3[2], x3[5], x3[0]);
    n9[3] = hmin3<4>(n3[3], n3[6], n3[1]); x9[3] = hmax3<4>(x3[3], x3[6], x3[1]);
    n9[4] = hmin3<4>(n3[4], n3[7], n3[2]); x9[4] = hmax3<4>(x3[4], x3[7], x3[2]);
    n9[5] = hmin3<6>(n3[5], n3[0], n3[3]); x9[5] = hmax3<6>(x3[5], x3[0], x3[3]);
    n9[6] = hmin3<6>(n3[6], n3[1], n3[4]); x9[6] = hmax3<6>(x3[6], x3[1], x3[4]);
    n9[7] = hmin3<6>(n3[7], n3[2], n3[5]); x9[7] = hmax3<6>(x3[7], x3[2], x3[5]);
    // darkest ring arc against the centre: the largest arc minimum; brightest: the smallest arc maximum
    const uint32_t bn = hmax3<0>(hmax3<0>(n9[0], n9[1], n9[2]), hmax3<0>(n9[3], n9[4], n9[5]), hmax3<0>(n9[6], n9[7], n9[7]));
    const uint32_t bx = hmin3<0>(hmin3<0>(x9[0], x9[1], x9[2]), hmin3<0>(x9[3], x9[4], x9[5]), hmin3<0>(x9[6], x9[7], x9[7]));
    uint32_t da, db, m;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(da) : "v"(bn), "v"(v2)); // (largest arc minimum) - v per half
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(db) : "v"(v2), "v"(bx)); // v - (smallest arc maximum)
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(m) : "v"(da), "v"(db));
    uint32_t sres;
    asm("v_pk_max_f16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(sres) : "v"(m));    // low half: max of the two halves
    return (int)(short)(sres & 0xffffu) - 1; // the bits of a positive denormal are the integer; anything else is far below any threshold
}

__global__ __launch_bounds__(256) void k_fast_cells(ImgView src0, ImgView pyr, const LevelGeom* __restrict__ geom,
                                                    const uint32_t* __restrict__ cellinfo, uint32_t* __restrict__ slots,
                                                    size_t slots_fstride, int32_t* __restrict__ cellcnt,
                                                    int ncells_total, int iniTh, int minTh, int roi_pitch, int roi_rows,
                                                    int map_pitch, int map_rows, int list_cap, int nx, int total,
                                                    int cell_base, int cell_end /* this launch covers cells [cell_base, cell_end) */)
{
    extern __shared__ __align__(16) unsigned char fc_smem[];
#ifdef FC_VGPR_TOP
    // Occupancy cap without an LDS request: claiming a high register makes the allocation FC_VGPR_TOP + 1 registers per lane, so fewer
    // waves (and with them fewer of this kernel's LDS-holding workgroups) fit a CU, and the detector's 65 - 77 KB workgroups find room
    asm volatile("" ::: FC_VGPR_TOP);
#endif
#ifdef ORBFE_PRIO_FAST
    __builtin_amdgcn_s_setprio(ORBFE_PRIO_FAST); // (experiment: 1.53 against 1.36 ms per C2 step at priority 1)
#endif
    const int lane = threadIdx.x & 63, wid = wave_id();
    WT_BEGIN();
    int bx, f;
    if (!xcd_remap(nx, total, bx, f)) return;
    const int cell = cell_base + bx * 4 + wid;
    if (cell >= cell_end) return;
    WT_COUNT(1);
    const size_t roi_bytes = ((size_t)roi_pitch * roi_rows + 15) & ~(size_t)15;
    const size_t map_bytes = ((size_t)map_pitch * map_rows + 15) & ~(size_t)15;
    const size_t per_wave = roi_bytes + map_bytes + (((size_t)list_cap * 2 + 15) & ~(size_t)15);
    uint8_t* simg = fc_smem + per_wave * wid;
    uint8_t* smap = simg + roi_bytes;
    uint16_t* slist = (uint16_t*)(smap + map_bytes);

    const uint32_t ci = cellinfo[cell];
    const int level = ci & 15, ci_i = (ci >> 4) & 1023, ci_j = (ci >> 14) & 1023;
    const LevelGeom g = geom[level];
    const int iniX = 16 + ci_j * g.wCell, iniY = 16 + ci_i * g.hCell;
    const int maxX = min(iniX + g.wCell + 6, g.maxBX), maxY = min(iniY + g.hCell + 6, g.maxBY);
    const int w = maxX - iniX, h = maxY - iniY;
    const int aw = w - 6, ah = h - 6;
    int32_t* cnt_out = cellcnt + (size_t)f * ncells_total + cell;
    if (aw <= 0 || ah <= 0) {
        if (lane == 0) *cnt_out = 0;
        return;
    }
    const uint8_t* img = (level == 0) ? src0.base + (size_t)f * src0.fstride
                                      : pyr.base + (size_t)f * pyr.fstride + g.img_off;
    const int pitch = (level == 0) ? src0.pitch : g.pitch;
    const int SP = roi_pitch, MP = map_pitch;

    // stage the ROI (columns -1 .. w-1: the one-byte shift) with unaligned dword loads, 8 in flight per lane.  A lane keeps its
    // dword column and takes every (64 / ndw)-th row: its image and LDS offsets advance by constants (uniform base + 32-bit lane
    // offset, one addition each per load -- the flat (row, dword) numbering cost a wrap test, three selects and a 64-bit
    // multiply-add per item: 115 of the kernel's ~890 VALU instructions per wave); the 64 % ndw lanes left over idle here.
    // (16-byte chunks, three to a row and two loads a lane, save 42 of these instructions and are no faster at 640 x 480 -- and 3 % slower
    // alone, 18 % in the pipeline, at 1920 x 1080, where the step then loses 3.5 %: unaligned 16-byte lanes straddle sectors)
    {
        typedef uint32_t u32_unaligned __attribute__((aligned(1)));
        const int ndw = (w + 4) >> 2;
        const int rpi = 64 / ndw;                      // rows per trip of the wave
        const int y0 = (int)(((float)lane + 0.5f) * (1.0f / (float)ndw)), c = lane - __mul24(y0, ndw); // exact: lane < 64
        const uint8_t* roi = img + (size_t)iniY * pitch + iniX - 1;
        // (24-bit multiplies fold into v_mad_u32_u24 with the column offset)
        uint32_t go = (uint32_t)(__mul24(y0, pitch) + 4 * c), so_ = (uint32_t)(__mul24(y0, SP) + 4 * c);
        const uint32_t gstep = (uint32_t)(rpi * pitch), sstep = (uint32_t)(rpi * SP);
        // no predicates (a load under a condition becomes a branch with a wait behind it): a row past the end -- and with it the
        // lanes left over -- takes the ROI's last row, i.e. loads and stores that row's dword once more
        const uint32_t go_last = (uint32_t)((h - 1) * pitch + 4 * c), so_last = (uint32_t)((h - 1) * SP + 4 * c);
        for (int r0 = 0; r0 < h; r0 += 8 * rpi) {
            uint32_t v[8], so[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool in = r0 + k * rpi + y0 < h;
                const uint32_t g_ = in ? go + (uint32_t)k * gstep : go_last;
                so[k] = in ? so_ + (uint32_t)k * sstep : so_last;
                v[k] = *reinterpret_cast<const u32_unaligned*>(roi + g_);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) *reinterpret_cast<uint32_t*>(simg + so[k]) = v[k];
            go += 8 * gstep; so_ += 8 * sstep;
        }
    }
    for (int i = lane; i < (int)(map_bytes / 4); i += 64) reinterpret_cast<uint32_t*>(smap)[i] = 0;
    __builtin_amdgcn_wave_barrier();
    WT_MARK(1, 0);   // cell record, geometry, ROI loads, LDS stores

    const int G = (aw + 7) >> 3, nitems = G * ah; // 8-pixel groups per active row
    const float inv_G = 1.0f / (float)G;
    int nlist = 0;
    unsigned long long keepbits = 0; // bit k: list entry lane + 64 k survives NMS (list_cap <= 64 * 64)
    for (int pass_no = 0; pass_no < 2; pass_no++) {
        const int t = pass_no == 0 ? iniTh : minTh;
        const u16x2 t2 = as_u16x2((uint32_t)t | ((uint32_t)t << 16));
        // ---- 1: compass prefilter, 8 adjacent pixels per lane (two aligned dwords of the centre row and their neighbours)
        int n1 = 0;
        for (int base = 0; base < nitems; base += 64) {
            const int it = base + lane;
            unsigned mask8 = 0;
            int x = 0, y = 0;
            if (it < nitems) {
                y = (int)(((float)it + 0.5f) * inv_G); // exact: it < 4096, quotient >= 0.5/G away from an integer
                x = 8 * (it - __mul24(y, G));
                // pixels x .. x+7 of active row y = ROI pixels (x+3 .. x+10, y+3) = bytes x+4 .. x+11 of ROI row y+3
                const uint8_t* rb = simg + __mul24(y, SP) + x + 4;
                const uint32_t* cr = reinterpret_cast<const uint32_t*>(rb + 3 * SP);
                const uint32_t* nr = reinterpret_cast<const uint32_t*>(rb);
                const uint32_t* sr = reinterpret_cast<const uint32_t*>(rb + 6 * SP);
                const uint32_t Wd = cr[-1], C0 = cr[0], C1 = cr[1], Ed = cr[2];
                const uint32_t N0 = nr[0], N1 = nr[1], S0 = sr[0], S1 = sr[1];
                const uint32_t E0 = __builtin_amdgcn_alignbyte(C1, C0, 3), W0 = __builtin_amdgcn_alignbyte(C0, Wd, 1);
                const uint32_t E1 = __builtin_amdgcn_alignbyte(Ed, C1, 3), W1 = __builtin_amdgcn_alignbyte(C1, C0, 1);
                // ring positions: 0 = (x, y+3) = S row, 4 = (x+3, y) = E, 8 = (x, y-3) = N row, 12 = (x-3, y) = W
                constexpr uint32_t LO = 0x0c010c00u, HI = 0x0c030c02u; // bytes (0, 1) / (2, 3) -> 16-bit lanes
#define FC_PASS2(C, S, E, N, W, SEL)                                                                                             \
    compass_pass2(as_u16x2(__builtin_amdgcn_perm(0, C, SEL)), as_u16x2(__builtin_amdgcn_perm(0, S, SEL)),                       \
                  as_u16x2(__builtin_amdgcn_perm(0, E, SEL)), as_u16x2(__builtin_amdgcn_perm(0, N, SEL)),                       \
                  as_u16x2(__builtin_amdgcn_perm(0, W, SEL)), t2)
                const uint32_t p01 = FC_PASS2(C0, S0, E0, N0, W0, LO), p23 = FC_PASS2(C0, S0, E0, N0, W0, HI);
                const uint32_t p45 = FC_PASS2(C1, S1, E1, N1, W1, LO), p67 = FC_PASS2(C1, S1, E1, N1, W1, HI);
#undef FC_PASS2
                // non-zero 16-bit lanes -> bits 0 .. 7: clamp every lane to 0 / 1, interleave the four words, fold the high halves in
                // (written as the instruction: from the builtin the compiler makes two compares, two selects and a v_perm per word)
                const uint32_t b = pk_min1(p01) | (pk_min1(p23) << 2) | (pk_min1(p45) << 4) | (pk_min1(p67) << 6);
                mask8 = (b & 0x55u) | ((b >> 15) & 0xaau);
                mask8 &= (1u << min(8, aw - x)) - 1u; // the last group of a row may be partial
            }
            const int cnt = __popc(mask8);
            const int incl = wave_incl_scan_add(cnt);
            int off = n1 + incl - cnt;
            const int yx0 = (y << 8) | x;
            while (mask8) { // raster order inside the group
                const int b = __ffs(mask8) - 1;
                mask8 &= mask8 - 1;
                slist[off++] = (uint16_t)(yx0 + b);
            }
            n1 += __builtin_amdgcn_readlane(incl, 63);
        }
        __builtin_amdgcn_wave_barrier();
        WT_MARK(1, 1);   // prefilter + compaction
        // ---- 2: exact score of the survivors; corners (score >= t) into the score map, list compacted in place
        // (order-preserving; writes trail reads)
        nlist = 0;
        for (int base = 0; base < n1; base += 64) {
            const int e = base + lane;
            bool corner = false;
            int yx = 0, score = 0;
            if (e < n1) {
                yx = slist[e];
                const uint8_t* c = simg + __mul24(yx >> 8, SP) + (yx & 255) + (3 * SP + 4);
                const uint32_t v = c[0];
                const uint32_t v2 = v | (v << 16);
                uint32_t P[8]; // ring pixels (r[k], r[k+8]) as f16 bits
                // (a d16 byte load per half would join the pairs for free, but with SRAM ECC -- gfx950 -- a d16 load clears the other half)
                P[0] = (uint32_t)c[3 * SP + 0] | ((uint32_t)c[-3 * SP + 0] << 16);
                P[1] = (uint32_t)c[3 * SP + 1] | ((uint32_t)c[-3 * SP - 1] << 16);
                P[2] = (uint32_t)c[2 * SP + 2] | ((uint32_t)c[-2 * SP - 2] << 16);
                P[3] = (uint32_t)c[1 * SP + 3] | ((uint32_t)c[-1 * SP - 3] << 16);
                P[4] = (uint32_t)c[3] | ((uint32_t)c[-3] << 16);
                P[5] = (uint32_t)c[-1 * SP + 3] | ((uint32_t)c[1 * SP - 3] << 16);
                P[6] = (uint32_t)c[-2 * SP + 2] | ((uint32_t)c[2 * SP - 2] << 16);
                P[7] = (uint32_t)c[-3 * SP + 1] | ((uint32_t)c[3 * SP - 1] << 16);
                score = fast_score16_h(P, v2);
                corner = score >= t;
            }
            const unsigned long long m = __ballot(corner);
            __builtin_amdgcn_wave_barrier();
            if (corner) {
                slist[nlist + lane_prefix(m)] = (uint16_t)yx;
                smap[__mul24(yx >> 8, MP) + (yx & 255) + (MP + 1)] = (uint8_t)score;
            }
            nlist += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        WT_MARK(1, 2);   // scores
        // ---- 3: strict 3x3 maximum inside the cell
        keepbits = 0;
        int nkeep = 0;
        for (int k = 0; k * 64 < nlist; k++) {
            const int e = k * 64 + lane;
            bool keep = false;
            if (e < nlist) {
                const int yx = slist[e], y = yx >> 8, x = yx & 255;
                const uint8_t* m = smap + __mul24(y, MP) + x + (MP + 1);
                const int s = m[0];
                keep = s > m[-MP - 1] && s > m[-MP] && s > m[-MP + 1] && s > m[-1] && s > m[1] && s > m[MP - 1] &&
                       s > m[MP] && s > m[MP + 1];
            }
            keepbits |= (unsigned long long)keep << k;
            nkeep += __popcll(__ballot(keep));
        }
        WT_MARK(1, 3);   // NMS
        if (nkeep > 0 || pass_no == 1) break;
        WT_RETRY(1);
        // nothing at iniThFAST: wipe the scores and try again at minThFAST
        for (int e = lane; e < nlist; e += 64) {
            const int yx = slist[e];
            smap[__mul24(yx >> 8, MP) + (yx & 255) + (MP + 1)] = 0;
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- 4: ordered write-out
    uint32_t* out = slots + (size_t)f * slots_fstride + g.slot_off + (size_t)(cell - g.cell_first) * g.cell_cap;
    const int xrel = ci_j * g.wCell + 3, yrel = ci_i * g.hCell + 3; // (*vit).pt.x += j*wCell (:822-823)
    int nout = 0;
    for (int k = 0; k * 64 < nlist; k++) {
        const int e = k * 64 + lane;
        const bool take = (keepbits >> k) & 1ull;
        const unsigned long long m = __ballot(take);
        if (take) {
            const int yx = slist[e], y = yx >> 8, x = yx & 255;
            out[nout + lane_prefix(m)] = (uint32_t)(x + xrel) | ((uint32_t)(y + yrel) << 12) |
                                         ((uint32_t)smap[__mul24(y, MP) + x + (MP + 1)] << 24);
        }
        nout += __popcll(m);
    }
    if (lane == 0) *cnt_out = nout;
    WT_MARK(1, 4);   // write-out
}

// ------------------------------------------------------------------------------------------------ quadtree --
// DistributeOctTree (ORBextractor.cc:539-763), one wave per (frame, level).  The std::list of nodes is a doubly
// linked list in LDS; a node's keypoints are a contiguous, order-preserving range of a ping-pong key buffer
// (LDS when the level's candidates fit, HBM scratch otherwise), so DivideNode (:481-537) is a stable 4-way
// partition done with wave ballots.  Control flow is wave-uniform.  Tie-break of the (size, pointer) sort at :684:
// creation sequence, as the oracle defines it.
struct QtShared {
    // sizes are set by the launch (dynamic LDS); see qt_lds_bytes()
    uint32_t* keys[2];
    short *x0, *y0, *x1, *y1;
    int *begin, *count;
    short *next, *prev;
    uint8_t* flags; // bit0 = key buffer, bit1 = bNoMore
    unsigned long long *vec, *vprev;
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// The workgroup is a single wave: LDS operations of one wave execute in order, so a compiler-level barrier is enough.
#define QT_SYNC() __builtin_amdgcn_wave_barrier()
// One (frame, level) of the general kernel; `qt_smem` is the workgroup's dynamic LDS (qt_lds_bytes()).
__device__ __forceinline__ void qt_general_level(unsigned char* qt_smem, int level, int f, const LevelGeom* __restrict__ geom,
                                                 const uint32_t* __restrict__ slots, size_t slots_fstride,
                                                 const int32_t* __restrict__ cellcnt, int ncells_total,
                                                 uint32_t* __restrict__ keyscratch, size_t keys_fstride,
                                                 uint32_t* __restrict__ lvl_out, int out_fstride, int32_t* __restrict__ lvl_cnt,
                                                 int nlevels, int32_t* __restrict__ lvl_ncand, int keycap_lds, int nodecap, int veccap)
{
    const int lane = threadIdx.x;
    const LevelGeom g = geom[level];

    // carve LDS
    unsigned char* sp = qt_smem;
    unsigned long long* vec = (unsigned long long*)sp; sp += (size_t)veccap * 8;
    unsigned long long* vprev = (unsigned long long*)sp; sp += (size_t)veccap * 8;
    int* nbegin = (int*)sp; sp += (size_t)nodecap * 4;
    int* ncount = (int*)sp; sp += (size_t)nodecap * 4;
    int* nseq = (int*)sp; sp += (size_t)nodecap * 4;
    short* nx0 = (short*)sp; sp += (size_t)nodecap * 2;
    short* ny0 = (short*)sp; sp += (size_t)nodecap * 2;
    short* nx1 = (short*)sp; sp += (size_t)nodecap * 2;
    short* ny1 = (short*)sp; sp += (size_t)nodecap * 2;
    short* nnext = (short*)sp; sp += (size_t)nodecap * 2;
    short* nprev = (short*)sp; sp += (size_t)nodecap * 2;
    short* nfree = (short*)sp; sp += (size_t)nodecap * 2;
    uint8_t* nflag = (uint8_t*)sp; sp += ((size_t)nodecap + 15) & ~(size_t)15;
    sp = qt_smem + (((size_t)(sp - qt_smem) + 15) & ~(size_t)15);
    uint32_t* lkeys = (uint32_t*)sp; // 2 * keycap_lds

#ifdef ORBFE_QT_TIMING
    const long long qt0 = clock64();
#endif
    // ---- gather the level's candidates in cell row-major order (= vToDistributeKeys order, :819-826)
    const int32_t* ccnt = cellcnt + (size_t)f * ncells_total + g.cell_first;
    const uint32_t* cslots = slots + (size_t)f * slots_fstride + g.slot_off;
    int n = 0;
    for (int c0 = 0; c0 < g.ncells; c0 += 64) {
        const int c = c0 + lane;
        n += (c < g.ncells) ? ccnt[c] : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    n = rfl(n);
    uint32_t* kb0;
    uint32_t* kb1;
    if (n <= keycap_lds) {
        kb0 = lkeys;
        kb1 = lkeys + keycap_lds;
    } else {
        kb0 = keyscratch + (size_t)f * keys_fstride + 2 * g.cand_off;
        kb1 = kb0 + g.cand_cap;
    }
    {
        int base = 0;
        for (int c0 = 0; c0 < g.ncells; c0 += 64) {
            const int c = c0 + lane;
            const int cnt = (c < g.ncells) ? ccnt[c] : 0;
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            const int off = base + incl - cnt;
            int mx = cnt;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
            const uint32_t* s = cslots + (size_t)c * g.cell_cap;
            for (int k0 = 0; k0 < mx; k0 += 8) { // 8 loads in flight per lane, then the 8 LDS stores
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = (k0 + k < cnt) ? s[k0 + k] : 0u;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (k0 + k < cnt) kb0[off + k0 + k] = v[k];
            }
            base += __shfl(incl, 63);
        }
    }
    if (lane == 0) lvl_ncand[f * nlevels + level] = n;
    QT_SYNC();
#ifdef ORBFE_QT_TIMING
    const long long qt1 = clock64();
#endif

    uint32_t* outp = lvl_out + (size_t)f * out_fstride + g.out_off;
    if (n == 0) {
        if (lane == 0) lvl_cnt[f * nlevels + level] = 0;
        return;
    }

    // ---- node list
    int head = -1, tail = -1, size = 0, nfreecnt = 0, nalloc = 0, seq = 0, nvec = 0;
    const int N = g.quota;
    uint32_t* kbuf[2] = {kb0, kb1};

    auto alloc_node = [&]() -> int {
        int id;
        if (nfreecnt > 0) id = nfree[--nfreecnt];
        else id = nalloc++;
        return id;
    };
    auto push_front = [&](int id) {
        if (lane == 0) {
            nprev[id] = -1;
            nnext[id] = (short)head;
            if (head >= 0) nprev[head] = (short)id;
        }
        if (head < 0) tail = id;
        head = id;
        size++;
    };
    auto push_back = [&](int id) {
        if (lane == 0) {
            nnext[id] = -1;
            nprev[id] = (short)tail;
            if (tail >= 0) nnext[tail] = (short)id;
        }
        if (tail < 0) head = id;
        tail = id;
        size++;
    };
    auto erase = [&](int id) {
        const int p = nprev[id], q = nnext[id];
        QT_SYNC();
        if (lane == 0) {
            if (p >= 0) nnext[p] = (short)q;
            if (q >= 0) nprev[q] = (short)p;
            nfree[nfreecnt] = (short)id;
        }
        if (p < 0) head = q;
        if (q < 0) tail = p;
        nfreecnt++;
        size--;
        QT_SYNC();
    };

    // root nodes (:546-570): nIni vertical strips, keys assigned by (int)(x / hX)
    {
        int cnt[QT_MAXROOTS], beginq[QT_MAXROOTS], run[QT_MAXROOTS];
#pragma unroll
        for (int q = 0; q < QT_MAXROOTS; q++) cnt[q] = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            int r = -1;
            if (i < n) r = (int)__fdiv_rn((float)(kb0[i] & 0xfff), g.hX);
#pragma unroll
            for (int q = 0; q < QT_MAXROOTS; q++)
                if (q < g.nIni) cnt[q] += __popcll(__ballot(r == q));
        }
        int acc = 0;
#pragma unroll
        for (int q = 0; q < QT_MAXROOTS; q++) { beginq[q] = acc; run[q] = acc; acc += cnt[q]; }
        if (g.nIni > 1) {
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                int r = -1;
                uint32_t kv = 0;
                if (i < n) { kv = kb0[i]; r = (int)__fdiv_rn((float)(kv & 0xfff), g.hX); }
#pragma unroll
                for (int q = 0; q < QT_MAXROOTS; q++)
                    if (q < g.nIni) {
                        const unsigned long long m = __ballot(r == q);
                        if (r == q) kb1[run[q] + lane_prefix(m)] = kv;
                        run[q] += __popcll(m);
                    }
            }
        }
        QT_SYNC();
        const int rootbuf = (g.nIni == 1) ? 0 : 1;
#pragma unroll
        for (int q = 0; q < QT_MAXROOTS; q++) {
            if (q >= g.nIni) continue;
            // the reference creates every root, then erases the empty ones (:574-585); seq counts all of them
            const int myseq = seq++;
            if (cnt[q] == 0) continue;
            const int id = alloc_node();
            if (lane == 0) {
                nx0[id] = (short)(int)(g.hX * (float)q);
                nx1[id] = (short)(int)(g.hX * (float)(q + 1));
                ny0[id] = 0;
                ny1[id] = (short)(g.maxBY - 16);
                nbegin[id] = beginq[q];
                ncount[id] = cnt[q];
                nseq[id] = myseq;
                nflag[id] = (uint8_t)(rootbuf | (cnt[q] == 1 ? 2 : 0));
            }
            push_back(id);
        }
        QT_SYNC();
    }

    // DivideNode + "add childs if they contain points" (:604-656 / :689-728); returns nothing, updates list + vec
    auto split = [&](int id, int* nToExpand) {
        const int x0 = nx0[id], y0 = ny0[id], x1 = nx1[id], y1 = ny1[id];
        const int b = nbegin[id], cnt = ncount[id], buf = nflag[id] & 1;
        const int halfX = (x1 - x0 + 1) >> 1, halfY = (y1 - y0 + 1) >> 1; // ceil(float(d)/2), d >= 0
        const int xm = x0 + halfX, ym = y0 + halfY;
        const uint32_t* srck = kbuf[buf] + b;
        uint32_t* dstk = kbuf[buf ^ 1] + b;
        int c[4] = {0, 0, 0, 0};
        int cb[4];
        if (cnt <= 64) {
            // the whole node fits one wave: classify once, counts and positions straight from four ballots
            int q = -1;
            uint32_t kv = 0;
            if (lane < cnt) {
                kv = srck[lane];
                const int kx = kv & 0xfff, ky = (kv >> 12) & 0xfff;
                q = (kx < xm ? 0 : 1) + (ky < ym ? 0 : 2);
            }
            const unsigned long long m0 = __ballot(q == 0), m1 = __ballot(q == 1), m2 = __ballot(q == 2),
                                     m3 = __ballot(q == 3);
            c[0] = __popcll(m0); c[1] = __popcll(m1); c[2] = __popcll(m2); c[3] = __popcll(m3);
            cb[0] = 0; cb[1] = c[0]; cb[2] = c[0] + c[1]; cb[3] = c[0] + c[1] + c[2];
            if (q >= 0) {
                const unsigned long long mq = q == 0 ? m0 : q == 1 ? m1 : q == 2 ? m2 : m3;
                const int base = q == 0 ? cb[0] : q == 1 ? cb[1] : q == 2 ? cb[2] : cb[3];
                dstk[base + lane_prefix(mq)] = kv;
            }
        } else {
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                const int i = i0 + lane;
                int q = -1;
                if (i < cnt) {
                    const uint32_t kv = srck[i];
                    const int kx = kv & 0xfff, ky = (kv >> 12) & 0xfff;
                    q = (kx < xm ? 0 : 1) + (ky < ym ? 0 : 2);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) c[k] += __popcll(__ballot(q == k));
            }
            int run[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};
            cb[0] = run[0]; cb[1] = run[1]; cb[2] = run[2]; cb[3] = run[3];
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                const int i = i0 + lane;
                int q = -1;
                uint32_t kv = 0;
                if (i < cnt) {
                    kv = srck[i];
                    const int kx = kv & 0xfff, ky = (kv >> 12) & 0xfff;
                    q = (kx < xm ? 0 : 1) + (ky < ym ? 0 : 2);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned long long m = __ballot(q == k);
                    if (q == k) dstk[run[k] + lane_prefix(m)] = kv;
                    run[k] += __popcll(m);
                }
            }
        }
        QT_SYNC();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (c[k] == 0) continue;
            const int cid = alloc_node();
            const int myseq = seq++;
            if (lane == 0) {
                nx0[cid] = (short)((k & 1) ? xm : x0);
                nx1[cid] = (short)((k & 1) ? x1 : xm);
                ny0[cid] = (short)((k & 2) ? ym : y0);
                ny1[cid] = (short)((k & 2) ? y1 : ym);
                nbegin[cid] = b + cb[k];
                ncount[cid] = c[k];
                nseq[cid] = myseq;
                nflag[cid] = (uint8_t)((buf ^ 1) | (c[k] == 1 ? 2 : 0));
            }
            push_front(cid);
            if (c[k] > 1) {
                if (nToExpand) (*nToExpand)++;
                if (lane == 0)
                    vec[nvec] = ((unsigned long long)c[k] << 40) | ((unsigned long long)myseq << 16) |
                                (unsigned long long)cid;
                nvec++;
            }
        }
        QT_SYNC();
    };

    bool finish = false;
    while (!finish) {
        const int prevSize = size;
        int nToExpand = 0;
        nvec = 0;
        int cur = head;
        while (cur >= 0) {
            const int nxt = rfl(nnext[cur]);
            const int fl = rfl(nflag[cur]);
            if (!(fl & 2)) {
                split(cur, &nToExpand);
                erase(cur);
            }
            cur = nxt;
        }
        if (size >= N || size == prevSize) {
            finish = true;
        } else if (size + nToExpand * 3 > N) {
            while (!finish) {
                const int prevSize2 = size;
                // vPrev = vSizeAndPointerToNode, sorted ascending by (size, seq); bitonic sort over a power of two
                const int np = nvec;
                int P = 1;
                while (P < np) P <<= 1;
                for (int i = lane; i < P; i += 64) vprev[i] = (i < np) ? vec[i] : 0ull;
                QT_SYNC();
                for (int k = 2; k <= P; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int t = lane; t < (P >> 1); t += 64) {
                            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                            const int l = i | j;
                            const unsigned long long a = vprev[i], bb = vprev[l];
                            const bool up = ((i & k) == 0);
                            if ((a > bb) == up) { vprev[i] = bb; vprev[l] = a; }
                        }
                        QT_SYNC();
                    }
                nvec = 0;
                for (int j = P - 1; j >= P - np; j--) {
                    const unsigned long long e = vprev[j];
                    const int id = rfl((int)(e & 0xffff));
                    split(id, nullptr);
                    erase(id);
                    if (size >= N) break;
                }
                if (size >= N || size == prevSize2) finish = true;
            }
        }
    }

#ifdef ORBFE_QT_TIMING
    const long long qt2 = clock64();
#endif
    // ---- retain the best point of each node (:741-760), in list order
    // walk the list once (wave-uniform) recording node ids, then one lane per node
    short* order = (short*)vprev; // reuse (veccap*8 bytes >= nodecap*2)
    {
        int cur = head, k = 0;
        while (cur >= 0) {
            if (lane == 0) order[k] = (short)cur;
            k++;
            cur = rfl(nnext[cur]);
        }
    }
    QT_SYNC();
    for (int k0 = 0; k0 < size; k0 += 64) {
        const int k = k0 + lane;
        if (k < size) {
            const int id = order[k];
            const uint32_t* kk = kbuf[nflag[id] & 1] + nbegin[id];
            const int cnt = ncount[id];
            uint32_t best = kk[0];
            for (int i = 1; i < cnt; i++) {
                const uint32_t kv = kk[i];
                if ((kv >> 24) > (best >> 24)) best = kv; // strict '>' : first one wins ties (:752)
            }
            outp[k] = best;
        }
    }
    if (lane == 0) lvl_cnt[f * nlevels + level] = size;
#ifdef ORBFE_QT_TIMING
    if (lane == 0) lvl_ncand[f * nlevels + level] = (int)((qt1 - qt0) >> 8) | ((int)((qt2 - qt1) >> 8) << 10) | ((int)((clock64() - qt2) >> 8) << 20);
#endif
}

// The general kernel is the fall-back of k_distribute_pyr: on ordinary frames no level needs it.  As a grid of one workgroup per
// (level, frame) -- 2400 workgroups of ~60 KB of LDS each at C2 -- the launch that did NOTHING still took 73 us on average in the
// pipeline (4 .. 170; 11 alone: every workgroup waits for 60 KB of LDS on a CU the other engines' kernels fill), on the extractor's
// critical chain between the quadtree and the descriptors.  Now the fast path appends the levels it gives up on to a work list, and a
// small grid of workgroups drains it (an empty list costs one launch of `gridDim.x` idle waves).  worklist == nullptr: every
// (level, frame) pair, one workgroup each (the test hook that runs the general kernel for everything).
__global__ __launch_bounds__(64) void k_distribute(const LevelGeom* __restrict__ geom, const uint32_t* __restrict__ slots,
                                                   size_t slots_fstride, const int32_t* __restrict__ cellcnt,
                                                   int ncells_total, uint32_t* __restrict__ keyscratch,
                                                   size_t keys_fstride, uint32_t* __restrict__ lvl_out,
                                                   int out_fstride, int32_t* __restrict__ lvl_cnt, int nlevels,
                                                   int32_t* __restrict__ lvl_ncand, int keycap_lds, int nodecap,
                                                   int veccap, const int32_t* __restrict__ worklist,
                                                   const int32_t* __restrict__ worklist_n)
{
    extern __shared__ __align__(16) unsigned char qt_smem[];
    if (!worklist) {
        qt_general_level(qt_smem, blockIdx.x, blockIdx.y, geom, slots, slots_fstride, cellcnt, ncells_total, keyscratch, keys_fstride,
                         lvl_out, out_fstride, lvl_cnt, nlevels, lvl_ncand, keycap_lds, nodecap, veccap);
        return;
    }
    const int n = *worklist_n;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int fl = worklist[i];
        const int f = fl / nlevels;
        qt_general_level(qt_smem, fl - f * nlevels, f, geom, slots, slots_fstride, cellcnt, ncells_total, keyscratch, keys_fstride,
                         lvl_out, out_fstride, lvl_cnt, nlevels, lvl_ncand, keycap_lds, nodecap, veccap);
        QT_SYNC(); // the next item reuses the LDS
    }
}

// ------------------------------------------------------------------------------------------------ quadtree, fast path
// DistributeOctTree without moving keypoints.  A node's rectangle depends only on its path from the root (DivideNode
// halves with ceil, :483-484), so every candidate has a fixed cell at every depth.  One pass over the candidates builds,
// in LDS, a COUNT PYRAMID (keypoints per cell for depths 0..D, cell index = root * 4^d + path code) and, per depth-D
// leaf, the best keypoint (max response, first in vToDistributeKeys order on ties, :745-757).  The node list logic of
// :594-739 then runs on (cell, depth) records with O(1) child counts, lane-per-node:
//   * a pass of the first loop (:598-657) splits every expandable node at once; the list after the pass is
//     [children in reverse creation order] ++ [bNoMore nodes in their old order], which two prefix sums reproduce;
//   * a round of the priority loop (:673-737) sorts (size, seq) and splits from the back until size >= N: the stopping
//     point is the first prefix of "children - 1" gains that reaches N, found with a scan.
// If the algorithm wants to split a depth-D cell (candidates concentrated in a tiny area) the pyramid is too shallow:
// the workgroup raises `fallback[f, level]` and k_distribute (the general kernel) redoes that level.
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total)
{
    (void)lane;
    const int incl = wave_incl_scan_add(v);
    total = __builtin_amdgcn_readlane(incl, 63);
    return incl - v;
}

__global__ __launch_bounds__(QP_THREADS) void k_distribute_pyr(const LevelGeom* __restrict__ geom, const uint32_t* __restrict__ slots,
                                                       size_t slots_fstride, const int32_t* __restrict__ cellcnt,
                                                       int ncells_total, uint32_t* __restrict__ lvl_out, int out_fstride,
                                                       int32_t* __restrict__ lvl_cnt, int nlevels,
                                                       int32_t* __restrict__ lvl_ncand, int32_t* __restrict__ fallback,
                                                       int D, int nodecap, int veccap, int32_t* __restrict__ worklist,
                                                       int32_t* __restrict__ worklist_n, int by_level)
{
#ifndef ORBFE_PRIO_QT
#define ORBFE_PRIO_QT 2
#endif
    __builtin_amdgcn_s_setprio(ORBFE_PRIO_QT); // latency-bound: its few waves go first when a VALU-bound kernel shares the CU
    extern __shared__ __align__(16) unsigned char qp_smem[];
    __shared__ int s_ncand;
    __shared__ int s_wtot[QP_THREADS / 64];
    // QP_THREADS threads build the leaf counts (the only part that is parallel over candidates), then one wave runs the
    // tree logic
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // by_level: grid (frames, levels) -- every frame's level 0, the level with the most candidates and the longest workgroups, is
    // dispatched first and the short ones fill the tail; else grid (levels, frames)
    const int level = by_level ? blockIdx.y : blockIdx.x, f = by_level ? blockIdx.x : blockIdx.y;
    WT_BEGIN();
    const LevelGeom g = geom[level];
    const int nIni = g.nIni;
    // depth d cells live at cnt[off(d) + root * 4^d + code]; off(d) = 4 * nIni * (4^d - 1) / 3 rounded so that every
    // 4-child group is 8-byte aligned (u16 counts)
    auto off = [&](int d) { return nIni * (((1 << (2 * d)) - 1) / 3) ; };
    const int T = off(D + 1);                     // total cells, depths 0..D
    const int nleaf = nIni << (2 * D);
    unsigned char* sp = qp_smem;
    uint32_t* best = (uint32_t*)sp; sp += (size_t)nleaf * 4; // per leaf: response << 22 | (0x3fffff - order): max = best response, first in order
    unsigned long long* vec = (unsigned long long*)sp; sp += (size_t)veccap * 8;
    unsigned long long* vprev = (unsigned long long*)sp; sp += (size_t)veccap * 8;
    uint32_t* cellA = (uint32_t*)sp; sp += (size_t)nodecap * 4;
    uint32_t* infoA = (uint32_t*)sp; sp += (size_t)nodecap * 4;
    uint32_t* seqA = (uint32_t*)sp; sp += (size_t)nodecap * 4;
    uint32_t* cellB = (uint32_t*)sp; sp += (size_t)nodecap * 4;
    uint32_t* infoB = (uint32_t*)sp; sp += (size_t)nodecap * 4;
    uint32_t* seqB = (uint32_t*)sp; sp += (size_t)nodecap * 4;
    uint16_t* tpos = (uint16_t*)sp; sp += (((size_t)nodecap * 2 + 15) & ~(size_t)15);
    uint16_t* smark = (uint16_t*)sp; sp += (((size_t)nodecap * 2 + 15) & ~(size_t)15);
    uint32_t* cnt32 = (uint32_t*)sp;              // u16 counts, two per word; T rounded up
    uint16_t* cnt = (uint16_t*)cnt32;
    // info word: count (20 bits) | depth << 20 (4 bits) | nomore << 24

    const int fl_idx = f * nlevels + level;
    for (int i = tid; i < (T + 1) / 2 + 2; i += QP_THREADS) cnt32[i] = 0;
    for (int i = tid; i < nleaf; i += QP_THREADS) best[i] = 0u;
    if (tid == 0) s_ncand = 0;
    __syncthreads();

    // ---- one pass over the level's candidates (straight from the per-cell slots; order is carried as (cell, k))
    const int32_t* ccnt = cellcnt + (size_t)f * ncells_total + g.cell_first;
    const uint32_t* cslots = slots + (size_t)f * slots_fstride + g.slot_off;
    const int H = g.maxBY - 16;
    int n = 0;
    // The pass runs a THREAD PER CANDIDATE: a lane per cell walking its cell's slots ran for as many trips as the fullest of a wave's 64
    // cells has candidates (20 - 100) with a seventh of the lanes at work, and the wave that held the textured cells kept the other three
    // waiting -- 60 % of the workgroup's life (tools/wave_timing.sh), on the extractor's critical chain.  Per 1024 cells: the counts are
    // scanned over the workgroup, every cell writes (cell << 10 | slot) of its candidates to the LDS the tree logic uses later (in
    // windows of its capacity), and the 256 threads read the candidates from their slots, four in flight each.
    uint32_t* clist = reinterpret_cast<uint32_t*>(vec);
    const int lcap = (int)(((unsigned char*)cnt32 - (unsigned char*)vec) / 4);
    constexpr int QP_CPT = 4;   // cells a thread: a level of up to 1024 cells (every level up to 1280 x 720's level 1) is one round of barriers
    for (int cbase = 0; cbase < g.ncells; cbase += QP_THREADS * QP_CPT) {
        const int cfirst = cbase + tid * QP_CPT;
        int kc[QP_CPT], k_cnt = 0;
#pragma unroll
        for (int j = 0; j < QP_CPT; j++) kc[j] = ccnt[min(cfirst + j, g.ncells - 1)];
#pragma unroll
        for (int j = 0; j < QP_CPT; j++) { kc[j] = cfirst + j < g.ncells ? kc[j] : 0; k_cnt += kc[j]; }
        n += k_cnt;
        int wtotal;
        int excl = wave_excl_scan(k_cnt, lane, wtotal);
        if (lane == 0) s_wtot[wid] = wtotal;
        __syncthreads();
        int total = 0;
#pragma unroll
        for (int w = 0; w < QP_THREADS / 64; w++) { const int t = s_wtot[w]; excl += w < wid ? t : 0; total += t; }
        if (total == 0) __syncthreads();   // (the totals are rewritten for the next cells)
        for (int base = 0; base < total; base += lcap) {
            int e = excl;
#pragma unroll
            for (int j = 0; j < QP_CPT; j++) {
                const int k_end = min(kc[j], base + lcap - e);
                for (int k = max(0, base - e); k < k_end; k++) clist[e + k - base] = ((uint32_t)(cfirst + j) << 10) | (uint32_t)k;
                e += kc[j];
            }
            __syncthreads();
            const int m = min(lcap, total - base);
            for (int i0 = 0; i0 < m; i0 += 4 * QP_THREADS) {
                uint32_t ord[4], kv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) ord[j] = clist[min(i0 + QP_THREADS * j + tid, m - 1)];
#pragma unroll
                for (int j = 0; j < 4; j++) kv[j] = cslots[(size_t)(ord[j] >> 10) * g.cell_cap + (ord[j] & 1023u)];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (i0 + QP_THREADS * j + tid >= m) continue;
                    const int x = kv[j] & 0xfff, y = (kv[j] >> 12) & 0xfff;
                    const int r = (int)__fdiv_rn((float)x, g.hX);
                    int x0 = (int)(g.hX * (float)r), x1 = (int)(g.hX * (float)(r + 1)), y0 = 0, y1 = H;
                    uint32_t code = 0;
                    for (int d = 0; d < D; d++) {
                        const int xm = x0 + ((x1 - x0 + 1) >> 1), ym = y0 + ((y1 - y0 + 1) >> 1); // ceil(float(w)/2)
                        const int qx = x >= xm, qy = y >= ym;
                        code = (code << 2) | (uint32_t)(qx + 2 * qy);
                        x0 = qx ? xm : x0; x1 = qx ? x1 : xm;
                        y0 = qy ? ym : y0; y1 = qy ? y1 : ym;
                    }
                    const uint32_t leaf = ((uint32_t)r << (2 * D)) + code;
                    const uint32_t ci = (uint32_t)off(D) + leaf;
                    atomicAdd(&cnt32[ci >> 1], (ci & 1) ? 0x10000u : 1u);
                    // ord: vToDistributeKeys order = (cell, slot); the keypoint itself is read back from its slot at the end
                    atomicMax(&best[leaf], ((kv[j] >> 24) << 22) | (0x3fffffu - ord[j]));
                }
            }
            __syncthreads();   // the next window, the next 256 cells or the tree logic reuse the list
        }
    }
    n = wave_sum(n);
    if (lane == 0) atomicAdd(&s_ncand, n);
    __syncthreads();
    if (wid != 0) return;
    WT_MARK(2 + (level == 0), 0);   // candidates -> leaf counts (all four waves), barrier
    WT_COUNT(2 + (level == 0));
    n = s_ncand;
    if (lane == 0) { lvl_ncand[fl_idx] = n; fallback[fl_idx] = 0; }
    if (n == 0) {
        if (lane == 0) lvl_cnt[fl_idx] = 0;
        return;
    }
    // ---- count pyramid, bottom-up
    for (int d = D - 1; d >= 0; d--) {
        const int nc = nIni << (2 * d);
        for (int i = lane; i < nc; i += 64) {
            const uint16_t* ch = cnt + off(d + 1) + 4 * i;
            cnt[off(d) + i] = (uint16_t)(ch[0] + ch[1] + ch[2] + ch[3]);
        }
        QT_SYNC();
    }

    WT_MARK(2 + (level == 0), 1);   // count pyramid
    // ---- initial list: non-empty roots in order (:546-585); seq counts every root
    int L = 0, seq = nIni;
    const int N = g.quota;
    for (int r = 0; r < nIni; r++) {
        const int c = cnt[r];
        if (c == 0) continue;
        if (lane == 0) { cellA[L] = r; infoA[L] = (uint32_t)c | (c == 1 ? (1u << 24) : 0u); seqA[L] = r; }
        L++;
    }
    QT_SYNC();
    uint32_t *cA = cellA, *iA = infoA, *sA = seqA, *cB = cellB, *iB = infoB, *sB = seqB;
    int nvec = 0;
    bool finish = false, deep = false;

    // children counts of node (cell, depth d): four u16 at cnt[off(d+1) + 4*cell .. +3]
    auto child_counts = [&](uint32_t cell, int d, int c[4]) {
        const uint16_t* ch = cnt + off(d + 1) + 4 * cell;
        c[0] = ch[0]; c[1] = ch[1]; c[2] = ch[2]; c[3] = ch[3];
    };

    while (!finish && !deep) {
        const int prevSize = L;
        // ---------------- one pass of the first loop: split every expandable node
        int total_ch = 0, nkeep = 0, nToExpand = 0;
        for (int p0 = 0; p0 < L; p0 += 64) {
            const int p = p0 + lane;
            int nch = 0, nexp = 0, keep = 0;
            if (p < L) {
                const uint32_t info = iA[p];
                if (info >> 24) keep = 1;
                else {
                    const int d = (info >> 20) & 15;
                    if (d >= D) deep = true;
                    else {
                        int c[4];
                        child_counts(cA[p], d, c);
                        nch = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
                        nexp = (c[0] > 1) + (c[1] > 1) + (c[2] > 1) + (c[3] > 1);
                    }
                }
            }
            int tch, tk, te;
            const int pch = wave_excl_scan(nch, lane, tch);
            const int pk = wave_excl_scan(keep, lane, tk);
            const int pe = wave_excl_scan(nexp, lane, te);
            if (p < L) { tpos[p] = (uint16_t)(keep ? nkeep + pk : total_ch + pch); smark[p] = (uint16_t)(nToExpand + pe); }
            total_ch += tch; nkeep += tk; nToExpand += te;
        }
        deep = __any(deep);
        if (deep) break;
        if (total_ch + nkeep > nodecap) { deep = true; break; } // cannot happen (size <= N + 2), kept as a guard
        QT_SYNC();
        for (int p0 = 0; p0 < L; p0 += 64) {
            const int p = p0 + lane;
            if (p < L) {
                const uint32_t info = iA[p];
                if (info >> 24) {
                    const int q = total_ch + tpos[p];
                    cB[q] = cA[p]; iB[q] = info; sB[q] = sA[p];
                } else {
                    const int d = (info >> 20) & 15;
                    int c[4];
                    child_counts(cA[p], d, c);
                    int ci = tpos[p], vi = smark[p];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (c[k] == 0) continue;
                        const int q = total_ch - 1 - ci;
                        cB[q] = cA[p] * 4 + k;
                        iB[q] = (uint32_t)c[k] | ((uint32_t)(d + 1) << 20) | (c[k] == 1 ? (1u << 24) : 0u);
                        sB[q] = (uint32_t)(seq + ci);
                        if (c[k] > 1) {
                            if (vi < veccap)
                                vec[vi] = ((unsigned long long)c[k] << 40) | ((unsigned long long)(seq + ci) << 16) |
                                          (unsigned long long)q;
                            vi++;
                        }
                        ci++;
                    }
                }
            }
        }
        seq += total_ch;
        nvec = nToExpand;
        L = total_ch + nkeep;
        { uint32_t* t; t = cA; cA = cB; cB = t; t = iA; iA = iB; iB = t; t = sA; sA = sB; sB = t; }
        QT_SYNC();
        WT_MARK(2 + (level == 0), 2);   // passes of the first loop
        if (L >= N || L == prevSize) {
            finish = true;
        } else if (L + nToExpand * 3 > N) {
            // ---------------- priority rounds (:673-737)
            while (!finish && !deep) {
                const int prevSize2 = L;
                const int np = min(nvec, veccap);
                int P = 1;
                while (P < np) P <<= 1;
                for (int i = lane; i < P; i += 64) vprev[i] = (i < np) ? vec[i] : 0ull;
                QT_SYNC();
                for (int k = 2; k <= P; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int t = lane; t < (P >> 1); t += 64) {
                            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                            const int l = i | j;
                            const unsigned long long a = vprev[i], b = vprev[l];
                            const bool up = ((i & k) == 0);
                            if ((a > b) == up) { vprev[i] = b; vprev[l] = a; }
                        }
                        QT_SYNC();
                    }
                for (int p = lane; p < L; p += 64) smark[p] = 0;
                QT_SYNC();
                // scan in processing order (largest first); stop at the first prefix that reaches N
                int cum = L, total_ch2 = 0, nsplit = np, nexp2 = 0;
                bool cut = false;
                for (int j0 = 0; j0 < np && !cut; j0 += 64) {
                    const int j = j0 + lane;
                    int nch = 0, nexp = 0, pos = 0;
                    if (j < np) {
                        pos = (int)(vprev[P - 1 - j] & 0xffff);
                        const uint32_t info = iA[pos];
                        const int d = (info >> 20) & 15;
                        if (d >= D) deep = true;
                        else {
                            int c[4];
                            child_counts(cA[pos], d, c);
                            nch = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
                            nexp = (c[0] > 1) + (c[1] > 1) + (c[2] > 1) + (c[3] > 1);
                        }
                    }
                    int tg, tch, te;
                    const int gain = (j < np) ? nch - 1 : 0;
                    const int pg = wave_excl_scan(gain, lane, tg);
                    const unsigned long long hit = __ballot(j < np && cum + pg + gain >= N);
                    int upto = 64; // lanes of this chunk that are split
                    if (hit) { upto = __ffsll((long long)hit); cut = true; nsplit = j0 + upto; }
                    const bool sel = (j < np) && lane < upto;
                    const int pch = wave_excl_scan(sel ? nch : 0, lane, tch);
                    const int pe = wave_excl_scan(sel ? nexp : 0, lane, te);
                    if (sel) { smark[pos] = (uint16_t)(j + 1); tpos[pos] = (uint16_t)(total_ch2 + pch); }
                    if (sel) vec[j] = ((unsigned long long)(nexp2 + pe) << 32) | (unsigned)pos; // scratch: (vec slot, node position)
                    total_ch2 += tch; nexp2 += te;
                    cum += (hit ? 0 : tg);
                }
                deep = __any(deep);
                if (deep) break;
                if (total_ch2 + (L - nsplit) > nodecap) { deep = true; break; }
                QT_SYNC();
                // children of the split nodes, in reverse creation order at the front of the new list
                // (vec[0..nsplit) currently holds (vec slot, node position) scratch; new vec entries go to vprev's tail-free
                //  area: build them in cB-side scratch first)
                for (int j0 = 0; j0 < nsplit; j0 += 64) {
                    const int j = j0 + lane;
                    if (j < nsplit) {
                        const unsigned long long sc = vec[j];
                        const int pos = (int)(sc & 0xffffffffu);
                        int vi = (int)(sc >> 32);
                        const uint32_t info = iA[pos];
                        const int d = (info >> 20) & 15;
                        int c[4];
                        child_counts(cA[pos], d, c);
                        int ci = tpos[pos];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (c[k] == 0) continue;
                            const int q = total_ch2 - 1 - ci;
                            cB[q] = cA[pos] * 4 + k;
                            iB[q] = (uint32_t)c[k] | ((uint32_t)(d + 1) << 20) | (c[k] == 1 ? (1u << 24) : 0u);
                            sB[q] = (uint32_t)(seq + ci);
                            if (c[k] > 1) {
                                if (vi < veccap)
                                    vprev[vi] = ((unsigned long long)c[k] << 40) | ((unsigned long long)(seq + ci) << 16) |
                                                (unsigned long long)q;
                                vi++;
                            }
                            ci++;
                        }
                    }
                }
                // (vprev[0..nexp2) now holds the next round's vector; the sorted copy it overwrote is no longer needed
                //  because every (vec slot, position) pair was taken from vec[] before this loop)
                // survivors keep their relative order behind the new children
                int kept = 0;
                for (int p0 = 0; p0 < L; p0 += 64) {
                    const int p = p0 + lane;
                    const int keep = (p < L) && smark[p] == 0;
                    int tk;
                    const int pk = wave_excl_scan(keep, lane, tk);
                    if (keep) {
                        const int q = total_ch2 + kept + pk;
                        cB[q] = cA[p]; iB[q] = iA[p]; sB[q] = sA[p];
                    }
                    kept += tk;
                }
                QT_SYNC();
                for (int i = lane; i < min(nexp2, veccap); i += 64) vec[i] = vprev[i];
                seq += total_ch2;
                nvec = nexp2;
                L = total_ch2 + kept;
                { uint32_t* t; t = cA; cA = cB; cB = t; t = iA; iA = iB; iB = t; t = sA; sA = sB; sB = t; }
                QT_SYNC();
                if (L >= N || L == prevSize2) finish = true;
            }
        }
    }
    WT_MARK(2 + (level == 0), 3);   // priority rounds (sort + split)
    if (deep) {
        if (lane == 0) { fallback[fl_idx] = 1; worklist[atomicAdd(worklist_n, 1)] = fl_idx; } // k_distribute redoes this level
        return;
    }

    // ---- best keypoint of every node, in list order (:741-760)
    uint32_t* outp = lvl_out + (size_t)f * out_fstride + g.out_off;
    for (int p0 = 0; p0 < L; p0 += 64) {
        const int p = p0 + lane;
        if (p < L) {
            const int d = (iA[p] >> 20) & 15;
            const int sh = 2 * (D - d);
            const uint32_t l0 = cA[p] << sh, l1 = (cA[p] + 1) << sh;
            uint32_t b = 0;
            for (uint32_t l = l0; l < l1; l++) b = max(b, best[l]);
            const uint32_t ord = 0x3fffffu - (b & 0x3fffffu);
            outp[p] = cslots[(size_t)(ord >> 10) * g.cell_cap + (ord & 1023u)];
        }
    }
    if (lane == 0) lvl_cnt[fl_idx] = L;
    WT_MARK(2 + (level == 0), 4);   // best keypoint per node
}

// per-frame level offsets (ascending-level concatenation, :1076-1104), totals, and a flat (keypoint, level) list so that
// the per-keypoint kernel needs one record load instead of a chain of dependent lookups
__global__ __launch_bounds__(256) void k_level_offsets(const int32_t* __restrict__ lvl_cnt, int32_t* __restrict__ lvl_off,
                                                       int32_t* __restrict__ n_out, int nlevels, int nframes,
                                                       int capacity, int32_t* __restrict__ overflow,
                                                       const LevelGeom* __restrict__ geom,
                                                       const uint32_t* __restrict__ lvl_out, int out_fstride,
                                                       uint32_t* __restrict__ flat_kv, uint8_t* __restrict__ flat_lvl,
                                                       int32_t* __restrict__ worklist_n)
{
    __builtin_amdgcn_s_setprio(2); // latency-bound: its few waves go first when a VALU-bound kernel shares the CU
    const int f = blockIdx.x, tid = threadIdx.x;
    if (f >= nframes) return;
    if (f == 0 && tid == 0) *worklist_n = 0; // the quadtree's work list (k_distribute has drained it) is empty for the next batch
    int acc = 0;
    for (int l = 0; l < nlevels; l++) {
        const int c = lvl_cnt[f * nlevels + l];
        if (tid == 0) lvl_off[f * nlevels + l] = acc;
        const uint32_t* src = lvl_out + (size_t)f * out_fstride + geom[l].out_off;
        for (int i = tid; i < c; i += 256)
            if (acc + i < capacity) {
                flat_kv[(size_t)f * capacity + acc + i] = src[i];
                flat_lvl[(size_t)f * capacity + acc + i] = (uint8_t)l;
            }
        acc += c;
    }
    if (tid == 0) {
        if (acc > capacity) {
            atomicMax(overflow, acc);
            acc = capacity;
        }
        n_out[f] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ blur ----
// GaussianBlur 7x7 sigma 2, taps 18 34 49 55 49 34 18 (x256, sum 257), BORDER_REFLECT_101,
// out = sat_u8((sum + 32768) >> 16).  64x64 output tile per workgroup (four 64x16 strips), separable through LDS.
__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}

// Instruction count, not bandwidth, limits this kernel, so both passes run on the packed dot-product instructions:
//   horizontal: a thread loads the 12 bytes around 4 adjacent outputs of one row as three dwords; every 7-tap sum is
//               two v_dot4_u32_u8 on byte windows cut out with v_alignbyte (sums <= 257 * 255 fit 16 bits);
//   transpose:  the sums go to LDS column-major (one column = 70 consecutive u16), so that
//   vertical:   a thread takes one column and 4 consecutive rows, reads ten sums as five dwords and forms each output
//               from four v_dot2_u32_u16 on (row, row+1) pairs; odd rows use pairs re-cut with v_alignbit.
#define BL_CP 74 // u16 per LDS column: 70 rows + pad, an odd number of dwords (conflict-free across columns)

__device__ __forceinline__ uint32_t bl_dot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
__device__ __forceinline__ uint32_t bl_dot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

// right-border windows of bl_load_rows: v_perm selectors of the three window dwords and which dword pair each takes ({l1, l0} = 0,
// {l2, l1} = 1; bits 0..2), indexed by r = w - x
__device__ const uint4 bl_right_tab[8] = {
    {0x00000000u, 0x00000000u, 0x00000000u, 0u},
    {0x06050403u, 0x04050607u, 0x04050607u, 3u}, // r = 1
    {0x05040302u, 0x05060706u, 0x01020304u, 7u}, // r = 2
    {0x04030201u, 0x06070605u, 0x02030405u, 7u}, // r = 3
    {0x07060504u, 0x07060504u, 0x03040506u, 6u}, // r = 4
    {0x06050403u, 0x06050403u, 0x04050607u, 6u}, // r = 5
    {0x05040302u, 0x05040302u, 0x05060706u, 6u}, // r = 6
    {0x04030201u, 0x04030201u, 0x06070605u, 6u}, // r = 7
};

// One workgroup = one 64-column strip of a level, walked top to bottom in 64-row tiles.  The loads of the next tile's
// rows are in flight while the vertical pass of the current one runs, and the six rows of horizontal sums two tiles
// share are carried over in LDS instead of being recomputed, so every source row is fetched once per strip.
__device__ __forceinline__ void bl_load_rows(const uint8_t* img, int pitch, const LevelGeom& g, int tx0, int row0,
                                             int first_rr, int nrr, int tid, int nitems, uint32_t (*w)[3])
{
    typedef uint32_t u32_unaligned __attribute__((aligned(1))); // level 0 is the caller's buffer
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (k >= nitems) break;
        const int it = tid + 256 * k;
        const int rr = first_rr + (it >> 4), x = tx0 + 4 * (it & 15);
        w[k][0] = w[k][1] = w[k][2] = 0;
        if (rr < nrr && x < g.bpitch) {
            const uint8_t* row = img + off24(reflect101(row0 + rr, g.h), pitch);
            if (x >= 4 && x + 8 <= g.w) {
                const u32_unaligned* p = reinterpret_cast<const u32_unaligned*>(row + x - 4);
                w[k][0] = p[0]; w[k][1] = p[1]; w[k][2] = p[2];
            } else if (x == 0) {
                // left border, BORDER_REFLECT_101: pixels -4 .. -1 are pixels 4 .. 1 -- one v_perm on the two dwords the window has anyway
                // (a level is at least 62 pixels wide, orb_extractor.hip build_geometry)
                const u32_unaligned* p = reinterpret_cast<const u32_unaligned*>(row);
                const uint32_t d0 = p[0], d1 = p[1];
                w[k][0] = __builtin_amdgcn_perm(d1, d0, 0x01020304u); w[k][1] = d0; w[k][2] = d1;
            } else if (x < g.w) {
                // right border: r = w - x (1 .. 7) pixels of the window's centre dword are inside.  Every window byte, reflected or not,
                // is one of the row's last 12 pixels: three dwords, and per window dword one v_perm on two of them with a selector
                // that depends only on r (bl_right_tab: byte j <- pixel index 8 - r + j inside, 14 + r - j reflected)
                const u32_unaligned* p = reinterpret_cast<const u32_unaligned*>(row + g.w - 12);
                const uint32_t l0 = p[0], l1 = p[1], l2 = p[2];
                const uint4 s = bl_right_tab[g.w - x];
                w[k][0] = (s.w & 1u) ? __builtin_amdgcn_perm(l2, l1, s.x) : __builtin_amdgcn_perm(l1, l0, s.x);
                w[k][1] = (s.w & 2u) ? __builtin_amdgcn_perm(l2, l1, s.y) : __builtin_amdgcn_perm(l1, l0, s.y);
                w[k][2] = (s.w & 4u) ? __builtin_amdgcn_perm(l2, l1, s.z) : __builtin_amdgcn_perm(l1, l0, s.z);
            } // else: columns of the blurred row's padding (x >= w): nobody reads them
        }
    }
}

// The 8-bit taps of GaussianBlur(7x7, sigma 2) depend on the OpenCV release (orbfe_extractor_set_gaussian_taps):
//   ED = false: every tap rounded on its own, 18 34 49 55 49 34 18 (sum 257) -- 2.4 / 3.2 (the version CMakeLists.txt:32-38 asks
//               for) and the first fixed-point implementation of 3.4
//   ED = true:  the bit-exact kernel with the rounding error carried tap to tap and the centre taking the rest, 18 34 48 56 48 34 18
//               (sum 256) -- late 3.4.x and 4.x
template <bool ED>
__device__ __forceinline__ void bl_hsum_rows(uint16_t* sh, int first_rr, int nrr, int tid, int nitems, const uint32_t (*w)[3])
{
    constexpr uint32_t T2 = ED ? 48u : 49u, T3 = ED ? 56u : 55u;
    constexpr uint32_t TA = 18u | (34u << 8) | (T2 << 16) | (T3 << 24); // taps 0..3
    constexpr uint32_t TB = T2 | (34u << 8) | (18u << 16);              // taps 4..6
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (k >= nitems) break;
        const int it = tid + 256 * k;
        const int rr = first_rr + (it >> 4), c = 4 * (it & 15);
        if (rr < nrr) {
            // byte j of the 12-byte window is pixel x - 4 + j; output i uses bytes i+1 .. i+7
            const uint32_t w0 = w[k][0], w1 = w[k][1], w2 = w[k][2];
            const uint32_t s0 = bl_dot4(__builtin_amdgcn_alignbyte(w1, w0, 1), TA, bl_dot4(__builtin_amdgcn_alignbyte(w2, w1, 1), TB, 0));
            const uint32_t s1 = bl_dot4(__builtin_amdgcn_alignbyte(w1, w0, 2), TA, bl_dot4(__builtin_amdgcn_alignbyte(w2, w1, 2), TB, 0));
            const uint32_t s2 = bl_dot4(__builtin_amdgcn_alignbyte(w1, w0, 3), TA, bl_dot4(__builtin_amdgcn_alignbyte(w2, w1, 3), TB, 0));
            const uint32_t s3 = bl_dot4(w1, TA, bl_dot4(w2, TB, 0));
            sh[(c + 0) * BL_CP + rr] = (uint16_t)s0;
            sh[(c + 1) * BL_CP + rr] = (uint16_t)s1;
            sh[(c + 2) * BL_CP + rr] = (uint16_t)s2;
            sh[(c + 3) * BL_CP + rr] = (uint16_t)s3;
        }
    }
}

#ifndef BL_ATTR
#define BL_ATTR
#endif
template <bool ED>
__global__ __launch_bounds__(256) BL_ATTR void k_blur7(ImgView src0, ImgView pyr, ImgView blur,
                                               const LevelGeom* __restrict__ geom, const uint32_t* __restrict__ strips,
                                               int nx, int total)
{
    __shared__ __align__(16) uint16_t sh[64 * BL_CP];
    int bx, f;
    if (!xcd_remap(nx, total, bx, f)) return;
    const uint32_t t = strips[bx];
    const int level = t & 15, tx0 = (int)(t >> 4) * 64;
    const LevelGeom g = geom[level];
    const uint8_t* img = (level == 0) ? src0.base + (size_t)f * src0.fstride
                                      : pyr.base + (size_t)f * pyr.fstride + g.img_off;
    const int pitch = (level == 0) ? src0.pitch : g.pitch;
    const int tid = threadIdx.x;
    uint8_t* D = blur.base_w + (size_t)f * blur.fstride + g.blur_off;
    constexpr uint32_t T2 = ED ? 48u : 49u, T3 = ED ? 56u : 55u;
    constexpr uint32_t V01 = 18u | (34u << 16), V23 = T2 | (T3 << 16), V45 = T2 | (34u << 16), V6 = 18u, V6H = 18u << 16;
    const int c4 = 4 * (tid & 15), q = tid >> 4;
    const int x = tx0 + c4;

    // LDS row rr of a tile that starts at image row tyb holds the horizontal sums of image row tyb - 3 + rr
    uint32_t w[5][3];
    {
        const int nrr = min(64, g.h) + 6;
        bl_load_rows(img, pitch, g, tx0, -3, 0, nrr, tid, 5, w);
        bl_hsum_rows<ED>(sh, 0, nrr, tid, 5, w);
    }
    for (int tyb = 0; tyb < g.h; tyb += 64) {
        const int nrows_out = min(64, g.h - tyb);
        const bool more = tyb + 64 < g.h;
        const int nrr_next = more ? min(64, g.h - tyb - 64) + 6 : 0;
        __syncthreads(); // sums of this tile complete
        if (more) bl_load_rows(img, pitch, g, tx0, tyb + 64 - 3, 6, nrr_next, tid, 4, w); // rows 6.. of the next tile
        // ---- vertical pass: one thread = 4 adjacent columns x 4 consecutive rows, stored as four aligned dwords
        if (4 * q < nrows_out && x < g.bpitch) {
            // sat_u8(sum >> 16) of four columns into one dword per row: the high halves of two columns' sums side by side (v_perm), both
            // saturated and packed by ONE v_sat_pk_u8_i16 (sum >> 16 <= 257), the two column pairs joined by a v_perm: 20 instructions for the
            // 16 pixels where shift / min / shift-or per pixel took 48
            uint32_t out[4], qp[2][4];
#pragma unroll
            for (int jp = 0; jp < 2; jp++) {
                uint32_t o[2][4];
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    const uint32_t* hp = reinterpret_cast<const uint32_t*>(&sh[(c4 + 2 * jp + jj) * BL_CP + 4 * q]);
                    const uint32_t d0 = hp[0], d1 = hp[1], d2 = hp[2], d3 = hp[3], d4 = hp[4];
                    const uint32_t a01 = __builtin_amdgcn_alignbit(d1, d0, 16), a12 = __builtin_amdgcn_alignbit(d2, d1, 16);
                    const uint32_t a23 = __builtin_amdgcn_alignbit(d3, d2, 16), a34 = __builtin_amdgcn_alignbit(d4, d3, 16);
                    o[jj][0] = bl_dot2(d0, V01, bl_dot2(d1, V23, bl_dot2(d2, V45, bl_dot2(d3, V6, 32768u))));
                    o[jj][1] = bl_dot2(a01, V01, bl_dot2(a12, V23, bl_dot2(a23, V45, bl_dot2(a34, V6, 32768u))));
                    o[jj][2] = bl_dot2(d1, V01, bl_dot2(d2, V23, bl_dot2(d3, V45, bl_dot2(d4, V6, 32768u))));
                    o[jj][3] = bl_dot2(a12, V01, bl_dot2(a23, V23, bl_dot2(a34, V45, bl_dot2(d4, V6H, 32768u))));
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t pr = __builtin_amdgcn_perm(o[1][r], o[0][r], 0x07060302u);   // (sum of column 2 jp) >> 16 | (column 2 jp + 1) >> 16 << 16
                    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(qp[jp][r]) : "v"(pr));
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) out[r] = __builtin_amdgcn_perm(qp[1][r], qp[0][r], 0x05040100u);
            const int y0 = tyb + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (y0 + j < g.h) *reinterpret_cast<uint32_t*>(D + (off24(y0 + j, g.bpitch) + (uint32_t)x)) = out[j];
        }
        if (!more) break;
        // ---- carry rows 64..69 (= rows 0..5 of the next tile) over: 3 dwords per column
        uint32_t carry = 0;
        const int cc = tid / 3, cd = tid - cc * 3;
        if (tid < 192) carry = reinterpret_cast<const uint32_t*>(&sh[cc * BL_CP + 64])[cd];
        __syncthreads(); // every read of this tile's sums done
        if (tid < 192) reinterpret_cast<uint32_t*>(&sh[cc * BL_CP])[cd] = carry;
        bl_hsum_rows<ED>(sh, 6, nrr_next, tid, 4, w);
    }
}

template __global__ void k_blur7<false>(ImgView, ImgView, ImgView, const LevelGeom*, const uint32_t*, int, int);
template __global__ void k_blur7<true>(ImgView, ImgView, ImgView, const LevelGeom*, const uint32_t*, int, int);

// The same blur on the matrix cores (round 6).  The pipeline is bound by vector-ALU issue (the twelve stages issue 985 us of VALU work in
// a 1.3 ms step) while the matrix pipe idles; a 7-tap filter is a banded (Toeplitz) matrix, and v_mfma_i32_32x32x32_i8 does 32 K integer
// multiply-adds exactly in the time of eight VALU instructions.  One WAVE walks a 32-column strip of a level from top to bottom in blocks
// of 32 rows:
//   pass 1  H = P x T1: A = the block's pixels minus 128 (signed bytes; lane = row, 16 consecutive bytes of it: one unaligned 16-byte
//           load), B = the strip's tap matrix.  32 outputs need 38 inputs: two K blocks, i.e. three 16-byte pieces of the row (the
//           third one only feeds the second block's first seven columns).  BORDER_REFLECT_101 in x is folded into the tap matrix by the
//           host (a reflected tap adds its weight to the column it lands on), in y it is the lane's row index.  With 128 as the
//           accumulator's start value H'' = H - 128 T + 128 lies in [-32768, 32767] (T = the taps' sum, 257 or 256).
//   planes  the accumulator layout of pass 1 -- lane = output column, registers = 16 rows -- IS the A layout of pass 2 (lane = row of A,
//           16 values along K), with K = image row in the registers' order; the tap matrix of pass 2 is built in that order.  An
//           accumulator is split into its high byte and its low byte minus 128 (both signed bytes): 7 instructions per 4 values.
//   pass 2  out^T = H^T x T2 for the two byte planes, over this block and the previous one (an output row reaches 3 rows into the next
//           block): four MFMA; S = (hi << 8) + lo with every constant (the 128s, the rounding 32768) in lo's start value.
//   store   lane = output row, four registers = four consecutive pixels: v_perm + v_sat_pk_u8_i16 as in k_blur7.
// ~100 VALU instructions per 32 x 32 pixels (6.4 lane operations a pixel; k_blur7: 16) and six MFMA; no LDS, no barrier.
typedef int bm_v4i __attribute__((ext_vector_type(4)));
typedef int bm_v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_blur7_mfma(ImgView src0, ImgView pyr, ImgView blur, const LevelGeom* __restrict__ geom,
                                                    const BlurStrip* __restrict__ strips, const uint4* __restrict__ tabs,
                                                    const uint4* __restrict__ tab2, int K2, int nstrips, int nx, int total)
{
    constexpr int BM_PP = 9;   // dwords per row of a wave's 32 x 32-byte output patch (odd: the rows fall on different banks)
    __shared__ uint32_t s_patch[4][32 * BM_PP];
    const int lane = threadIdx.x & 63, wid = wave_id();
    int bx, f;
    if (!xcd_remap(nx, total, bx, f)) return;
    const int si = bx * 4 + wid;
    if (si >= nstrips) return;
    uint32_t* patch = s_patch[wid];
    const BlurStrip S = strips[si];
    const LevelGeom& g = geom[S.level];
    const int r = lane & 31, half = lane >> 5;
    const uint8_t* img = (S.level == 0) ? src0.base + (size_t)f * src0.fstride : pyr.base + (size_t)f * pyr.fstride + g.img_off;
    const int pitch = (S.level == 0) ? src0.pitch : g.pitch;
    uint8_t* D = blur.base_w + (size_t)f * blur.fstride + g.blur_off;
    const int h = g.h, wst = (g.w + 3) & ~3, bpitch = g.bpitch;
    const bm_v4i B1a = __builtin_bit_cast(bm_v4i, tabs[(size_t)S.tab * 64 + lane]), B1b = __builtin_bit_cast(bm_v4i, tabs[(size_t)(S.tab + 1) * 64 + lane]);
    const bm_v4i B2a = __builtin_bit_cast(bm_v4i, tab2[lane]), B2b = __builtin_bit_cast(bm_v4i, tab2[64 + lane]);
    const uint32_t ca = (uint32_t)(half ? S.c1 : S.c0), cb = (uint32_t)S.c2;   // K block a: pieces 0 | 1 by half; K block b: piece 2 (its second half has no weight)
    typedef uint32_t bm_u32x4 __attribute__((ext_vector_type(4)));
    typedef bm_u32x4 bm_u32x4_unaligned __attribute__((aligned(1)));   // (level 0 is the caller's buffer: no alignment is assumed)
    const int nblk = ((h + 31) >> 5) + 1;
    auto load_rows = [&](int j, bm_u32x4& pa, bm_u32x4& pb) {
        const uint32_t ro = off24(reflect101(32 * j - 4 + r, h), pitch);
        pa = *reinterpret_cast<const bm_u32x4_unaligned*>(img + (ro + ca));
        pb = *reinterpret_cast<const bm_u32x4_unaligned*>(img + (ro + cb));
    };
    bm_u32x4 na, nb;
    load_rows(0, na, nb);
    bm_v4i ph = {0, 0, 0, 0}, pl = {0, 0, 0, 0};
    // the accumulators' start values as operands of their own (srcC != vdst): kept in registers for the whole strip instead of sixteen
    // moves per accumulator and block
    const bm_v16i c128 = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
    const bm_v16i ck2 = {K2, K2, K2, K2, K2, K2, K2, K2, K2, K2, K2, K2, K2, K2, K2, K2};
    const bm_v16i czero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < nblk; j++) {
        const bm_u32x4 qa = na, qb = nb;
        if (j + 1 < nblk) load_rows(j + 1, na, nb);   // the next block's rows are in flight while this one is worked on
        constexpr uint32_t SGN = 0x80808080u;
        const bm_v4i A1a = {(int)(qa.x ^ SGN), (int)(qa.y ^ SGN), (int)(qa.z ^ SGN), (int)(qa.w ^ SGN)};
        const bm_v4i A1b = {(int)(qb.x ^ SGN), (int)(qb.y ^ SGN), (int)(qb.z ^ SGN), (int)(qb.w ^ SGN)};
        bm_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1a, B1a, c128, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1b, B1b, acc, 0, 0, 0);
        // planes: byte 1 (the signed high byte) and byte 0 minus 128 of four accumulators into one register each
        bm_v4i nh, nl;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t a0 = (uint32_t)acc[4 * q], a1 = (uint32_t)acc[4 * q + 1], a2 = (uint32_t)acc[4 * q + 2], a3 = (uint32_t)acc[4 * q + 3];
            const uint32_t h01 = __builtin_amdgcn_perm(a1, a0, 0x0c0c0501u), h23 = __builtin_amdgcn_perm(a3, a2, 0x05010c0cu);
            const uint32_t l01 = __builtin_amdgcn_perm(a1, a0, 0x0c0c0400u), l23 = __builtin_amdgcn_perm(a3, a2, 0x04000c0cu);
            nh[q] = (int)(h01 | h23);
            nl[q] = (int)((l01 | l23) ^ SGN);
        }
        if (j >= 1) {
            bm_v16i ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(ph, B2a, czero, 0, 0, 0);
            bm_v16i al = __builtin_amdgcn_mfma_i32_32x32x32_i8(pl, B2a, ck2, 0, 0, 0);
            ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(nh, B2b, ah, 0, 0, 0);
            al = __builtin_amdgcn_mfma_i32_32x32x32_i8(nl, B2b, al, 0, 0, 0);
            // the tile through the wave's LDS patch: written in the accumulators' layout (lane = row, four pixels a register group), read
            // back with lanes along x -- a store instruction then covers 8 rows x 32 bytes instead of 32 rows x 8 bytes (stored straight
            // from the accumulators the kernel was bound by its partial-line stores: 400 us next to FAST, 227 without them)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t s4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) s4[k] = ((uint32_t)ah[4 * q + k] << 8) + (uint32_t)al[4 * q + k];
                uint32_t p01, p23;
                const uint32_t pr01 = __builtin_amdgcn_perm(s4[1], s4[0], 0x07060302u), pr23 = __builtin_amdgcn_perm(s4[3], s4[2], 0x07060302u);
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(p01) : "v"(pr01));
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(p23) : "v"(pr23));
                patch[r * BM_PP + half + 2 * q] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
            }
            __builtin_amdgcn_wave_barrier();
            const int x = S.x0 + 4 * (lane & 7);
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int rr = (lane >> 3) + 8 * it, y = 32 * (j - 1) + rr;
                const uint32_t v = patch[rr * BM_PP + (lane & 7)];
                if (y < h && x < wst) *reinterpret_cast<uint32_t*>(D + (off24(y, bpitch) + (uint32_t)x)) = v;
            }
        }
        ph = nh; pl = nl;
    }
}

// ------------------------------------------------------------------------------------------------ describe --
// IC_Angle on the un-blurred level (:77-104), then the 256 steered BRIEF tests on the blurred level (:108-147), then the final record.
// Two keypoints per wave.  A third of the instructions of one keypoint do not depend on the lane: fastAtan2 and sin / cos of the one
// angle, and IC_Angle uses 31 lanes.  Here lanes 0..31 belong to keypoint A and 32..63 to keypoint B for those parts -- the patch rows
// of both in one pass, the two moment sums out of one scan (lanes 31 and 63), one fastAtan2 / sincos sequence for both angles -- and
// the loads and the 256 tests run per keypoint on all 64 lanes.  An odd last keypoint is paired with itself.
//
// What the kernel runs at is the CU's vector memory path, not instruction issue (round 5, tools/ext_alone.py with parts compiled out:
// 452 us alone at C2, 449 without the trigonometry, 431 without the 256 tests, 243 without the image loads; the C2 step 1.39 -> 1.26 ms
// without the image loads, unchanged without the arithmetic).  The texture addresser takes a wave's load FOUR LANES A CLOCK whatever
// the width: 16 clocks for 256 bytes as dwords, 16 clocks for 1 KB as 16-byte lanes.  So:
//   - the 37 x 40-byte window and the 31 x 36-byte patch are fetched as (row, 16-byte chunk) items, three chunks a row: two loads
//     each (they were 6 + 5 dword loads: 24 vector memory instructions per wave for the images, now 8);
//   - the two tables every wave needs -- the IC_Angle row weights (64 bytes a lane) and the test pattern -- come through LDS, fetched
//     once per workgroup (8 loads per wave before);
//   - the keypoint records are fetched WITH the frame's count, not behind it (one round trip less in a wave's life).
// 66 registers would cost the eighth wave per SIMD, and occupancy is what hides the round trips.
#ifndef OD2_ATTR
#define OD2_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
__global__ __launch_bounds__(256) OD2_ATTR void k_orient_describe2(ImgView src0, ImgView pyr, ImgView blur,
                                                          const LevelGeom* __restrict__ geom,
                                                          const uint32_t* __restrict__ flat_kv,
                                                          const uint8_t* __restrict__ flat_lvl,
                                                          const int32_t* __restrict__ n_out, int nlevels,
                                                          const uint32_t* __restrict__ pattern32, const uint4* __restrict__ icw,
                                                          orbfe_keypoint* __restrict__ kps,
                                                          uint8_t* __restrict__ desc, int capacity, int nx, int total)
{
    const int lane = threadIdx.x & 63, wid = wave_id();
    WT_BEGIN();
    int bx, f;
    if (!xcd_remap(nx, total, bx, f)) return;
    constexpr int PROW = 40, PATB = 31 * PROW + 8, WINB = 37 * 40 + 8;   // (rows of 8-byte multiples: the chunks go in as 8-byte stores)
    // LDS per wave: the two IC_Angle patches; once the moments are summed the same bytes hold one descriptor window at a time
    // (13 KB per workgroup with the tables: the footprint matters, an LDS request of 23 KB cost the C2 step 50 us in round 3)
    static_assert(2 * PATB >= WINB, "the window overlays the patches");
    __shared__ __align__(16) uint8_t s_pat[4][2][PATB];
    __shared__ uint4 s_icw[31 * 4];
    __shared__ uint32_t s_pattern[256];
#if OD2_LDS_PAD > 0
    // LDS the kernel asks for and never uses.  13 KB of its own let eight workgroups of it -- every wave slot -- sit on a CU; with 8 KB
    // more there are seven, and in the batched pipeline, where this kernel runs next to FAST for a whole step, that is worth 2 % of the
    // C2 step: 1.174 against 1.198 ms over eight interleaved runs each (1.163 - 1.181 against 1.185 - 1.196 in four more series);
    // + 4 KB changes nothing, + 12 KB gains half of it, + 16 KB and more lose (1.196 ... 1.31); 1280 x 720 and 1920 x 1080 do not care
    // (3.50 / 3.51, 2.81 / 2.83).  The same cap by registers (amdgpu_waves_per_eu(7, 7)) loses 4 %, the same 8 KB as dynamic LDS of
    // the launch gains nothing (12 KB: 1.2 %) -- tools/sweeps.md has the tables; alone the kernel takes 231 us either way.
    __shared__ uint32_t s_pad[OD2_LDS_PAD / 4];
    if (nlevels < 0) s_pad[threadIdx.x] = 1;                             // (never: keeps the array)
    if (nlevels < -1) kps[0].x = (float)s_pad[threadIdx.x ^ 1];
#endif
    if (threadIdx.x < 31 * 4) s_icw[threadIdx.x] = icw[threadIdx.x];
    s_pattern[threadIdx.x] = pattern32[threadIdx.x];
    __syncthreads();
    const int o0 = (bx * 4 + wid) * 2;
    // (the grid only covers slots below the capacity, whatever they hold)
    const size_t slot0 = (size_t)f * capacity + min(o0, capacity - 1), slot1 = (size_t)f * capacity + min(o0 + 1, capacity - 1);
    const uint32_t kv_pre[2] = {flat_kv[slot0], flat_kv[slot1]};
    const int lvl_pre[2] = {flat_lvl[slot0], flat_lvl[slot1]};
    const int nk = n_out[f];
    WT_MARK(0, 0);   // tables, barrier, records
    if (o0 >= nk) return;
    WT_COUNT(0);
    const bool two = o0 + 1 < nk;
    const int half = lane >> 5, r = lane & 31;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));   // level 0 is the caller's buffer: no alignment is assumed there
    int kx[2], ky[2], level[2], score[2], xoff[2], xo[2];
    u32x4 wv[2][2], v[2][2];
    uint32_t pat[4];
#pragma unroll
    for (int j = 0; j < 4; j++) pat[j] = s_pattern[j * 64 + lane];
    const int wrow = min(r, 30);
    const uint4 wa = s_icw[wrow * 4 + 0], wb = s_icw[wrow * 4 + 1], oa = s_icw[wrow * 4 + 2], ob = s_icw[wrow * 4 + 3];
    // item i = lane + 64 k (k = 0, 1) is chunk i % 3 of row i / 3 -- the same for the window, the patch and both keypoints; the items
    // past the last row (37 x 3 = 111, 31 x 3 = 93) repeat that row's chunk: the same bytes to the same place once more
    const int r0 = (lane * 171) >> 9, c0 = 16 * (lane - 3 * r0);                   // lane / 3 (exact below 171)
    const int r1f = ((lane + 64) * 171) >> 9, c1 = 16 * (lane + 64 - 3 * r1f);
    const int wr1 = min(r1f, 36), pr1 = min(r1f, 30);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int hs = (h == 1 && two) ? 1 : 0;
        const uint32_t kv = (uint32_t)__builtin_amdgcn_readfirstlane((int)kv_pre[hs]);
        level[h] = __builtin_amdgcn_readfirstlane(lvl_pre[hs]);
        const LevelGeom& g = geom[level[h]];
        kx[h] = (int)(kv & 0xfff) + 16; ky[h] = (int)((kv >> 12) & 0xfff) + 16;   // += minBorder (:843-844)
        score[h] = kv >> 24;
        const int ax = (kx[h] - 18) & ~3;
        xoff[h] = (kx[h] - 18) - ax;
        const uint8_t* img = (level[h] == 0) ? src0.base + (size_t)f * src0.fstride : pyr.base + (size_t)f * pyr.fstride + g.img_off;
        const int pitch = (level[h] == 0) ? src0.pitch : g.pitch;
        const int bpitch = g.bpitch;
        // rows ky - 18 .. ky + 18 of the blurred level from the 4-byte-aligned column at or below kx - 18: 48 bytes a row (40 used; a
        // row's last chunk may end up to 9 bytes past the level's width: the next row, or the slack behind the last one)
        const uint8_t* bimg = blur.base + (size_t)f * blur.fstride + g.blur_off + (off24(ky[h] - 18, bpitch) + (uint32_t)ax);
        wv[h][0] = *reinterpret_cast<const u32x4_unaligned*>(bimg + (uint32_t)(__mul24(r0, bpitch) + c0));
        wv[h][1] = *reinterpret_cast<const u32x4_unaligned*>(bimg + (uint32_t)(__mul24(wr1, bpitch) + c1));
        // rows ky - 15 .. ky + 15 of the level itself from the aligned column at or below kx - 15 (36 bytes used)
        const int axp = (kx[h] - 15) & ~3;
        xo[h] = (kx[h] - 15) - axp;
        const uint8_t* pimg = img + (off24(ky[h] - 15, pitch) + (uint32_t)axp);
        v[h][0] = *reinterpret_cast<const u32x4_unaligned*>(pimg + (uint32_t)(__mul24(r0, pitch) + c0));
        v[h][1] = *reinterpret_cast<const u32x4_unaligned*>(pimg + (uint32_t)(__mul24(pr1, pitch) + c1));
    }
    WT_MARK(0, 1);   // geometry, addresses, loads issued
    // a chunk goes into its 40-byte LDS row as 8-byte halves; the second half of a row's last chunk would be the next row's first bytes
    auto put_chunk = [&](uint8_t* base, int row, int c, const u32x4& q) {
        uint2* d = reinterpret_cast<uint2*>(base + row * 40 + c);
        d[0] = make_uint2(q.x, q.y);
        if (c != 32) d[1] = make_uint2(q.z, q.w);
    };
#pragma unroll
    for (int h = 0; h < 2; h++) {
        put_chunk(&s_pat[wid][h][0], r0, c0, v[h][0]);
        put_chunk(&s_pat[wid][h][0], pr1, c1, v[h][1]);
    }
    __builtin_amdgcn_wave_barrier();
    WT_MARK(0, 2);   // patches arrived and stored
    // ---- IC_Angle of both keypoints: lane = (keypoint, patch row v = r - 15): the row's 31 bytes as eight dwords re-cut at the byte
    // offset xo, m10 = sum (i - 15) I = dot(I, i) - 15 dot(I, 1) and m01 = v dot(I, 1) over the row's part of the circular patch
    // (umax, ORBextractor.cc:454-469), with the two weight vectors of the row (byte index i, ones; zero outside |u| <= umax(|v|))
    int m10 = 0, m01 = 0;
    if (r < 31) {
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(&s_pat[wid][0][0] + half * PATB + r * PROW);
        const int sh = half ? xo[1] : xo[0];
        uint32_t d[9];
#pragma unroll
        for (int q = 0; q < 9; q++) d[q] = rw[q];
        uint32_t e[8];
#pragma unroll
        for (int q = 0; q < 8; q++) e[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh);
        uint32_t A = 0, B = 0;
        A = bl_dot4(e[0], wa.x, A); A = bl_dot4(e[1], wa.y, A); A = bl_dot4(e[2], wa.z, A); A = bl_dot4(e[3], wa.w, A);
        A = bl_dot4(e[4], wb.x, A); A = bl_dot4(e[5], wb.y, A); A = bl_dot4(e[6], wb.z, A); A = bl_dot4(e[7], wb.w, A);
        B = bl_dot4(e[0], oa.x, B); B = bl_dot4(e[1], oa.y, B); B = bl_dot4(e[2], oa.z, B); B = bl_dot4(e[3], oa.w, B);
        B = bl_dot4(e[4], ob.x, B); B = bl_dot4(e[5], ob.y, B); B = bl_dot4(e[6], ob.z, B); B = bl_dot4(e[7], ob.w, B);
        m10 = (int)A - 15 * (int)B;
        m01 = (r - 15) * (int)B;
    }
    const int s10 = wave_incl_scan_add(m10), s01 = wave_incl_scan_add(m01);
    const int m10a = __builtin_amdgcn_readlane(s10, 31), m10t = __builtin_amdgcn_readlane(s10, 63);
    const int m01a = __builtin_amdgcn_readlane(s01, 31), m01t = __builtin_amdgcn_readlane(s01, 63);
    const float fm10 = (float)(half ? m10t - m10a : m10a), fm01 = (float)(half ? m01t - m01a : m01a);
    const float angle = orbfe_fast_atan2(fm01, fm10);     // lanes 0..31: keypoint A's, lanes 32..63: keypoint B's
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a, b;
    orbfe_sincosf(angle * factorPI, &b, &a); // a = cos, b = sin
    WT_MARK(0, 3);   // IC_Angle, atan2, sincos
    // ---- steered BRIEF on the staged window, one keypoint after the other on all lanes
    // The rotation is separate multiplies and adds (the reference is compiled without fused multiply-add) on packed f32: the two
    // points of a test side by side, (x0, x1) (b, b) + (y0, y1) (a, a), six v_pk_* instead of twelve scalar operations.  cvRound
    // is "add 1.5 * 2^23": the sum is rounded to an integer by the adder (to nearest even, like cvRound), and its low mantissa
    // bits are that integer plus 2^22; row * 40 + column is then one 24-bit multiply-add on the raw bits, the constants folded
    // into the window's base address.  (The pattern as a float4 table saves 32 conversions per lane but quadruples the table: slower.)
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr uint32_t MAGIC_BITS = 0x4B400000u;                                      // 12582912.0f = 1.5 * 2^23
    constexpr uint32_t IDX_BIAS = (MAGIC_BITS & 0xffffffu) * 40u + MAGIC_BITS;         // what the raw-bit multiply-add carries along
    const v2f magic = {12582912.0f, 12582912.0f};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const float ah = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), h * 32));
        const float bh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), h * 32));
        const v2f aa = {ah, ah}, bb = {bh, bh};
        uint8_t* win = &s_pat[wid][0][0];
        __builtin_amdgcn_wave_barrier();   // the patches (h = 0) / the first window (h = 1) have been read
        put_chunk(win, r0, c0, wv[h][0]);
        put_chunk(win, wr1, c1, wv[h][1]);
        __builtin_amdgcn_wave_barrier();
        const uint8_t* bc = win + 18 * 40 + 18 + xoff[h];
        unsigned long long words[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t pp = pat[j];
            const v2f X = {(float)(signed char)(pp & 0xff), (float)(signed char)((pp >> 16) & 0xff)};
            const v2f Y = {(float)(signed char)((pp >> 8) & 0xff), (float)(signed char)(pp >> 24)};
            const v2f R = (X * bb + Y * aa) + magic;   // rows of the two points (-ffp-contract=off: no fused multiply-add)
            const v2f C = (X * aa - Y * bb) + magic;   // columns
            const uint32_t i0 = __umul24(__float_as_uint(R.x), 40u) + __float_as_uint(C.x) - IDX_BIAS;
            const uint32_t i1 = __umul24(__float_as_uint(R.y), 40u) + __float_as_uint(C.y) - IDX_BIAS;
            const int t0 = bc[(int)i0], t1 = bc[(int)i1];
            words[j] = __ballot(t0 < t1);
        }
        WT_MARK(0, 4 + h);   // window stored (h = 0: arrived), 256 tests
        if (h == 0 || two) {
            if (lane < 4) {
                unsigned long long wd = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
                reinterpret_cast<unsigned long long*>(desc + ((size_t)f * capacity + o0 + h) * 32)[lane] = wd;
            }
            if (lane == h * 32) {
                // the final record (octave, size, scaled coordinates; :837-847, :1095-1101)
                const LevelGeom& g = geom[level[h]];
                orbfe_keypoint kp;
                float px = (float)kx[h], py = (float)ky[h];
                if (level[h] != 0) { px = __fmul_rn(px, g.scale); py = __fmul_rn(py, g.scale); }
                kp.x = px; kp.y = py;
                kp.size = g.kp_size;
                kp.angle = angle;
                kp.response = (float)score[h];
                kp.octave = level[h];
                kp.class_id = -1;
                kps[(size_t)f * capacity + o0 + h] = kp;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ debug ---
__global__ void k_unpack_keys(const uint32_t* __restrict__ in, int n, int add, orbfe_keypoint* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t kv = in[i];
    orbfe_keypoint kp;
    kp.x = (float)((int)(kv & 0xfff) + add);
    kp.y = (float)((int)((kv >> 12) & 0xfff) + add);
    kp.size = 7.f; kp.angle = -1.f; kp.response = (float)(kv >> 24); kp.octave = 0; kp.class_id = -1;
    out[i] = kp;
}

} // namespace orbfe
