// aruco_kernels.hip -- gfx950 kernels of the ArUco marker detector (DM_NORMAL + CORNER_LINES, the configuration of
// reference src/Frame.cc:129-142).  Pipeline per batch of B same-sized frames:
//
//   k_adaptive_threshold  box-mean adaptive threshold, writes a BIT image (1 bit/px)       (markerdetector_impl.cpp:2983)
//   k_half_area / resize  the detector's /2 pyramid used by the 35x35 warps                 (:1299-1488)
//   k_contours            one workgroup per frame: bit image -> LDS, border starts, read-only border following,
//                         length gate (> 70), approxPolyDP, 4-gon + convexity               (:3104-3556)
//   k_prefilter           per frame: CCW orientation, near-duplicate and image-border filters  (:4347-5347)
//   k_decode_warp / _otsu / _vote   per candidate of the batch's work list: level pick, homography, 35x35 warp | Otsu | cell vote, dictionary
//                                                                                            (:6448-6900, dictionary_based.cpp)
//   k_finalize            per frame: rotate corners, sort by id, de-duplicate, contour-line corner refinement
//                                                                                            (:6723-6858, :8140-8377, :8978-10044)
// HBM-bound part: threshold (reads W*H bytes, writes W*H/8).  Everything after it works on the bit image in LDS or
// on a few hundred contour points, so it is latency- rather than bandwidth-bound; see DESIGN.md.
#include "aruco_kernels.hpp"
#include "wave_dpp.hpp"

#ifndef ORBFE_PRIO_DET_TAIL
#define ORBFE_PRIO_DET_TAIL 0 // wave priority of the detector's short kernels behind the contours (k_tail_approx, k_tail_finish, k_decode_warp, k_decode_vote)
#endif
namespace orbfe {

__device__ __forceinline__ int a_lane_prefix(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ uint32_t bl_dot4_a(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }

// ---------------------------------------------------------------------------------------- threshold -----------
// cv::adaptiveThreshold(MEAN_C, BINARY_INV, win, C): mean = round(boxsum / win^2) with BORDER_REPLICATE,
// out = (src - mean <= -C).  Tile 64 x 16 per workgroup; one wave = one row of 64 px -> one 64-bit ballot store.
// TH_MAXR = largest window radius the instantiation holds: 7 (windows up to 15: frames up to 2047 pixels wide, where the
// specialised kernels below do not apply) and 15 (windows up to 31: frames up to 4095 pixels wide)
template <int TH_MAXR>
__global__ __launch_bounds__(256) void k_adaptive_threshold(ImgView src, int W, int H, int win, int C, double scale,
                                                            uint32_t* __restrict__ bits, size_t bits_fstride, int wpr)
{
    constexpr int TH_NLD = (16 + 2 * TH_MAXR + 3) / 4; // rows of a strip per wave
    __shared__ uint8_t sin[16 + 2 * TH_MAXR][64 + 2 * TH_MAXR + 2];
    __shared__ uint16_t sh[16 + 2 * TH_MAXR][64];
    const int r = win >> 1;
    const int tx0 = blockIdx.x * 64, tyb = blockIdx.y * 64, f = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint8_t* img = src.base + (size_t)f * src.fstride;
    // lane = column (clamped once per thread = BORDER_REPLICATE), wave = rows; 64 x 64 tile as four 16-row strips
    const int xa = min(max(tx0 + lane - r, 0), W - 1);
    const int xb = min(max(tx0 + 64 + (lane & (TH_MAXR > 7 ? 31 : 15)) - r, 0), W - 1); // halo column 64 + lane, lane < 2 r
    for (int ty0 = tyb; ty0 < tyb + 64 && ty0 < H; ty0 += 16) {
        const int rows = 16 + 2 * r;
        __syncthreads();
        {
            uint8_t va[TH_NLD], vb[TH_NLD]; // all loads issued before the LDS stores
#pragma unroll
            for (int k = 0; k < TH_NLD; k++) {
                const int rr = wid + 4 * k;
                const uint8_t* row = img + (size_t)min(max(ty0 + min(rr, rows - 1) - r, 0), H - 1) * src.pitch;
                va[k] = row[xa];
                vb[k] = row[xb];
            }
#pragma unroll
            for (int k = 0; k < TH_NLD; k++) {
                const int rr = wid + 4 * k;
                if (rr < rows) {
                    sin[rr][lane] = va[k];
                    if (lane < 2 * r) sin[rr][64 + lane] = vb[k];
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < rows * 64; i += 256) {
            const int rr = i >> 6, cc = i & 63;
            int s = 0;
#pragma unroll
            for (int k = 0; k < 2 * TH_MAXR + 1; k++) // fixed trip count: the LDS reads issue back to back
                if (k < win) s += sin[rr][cc + k];
            sh[rr][cc] = (uint16_t)s;
        }
        __syncthreads();
        for (int ry = wid; ry < 16; ry += 4) {
            const int y = ty0 + ry, x = tx0 + lane;
            int s = 0;
#pragma unroll
            for (int k = 0; k < 2 * TH_MAXR + 1; k++)
                if (k < win) s += sh[ry + k][lane];
            int mean = orbfe_round_d((double)s * scale);
            mean = mean > 255 ? 255 : mean;
            const int v = sin[ry + r][lane + r];
            const bool on = (x < W) && (y < H) && (v - mean <= -C);
            const unsigned long long m = __ballot(on);
            if (y < H && lane < 2) {
                const int word = (tx0 >> 5) + lane;
                if (word < wpr) bits[(size_t)f * bits_fstride + (uint32_t)(__mul24(y, wpr) + word)] = (uint32_t)(m >> (32 * lane));
            }
        }
    }
}

template __global__ void k_adaptive_threshold<7>(ImgView, int, int, int, int, double, uint32_t*, size_t, int);
template __global__ void k_adaptive_threshold<15>(ImgView, int, int, int, int, double, uint32_t*, size_t, int);

// The same threshold on the packed dot-product instructions (the generic kernel above issues 2 * 15 predicated LDS reads
// and adds per pixel and an fp64 multiply for the mean).  64 x 64 tile per workgroup, WIN = 2 R + 1 a template parameter:
//   horizontal: a thread loads the bytes around 4 adjacent outputs of one row as 3 (R <= 3) or 5 dwords; a box sum of WIN
//               bytes is ceil(WIN / 4) v_dot4_u32_u8 with all-ones weights on windows cut out with v_alignbyte; the sums
//               go to LDS column-major, the tile's own pixels row-major;
//   vertical:   a lane owns one column and 16 consecutive rows: 16 + 2 R sums as dwords, sliding window down the column;
//               mean = (s + WIN^2 / 2) / WIN^2 as a multiply-high (WIN^2 is odd, so the rounding has no ties and equals
//               rint(s * (1.0 / WIN^2)) -- verified exhaustively by the host before this kernel is chosen);
//               one 64-bit ballot per row = 64 bits of the bit image.
typedef uint32_t u32_unaligned_t __attribute__((aligned(1)));
template <int WIN>
__global__ __launch_bounds__(256) void k_adaptive_threshold_t(ImgView src, int W, int H, int C, uint32_t magic,
                                                              uint32_t* __restrict__ bits, size_t bits_fstride, int wpr, int ntx, int ntiles,
                                                              int total)
{
    constexpr int R = WIN / 2, NDW = R <= 3 ? 3 : 5, LEAD = R <= 3 ? 4 : 8; // window = bytes x - LEAD .. x - LEAD + 4 NDW - 1
    constexpr int ROWS = 64 + 2 * R, P0 = (ROWS + 1) & ~1, PITCH = ((P0 / 2) & 1) ? P0 : P0 + 2; // u16 per LDS column:
    __shared__ __align__(16) uint16_t sh[64 * PITCH];                                           // an odd number of dwords
    __shared__ __align__(16) uint32_t spx[64][16];
    // 1-D grid renumbered so that the tiles of a frame run on one XCD (xcd_remap): neighbouring tiles share the 128-byte lines
    // their 64-byte rows lie in and their halo rows; on eight different L2s every line was fetched 3.5 times
    int tile, f;
    if (!xcd_remap(ntiles, total, tile, f)) return;
    const int tyi = tile / ntx;
    const int tx0 = (tile - tyi * ntx) * 64, ty0 = tyi * 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint8_t* img = src.base + (size_t)f * src.fstride;
    constexpr int NIT = (ROWS * 16 + 255) / 256;
    uint32_t w[NIT][NDW];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int it = tid + 256 * k;
        const int rr = it >> 4, x = tx0 + 4 * (it & 15);
#pragma unroll
        for (int j = 0; j < NDW; j++) w[k][j] = 0;
        if (rr < ROWS) {
            const uint8_t* row = img + (uint32_t)__mul24(min(max(ty0 + rr - R, 0), H - 1), src.pitch); // BORDER_REPLICATE (24-bit multiply: a 32-bit offset off the uniform base instead of a 64-bit v_mad_i64_i32)
            if (x >= LEAD && x - LEAD + 4 * NDW <= W) {
#pragma unroll
                for (int j = 0; j < NDW; j++) w[k][j] = reinterpret_cast<const u32_unaligned_t*>(row + x - LEAD)[j];
            } else {
                // the window crosses the left or right image border (BORDER_REPLICATE).  Its dwords start at multiples of four, so
                // a dword is either inside (loaded), left of the image (pixel 0 four times) or at / beyond the right edge: its
                // pixels then come out of the row's last four with one v_perm (n_in = 3, 2, 1, <= 0 pixels of it are inside)
                const uint32_t first = reinterpret_cast<const u32_unaligned_t*>(row)[0];
                const uint32_t last = reinterpret_cast<const u32_unaligned_t*>(row + W - 4)[0];
#pragma unroll
                for (int j = 0; j < NDW; j++) {
                    const int q = x - LEAD + 4 * j, n_in = W - q;
                    if (q < 0) w[k][j] = __builtin_amdgcn_perm(0u, first, 0x00000000u);
                    else if (n_in >= 4) w[k][j] = reinterpret_cast<const u32_unaligned_t*>(row + q)[0];
                    else w[k][j] = __builtin_amdgcn_perm(0u, last, n_in <= 1 ? 0x03030303u : n_in == 2 ? 0x03030302u : 0x03030201u);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int it = tid + 256 * k;
        const int rr = it >> 4, c = 4 * (it & 15);
        if (rr < ROWS) {
            uint32_t sum[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // output i sums window bytes a .. a + WIN - 1, a = LEAD + i - R
                constexpr uint32_t ONES = 0x01010101u;
                const int a = LEAD + i - R;
                uint32_t acc = 0;
#pragma unroll
                for (int q = 0; q < (WIN + 3) / 4; q++) {
                    const int b = a + 4 * q, j = b >> 2, sft = b & 3;                 // window bytes b .. b+3
                    const uint32_t hi = j + 1 < NDW ? w[k][j + 1] : 0u;
                    const uint32_t v = sft ? __builtin_amdgcn_alignbyte(hi, w[k][j], sft) : w[k][j];
                    const int left = WIN - 4 * q;                                       // bytes still to add
                    acc = bl_dot4_a(v, left >= 4 ? ONES : (ONES >> (8 * (4 - left))), acc);
                }
                sum[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) sh[(c + i) * PITCH + rr] = (uint16_t)sum[i];
            if (rr >= R && rr < 64 + R) spx[rr - R][it & 15] = w[k][LEAD / 4]; // the tile's own pixels x .. x+3
        }
    }
    __syncthreads();
    // vertical: lane = column, wave = 16 rows
    {
        const int c = lane, r0 = 16 * wid;
        const uint16_t* hp = &sh[c * PITCH + r0];
        int s = 0;
#pragma unroll
        for (int k = 0; k < WIN; k++) s += hp[k];
        const int x = tx0 + c;
        const uint8_t* px = reinterpret_cast<const uint8_t*>(&spx[r0][0]) + c;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int y = ty0 + r0 + k;
            // mean = floor((s + WIN^2 / 2) / WIN^2) (<= 255: no saturation) and "v - mean <= -C" is "mean >= v + C", i.e.
            // s + WIN^2 / 2 >= WIN^2 (v + C): one 24-bit multiply-add and a compare instead of the division (a quarter-rate v_mul_hi)
            const int v = px[k * 64];
            const bool on = (x < W) && (y < H) && (s + WIN * WIN / 2 >= __mul24(WIN * WIN, v + C));   // (v_mul_u32_u24; v_mul_lo_u32 without the hint)
            const unsigned long long m = __ballot(on);
            if (y < H && lane < 2) {
                const int word = (tx0 >> 5) + lane;
                if (word < wpr) bits[(size_t)f * bits_fstride + (uint32_t)(__mul24(y, wpr) + word)] = (uint32_t)(m >> (32 * lane));
            }
            if (k < 15) s += (int)hp[k + WIN] - (int)hp[k];
        }
    }
}
template __global__ void k_adaptive_threshold_t<5>(ImgView, int, int, int, uint32_t, uint32_t*, size_t, int, int, int, int);
template __global__ void k_adaptive_threshold_t<7>(ImgView, int, int, int, uint32_t, uint32_t*, size_t, int, int, int, int);
template __global__ void k_adaptive_threshold_t<11>(ImgView, int, int, int, uint32_t, uint32_t*, size_t, int, int, int, int);
template __global__ void k_adaptive_threshold_t<15>(ImgView, int, int, int, uint32_t, uint32_t*, size_t, int, int, int, int);

// The threshold kernel of the batched path (round 6): the same horizontal pass, then
//   vertical:   TWO row strips per lane on packed 16-bit lanes.  A wave owns rows 8 w .. 8 w + 7 (low halves) and 8 w + 32 .. 8 w + 39
//               (high halves) of the tile; the box sums lie in LDS as (row q, row q + 32) pairs, so one ds_read_b32 feeds both strips,
//               the sliding window is one v_pk_add_u16 + one v_pk_sub_u16 for 128 pixels, and "mean >= v + C", i.e.
//               s + n/2 >= n (v + C) with n = WIN^2, is v_pk_mad_u16 (n v + K, K = n C - n/2 >= 0: the host checks the range) and one
//               saturating v_pk_sub_u16 whose halves are zero where the pixel is set.  The two ballots of a row go into lanes k and 8 + k of
//               two registers (v_writelane) and the 16 rows of a wave leave in ONE store instruction -- the per-row store of
//               k_adaptive_threshold_t (address arithmetic on every lane for a store two lanes execute) was half of its vertical pass.
//               12 -> ~5.5 instructions per 64 pixels and row.
//   pyramid:    the detector's /2 pyramid (buildPyramid, markerdetector_impl.cpp:1299-1488; exact halving = the 2 x 2 mean) from the
//               tile's own pixels, which are in LDS anyway: a 64 x 64 tile holds its 32 x 32, 16 x 16, 8 x 8 and 4 x 4 descendants whole.
//               Four launches (k_half_area4) and a second read of every frame less on the detector's chain.
typedef unsigned short th_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t th_quad(uint32_t r0, uint32_t r1) // two outputs (16-bit lanes) from one dword of each source row
{
    const uint32_t sum = (r0 & 0x00ff00ffu) + ((r0 >> 8) & 0x00ff00ffu) + (r1 & 0x00ff00ffu) + ((r1 >> 8) & 0x00ff00ffu) + 0x00020002u;
    return (sum >> 2) & 0x00ff00ffu;
}
__device__ __forceinline__ uint32_t th_half4(uint2 a0, uint2 a1) // four outputs from eight pixels of two rows
{
    return __builtin_amdgcn_perm(th_quad(a0.y, a1.y), th_quad(a0.x, a1.x), 0x06040200u);
}
template <int WIN>
__global__ __launch_bounds__(256) void k_threshold_pyr(ImgView src, int W, int H, uint32_t kk /* K | K << 16 */,
                                                       uint32_t* __restrict__ bits, size_t bits_fstride, int wpr, int ntx, int ntiles,
                                                       int total, ImgView pyr, ThrPyr P)
{
    constexpr int R = WIN / 2, NDW = R <= 3 ? 3 : 5, LEAD = R <= 3 ? 4 : 8; // window = bytes x - LEAD .. x - LEAD + 4 NDW - 1
    constexpr int ROWS = 64 + 2 * R, QP = (32 + 2 * R) | 1;                   // pairs per LDS column: an odd number of dwords
    __shared__ __align__(16) uint32_t shp[64 * QP];                           // (sum of row q, sum of row q + 32) of a column
    __shared__ __align__(16) uint32_t spx[64][16];
    __shared__ __align__(16) uint32_t l1[32][8];
    __shared__ __align__(16) uint32_t l2[16][4];
    __shared__ __align__(16) uint32_t l3[8][2];
    int tile, f;
    if (!xcd_remap(ntiles, total, tile, f)) return;
    const int tyi = tile / ntx;
    const int tx0 = (tile - tyi * ntx) * 64, ty0 = tyi * 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint8_t* img = src.base + (size_t)f * src.fstride;
    constexpr int NIT = (ROWS * 16 + 255) / 256;
    uint32_t w[NIT][NDW];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int it = tid + 256 * k;
        const int rr = it >> 4, x = tx0 + 4 * (it & 15);
#pragma unroll
        for (int j = 0; j < NDW; j++) w[k][j] = 0;
        if (rr < ROWS) {
            const uint8_t* row = img + (uint32_t)__mul24(min(max(ty0 + rr - R, 0), H - 1), src.pitch); // BORDER_REPLICATE
            if (x >= LEAD && x - LEAD + 4 * NDW <= W) {
#pragma unroll
                for (int j = 0; j < NDW; j++) w[k][j] = reinterpret_cast<const u32_unaligned_t*>(row + x - LEAD)[j];
            } else {
                // the window crosses the left or right image border: see k_adaptive_threshold_t
                const uint32_t first = reinterpret_cast<const u32_unaligned_t*>(row)[0];
                const uint32_t last = reinterpret_cast<const u32_unaligned_t*>(row + W - 4)[0];
#pragma unroll
                for (int j = 0; j < NDW; j++) {
                    const int q = x - LEAD + 4 * j, n_in = W - q;
                    if (q < 0) w[k][j] = __builtin_amdgcn_perm(0u, first, 0x00000000u);
                    else if (n_in >= 4) w[k][j] = reinterpret_cast<const u32_unaligned_t*>(row + q)[0];
                    else w[k][j] = __builtin_amdgcn_perm(0u, last, n_in <= 1 ? 0x03030303u : n_in == 2 ? 0x03030302u : 0x03030201u);
                }
            }
        }
    }
    uint16_t* shh = reinterpret_cast<uint16_t*>(shp);
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int it = tid + 256 * k;
        const int rr = it >> 4, c = 4 * (it & 15);
        if (rr < ROWS) {
            uint32_t sum[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                constexpr uint32_t ONES = 0x01010101u;
                const int a = LEAD + i - R;
                uint32_t acc = 0;
#pragma unroll
                for (int q = 0; q < (WIN + 3) / 4; q++) {
                    const int b = a + 4 * q, j = b >> 2, sft = b & 3;                 // window bytes b .. b+3
                    const uint32_t hi = j + 1 < NDW ? w[k][j + 1] : 0u;
                    const uint32_t v = sft ? __builtin_amdgcn_alignbyte(hi, w[k][j], sft) : w[k][j];
                    const int left = WIN - 4 * q;                                       // bytes still to add
                    acc = bl_dot4_a(v, left >= 4 ? ONES : (ONES >> (8 * (4 - left))), acc);
                }
                sum[i] = acc;
            }
            // row rr is the low half of pair rr and the high half of pair rr - 32 (rows 32 .. 32 + 2 R - 1 are both)
            if (rr < 32 + 2 * R) {
#pragma unroll
                for (int i = 0; i < 4; i++) shh[((c + i) * QP + rr) * 2] = (uint16_t)sum[i];
            }
            if (rr >= 32) {
#pragma unroll
                for (int i = 0; i < 4; i++) shh[((c + i) * QP + rr - 32) * 2 + 1] = (uint16_t)sum[i];
            }
            if (rr >= R && rr < 64 + R) spx[rr - R][it & 15] = w[k][LEAD / 4]; // the tile's own pixels x .. x+3
        }
    }
    __syncthreads();
    // vertical: lane = column, wave = rows 8 wid .. + 7 (low halves) and 8 wid + 32 .. + 39 (high halves)
    {
        const int c = lane, r0 = 8 * wid;
        const uint32_t* hp = &shp[c * QP + r0];
        uint32_t v[8 + 2 * R];
#pragma unroll
        for (int k = 0; k < 8 + 2 * R; k++) v[k] = hp[k];
        th_u16x2 s = __builtin_bit_cast(th_u16x2, v[0]);
#pragma unroll
        for (int k = 1; k < WIN; k++) s += __builtin_bit_cast(th_u16x2, v[k]);
        const unsigned long long xm = __ballot(tx0 + c < W);
        const uint8_t* px = reinterpret_cast<const uint8_t*>(&spx[r0][0]) + c;
        const th_u16x2 nn = {(unsigned short)(WIN * WIN), (unsigned short)(WIN * WIN)}, kv = __builtin_bit_cast(th_u16x2, kk);
        uint32_t mlo = 0, mhi = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t pp = (uint32_t)px[k * 64] | ((uint32_t)px[(k + 32) * 64] << 16);
            const th_u16x2 rhs = __builtin_bit_cast(th_u16x2, pp) * nn + kv;               // n v + K <= 65535 (host)
            const uint32_t d = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(rhs, s)); // half == 0 <=> s >= n v + K
            const unsigned long long ma = __ballot((d & 0xffffu) == 0u) & xm, mb = __ballot(d < 0x10000u) & xm;
            // (this compiler has no writelane builtin; the ballots are scalar registers already)
            asm("v_writelane_b32 %0, %1, %2" : "+v"(mlo) : "s"((uint32_t)ma), "n"(k));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(mhi) : "s"((uint32_t)(ma >> 32)), "n"(k));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(mlo) : "s"((uint32_t)mb), "n"(8 + k));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(mhi) : "s"((uint32_t)(mb >> 32)), "n"(8 + k));
            if (k < 7) s = s + __builtin_bit_cast(th_u16x2, v[k + WIN]) - __builtin_bit_cast(th_u16x2, v[k]);
        }
        if (lane < 16) {
            const int y = ty0 + r0 + 32 * (lane >> 3) + (lane & 7), word = tx0 >> 5;
            if (y < H) {
                uint32_t* o = bits + (size_t)f * bits_fstride + (uint32_t)(__mul24(y, wpr) + word);
                o[0] = mlo;
                if (word + 1 < wpr) o[1] = mhi;
            }
        }
    }
    // the /2 pyramid of the tile (uniform branches: P is a kernel argument)
    if (P.n >= 1) {
        uint8_t* pf = pyr.base_w + (size_t)f * pyr.fstride;
        {
            const int j = tid >> 3, d = tid & 7;
            const uint2 a0 = *reinterpret_cast<const uint2*>(&spx[2 * j][2 * d]), a1 = *reinterpret_cast<const uint2*>(&spx[2 * j + 1][2 * d]);
            const uint32_t o = th_half4(a0, a1);
            l1[j][d] = o;
            const int y = (ty0 >> 1) + j, x = (tx0 >> 1) + 4 * d;
            if (y < P.h[0] && x < P.w[0]) *reinterpret_cast<uint32_t*>(pf + P.off[0] + (uint32_t)(__mul24(y, P.pitch[0]) + x)) = o;
        }
        if (P.n >= 2) {
            __syncthreads();
            if (tid < 64) {
                const int j = tid >> 2, d = tid & 3;
                const uint2 a0 = *reinterpret_cast<const uint2*>(&l1[2 * j][2 * d]), a1 = *reinterpret_cast<const uint2*>(&l1[2 * j + 1][2 * d]);
                const uint32_t o = th_half4(a0, a1);
                l2[j][d] = o;
                const int y = (ty0 >> 2) + j, x = (tx0 >> 2) + 4 * d;
                if (y < P.h[1] && x < P.w[1]) *reinterpret_cast<uint32_t*>(pf + P.off[1] + (uint32_t)(__mul24(y, P.pitch[1]) + x)) = o;
            }
        }
        if (P.n >= 3) {
            __syncthreads();
            if (tid < 16) {
                const int j = tid >> 1, d = tid & 1;
                const uint2 a0 = *reinterpret_cast<const uint2*>(&l2[2 * j][2 * d]), a1 = *reinterpret_cast<const uint2*>(&l2[2 * j + 1][2 * d]);
                const uint32_t o = th_half4(a0, a1);
                l3[j][d] = o;
                const int y = (ty0 >> 3) + j, x = (tx0 >> 3) + 4 * d;
                if (y < P.h[2] && x < P.w[2]) *reinterpret_cast<uint32_t*>(pf + P.off[2] + (uint32_t)(__mul24(y, P.pitch[2]) + x)) = o;
            }
        }
        if (P.n >= 4) {
            __syncthreads();
            if (tid < 4) {
                const int j = tid;
                const uint2 a0 = *reinterpret_cast<const uint2*>(&l3[2 * j][0]), a1 = *reinterpret_cast<const uint2*>(&l3[2 * j + 1][0]);
                const uint32_t o = th_half4(a0, a1);
                const int y = (ty0 >> 4) + j, x = tx0 >> 4;
                if (y < P.h[3] && x < P.w[3]) *reinterpret_cast<uint32_t*>(pf + P.off[3] + (uint32_t)(__mul24(y, P.pitch[3]) + x)) = o;
            }
        }
    }
}
template __global__ void k_threshold_pyr<5>(ImgView, int, int, uint32_t, uint32_t*, size_t, int, int, int, int, ImgView, ThrPyr);
template __global__ void k_threshold_pyr<7>(ImgView, int, int, uint32_t, uint32_t*, size_t, int, int, int, int, ImgView, ThrPyr);
template __global__ void k_threshold_pyr<11>(ImgView, int, int, uint32_t, uint32_t*, size_t, int, int, int, int, ImgView, ThrPyr);
template __global__ void k_threshold_pyr<15>(ImgView, int, int, uint32_t, uint32_t*, size_t, int, int, int, int, ImgView, ThrPyr);

// The adaptive threshold on the matrix cores (round 6; the pipeline is bound by vector-ALU issue, see k_blur7_mfma in orb_kernels.hip,
// whose structure this is): a wave walks a 32-column strip of the frame in blocks of 32 rows;
//   pass 1  row sums of the WIN-pixel box (pixels minus 128 x a band matrix of ones, BORDER_REPLICATE folded into the matrix in x, the
//           lane's row index clamped in y), and -- one more MFMA with a selection matrix -- the block's own pixels TRANSPOSED into the
//           layout pass 2 reads (lane = column, bytes = rows);
//   pass 2  column sums of the row sums' two byte planes over this block and the previous one, plus the centre pixels times -WIN^2:
//           the accumulator is  box sum - n v - (n C - n / 2),  n = WIN^2, whose sign is the pixel ("mean >= v + C");
//   bits    the sign bytes of four accumulators gathered by v_perm and one multiplication, a lane's 16 bits spread to their places in
//           the row's word, the two halves of a row joined by one cross-lane read: a 32-bit word of the bit image per row and block.
// Windows up to 15 (WIN^2 above 127: the centre's weight in two matrices); ~8 lane operations a pixel where the dot-product kernels
// issue ~25.
typedef int tm_v4i __attribute__((ext_vector_type(4)));
typedef int tm_v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_threshold_mfma(ImgView src, int W, int H, int rb /* rows a block starts above 32 j */, int kinit,
                                                        const ThrStrip* __restrict__ strips, const uint4* __restrict__ tabs,
                                                        const uint4* __restrict__ tab2, uint32_t* __restrict__ bits, size_t bits_fstride,
                                                        int wpr, int nstrips, int nx, int total, int two_centre)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int bx, f;
    if (!xcd_remap(nx, total, bx, f)) return;
    const int si = bx * 4 + wid;
    if (si >= nstrips) return;
    const ThrStrip S = strips[si];
    const int r = lane & 31, half = lane >> 5;
    const uint8_t* img = src.base + (size_t)f * src.fstride;
    // pass-1 matrices of the strip: box (K blocks a, b), selection (a, b); pass-2 matrices: box (previous block, this block), centre (prev, this)
    const tm_v4i Ba = __builtin_bit_cast(tm_v4i, tabs[(size_t)S.tab * 64 + lane]), Bb = __builtin_bit_cast(tm_v4i, tabs[(size_t)(S.tab + 1) * 64 + lane]);
    const tm_v4i Ia = __builtin_bit_cast(tm_v4i, tabs[(size_t)(S.tab + 2) * 64 + lane]), Ib = __builtin_bit_cast(tm_v4i, tabs[(size_t)(S.tab + 3) * 64 + lane]);
    const tm_v4i V2a = __builtin_bit_cast(tm_v4i, tab2[lane]), V2b = __builtin_bit_cast(tm_v4i, tab2[64 + lane]);
    const tm_v4i C2a = __builtin_bit_cast(tm_v4i, tab2[128 + lane]), C2b = __builtin_bit_cast(tm_v4i, tab2[192 + lane]);
    // windows of 13 and 15 pixels: -WIN^2 does not fit a signed byte, the centre takes two matrices (-113 and the rest)
    const tm_v4i C3a = __builtin_bit_cast(tm_v4i, tab2[256 + lane]), C3b = __builtin_bit_cast(tm_v4i, tab2[320 + lane]);
    const uint32_t ca = (uint32_t)(half ? S.c1 : S.c0), cb = (uint32_t)S.c2;
    typedef uint32_t tm_u32x4 __attribute__((ext_vector_type(4)));
    typedef tm_u32x4 tm_u32x4_unaligned __attribute__((aligned(1)));
    const int nblk = ((H + 31) >> 5) + 1;
    auto load_rows = [&](int j, tm_u32x4& pa, tm_u32x4& pb) {
        const uint32_t ro = (uint32_t)__mul24(min(max(32 * j - rb + r, 0), H - 1), src.pitch);   // BORDER_REPLICATE
        pa = *reinterpret_cast<const tm_u32x4_unaligned*>(img + (ro + ca));
        pb = *reinterpret_cast<const tm_u32x4_unaligned*>(img + (ro + cb));
    };
    tm_u32x4 na, nb;
    load_rows(0, na, nb);
    // (the accumulators' start values as operands of their own: no moves per block)
    const tm_v16i c128 = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
    const tm_v16i ckin = {kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit, kinit};
    const tm_v16i czero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    tm_v4i ph = {0, 0, 0, 0}, pl = {0, 0, 0, 0}, pc = {0, 0, 0, 0};
    // where a lane's 16 sign bits go in the row's word: pixel 4 half + 8 q + k
    const uint32_t valid = (S.x0 + 32 <= W) ? 0xffffffffu : ((1u << (W - S.x0)) - 1u);
    uint32_t* brow = bits + (size_t)f * bits_fstride + (S.x0 >> 5);
    for (int j = 0; j < nblk; j++) {
        const tm_u32x4 qa = na, qb = nb;
        if (j + 1 < nblk) load_rows(j + 1, na, nb);
        constexpr uint32_t SGN = 0x80808080u;
        const tm_v4i A1a = {(int)(qa.x ^ SGN), (int)(qa.y ^ SGN), (int)(qa.z ^ SGN), (int)(qa.w ^ SGN)};
        const tm_v4i A1b = {(int)(qb.x ^ SGN), (int)(qb.y ^ SGN), (int)(qb.z ^ SGN), (int)(qb.w ^ SGN)};
        tm_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1a, Ba, c128, 0, 0, 0);
        tm_v16i sel = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1a, Ia, czero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1b, Bb, acc, 0, 0, 0);
        sel = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1b, Ib, sel, 0, 0, 0);
        tm_v4i nh, nl, nc;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t a0 = (uint32_t)acc[4 * q], a1 = (uint32_t)acc[4 * q + 1], a2 = (uint32_t)acc[4 * q + 2], a3 = (uint32_t)acc[4 * q + 3];
            nh[q] = (int)(__builtin_amdgcn_perm(a1, a0, 0x0c0c0501u) | __builtin_amdgcn_perm(a3, a2, 0x05010c0cu));
            nl[q] = (int)((__builtin_amdgcn_perm(a1, a0, 0x0c0c0400u) | __builtin_amdgcn_perm(a3, a2, 0x04000c0cu)) ^ SGN);
            const uint32_t c0 = (uint32_t)sel[4 * q], c1 = (uint32_t)sel[4 * q + 1], c2 = (uint32_t)sel[4 * q + 2], c3 = (uint32_t)sel[4 * q + 3];
            nc[q] = (int)(__builtin_amdgcn_perm(c1, c0, 0x0c0c0400u) | __builtin_amdgcn_perm(c3, c2, 0x04000c0cu));   // (pixel - 128: one signed byte)
        }
        if (j >= 1) {
            tm_v16i ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(ph, V2a, czero, 0, 0, 0);
            tm_v16i al = __builtin_amdgcn_mfma_i32_32x32x32_i8(pl, V2a, ckin, 0, 0, 0);
            ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(nh, V2b, ah, 0, 0, 0);
            al = __builtin_amdgcn_mfma_i32_32x32x32_i8(nl, V2b, al, 0, 0, 0);
            al = __builtin_amdgcn_mfma_i32_32x32x32_i8(pc, C2a, al, 0, 0, 0);
            al = __builtin_amdgcn_mfma_i32_32x32x32_i8(nc, C2b, al, 0, 0, 0);
            if (two_centre) {
                al = __builtin_amdgcn_mfma_i32_32x32x32_i8(pc, C3a, al, 0, 0, 0);
                al = __builtin_amdgcn_mfma_i32_32x32x32_i8(nc, C3b, al, 0, 0, 0);
            }
            uint32_t wbits = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t t4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) t4[k] = ((uint32_t)ah[4 * q + k] << 8) + (uint32_t)al[4 * q + k];
                // sign bytes of the four sums side by side, their top bits to a nibble (bit k = pixel k is OFF)
                const uint32_t sg = (__builtin_amdgcn_perm(t4[1], t4[0], 0x0c0c0703u) | __builtin_amdgcn_perm(t4[3], t4[2], 0x07030c0cu)) & SGN;
                const uint32_t nib = (sg * 0x00204081u) >> 28;
                wbits |= nib << (8 * q);
            }
            wbits <<= 4 * half;
            wbits |= (uint32_t)__shfl_xor((int)wbits, 32);
            const int y = 32 * (j - 1) + r;
            if (half == 0 && y < H) brow[(uint32_t)__mul24(y, wpr)] = ~wbits & valid;
        }
        ph = nh; pl = nl; pc = nc;
    }
}

// ---------------------------------------------------------------------------------------- specks --------------
// The speck passes of aruco_trace.hpp ("FEWER WALKS" (2)) between the threshold and the contour kernels: what they clear has no border
// of more than 68 points and changes no other border, and on textured frames it is most of the start candidates and a fifth of the
// grid markers the contour kernels would otherwise walk.  A workgroup owns SPK_ROWS rows of one frame at full width and holds them
// with ORBFE_SPECK_REACH rows above and below (what the two passes together can depend on) in LDS, in the padded layout of the contour
// kernels (pixel (x, y) = bit x + 1 of row y + 1; zero beyond the frame); every pass is three sweeps over the tile's words -- rim
// masks of each row, anchors (empty rims), clear -- with a barrier between them.  HBM-bound byte work in principle (38 KB in, 38 KB
// out per 640 x 480 frame); the sweeps are ~35 VALU + ~25 LDS instructions per word and pass.
template <int WW, int HH>
__device__ __forceinline__ void speck_pass_lds(uint32_t* P, uint32_t* Fm, uint32_t* Sm, uint32_t* An, int NR, int PWS, int tid)
{
    const int n = NR * PWS;
    const float inv_pws = 1.0f / (float)PWS;
    for (int i = tid; i < n; i += SPK_THREADS) {
        const int r = (int)(((float)i + 0.5f) * inv_pws), j = i - __mul24(r, PWS);   // exact: i < 2^16
        uint32_t f = 0, sd = 0;
        if (j + 1 < PWS) speck_row_masks<WW>(P[i], P[i + 1], &f, &sd);              // (the last word of a row is the zero word)
        Fm[i] = f; Sm[i] = sd;
    }
    __syncthreads();
    for (int i = tid; i < n; i += SPK_THREADS) {
        const int r = (int)(((float)i + 0.5f) * inv_pws);
        uint32_t a = 0;
        if (r + HH + 1 < NR) {   // anchors whose rim leaves the tile decide nothing here: the rows they could clear are another workgroup's
            uint32_t occ = Fm[i] | Fm[i + (HH + 1) * PWS];
#pragma unroll
            for (int k = 1; k <= HH; k++) occ |= Sm[i + k * PWS];
            a = ~occ;
        }
        An[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < n; i += SPK_THREADS) {
        const int r = (int)(((float)i + 0.5f) * inv_pws), j = i - __mul24(r, PWS);
        if (r < HH) continue;
        uint32_t e = 0, el = 0;
#pragma unroll
        for (int dy = 1; dy <= HH; dy++) {
            e |= An[i - dy * PWS];
            if (j) el |= An[i - dy * PWS - 1];
        }
        P[i] &= ~speck_dilate<WW>(e, el);
    }
    __syncthreads();
}

__global__ __launch_bounds__(SPK_THREADS) void k_speck_clean(const uint32_t* __restrict__ bits, size_t bits_fstride, int wpr_g, int W, int H,
                                                             uint32_t* __restrict__ out)
{
    extern __shared__ __align__(16) uint32_t spk_smem[];
    constexpr int RCH = ORBFE_SPECK_REACH, NR = SPK_ROWS + 2 * RCH;
    const int tid = threadIdx.x, f = blockIdx.y, y0 = blockIdx.x * SPK_ROWS;       // image rows y0 .. y0 + SPK_ROWS - 1
    const int pw = (W + 2 + 31) >> 5, PWS = pw + 1, n = NR * PWS;
    uint32_t* P = spk_smem;
    uint32_t* Fm = P + n;
    uint32_t* Sm = Fm + n;
    uint32_t* An = Sm + n;
    const uint32_t* gb = bits + (size_t)f * bits_fstride;
    const float inv_pws = 1.0f / (float)PWS;
    for (int i = tid; i < n; i += SPK_THREADS) {
        const int r = (int)(((float)i + 0.5f) * inv_pws), j = i - __mul24(r, PWS), py = y0 + 1 - RCH + r;   // word j of padded row py
        uint32_t v = 0;
        if (j < pw && py >= 1 && py <= H) {
            const uint32_t* row = gb + (uint32_t)__mul24(py - 1, wpr_g);
            const uint32_t cur = j < wpr_g ? row[j] : 0u;
            const uint32_t prv = (j >= 1 && j - 1 < wpr_g) ? row[j - 1] : 0u;
            v = (cur << 1) | (prv >> 31);
        }
        P[i] = v;
    }
    __syncthreads();
    speck_pass_lds<ORBFE_SPECK_W1, ORBFE_SPECK_H1>(P, Fm, Sm, An, NR, PWS, tid);
    speck_pass_lds<ORBFE_SPECK_W2, ORBFE_SPECK_H2>(P, Fm, Sm, An, NR, PWS, tid);
    // the owned rows back in the threshold kernel's layout (pixel x = bit x of its row's words)
    uint32_t* ob = out + (size_t)f * bits_fstride;
    const float inv_wg = 1.0f / (float)wpr_g;
    for (int i = tid; i < SPK_ROWS * wpr_g; i += SPK_THREADS) {
        const int r = (int)(((float)i + 0.5f) * inv_wg), j = i - __mul24(r, wpr_g), y = y0 + r;
        if (y >= H) break;
        const uint32_t* row = P + __mul24(RCH + r, PWS);
        ob[(uint32_t)__mul24(y, wpr_g) + j] = (row[j] >> 1) | (row[j + 1] << 31);
    }
}

// exact 2x downscale = INTER_AREA 2x2 mean (what cv::resize(INTER_LINEAR) does for an exact factor of two)
__global__ __launch_bounds__(256) void k_half_area(ImgView src, ImgView dst, int dw, int dh)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (x >= dw || y >= dh) return;
    const uint8_t* s0 = src.base + (size_t)f * src.fstride + (size_t)(2 * y) * src.pitch + 2 * x;
    const uint8_t* s1 = s0 + src.pitch;
    dst.base_w[(size_t)f * dst.fstride + (size_t)y * dst.pitch + x] = (uint8_t)((s0[0] + s0[1] + s1[0] + s1[1] + 2) >> 2);
}

// The same, four output pixels x two rows per thread: two aligned 8-byte loads per source row pair, the 2 x 2 sums on two 16-bit
// lanes per dword, one dword store per output row (the byte-per-thread kernel above issues 16 byte loads and 4 byte stores for the
// same four pixels; it stays for sources whose rows are not 8-byte aligned).  dw4 = ceil(dw / 4); rows of dst are dword aligned.
__global__ __launch_bounds__(256) void k_half_area4(ImgView src, ImgView dst, int dw4, int dh)
{
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    const int yp = i / dw4, x4 = i - yp * dw4, y = 2 * yp;
    if (y >= dh) return;
    const uint8_t* s = src.base + (size_t)f * src.fstride + (size_t)(2 * y) * src.pitch + 8 * x4;
    uint8_t* d = dst.base_w + (size_t)f * dst.fstride + (size_t)y * dst.pitch + 4 * x4;
    const bool two = y + 1 < dh;
    const uint2 a0 = *reinterpret_cast<const uint2*>(s), a1 = *reinterpret_cast<const uint2*>(s + src.pitch);
    uint2 b0 = a0, b1 = a1;
    if (two) { b0 = *reinterpret_cast<const uint2*>(s + 2 * (size_t)src.pitch); b1 = *reinterpret_cast<const uint2*>(s + 3 * (size_t)src.pitch); }
    auto quad = [](uint32_t r0, uint32_t r1) -> uint32_t { // two outputs (16-bit lanes) from one dword of each source row
        const uint32_t sum = (r0 & 0x00ff00ffu) + ((r0 >> 8) & 0x00ff00ffu) + (r1 & 0x00ff00ffu) + ((r1 >> 8) & 0x00ff00ffu) + 0x00020002u;
        return (sum >> 2) & 0x00ff00ffu;
    };
    auto pack = [](uint32_t lo, uint32_t hi) -> uint32_t { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }; // bytes 0, 2 of lo, then of hi
    *reinterpret_cast<uint32_t*>(d) = pack(quad(a0.x, a1.x), quad(a0.y, a1.y));
    if (two) *reinterpret_cast<uint32_t*>(d + dst.pitch) = pack(quad(b0.x, b1.x), quad(b0.y, b1.y));
}

// The first NF exact halvings in ONE launch (round 6): a thread owns a 2^NF x 2^NF block of the source (16 x 16 or 8 x 8: its rows as
// 16- or 8-byte loads, lanes side by side along x), halves it in registers level by level -- every level from the ROUNDED level above,
// as the chain of launches does -- and stores its 8 x 8, 4 x 4, 2 x 2 and 1 x 1 pieces.  The source is read once (k_half_area4 x 4 read
// every level back), ~1.5 vector instructions a source pixel instead of 5.6, and on the detector's stream of a 640 x 480 batch -- where
// the pyramid runs in line, not forked -- one launch stands where four dependent ones stood (88 - 96 us of its chain).
template <int NF> __global__ __launch_bounds__(256) void k_half_pyr(ImgView src, HalfPyrDst P, int bw, int nblocks)
{
    constexpr int BS = 1 << NF, DW = BS / 4;
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= nblocks) return;
    const int by = i / bw, bx = i - by * bw;
    const uint8_t* s = src.base + (size_t)f * src.fstride + (size_t)(by * BS) * src.pitch + (size_t)bx * BS;
    uint8_t* dbase = P.base + (size_t)f * P.fstride;
    auto quad = [](uint32_t r0, uint32_t r1) -> uint32_t { // two outputs (16-bit lanes) from one dword of each source row
        const uint32_t sum = (r0 & 0x00ff00ffu) + ((r0 >> 8) & 0x00ff00ffu) + (r1 & 0x00ff00ffu) + ((r1 >> 8) & 0x00ff00ffu) + 0x00020002u;
        return (sum >> 2) & 0x00ff00ffu;
    };
    auto pack = [](uint32_t lo, uint32_t hi) -> uint32_t { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }; // bytes 0, 2 of lo, then of hi
    uint32_t a[BS][DW];
#pragma unroll
    for (int r = 0; r < BS; r++) {
        if constexpr (DW == 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(s + (size_t)r * src.pitch);
            a[r][0] = v.x; a[r][1] = v.y; a[r][2] = v.z; a[r][3] = v.w;
        } else {
            const uint2 v = *reinterpret_cast<const uint2*>(s + (size_t)r * src.pitch);
            a[r][0] = v.x; a[r][1] = v.y;
        }
    }
    // level 1: BS / 2 rows of DW / 2 dwords
    constexpr int R1 = BS / 2, D1 = DW / 2;
    uint32_t b[R1][D1];
#pragma unroll
    for (int r = 0; r < R1; r++)
#pragma unroll
        for (int d = 0; d < D1; d++) b[r][d] = pack(quad(a[2 * r][2 * d], a[2 * r + 1][2 * d]), quad(a[2 * r][2 * d + 1], a[2 * r + 1][2 * d + 1]));
    {
        uint8_t* d1 = dbase + P.off[0] + (size_t)(by * R1) * P.pitch[0] + (size_t)bx * R1;
#pragma unroll
        for (int r = 0; r < R1; r++) {
            if constexpr (D1 == 2) *reinterpret_cast<uint2*>(d1 + (size_t)r * P.pitch[0]) = make_uint2(b[r][0], b[r][1]);
            else *reinterpret_cast<uint32_t*>(d1 + (size_t)r * P.pitch[0]) = b[r][0];
        }
    }
    // level 2 from level 1; NF = 4: 4 rows of one dword; NF = 3: 2 rows of two pixels
    constexpr int R2 = R1 / 2;
    uint32_t c[R2];   // D1 == 2: four pixels a row; D1 == 1: two pixels in the low half
#pragma unroll
    for (int r = 0; r < R2; r++) {
        if constexpr (D1 == 2) c[r] = pack(quad(b[2 * r][0], b[2 * r + 1][0]), quad(b[2 * r][1], b[2 * r + 1][1]));
        else c[r] = pack(quad(b[2 * r][0], b[2 * r + 1][0]), 0u);
    }
    {
        uint8_t* d2 = dbase + P.off[1] + (size_t)(by * R2) * P.pitch[1] + (size_t)bx * R2;
#pragma unroll
        for (int r = 0; r < R2; r++) {
            if constexpr (D1 == 2) *reinterpret_cast<uint32_t*>(d2 + (size_t)r * P.pitch[1]) = c[r];
            else *reinterpret_cast<uint16_t*>(d2 + (size_t)r * P.pitch[1]) = (uint16_t)c[r];
        }
    }
    if constexpr (NF == 4) {
        // level 3: 2 rows of two pixels; level 4: one pixel
        uint32_t e[2];
#pragma unroll
        for (int r = 0; r < 2; r++) e[r] = pack(quad(c[2 * r], c[2 * r + 1]), 0u);
        uint8_t* d3 = dbase + P.off[2] + (size_t)(by * 2) * P.pitch[2] + (size_t)bx * 2;
        *reinterpret_cast<uint16_t*>(d3) = (uint16_t)e[0];
        *reinterpret_cast<uint16_t*>(d3 + P.pitch[2]) = (uint16_t)e[1];
        const uint32_t q = quad(e[0] & 0xffffu, e[1] & 0xffffu);
        dbase[P.off[3] + (size_t)by * P.pitch[3] + bx] = (uint8_t)q;
    } else {
        // level 3: one pixel
        const uint32_t q = quad(c[0] & 0xffffu, c[1] & 0xffffu);
        dbase[P.off[2] + (size_t)by * P.pitch[2] + bx] = (uint8_t)q;
    }
}
template __global__ void k_half_pyr<4>(ImgView, HalfPyrDst, int, int);
template __global__ void k_half_pyr<3>(ImgView, HalfPyrDst, int, int);

// ---------------------------------------------------------------------------------------- contours ------------
struct ApPt { int x, y; };

// Maximum / minimum over the G lanes of a group (G = 64: the wave; G = 16: a DPP row, every lane of the row gets the result:
// xor 1, xor 2 as quad permutations, then row_half_mirror and row_mirror -- the groups of a wave may have diverged, a row has not).
template <int G> __device__ __forceinline__ int group_max(int v)
{
    if (G == 64) return wave_max(v);
    v = max(v, ORBFE_DPP(v, v, 0xB1, 0xf)); v = max(v, ORBFE_DPP(v, v, 0x4E, 0xf));
    v = max(v, ORBFE_DPP(v, v, 0x141, 0xf)); v = max(v, ORBFE_DPP(v, v, 0x140, 0xf));
    return v;
}
template <int G> __device__ __forceinline__ int group_min(int v)
{
    if (G == 64) return wave_min(v);
    v = min(v, ORBFE_DPP(v, v, 0xB1, 0xf)); v = min(v, ORBFE_DPP(v, v, 0x4E, 0xf));
    v = min(v, ORBFE_DPP(v, v, 0x141, 0xf)); v = min(v, ORBFE_DPP(v, v, 0x140, 0xf));
    return v;
}

// approxPolyDP (closed curve) by a group of G lanes (`lane`: 0 .. G-1); P = contour points (x | y<<16), n > 0.  Returns the number
// of vertices (<= OUTCAP, or OUTCAP+1 on overflow) in `out` (LDS).  Reductions keep the FIRST maximum like the serial loops of
// cv::approxPolyDP_ ("dist > max_dist").  The callers only ask "is it 4?": an overflow of `out` (or of the stack: every pending
// slice ends as at least one vertex) means more than the capacity before the clean-up pass, which at most halves the count --
// so any capacity of 10 or more answers exactly.
template <int G, int OUTCAP, int STACKCAP>
__device__ __forceinline__ int approx_poly_g(const uint32_t* __restrict__ P, int n, ApPt* out, int2* stack, int lane)
{
    auto rd = [&](int i) -> ApPt { const uint32_t v = P[i]; return ApPt{(int)(v & 0xffff), (int)(v >> 16)}; };
    // Largest metric(point (i0 + j) mod n) over j in [j_begin, j_end) and the first j it occurs at, per lane (lane l looks at
    // j = j_begin + l, + G, ...; ">" keeps the earliest).  Four chunks per trip while more than one is left: the loads of a long
    // border (LDS, or the pool in L2 when it does not fit) are in flight together instead of one latency per chunk.
    auto scan = [&](int i0, int j_begin, int j_end, auto metric, int& bd, int& bj) {
        int j0 = j_begin;
        for (; j_end - j0 > G; j0 += 4 * G) {
            uint32_t w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int idx = i0 + min(j0 + u * G + lane, j_end - 1);
                if (idx >= n) idx -= n;
                w[u] = P[idx];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u * G + lane, d = metric(w[u]);
                if (j < j_end && d > bd) { bd = d; bj = j; }
            }
        }
        const int j = j0 + lane;
        if (j < j_end) {
            int idx = i0 + j;
            if (idx >= n) idx -= n;
            const int d = metric(P[idx]);
            if (d > bd) { bd = d; bj = j; }
        }
    };
    double eps = (double)n * 0.05;
    eps *= eps;
    int nout = 0, top = 0;
    int pos = 0, right_start = 0;
    bool le_eps = false;
    ApPt start_pt{0, 0};
    for (int it = 0; it < 3; it++) {
        pos = (pos + right_start) % n;
        start_pt = rd(pos);
        // farthest point from start_pt among j = 1..n-1 at index (pos + j) % n ; first maximum wins.  Coordinates are below 2^14
        // (far above any frame here), so squared distances and cross products fit 32 bits: the maximum is two 32-bit DPP reductions
        // (largest distance, then smallest j among the lanes that hold it) instead of one on 64-bit keys
        int bd = 0, bj = 0x7fffffff; // this lane: largest distance so far and the first j it occurred at (j grows chunk by chunk)
        scan(pos, 1, n, [&](uint32_t w) { const int dx = (int)(w & 0xffff) - start_pt.x, dy = (int)(w >> 16) - start_pt.y; return dx * dx + dy * dy; }, bd, bj);
        const int max_dist = group_max<G>(bd);
        const int first_j = group_min<G>(bd == max_dist ? bj : 0x7fffffff);
        right_start = max_dist > 0 ? first_j : right_start; // no dist > 0: index unchanged
        le_eps = (double)max_dist <= eps;
        // pos returns to the start index after the sweep (READ_PT wrapped n times)
    }
    if (!le_eps) {
        const int A = pos % n, Bi = (right_start + A) % n;
        stack[top++] = make_int2(Bi, A); // right_slice
        stack[top++] = make_int2(A, Bi); // slice (popped first)
    } else {
        out[nout++] = start_pt;
    }
    while (top > 0) {
        const int2 sl = stack[--top];
        const ApPt end_pt = rd(sl.y);
        const ApPt sp = rd(sl.x);
        int p1 = sl.x + 1;
        if (p1 >= n) p1 = 0;
        bool le;
        int split = 0;
        if (p1 != sl.y) {
            const int dx = end_pt.x - sp.x, dy = end_pt.y - sp.y;
            int cnt = sl.y - p1; // points strictly between
            if (cnt < 0) cnt += n;
            int bd = 0, bj = 0x7fffffff;
            scan(p1, 0, cnt, [&](uint32_t w) { const int d = ((int)(w >> 16) - sp.y) * dx - ((int)(w & 0xffff) - sp.x) * dy; return d < 0 ? -d : d; }, bd, bj);
            const int maxd = group_max<G>(bd);
            const int first_j = group_min<G>(bd == maxd ? bj : 0x7fffffff);
            const double md = (double)maxd;
            le = md * md <= eps * (double)((long long)dx * dx + (long long)dy * dy);
            if (maxd > 0) {
                split = p1 + first_j;
                if (split >= n) split -= n;
            }
        } else {
            le = true;
        }
        if (le) {
            if (nout < OUTCAP) out[nout] = sp;
            nout++;
            if (nout > OUTCAP) return OUTCAP + 1;
        } else {
            if (top + 2 > STACKCAP) return OUTCAP + 1;
            stack[top++] = make_int2(split, sl.y);
            stack[top++] = make_int2(sl.x, split);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // clean-up pass (serial, tiny)
    int new_count = nout;
    const int cnt = nout;
    if (cnt > 0) {
        int ps = cnt - 1;
        ApPt s = out[ps]; if (++ps >= cnt) ps = 0;
        int wpos = ps;
        ApPt pt = out[ps]; if (++ps >= cnt) ps = 0;
        for (int i = 0; i < cnt && new_count > 2; i++) {
            ApPt e = out[ps]; if (++ps >= cnt) ps = 0;
            const double dx = e.x - s.x, dy = e.y - s.y;
            const double dist = fabs((pt.x - s.x) * dy - (pt.y - s.y) * dx);
            const double sip = (double)(pt.x - s.x) * (e.x - pt.x) + (double)(pt.y - s.y) * (e.y - pt.y);
            if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
                new_count--;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) out[wpos] = e;
                __builtin_amdgcn_wave_barrier();
                s = e;
                if (++wpos >= cnt) wpos = 0;
                pt = out[ps]; if (++ps >= cnt) ps = 0;
                i++;
                continue;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) out[wpos] = pt;
            __builtin_amdgcn_wave_barrier();
            s = pt;
            if (++wpos >= cnt) wpos = 0;
            pt = e;
        }
    }
    return new_count;
}

__device__ __forceinline__ int approx_poly_wave(const uint32_t* __restrict__ P, int n, ApPt* out, int2* stack, int lane)
{
    return approx_poly_g<64, AP_OUT, AP_STACK>(P, n, out, stack, lane);
}

__device__ bool convex4(const ApPt* p)
{
    const int n = 4;
    ApPt prev_pt = p[(n - 2 + n) % n], cur_pt = p[n - 1];
    int dx0 = cur_pt.x - prev_pt.x, dy0 = cur_pt.y - prev_pt.y, orientation = 0;
    for (int i = 0; i < n; i++) {
        prev_pt = cur_pt;
        cur_pt = p[i];
        const int dx = cur_pt.x - prev_pt.x, dy = cur_pt.y - prev_pt.y;
        const int dxdy0 = dx * dy0, dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return false;
        dx0 = dx;
        dy0 = dy;
    }
    return true;
}

// Last phases of both contour kernels: sort the kept borders into findContours' order (reverse discovery), run
// approxPolyDP(eps = 0.05 * len) + the convexity test on each (one wave per border), compact the 4-gons in order and
// write the per-frame counts.  All NT (a multiple of 64) remaining threads of the workgroup call it.
__device__ __forceinline__ void contours_tail(int f, int tid, int NT, int nkept, unsigned long long* kkey, const int* off_u, int* klen,
                              int* koff, int* rectflag, ApPt* ap_out, int2* ap_stack, const uint32_t* pl,
                              ArKept* __restrict__ kept_out, int kept_cap, ArRect* __restrict__ rects_out, int rect_cap,
                              int32_t* __restrict__ counts, int* s_flags, const int* s_ncand, uint16_t* rank_of,
                              unsigned* tail_q, int nbig, uint32_t* pbuf, int pbuf_pts, uint32_t* pbuf2, int pbuf2_pts)
{
    const int lane = tid & 63, wid = tid >> 6, nwaves = NT >> 6;
#ifdef ORBFE_CT_TIMING
    long long* tdbg = (long long*)(kept_out + (size_t)f * kept_cap + kept_cap - 4) + 8;
    const long long tt0 = clock64();
#endif
    int Pn = 1;
    while (Pn < nkept) Pn <<= 1;
    for (int i = nkept + tid; i < Pn; i += NT) kkey[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= Pn; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (Pn >> 1); t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const unsigned long long a = kkey[i], b = kkey[l];
                const bool up = ((i & k) == 0);
                if ((a > b) == up) { kkey[i] = b; kkey[l] = a; }
            }
            __syncthreads();
        }
    for (int k = tid; k < nkept; k += NT) {
        const unsigned long long key = kkey[k];
        klen[k] = (int)((key >> 13) & 0x7ffff);
        koff[k] = off_u[(int)((key >> 1) & 0xfff)]; // kept index: AR_MAX_KEPT_BIG = 4096 borders at most
    }
    __syncthreads();
#ifdef ORBFE_CT_TIMING
    const long long tt1 = clock64();
#endif
    // approxPolyDP makes many dependent passes over a border's points and its cost grows with the length; a frame has
    // a few long borders and many short ones.  So: the points are staged in LDS that is dead by now (waves < nbig own a
    // big buffer, the others a small one; a border that does not fit is read from the pool), and the borders are
    // ranked by length -- big-buffer waves take them from the long end, the others from the short end, until the
    // two meet (one packed ticket counter: front count | back count << 16).
    for (int k = tid; k < nkept; k += NT) {
        const int lk = klen[k];
        int r = 0;
        for (int j = 0; j < nkept; j++) {
            const int lj = klen[j];
            r += (lj > lk) || (lj == lk && j < k);
        }
        rank_of[r] = (uint16_t)k;
    }
    if (tid == 0) *tail_q = 0u;
    __syncthreads();
    {
        const bool big = wid < nbig;
        const int my_pts = big ? pbuf_pts : pbuf2_pts;
        uint32_t* b = big ? pbuf + wid * pbuf_pts : pbuf2 + (wid - nbig) * pbuf2_pts;
        ApPt* o = ap_out + wid * AP_OUT;
        for (;;) {
            unsigned tk = 0;
            if (lane == 0) tk = atomicAdd(tail_q, big ? 1u : 0x10000u);
            tk = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
            const int fr = (int)(tk & 0xffffu), bk = (int)(tk >> 16);
            if (fr + bk >= nkept) break; // the first nkept tickets are the valid ones: every border exactly once
            const int k = rank_of[big ? fr : nkept - 1 - bk];
            const int n = klen[k];
            int ok = 0;
            if (n > 0) {
                const uint32_t* src = pl + koff[k];
                int nv;
                if (n <= my_pts) {
                    for (int i = lane; i < n; i += 64) b[i] = src[i];
                    __builtin_amdgcn_wave_barrier();
                    nv = approx_poly_wave(b, n, o, ap_stack + wid * AP_STACK, lane);
                } else {
                    nv = approx_poly_wave(src, n, o, ap_stack + wid * AP_STACK, lane);
                }
                __builtin_amdgcn_wave_barrier();
                ok = (nv == 4) && convex4(o);
            }
            if (lane == 0) {
                rectflag[k] = ok;
                if (ok && k < kept_cap) {
                    ArKept kk;
                    kk.off = koff[k]; kk.len = n;
                    for (int j = 0; j < 4; j++) { kk.vx[j] = (short)o[j].x; kk.vy[j] = (short)o[j].y; }
                    kept_out[(size_t)f * kept_cap + k] = kk;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
#ifdef ORBFE_CT_TIMING
    if (tid == 0) { tdbg[0] = tt1 - tt0; tdbg[1] = clock64() - tt1; }
#endif
    if (tid == 0) { // ordered compaction of the rectangles (a few dozen)
        int nr = 0;
        for (int k = 0; k < nkept; k++)
            if (rectflag[k]) {
                if (nr < rect_cap) {
                    const ArKept kk = kept_out[(size_t)f * kept_cap + k];
                    ArRect r;
                    for (int j = 0; j < 4; j++) { r.c[j][0] = (float)kk.vx[j]; r.c[j][1] = (float)kk.vy[j]; }
                    r.off = kk.off; r.len = kk.len;
                    rects_out[(size_t)f * rect_cap + nr] = r;
                } else *s_flags |= 8;
                nr++;
            }
        counts[f * 4 + 0] = nkept;
        counts[f * 4 + 1] = min(nr, rect_cap);
        counts[f * 4 + 2] = *s_flags;
        counts[f * 4 + 3] = *s_ncand;
    }
}

template <bool LDS_BITS>
__global__ __launch_bounds__(CT_PROBE_THREADS) void k_contours_t(const uint32_t* __restrict__ gbits, size_t bits_fstride,
                                                         int wpr_g, int W, int H, int lds_bits_words, int min_len,
                                                         uint32_t* __restrict__ candq, size_t candq_fstride,
                                                         int candq_cap, uint32_t* __restrict__ pool,
                                                         size_t pool_fstride, int pool_cap,
                                                         ArKept* __restrict__ kept_out, int kept_cap,
                                                         ArRect* __restrict__ rects_out, int rect_cap,
                                                         int32_t* __restrict__ counts /*per frame: [nkept, nrect, flags, ncand]*/,
                                                         uint32_t* __restrict__ gpadded, size_t gpadded_fstride,
                                                         int only_flagged)
{
    __builtin_amdgcn_s_setprio(2); // latency-bound: its few waves go first when a VALU-bound kernel shares the CU
    extern __shared__ __align__(16) unsigned char ct_smem[];
    __shared__ int s_ncand, s_next, s_nkept, s_nlong, s_flags;
    __shared__ unsigned s_tailq;
    const int tid = threadIdx.x, f = blockIdx.x;
    if (only_flagged && !(counts[f * 4 + 2] & RL_FLAG_BUG)) return; // (debug builds: redo frames the relay kernel flagged)
    const int wpr = (W + 2 + 31) >> 5, prow = H + 2;
    // LDS carve-up: [bits][kept keys (u64) kept_cap][per-wave approx scratch]
    // the padded bit image lives in LDS when it fits (lds_bits_words > 0), else in an HBM scratch (L2-resident)
    uint32_t* lbits = LDS_BITS ? (uint32_t*)ct_smem : gpadded + (size_t)f * gpadded_fstride;
    unsigned long long* kkey = (unsigned long long*)(ct_smem + (((size_t)lds_bits_words * 4 + 15) & ~(size_t)15));
    int* off_u = (int*)(kkey + kept_cap);
    int* klen = off_u + kept_cap;
    int* koff = klen + kept_cap;
    int* rectflag = koff + kept_cap;
    uint32_t* longq = (uint32_t*)klen; // probe survivors; dead before klen/koff/rectflag are written
    const int lq_cap = 3 * kept_cap;
    ApPt* ap_out = (ApPt*)(rectflag + kept_cap);
    int2* ap_stack = (int2*)(ap_out + CT_WAVES * AP_OUT);
    const uint32_t* gb = gbits + (size_t)f * bits_fstride;
    uint32_t* cq = candq + (size_t)f * candq_fstride; // overflow of the long-walk queue
    uint32_t* pl = pool + (size_t)f * pool_fstride;

    if (tid == 0) { s_ncand = 0; s_next = 0; s_nkept = 0; s_nlong = 0; s_flags = 0; }
#ifdef ORBFE_CT_TIMING
    long long t0 = clock64(), t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
#define CT_STAMP(v) v = clock64()
#else
#define CT_STAMP(v)
#endif
    // ---- (a) padded bit image into LDS: pixel (x, y) -> bit x+1 of row y+1
    for (int i = tid; i < wpr * prow; i += CT_PROBE_THREADS) {
        const int py = i / wpr, j = i - py * wpr;
        uint32_t v = 0;
        if (py >= 1 && py <= H) {
            const uint32_t* row = gb + (size_t)(py - 1) * wpr_g;
            const uint32_t cur = j < wpr_g ? row[j] : 0u;
            const uint32_t prv = (j >= 1 && j - 1 < wpr_g) ? row[j - 1] : 0u;
            v = (cur << 1) | (prv >> 31);
        }
        lbits[i] = v;
    }
    if (tid < 2) lbits[wpr * prow + tid] = 0; // spare words read by ring8()'s funnel loads
    __threadfence_block();
    __syncthreads();
    const BitImage im{lbits, wpr, W, H};
    CT_STAMP(t1);

    // ---- (b)+(c) border starts and border following, fused, in two flat phases.
    // Start candidates come from word-wide bit tricks on the LDS image (aruco_trace.hpp: outer = foreground with
    // background W, NW, N, NE; hole = background with foreground W and N); a lane takes one 32-pixel word at a time.
    // Every loop iteration advances each busy lane's walk by ONE step and lets idle lanes take their next candidate,
    // so no lane waits for the longest walk of its wave.
    //   probe phase: walk at most CT_PROBE steps without storing anything; most candidates end here (abandoned as
    //                non-canonical, or closed with fewer than min_len points).  Survivors go to an LDS queue.
    //   long phase:  walk the survivors to the end, writing the points into the lane's private arena of the frame's
    //                point pool; a closed canonical border longer than min_len keeps its arena space.
    const int arena = pool_cap / CT_THREADS;
    uint32_t* my_arena = pl + (size_t)(tid & (CT_THREADS - 1)) * arena;
    int wp = 0; // arena words owned by kept borders of this lane
    for (int phase = 0; phase < 2; phase++) {
        TraceState t;
        bool busy = false, drained = false, storing = false;
        uint32_t m_outer = 0, m_hole = 0;
        int wj = 0, wy = 0, qx = 0, qy = 0, ncand_l = 0;
        const int nwords = wpr * H;
        const float inv_wpr = 1.0f / (float)wpr;
        __syncthreads();
        const int nlong = min(s_nlong, lq_cap + candq_cap);
        if (tid == 0) s_next = 0;
        __syncthreads();
        for (;;) {
            if (!busy && !drained) {
                if (phase == 0) {
                    if (!(m_outer | m_hole)) {
                        const int i = atomicAdd(&s_next, 1);
                        if (i >= nwords) drained = true;
                        else {
                            wy = 1 + (int)(((float)i + 0.5f) * inv_wpr); // exact: i < 2^20, quotient >= 0.5/wpr off an integer
                            wj = i - (wy - 1) * wpr;
                            const uint32_t* row = lbits + wy * wpr;
                            const uint32_t* up = row - wpr;
                            const uint32_t cur = row[wj], upw = up[wj];
                            const uint32_t cur_l = (cur << 1) | (wj ? row[wj - 1] >> 31 : 0u);
                            const uint32_t up_l = (upw << 1) | (wj ? up[wj - 1] >> 31 : 0u);
                            const uint32_t up_r = (upw >> 1) | (wj + 1 < wpr ? up[wj + 1] << 31 : 0u);
                            start_candidate_masks(cur, cur_l, upw, up_l, up_r, ORBFE_CAND_FILTER != 0, &m_outer, &m_hole);
                        }
                    }
                    if (m_outer | m_hole) {
                        const int is_hole = m_outer ? 0 : 1;
                        uint32_t& mm = m_outer ? m_outer : m_hole;
                        const int b = __ffs(mm) - 1;
                        mm &= mm - 1;
                        qx = wj * 32 + b;
                        qy = wy;
                        ncand_l++;
                        storing = false;
                        busy = !trace_init(im, t, qx - is_hole, qy, is_hole); // single-pixel borders are never kept
                    }
                } else {
                    const int k = atomicAdd(&s_next, 1);
                    if (k >= nlong) drained = true;
                    else {
                        const uint32_t q = k < lq_cap ? longq[k] : cq[k - lq_cap];
                        qx = q & 0x1fff; qy = (q >> 13) & 0x1fff;
                        const int is_hole = q >> 26;
                        storing = true;
                        busy = !trace_init(im, t, qx - is_hole, qy, is_hole);
                    }
                }
            }
            if (!__any(busy || !drained)) break;
            if (busy) {
                uint32_t pt;
                const int n0 = t.n;
                const int st = trace_step(im, t, &pt);
                if (storing && st >= 0 && wp + n0 < arena) my_arena[wp + n0] = pt;
                if (st != 0) {
                    busy = false;
                    if (st == 1 && t.n > min_len) { // only reachable in the long phase (CT_PROBE < min_len)
                        const int k = atomicAdd(&s_nkept, 1);
                        if (k < kept_cap) {
                            // Every slot that was counted gets a key: a border whose points did not fit the lane's arena is flagged
                            // (the frame is reported as incomplete) and entered with length 0, which the tail skips.  (Until round 5
                            // its slot stayed unwritten and the tail sorted, and then followed, whatever the LDS held there: a
                            // memory fault on frames of dense noise in big-frame mode.)
                            const bool fits = wp + t.n <= arena;
                            if (!fits) atomicOr(&s_flags, 4);
                            // discovery order = raster order of the transition pixel; findContours returns the reverse
                            kkey[k] = ((unsigned long long)(0xffffffffu - (uint32_t)(qy * 65536 + qx)) << 32) |
                                      ((unsigned long long)((fits ? t.n : 0) & 0x7ffff) << 13) | ((unsigned)k << 1) | (unsigned)t.is_hole;
                            off_u[k] = fits ? tid * arena + wp : 0;
                            if (fits) wp += t.n;
                        }
                    }
                } else if (!storing && t.n >= CT_PROBE) {
                    // survivor of the probe: queue it for the long phase (LDS queue first, HBM overflow behind it)
                    busy = false;
                    const int k = atomicAdd(&s_nlong, 1);
                    const uint32_t rec = (uint32_t)qx | ((uint32_t)qy << 13) | ((uint32_t)t.is_hole << 26);
                    if (k < lq_cap) longq[k] = rec;
                    else if (k - lq_cap < candq_cap) cq[k - lq_cap] = rec;
                    else atomicOr(&s_flags, 16);
                }
            }
        }
        if (phase == 0) {
            atomicAdd(&s_ncand, ncand_l);
            CT_STAMP(t2);
            // the probe phase is throughput-bound (one short walk per start candidate), the long phase is bound by the
            // longest border: only CT_THREADS lanes stay, the rest free their wave slots for other kernels
            if (tid >= CT_THREADS) return;
        }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        if (s_nkept > kept_cap) { s_flags |= 2; s_nkept = kept_cap; }
    }
    __syncthreads();
    const int nkept = s_nkept;
    CT_STAMP(t3);
    if (only_flagged && tid == 0) s_ncand |= 1 << 30; // debug: this frame took the fallback (contours_tail syncs first)
    const int pbuf_pts = LDS_BITS ? (lds_bits_words / CT_WAVES) & ~3 : 0; // the LDS bit image is dead: point buffers
    contours_tail(f, tid, CT_THREADS, nkept, kkey, off_u, klen, koff, rectflag, ap_out, ap_stack, pl, kept_out, kept_cap,
                  rects_out, rect_cap, counts, &s_flags, &s_ncand, (uint16_t*)(ap_stack + CT_WAVES * AP_STACK), &s_tailq,
                  CT_WAVES, (uint32_t*)ct_smem, pbuf_pts, nullptr, 0);
#ifdef ORBFE_CT_TIMING
    if (tid == 0) {
        t5 = clock64();
        t4 = t3;
        long long* dbg = (long long*)(kept_out + (size_t)f * kept_cap + kept_cap - 4);
        dbg[0] = t1 - t0; dbg[1] = t2 - t1; dbg[2] = t3 - t2; dbg[3] = t4 - t3; dbg[4] = t5 - t4;
    }
#endif
}

template __global__ void k_contours_t<true>(const uint32_t*, size_t, int, int, int, int, int, uint32_t*, size_t, int,
                                            uint32_t*, size_t, int, ArKept*, int, ArRect*, int, int32_t*, uint32_t*, size_t,
                                            int);
template __global__ void k_contours_t<false>(const uint32_t*, size_t, int, int, int, int, int, uint32_t*, size_t, int,
                                             uint32_t*, size_t, int, ArKept*, int, ArRect*, int, int32_t*, uint32_t*, size_t,
                                             int);

// ---------------------------------------------------------------------------------------- contours, relay -----
// findContours + approxPolyDP by relay segments (aruco_trace.hpp, second half).  One workgroup per frame; every phase
// is parallel over all lanes and no lane ever follows a long border alone:
//   (a) padded bit image -> LDS;  (b) grid markers -> hash table in LDS (the slot is the segment's id);
//   (c) small borders (no grid marker on them) followed whole from their start candidates;
//   (d) one segment per marker: walk to the next marker, remember the smallest start state passed;
//   (e) cyclic lists: pointer doubling for the minimum (= the canonical start), list ranking for the offsets;
//   (f) kept borders (> min_len points) get pool space, their segments are walked again (nothing is staged in (d)) and write
//       their points straight to the final position;  (g) the common tail (sort, approxPolyDP, rectangles).
// If the markers do not fit the table the grid is coarsened, down to no grid at all (then (c) follows every border whole).
__device__ __forceinline__ uint32_t rl_hash(uint32_t key, int tbits) { return (key * 0x9E3779B1u) >> (32 - tbits); }

__device__ __forceinline__ int rl_find_or_insert(uint32_t* hkey, int tbits, uint32_t key, int* fresh)
{
    const uint32_t mask = (1u << tbits) - 1u;
    uint32_t h = rl_hash(key, tbits);
    // a probe sequence this long means the table is (nearly) full: the caller coarsens the grid.  Without the bound every
    // insert into a full table walks all of it with atomics (1280x720 frames: 400x the time of the whole phase).
    for (uint32_t p = 0; p <= min(mask, 255u); p++) {
        const uint32_t old = atomicCAS(&hkey[h], 0u, key);
        if (old == 0u) { *fresh = 1; return (int)h; }
        if (old == key) { *fresh = 0; return (int)h; }
        h = (h + 1) & mask;
    }
    return -1;
}

__device__ __forceinline__ int rl_find(const uint32_t* hkey, int tbits, uint32_t key)
{
    const uint32_t mask = (1u << tbits) - 1u;
    uint32_t h = rl_hash(key, tbits);
    for (uint32_t p = 0; p <= mask; p++) {
        const uint32_t v = hkey[h];
        if (v == key) return (int)h;
        if (v == 0u) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

// One walk step as a table lookup: entry [(ring << 3) | s] of a 2048-entry LDS table built from the functions of
// aruco_trace.hpp.  Bits: 0-2 direction of the next border pixel; 3 run has W or E; 4 run has N or S;
// 5-6 start class; 7-10 run has N, W, E, S; 11-12 dx + 1; 13-14 dy + 1.
__device__ __forceinline__ uint16_t rl_lut_entry(unsigned ring, int st)
{
    if (!ring) return 0;
    unsigned run;
    const int d = relay_examine(ring, st, &run);
    unsigned e = (unsigned)d;
    if (run & 0x11u) e |= 8u;
    if (run & 0x44u) e |= 16u;
    e |= (unsigned)relay_start_class(ring, run) << 5;
    e |= ((run >> 2) & 1u) << 7;
    e |= ((run >> 4) & 1u) << 8;
    e |= (run & 1u) << 9;
    e |= ((run >> 6) & 1u) << 10;
    e |= (unsigned)(dir_dx(d) + 1) << 11;
    e |= (unsigned)(dir_dy(d) + 1) << 13;
    return (uint16_t)e;
}
// is the state a grid marker: its run has W/E on a relay row, or N/S on a relay column
__device__ __forceinline__ bool rl_is_marker(unsigned e, int x, int y, int kmask)
{
    const unsigned g = ((y & kmask) == 0 ? 8u : 0u) | ((x & kmask) == 0 ? 16u : 0u);
    return (e & g) != 0;
}
__device__ __forceinline__ void rl_advance(const BitImage& im, RelayWalk& w, unsigned e)
{
    w.x += (int)((e >> 11) & 3u) - 1;
    w.y += (int)((e >> 13) & 3u) - 1;
    w.s = (int)((e + 4u) & 7u);
    w.n++;
    w.ring = ring8(im, w.x, w.y);
}

// One speck pass (aruco_trace.hpp "FEWER WALKS" (2)) on a frame's padded bit image where the relay kernel holds it anyway -- in LDS --
// instead of as a launch of its own between threshold and contours (k_speck_clean), which cost the pipeline more than the shorter
// walks gave back.  Same function of the image as k_speck_clean and the CPU twin.  The rim masks and anchors of the whole frame go
// through scratch in HBM (it stays in L2; every array is written once and read afterwards, by other waves of the workgroup behind a
// barrier), so the image can be cleared in place: nothing reads P between the sweep that computes the anchors and the one that clears.
// scr: 3 arrays of speck_frame_rows(H, HH) x wpr words; row index = padded row + HH + 1.
template <int WW, int HH, int NT>
__device__ __forceinline__ void speck_pass_frame(uint32_t* P, int wpr, int prow, uint32_t* __restrict__ scr, int tid)
{
    const int nr = prow + 2 * (HH + 1), n = nr * wpr;
    uint32_t* Fm = scr;
    uint32_t* Sm = scr + n;
    uint32_t* An = scr + 2 * n;
    const float inv_wpr = 1.0f / (float)wpr;
    for (int i = tid; i < n; i += NT) {
        const int rr = (int)(((float)i + 0.5f) * inv_wpr), j = i - __mul24(rr, wpr), r = rr - (HH + 1);   // exact: i < 2^20
        uint32_t f = 0, sd = 0;
        if (r >= 0 && r < prow) {
            const uint32_t* row = P + __mul24(r, wpr);
            speck_row_masks<WW>(row[j], j + 1 < wpr ? row[j + 1] : 0u, &f, &sd);
        }
        Fm[i] = f; Sm[i] = sd;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const int rr = (int)(((float)i + 0.5f) * inv_wpr);
        uint32_t a = 0;
        if (rr + HH + 1 < nr) {
            uint32_t occ = Fm[i] | Fm[i + (HH + 1) * wpr];
#pragma unroll
            for (int k = 1; k <= HH; k++) occ |= Sm[i + k * wpr];
            a = ~occ;
        }
        An[i] = a;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < prow * wpr; i += NT) {
        const int r = (int)(((float)i + 0.5f) * inv_wpr), j = i - __mul24(r, wpr);
        const uint32_t* a = An + __mul24(r + HH + 1, wpr) + j;
        uint32_t e = 0, el = 0;
#pragma unroll
        for (int dy = 1; dy <= HH; dy++) {
            e |= a[-dy * wpr];
            if (j) el |= a[-dy * wpr - 1];
        }
        P[i] &= ~speck_dilate<WW>(e, el);
    }
    __syncthreads();
}

#ifndef RL_FETCH_GATE
#define RL_FETCH_GATE 1
#endif
#ifndef RL_FETCH_TRIES
#define RL_FETCH_TRIES 1   // experiment (build parameter): refill attempts of phase (c) per loop trip
#endif
// RL_FETCH_GATE > 1 (experiment, build parameter): a walk loop refills its idle lanes only when at least that many lanes of the wave
// are idle (or none is walking), so that the refill block is issued for many lanes at a time instead of a few on every trip
#if RL_FETCH_GATE > 1
#define RL_GATE_OPEN(busy, drained) (__popcll(__ballot(!(busy) && !(drained))) >= RL_FETCH_GATE || !__any(busy))
#else
#define RL_GATE_OPEN(busy, drained) true
#endif
// returns 1 if the frame has to be done again without a grid (force_nogrid), else 0
#if RL_THREADS == 1024
#define RL_VGPR_ATTR __attribute__((amdgpu_num_vgpr(64))) // two workgroups (32 waves) share a CU
#else
#define RL_VGPR_ATTR
#endif
template <bool force_nogrid, int RL_SLOTS, int RL_NT, bool GBITS = false>
__device__ __forceinline__ int relay_frame(
    const uint32_t* __restrict__ gbits, size_t bits_fstride, int wpr_g, int W, int H, int lds_bits_words, int min_len,
    int kshift, int tbits, RelaySeg* __restrict__ segs, uint32_t* __restrict__ pool, size_t pool_fstride, int pool_cap,
    ArKept* __restrict__ kept_out, int kept_cap, int kcap /*kept borders this kernel's LDS holds*/,
    unsigned long long* __restrict__ tail_keys, int32_t* __restrict__ tail_off, int32_t* __restrict__ counts, int32_t* __restrict__ hint, uint4* __restrict__ small_g,
    int32_t* __restrict__ rstate /* per frame: grid shift the frame was done with, final pool words in use (k_contours_small goes on from there) */,
    uint32_t* __restrict__ gpad = nullptr /* GBITS: the padded bit image lives here (HBM / L2) instead of LDS */, size_t gpad_fstride = 0,
    int small_elsewhere = 0 /* phase (c) of gridded frames is k_contours_small's */,
    const uint16_t* __restrict__ lut_g = nullptr /* the step table, built once per detector (k_relay_lut) */,
    int f0 = 0 /* first frame of this launch (a batch may be launched in chunks) */,
    uint32_t* __restrict__ candq_g = nullptr /* per frame: the start candidates of phase (c), listed before the walks (see (c)) */,
    size_t candq_fstride = 0)
{
    extern __shared__ __align__(16) unsigned char ct_smem[];
    __shared__ int s_nmpix, s_qn;
    __shared__ int s_next, s_next_d, s_nkept, s_flags, s_ncand, s_nmark, s_nsmall, s_pool, s_changed[2];
    // grid spacing that fitted the marker table in the previous batch of this handle (video: it will fit again); saves the
    // enumeration passes that overflow at finer spacings.  The result does not depend on the spacing.
    if (!force_nogrid && hint) kshift = max(kshift, min(*hint, 7));
    __shared__ uint16_t s_lut[2048];
    __shared__ unsigned s_tailq;
    const int tid = threadIdx.x, f = blockIdx.x + f0, NT = RL_NT;
    const int wpr = (W + 2 + 31) >> 5, prow = H + 2;
    const int T = 1 << tbits;
    int kmask = (1 << kshift) - 1, K = 1 << kshift;
    // LDS: one region R seen three ways, then the marker keys (so that the tail's point buffers can run on into them):
    //   walks (a-d):  R = the padded bit image
    //   lists (e-f):  R = kept keys u64, kept pool offsets, cmin/val, jmp, arg      (the bit image is dead)
    //   tail  (g):    R = kept keys, offsets | klen, koff, rectflag, approx scratch, length ranks, point buffers ...
    //   [hkey: T marker state keys] lives until (f2)
    uint32_t* lbits = GBITS ? gpad + (size_t)(blockIdx.x + f0) * gpad_fstride : (uint32_t*)ct_smem;
    unsigned long long* kkey = (unsigned long long*)ct_smem;
    int* off_u = (int*)(kkey + kcap);
    unsigned char* uni = (unsigned char*)(off_u + kcap);
    uint32_t* cmin = (uint32_t*)uni; // smallest start state of the segment / of the border; later the ranking value
    uint16_t* jmp = (uint16_t*)(cmin + T);
    uint16_t* arg = jmp + T;   // slot of the segment that holds the border's smallest start state
    const size_t r_bytes = relay_region_bytes(lds_bits_words, kcap, tbits);
    uint32_t* hkey = (uint32_t*)(ct_smem + r_bytes);
    // whole borders kept by phase (c): pool offset, length, -, discovery key * 2 + is_hole (kcap entries).  Rarely touched, so they
    // live in HBM: their 16 KB of LDS are what lets a 1280 x 720 frame have an 8192-slot marker table.
    uint4* s_small = small_g + (size_t)f * kcap;
    int* klen = (int*)uni;
    int* koff = klen + kcap;
    int* rectflag = koff + kcap;
    ApPt* ap_out = (ApPt*)(rectflag + kcap);
    int2* ap_stack = (int2*)(ap_out + (RL_NT / 64) * AP_OUT);
    const uint32_t* gb = gbits + (size_t)f * bits_fstride;
    uint32_t* pl = pool + (size_t)f * pool_fstride;
    RelaySeg* sg = segs + ((size_t)f << tbits);

#ifdef ORBFE_CT_TIMING
    long long tq[8];
    int tqi = 0;
    __shared__ unsigned s_dbg_steps[6]; // phase (c): steps of closed / marker-stopped / non-canonical walks; max and sum of a wave's loop iterations; longest closed border
    if (threadIdx.x < 6) s_dbg_steps[threadIdx.x] = 0;
#define RL_STAMP() tq[tqi++] = clock64()
#else
#define RL_STAMP()
#endif
    // A frame whose segments overflow the staging arenas or the copy list (one giant noisy component) is done again
    // from (a) without a grid: every border is then followed whole, straight into the pool.
    if (tid == 0) {
        s_next = 0; s_next_d = 0; s_nkept = 0; s_flags = 0; s_ncand = 0; s_nmark = 0; s_nsmall = 0; s_pool = 0; s_qn = 0;
        s_changed[0] = 0; s_changed[1] = 0;
    }
    RL_STAMP();
    // ---- (a) padded bit image into LDS: pixel (x, y) -> bit x+1 of row y+1
    const float inv_wpr_a = 1.0f / (float)wpr;
    // four words a thread and trip, their eight loads in flight together and none under a predicate (clamped addresses, the value
    // masked afterwards): a load behind a condition is a branch with a full wait, and the image was twenty serial round trips --
    // 22 k clocks of a frame's 560 k, twice (here and again in (f2))
    auto load_padded_bits = [&]() {
        const int nw = wpr * prow;
        for (int i0 = 0; i0 < nw; i0 += 4 * NT) {
            uint32_t cur[4], prv[4];
            bool in[4], hasp[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = min(i0 + k * NT + tid, nw - 1);
                const int py = (int)(((float)i + 0.5f) * inv_wpr_a), j = i - __mul24(py, wpr); // exact: i < 2^20 (an integer division by a runtime value is ~25 instructions)
                const uint32_t* row = gb + (uint32_t)__mul24(min(max(py - 1, 0), H - 1), wpr_g);
                in[k] = py >= 1 && py <= H && j < wpr_g;
                hasp[k] = py >= 1 && py <= H && j >= 1 && j - 1 < wpr_g;
                cur[k] = row[min(j, wpr_g - 1)];
                prv[k] = row[min(max(j - 1, 0), wpr_g - 1)];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k * NT + tid;
                if (i < nw) lbits[i] = ((in[k] ? cur[k] : 0u) << 1) | ((hasp[k] ? prv[k] : 0u) >> 31);
            }
        }
        if (tid < 2) lbits[nw + tid] = 0; // spare words read by ring8()'s funnel loads
    };
    load_padded_bits();
    for (int i = tid; i < T; i += NT) hkey[i] = 0u;
    if (lut_g) for (int i = tid; i < 1024; i += NT) reinterpret_cast<uint32_t*>(s_lut)[i] = reinterpret_cast<const uint32_t*>(lut_g)[i];
    else for (int i = tid; i < 2048; i += NT) s_lut[i] = rl_lut_entry((unsigned)i >> 3, i & 7);
    if (GBITS) __threadfence_block(); // the padded image was written to HBM: visible to the workgroup's other waves
    __syncthreads();
    const BitImage im{lbits, wpr, W, H};
    if (!GBITS && (small_elsewhere & 2) && candq_g) {   // the speck passes on the image in LDS (see speck_pass_frame)
        uint32_t* scr = candq_g + (size_t)f * candq_fstride + relay_queue_words(W, H);
        speck_pass_frame<ORBFE_SPECK_W1, ORBFE_SPECK_H1, RL_NT>(lbits, wpr, prow, scr, tid);
        speck_pass_frame<ORBFE_SPECK_W2, ORBFE_SPECK_H2, RL_NT>(lbits, wpr, prow, scr + 3 * (size_t)wpr * speck_frame_rows(H, ORBFE_SPECK_H1), tid);
    }
    small_elsewhere &= 1;
    RL_STAMP();

    // ---- (b) grid markers: relay rows word by word, relay columns in chunks of 32 rows.  If they do not fit the table
    // the grid is coarsened (K doubles) and the enumeration repeated.
    if (force_nogrid) { kshift = 30; kmask = (1 << kshift) - 1; K = 1 << kshift; }
    else for (;;) {
    {
        const int nrelrow = H >> kshift, nrelcol = W >> kshift, nchunk = (H + 31) >> 5;
        const int nrow_items = nrelrow * wpr, nitems = nrow_items + nrelcol * nchunk;
        int nm = 0;
        auto add = [&](int x, int y) {
            const unsigned ring = ring8(im, x, y);
            if (!ring) return;
            if (*(volatile int*)&s_flags & RL_FLAG_TABLE) return; // the table overflowed already: this pass is void
            relay_states_of_pixel(ring, grid_active(ring, x, y, kmask), [&](int st) {
                int fresh = 0;
                if (rl_find_or_insert(hkey, tbits, relay_key(x, y, st), &fresh) < 0) atomicOr(&s_flags, RL_FLAG_TABLE);
                nm += fresh;
            });
        };
        // two passes: the pixels that carry markers are listed first (a textured word has sixteen of them, most have none: with one
        // lane doing a word's inserts the phase waited for the unluckiest lane), then every lane inserts the states of listed pixels.
        // The list lives in the segment records' HBM (unused until (d)): 5 T words.
        uint32_t* mlist = reinterpret_cast<uint32_t*>(sg);
        const int mcap = 5 * T;
        if (tid == 0) s_nmpix = 0;
        __syncthreads();
        auto push = [&](int x, int y) {
            const int q = atomicAdd(&s_nmpix, 1);
            if (q < mcap) mlist[q] = (uint32_t)x | ((uint32_t)y << 16);
        };
        for (int it = tid; it < nitems; it += NT) {
            if (it < nrow_items) {
                const int r = (int)(((float)it + 0.5f) * inv_wpr_a), j = it - __mul24(r, wpr), y = (r + 1) << kshift;
                const uint32_t* row = lbits + __mul24(y, wpr);
                const uint32_t cur = row[j];
                if (!cur) continue;
                const uint32_t cur_l = (cur << 1) | (j ? row[j - 1] >> 31 : 0u);
                const uint32_t cur_r = (cur >> 1) | (j + 1 < wpr ? row[j + 1] << 31 : 0u);
                uint32_t colmask = 0;
                for (int b = (-(j << 5)) & kmask; b < 32; b += K) colmask |= 1u << b;
                uint32_t m = (cur & (~cur_l | ~cur_r)) | (cur & colmask & (~row[j - wpr] | ~row[j + wpr]));
                while (m) {
                    const int b = __ffs(m) - 1;
                    m &= m - 1;
                    push(j * 32 + b, y);
                }
            } else {
                const int c = (it - nrow_items) / nchunk, ch = (it - nrow_items) - c * nchunk;
                const int x = (c + 1) << kshift, j = x >> 5, b = x & 31, y0 = 1 + (ch << 5);
                unsigned long long col = 0; // bit i = pixel (x, y0 - 1 + i)
                for (int i = 0; i < 34; i++) {
                    const int y = y0 - 1 + i;
                    if (y <= H + 1) col |= (unsigned long long)((lbits[__mul24(y, wpr) + j] >> b) & 1u) << i;
                }
                uint32_t m = (uint32_t)(col >> 1) & (~(uint32_t)col | ~(uint32_t)(col >> 2));
                while (m) {
                    const int i = __ffs(m) - 1;
                    m &= m - 1;
                    const int y = y0 + i;
                    if (y <= H && (y & kmask) != 0) push(x, y); // pixels on relay rows were taken above
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        if (s_nmpix > mcap && tid == 0) atomicOr(&s_flags, RL_FLAG_TABLE); // cannot fit the table either
        const int npx = min(s_nmpix, mcap);
        for (int i0 = 0; i0 < npx; i0 += 4 * NT) {   // (four list entries a thread in flight: the list is in HBM)
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = mlist[min(i0 + k * NT + tid, npx - 1)];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i0 + k * NT + tid < npx) add((int)(v[k] & 0xffffu), (int)(v[k] >> 16));
        }
        if (nm) atomicAdd(&s_nmark, nm);
    }
    __syncthreads();
    const bool full = (s_flags & RL_FLAG_TABLE) || s_nmark > T - (T >> 3);
    if (!full) break;
    __syncthreads();
    for (int i = tid; i < T; i += NT) hkey[i] = 0u;
    if (tid == 0) { s_flags &= ~RL_FLAG_TABLE; s_nmark = 0; }
    // even a 128-pixel grid has too many markers (noise): no grid at all -- every border is then "small" and phase (c)
    // follows it whole from its start candidate, which is the single-walker formulation of k_contours_t
    const bool nogrid = kshift >= 7;
    kshift = nogrid ? 30 : kshift + 1;
    kmask = (1 << kshift) - 1; K = 1 << kshift;
    __syncthreads();
    if (nogrid) break;
    }
    if (!force_nogrid && hint && f == 0 && tid == 0)
        *hint = kshift >= 30 ? 7 : (s_nmark < (T >> 2) && kshift > 5) ? kshift - 1 : kshift;
    RL_STAMP();

    // the part of the frame's pool the kept borders may fill (the rest was the staging arenas of rounds 1 - 2; the single-walker
    // kernel still uses it that way)
    const int stage0 = pool_cap >> 2;
    // ---- (c) small borders.  With a grid they are k_contours_small's (the next launch: one small workgroup per band of K rows,
    // because such walks never leave their grid cell -- the chip is full instead of one workgroup per frame waiting on LDS round
    // trips).  Without a grid every border is "small" and is followed whole here: a lane takes one 32-pixel word of start
    // candidates at a time; every loop iteration advances each busy lane by ONE step.  A walk stops at a proof that
    // the candidate is not canonical, or when the border closes; a closed border longer than min_len is queued.
    // The candidates are LISTED first (round 5): every lane takes words of the frame, the run tests of aruco_trace.hpp decide on whole
    // words, and what is left goes to the frame's queue in HBM (it stays in L2).  With the speck passes in front of this kernel three
    // words in four have no candidate at all; drawn one word per loop trip, as until round 4, the empty words were 40 % of a wave's
    // trips, each with the whole loop body behind it.  A walking lane fetches its next queue entry one trip ahead of needing it, so the
    // L2 round trip is hidden behind a trip of walking.  A frame with more candidates than the queue holds (noise) draws words as before.
    auto phase_c = [&]() {
    if (kshift >= 30 || !small_elsewhere) {
        RelayWalk wk;
        bool busy = false, drained = false;
        uint32_t m_outer = 0, m_hole = 0;
        int wj = 0, wy = 0, sx = 0, sy = 0, s0 = 0, is_hole = 0, start_key = 0, ncand_l = 0;
        const int nwords = wpr * H;
        const float inv_wpr = 1.0f / (float)wpr;
        uint32_t* cq = candq_g ? candq_g + (size_t)f * candq_fstride : nullptr;
        const int qcap = relay_queue_words(W, H);
        if (cq) {
            for (int i = tid; i < nwords; i += NT) {
                const int y = 1 + (int)(((float)i + 0.5f) * inv_wpr), j = i - __mul24(y - 1, wpr); // exact: i < 2^20
                const uint32_t* row = lbits + __mul24(y, wpr);
                const uint32_t* up = row - wpr;
                const uint32_t cur = row[j], upw = up[j];
                const uint32_t cur_l = (cur << 1) | (j ? row[j - 1] >> 31 : 0u);
                const uint32_t up_l = (upw << 1) | (j ? up[j - 1] >> 31 : 0u);
                const uint32_t up_r = (upw >> 1) | (j + 1 < wpr ? up[j + 1] << 31 : 0u);
                uint32_t mo, mh;
                start_candidate_masks(cur, cur_l, upw, up_l, up_r, ORBFE_CAND_FILTER != 0, &mo, &mh);
                const int c = __popc(mo) + __popc(mh);
                if (c) {
                    int q = atomicAdd(&s_qn, c);
                    if (q + c <= qcap) {
                        const uint32_t hi = ((uint32_t)y << 13) | ((uint32_t)j << 5);   // x | y << 13 | hole << 26 (x, y < 8192)
                        while (mo) { const int b = __ffs(mo) - 1; mo &= mo - 1; cq[q++] = hi | (uint32_t)b; }
                        while (mh) { const int b = __ffs(mh) - 1; mh &= mh - 1; cq[q++] = hi | (uint32_t)b | (1u << 26); }
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        const int qn = cq ? s_qn : 0;
        const bool listed = cq != nullptr && qn <= qcap;
        uint32_t nxt = 0;
        bool have = false;
#ifdef ORBFE_CT_TIMING
        unsigned dbg_iters = 0;
#endif
        for (;;) {
#ifdef ORBFE_CT_TIMING
            dbg_iters++;
#endif
            if (listed) {
                if (!busy && have) {
                    have = false;
                    const int qx = (int)(nxt & 0x1fffu);
                    wy = (int)((nxt >> 13) & 0x1fffu); is_hole = (int)(nxt >> 26);
                    sx = qx - is_hole; sy = wy;
                    start_key = wy * 65536 + qx;
                    wk.x = sx; wk.y = sy; wk.n = 0;
                    wk.ring = ring8(im, sx, sy);
                    s0 = relay_start_dir(wk.ring, is_hole);
                    wk.s = s0;
                    busy = s0 >= 0; // single-pixel borders are never kept
                }
                if (!have && !drained) {
                    const int i = atomicAdd(&s_next, 1);
                    if (i >= qn) drained = true;
                    else { nxt = cq[i]; have = true; }
                }
            } else
            for (int tries_ = 0; tries_ < RL_FETCH_TRIES; tries_++)   // (a lane that drew an empty word, or a one-pixel border, tries again at once)
            if (!busy && !drained && RL_GATE_OPEN(busy, drained)) {
                if (!(m_outer | m_hole)) {
                    const int i = atomicAdd(&s_next, 1);
                    if (i >= nwords) drained = true;
                    else {
                        wy = 1 + (int)(((float)i + 0.5f) * inv_wpr); // exact: i < 2^20
                        wj = i - __mul24(wy - 1, wpr);
                        const uint32_t* row = lbits + __mul24(wy, wpr);
                        const uint32_t* up = row - wpr;
                        const uint32_t cur = row[wj], upw = up[wj];
                        const uint32_t cur_l = (cur << 1) | (wj ? row[wj - 1] >> 31 : 0u);
                        const uint32_t up_l = (upw << 1) | (wj ? up[wj - 1] >> 31 : 0u);
                        const uint32_t up_r = (upw >> 1) | (wj + 1 < wpr ? up[wj + 1] << 31 : 0u);
                        start_candidate_masks(cur, cur_l, upw, up_l, up_r, ORBFE_CAND_FILTER != 0, &m_outer, &m_hole);
                    }
                }
                if (m_outer | m_hole) {
                    is_hole = m_outer ? 0 : 1;
                    uint32_t& mm = m_outer ? m_outer : m_hole;
                    const int b = __ffs(mm) - 1;
                    mm &= mm - 1;
                    const int qx = wj * 32 + b;
                    ncand_l++;
                    sx = qx - is_hole; sy = wy;
                    start_key = wy * 65536 + qx;
                    wk.x = sx; wk.y = sy; wk.n = 0;
                    wk.ring = ring8(im, sx, sy);
                    s0 = relay_start_dir(wk.ring, is_hole);
                    wk.s = s0;
                    busy = s0 >= 0; // single-pixel borders are never kept
                }
            }
            if (!__any(busy || have || !drained)) break;
#pragma unroll
            for (int u = 0; u < RL_STEPS_PER_ITER; u++) {
                if (busy) {
                    const unsigned e = s_lut[(wk.ring << 3) | (unsigned)wk.s];
                    const int key3 = wk.y * 65536 + wk.x;
                    // a walk that meets a grid marker: the border belongs to the segment walkers (with the candidate bits of (d)
                    // no such walk is started, but a marker test costs three instructions and keeps the two modes one code path)
                    bool stop = rl_is_marker(e, wk.x, wk.y, kmask);
                    if (is_hole)                                      // relay_not_canonical() on the table's run bits
                        stop |= ((e & 0x080u) && key3 - 65536 < start_key) || ((e & 0x100u) && key3 - 1 < start_key) ||
                                ((e & 0x200u) && key3 + 1 < start_key) || ((e & 0x400u) && key3 + 65536 < start_key);
                    else stop |= key3 < start_key;
#ifdef ORBFE_CT_TIMING
                    if (stop) atomicAdd(&s_dbg_steps[rl_is_marker(e, wk.x, wk.y, kmask) ? 1 : 2], (unsigned)wk.n + 1);
#endif
                    if (stop) busy = false;
                    else {
                        rl_advance(im, wk, e);
                        if (wk.x == sx && wk.y == sy && wk.s == s0) {
                            busy = false;
#ifdef ORBFE_CT_TIMING
                            atomicAdd(&s_dbg_steps[0], (unsigned)wk.n);
                            atomicMax(&s_dbg_steps[5], (unsigned)wk.n);
#endif
                            if (wk.n > min_len) {
                                // rare with a grid (> min_len points between grid lines), every kept border without one: the
                                // border is whole, so its final place is known -- walk it once more, straight into the pool
                                const int q = atomicAdd(&s_nsmall, 1), n = wk.n;
                                const int base = atomicAdd(&s_pool, n);
                                if (q >= kcap) atomicOr(&s_flags, 2);
                                else if (base + n > stage0) atomicOr(&s_flags, 4);
                                else {
                                    RelayWalk w2;
                                    relay_walk_from_key(im, w2, relay_key(sx, sy, s0));
                                    for (int o = 0; o < n; o++) {
                                        pl[base + o] = relay_point(w2);
                                        rl_advance(im, w2, s_lut[(w2.ring << 3) | (unsigned)w2.s]);
                                    }
                                    s_small[q] = make_uint4((uint32_t)base, (uint32_t)n, 0u, (uint32_t)start_key * 2u + (uint32_t)is_hole);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (listed) { if (tid == 0) s_ncand = qn; }
        else atomicAdd(&s_ncand, ncand_l);
#ifdef ORBFE_CT_TIMING
        if ((tid & 63) == 0) { atomicMax(&s_dbg_steps[3], dbg_iters); atomicAdd(&s_dbg_steps[4], dbg_iters); }
#endif
    }
    };
    RL_STAMP();

    // ---- (d) segments: table slot -> walk to the next grid marker.  The points go to the lane's staging arena (upper
    // part of the frame's pool); (f2) copies the segments of kept borders to their final place.
    auto phase_d = [&]() {
    {
        RelayWalk wk;
        bool busy = false, drained = false;
        int slot = 0, mnoff = 0;
        uint32_t mn = 0xffffffffu, mnhole = 0;
        for (;;) {
            if (!busy && !drained && RL_GATE_OPEN(busy, drained)) {
                const int i = atomicAdd(&s_next_d, 1);
                if (i >= T) drained = true;
                else {
                    const uint32_t key = hkey[i];
                    if (key) {
                        slot = i;
                        relay_walk_from_key(im, wk, key);
                        mn = 0xffffffffu; mnoff = 0; mnhole = 0;
                        busy = true;
                    }
                }
            }
            if (!__any(busy || !drained)) break;
#pragma unroll
            for (int u = 0; u < RL_STEPS_PER_ITER; u++) {
                if (busy) {
                    const unsigned e = s_lut[(wk.ring << 3) | (unsigned)wk.s];
                    if (wk.n > 0 && rl_is_marker(e, wk.x, wk.y, kmask)) {
                        int nx = rl_find(hkey, tbits, relay_key(wk.x, wk.y, wk.s));
                        if (nx < 0) { atomicOr(&s_flags, RL_FLAG_BUG); nx = slot; }
                        RelaySeg r;
                        r.nxt = (uint32_t)nx; r.len = (uint32_t)wk.n; r.minoff = (uint32_t)mnoff | (mnhole << 31);
                        r.stg = 0u; // (no staging since round 3: kept segments are walked again in (f2))
                        r.mn = mn;
                        sg[slot] = r;
                        busy = false;
                    } else {
                        if (e & 0x60u) {
                            const uint32_t k = relay_key(wk.x, wk.y, wk.s);
                            const uint32_t hole = ((e >> 5) & 3u) == 2u ? 1u : 0u;
                            if (k < mn) { mn = k; mnoff = wk.n; mnhole = hole; }
                        }
                        rl_advance(im, wk, e);
                    }
                }
            }
        }
    }
    };
    // (c) then (d) with no barrier between them: (d) depends on nothing (c) writes and has its own ticket counter; a wave that has
    // drained the small borders goes straight on to the segments.  (Round 3 tried (d) first, its walkers marking the start candidates
    // they pass in an HBM plane so that (c) need not walk from them: a quarter fewer candidates, and slower -- 553 against 399 us
    // alone, profiles/r03_contour_candidate_plane.txt -- because (c)'s time was set by its slowest wave and its word fetches, not by
    // the number of walks.  The experiment's code went in round 5.)
    phase_c();
    phase_d();
    __syncthreads();
    if ((s_flags & 4) && kshift < 30) return 1; // staging arena full: again without a grid
    if (s_flags) { // capacity exceeded or an invariant broken: the frame is reported as failed (flags != 0)
        if (tid == 0) { counts[f * 4 + 0] = 0; counts[f * 4 + 1] = 0; counts[f * 4 + 2] = s_flags; counts[f * 4 + 3] = 0; }
        return 0;
    }
    RL_STAMP();

    // the bit image is dead: its space now holds the list arrays, loaded from the segment records
    {
        // (the records' fields as unconditional loads of every slot of the thread, all in flight together: under `if (hkey[i])` each
        //  was a branch with a wait for HBM behind it)
        uint32_t r_mn[RL_SLOTS], r_nxt[RL_SLOTS];
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {
            const int i = min(tid + q * NT, T - 1);
            r_mn[q] = sg[i].mn; r_nxt[q] = sg[i].nxt;
        }
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {
            const int i = tid + q * NT;
            if (i < T && hkey[i]) { cmin[i] = r_mn[q]; jmp[i] = (uint16_t)r_nxt[q]; arg[i] = (uint16_t)i; }
        }
    }
    __syncthreads();
    // ---- (e1) the border's smallest start state, by pointer doubling round the cyclic list.  When a round changes
    // nothing, every window already covers its cycle (windows double; "no change" makes the minima periodic).
    for (int round = 0; round < 24; round++) {
        uint32_t m_[RL_SLOTS];
        uint16_t a_[RL_SLOTS], j_[RL_SLOTS];
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {
            const int i = tid + q * NT;
            if (i < T && hkey[i]) {
                const int j = jmp[i];
                m_[q] = cmin[j]; a_[q] = arg[j]; j_[q] = jmp[j];
            }
        }
        __syncthreads();
        if (tid == 0) s_changed[(round + 1) & 1] = 0;
        bool ch = false;
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {
            const int i = tid + q * NT;
            if (i < T && hkey[i]) {
                if (m_[q] < cmin[i]) { cmin[i] = m_[q]; arg[i] = a_[q]; ch = true; }
                jmp[i] = j_[q];
            }
        }
        if (ch) s_changed[round & 1] = 1;
        __syncthreads();
        if (!s_changed[round & 1]) break;
    }
    // ---- (e2) list ranking: cut every cycle in front of the segment that holds the canonical start; val = points
    // from the segment to the end of the list
    uint32_t* val = cmin;
    uint32_t canon_[RL_SLOTS]; // (head segments) the border's canonical start state
#pragma unroll
    for (int q = 0; q < RL_SLOTS; q++) {
        const int i = tid + q * NT;
        canon_[q] = (i < T && hkey[i]) ? cmin[i] : 0xffffffffu;
    }
    {
        uint32_t l_[RL_SLOTS];
        uint16_t n_[RL_SLOTS];
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {
            const int i = tid + q * NT;
            if (i < T && hkey[i]) {
                const RelaySeg r = sg[i];
                l_[q] = r.len;
                n_[q] = (arg[r.nxt] == r.nxt) ? (uint16_t)RL_NIL : (uint16_t)r.nxt;
            }
        }
        __syncthreads(); // all reads of cmin (as minimum) and of arg[nxt] done
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {
            const int i = tid + q * NT;
            if (i < T && hkey[i]) { val[i] = l_[q]; jmp[i] = n_[q]; }
        }
        if (tid == 0) { s_changed[0] = 0; s_changed[1] = 0; }
        __syncthreads();
    }
    for (int round = 0; round < 24; round++) {
        uint32_t v_[RL_SLOTS];
        uint16_t j_[RL_SLOTS];
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {
            const int i = tid + q * NT;
            j_[q] = RL_NIL;
            if (i < T && hkey[i]) {
                const int j = jmp[i];
                if (j != RL_NIL) { v_[q] = val[j]; j_[q] = jmp[j]; }
                else v_[q] = 0xffffffffu; // marks "nothing to do"
            } else v_[q] = 0xffffffffu;
        }
        __syncthreads();
        if (tid == 0) s_changed[(round + 1) & 1] = 0;
        bool ch = false;
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {
            const int i = tid + q * NT;
            if (v_[q] != 0xffffffffu) { val[i] += v_[q]; jmp[i] = j_[q]; ch = true; }
        }
        if (ch) s_changed[round & 1] = 1;
        __syncthreads();
        if (!s_changed[round & 1]) break;
    }
    RL_STAMP();

    // ---- (f1) kept borders: pool space and sort key; jmp[root] = kept index or NIL
#pragma unroll
    for (int q = 0; q < RL_SLOTS; q++) {
        const int i = tid + q * NT;
        if (i < T && hkey[i] && arg[i] == i) {
            const int n = (int)val[i];
            const uint32_t canon = canon_[q];
            uint16_t kk = RL_NIL;
            if (canon == 0xffffffffu) atomicOr(&s_flags, RL_FLAG_BUG); // a border without a start state
            else if (n > min_len) {
                const int k = atomicAdd(&s_nkept, 1);
                const int base = atomicAdd(&s_pool, n);
                if (base + n > stage0) atomicOr(&s_flags, 4);
                else if (k < kcap) {
                    const unsigned hole = sg[i].minoff >> 31; // pattern of the canonical start (this segment holds it)
                    const uint32_t disc = (canon >> 16) * 65536u + ((canon >> 3) & 0x1fffu) + hole;
                    kkey[k] = ((unsigned long long)(0xffffffffu - disc) << 32) | ((unsigned long long)(n & 0x7ffff) << 13) |
                              ((unsigned)k << 1) | hole;
                    off_u[k] = base;
                    kk = (uint16_t)k;
                }
            }
            jmp[i] = kk;
        }
    }
    for (int q = tid; q < min(s_nsmall, kcap); q += NT) { // whole borders of phase (c): already in the pool
        const uint4 e = s_small[q];
        const int n = (int)e.y;
        const int k = atomicAdd(&s_nkept, 1);
        if (k < kcap) {
            kkey[k] = ((unsigned long long)(0xffffffffu - (e.w >> 1)) << 32) | ((unsigned long long)(n & 0x7ffff) << 13) |
                      ((unsigned)k << 1) | (e.w & 1u);
            off_u[k] = (int)e.x;
        }
    }
    __syncthreads();
    if (s_nkept > kcap) { // more kept borders than this kernel's arrays hold: capacity error
        if (tid == 0) { counts[f * 4 + 0] = 0; counts[f * 4 + 1] = 0; counts[f * 4 + 2] = s_flags | 2; counts[f * 4 + 3] = 0; }
        return 0;
    }
    // ---- (f2) the points of the kept borders.  Until round 3 every walked point of (d) was staged in a per-lane arena in HBM (57 k
    // uncoalesced 4-byte stores per 640 x 480 frame, 133 MB per 300-frame launch) and the segments of kept borders were copied from
    // there; but only ~60 borders (~9 k points, ~280 segments) per frame are kept.  Now (d) stores nothing: the kept segments are
    // listed (in the marker keys' space, dead after this point), the bit image is loaded into LDS once more (the list arrays that
    // overlaid it are done), and one lane per listed segment walks it AGAIN, straight into its final place.  Segment i starts
    // (n - val[i]) points after the list head; the border starts `minoff` points into the head segment, so everything shifts down
    // by minoff and the head's first points wrap to the end.
    {
        int e_dst[RL_SLOTS], e_len[RL_SLOTS], e_base[RL_SLOTS], e_n[RL_SLOTS];
        uint32_t e_key[RL_SLOTS];
        int mine = 0;
        int e_g[RL_SLOTS], e_k[RL_SLOTS];
        uint32_t r_len[RL_SLOTS], g_moff[RL_SLOTS];
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {   // what the lists say (LDS) ...
            const int i = tid + q * NT;
            e_len[q] = 0; e_g[q] = 0; e_k[q] = RL_NIL;
            if (i < T && hkey[i]) { e_g[q] = arg[i]; e_k[q] = jmp[e_g[q]]; }
        }
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {   // ... the two record fields of every slot (HBM), unconditional and in flight together ...
            r_len[q] = sg[min(tid + q * NT, T - 1)].len;
            g_moff[q] = sg[e_g[q]].minoff;
        }
#pragma unroll
        for (int q = 0; q < RL_SLOTS; q++) {   // ... the entries
            const int i = tid + q * NT;
            if (e_k[q] != RL_NIL) {
                const int g = e_g[q], k = e_k[q];
                const int n = (int)val[g];
                e_base[q] = off_u[k];
                e_dst[q] = off_u[k] + (n - (int)val[i]) - (int)(g_moff[q] & 0x7fffffffu);
                e_len[q] = (int)r_len[q]; e_n[q] = n; e_key[q] = hkey[i];
                mine++;
            }
        }
        // the kept borders (sort key, pool offset) go to k_tail_prep now: their arrays are about to be overwritten by the bit image
        {
            const int nk = s_nkept;
            for (int k = tid; k < nk; k += NT) {
                tail_keys[(size_t)f * kcap + k] = kkey[k];
                tail_off[(size_t)f * kcap + k] = off_u[k];
            }
        }
        if (tid == 0) s_next = 0;
        __syncthreads(); // every read of the marker keys and of the list arrays done
        const int ebase = mine ? atomicAdd(&s_next, mine) : 0;
        if (!GBITS) load_padded_bits(); // (a) again: the padded bit image
        __syncthreads();
        const int E = s_next;
        // the list: key, first destination, length, border base, border length -- five words per entry in the marker keys' space
        const int ECAP = min(NT, T / 5);
        int* c_key = (int*)hkey;
        int* c_dst = c_key + ECAP;
        int* c_len = c_dst + ECAP;
        int* c_base = c_len + ECAP;
        int* c_n = c_base + ECAP;
        for (int r0 = 0; r0 < E; r0 += ECAP) { // frames with more kept segments than the list holds (large images) take several rounds
            const int Er = min(ECAP, E - r0);
            int e0 = ebase - r0;
#pragma unroll
            for (int q = 0; q < RL_SLOTS; q++)
                if (e_len[q] > 0) {
                    if (e0 >= 0 && e0 < ECAP) {
                        c_key[e0] = (int)e_key[q]; c_dst[e0] = e_dst[q]; c_len[e0] = e_len[q]; c_base[e0] = e_base[q]; c_n[e0] = e_n[q];
                    }
                    e0++;
                }
            __syncthreads();
            if (tid < Er) {
                RelayWalk w2;
                relay_walk_from_key(im, w2, (uint32_t)c_key[tid]);
                const int len = c_len[tid], base = c_base[tid], n = c_n[tid];
                int pdst = c_dst[tid];
                for (int o = 0; o < len; o++, pdst++) {
                    pl[pdst < base ? pdst + n : pdst] = relay_point(w2);
                    rl_advance(im, w2, s_lut[(w2.ring << 3) | (unsigned)w2.s]);
                }
            }
            __syncthreads(); // the list is rewritten by the next round
        }
    }
    __threadfence_block();
    __syncthreads();
    RL_STAMP();
    {
        if (tid == 0) {
            counts[f * 4 + 0] = s_nkept; counts[f * 4 + 1] = 0; counts[f * 4 + 2] = s_flags; counts[f * 4 + 3] = s_ncand;
            rstate[f * 2 + 0] = small_elsewhere ? kshift : 30; rstate[f * 2 + 1] = s_pool;
        }
    }
#ifdef ORBFE_CT_TIMING
    if (tid == 0) {
        RL_STAMP();
        long long* dbg = (long long*)(kept_out + (size_t)f * kept_cap + kept_cap - 4);
        for (int i = 0; i < 6; i++) dbg[i] = tq[i + 1] - tq[i];
        dbg[6] = ((long long)s_dbg_steps[3] << 40) | ((long long)s_dbg_steps[4] << 16) | s_dbg_steps[5];
        dbg[7] = s_dbg_steps[0]; dbg[10] = s_dbg_steps[1]; dbg[11] = s_dbg_steps[2];
    }
#endif
    return 0;
}

__global__ __launch_bounds__(RL_THREADS) RL_VGPR_ATTR void k_contours_relay(
    const uint32_t* __restrict__ gbits, size_t bits_fstride, int wpr_g, int W, int H, int lds_bits_words, int min_len,
    int kshift, int tbits, RelaySeg* __restrict__ segs, uint32_t* __restrict__ pool, size_t pool_fstride, int pool_cap,
    ArKept* __restrict__ kept_out, int kept_cap, int kcap /*kept borders this kernel's LDS holds*/,
    unsigned long long* __restrict__ tail_keys, int32_t* __restrict__ tail_off, int32_t* __restrict__ counts, int32_t* __restrict__ hint,
    uint4* __restrict__ small_g, int32_t* __restrict__ rstate, int small_elsewhere, const uint16_t* __restrict__ lut_g, int f0, uint32_t* __restrict__ candq_g, size_t candq_fstride)
{
    __builtin_amdgcn_s_setprio(2); // latency-bound: its few waves go first when a VALU-bound kernel shares the CU
    if (relay_frame<false, RL_SLOTS_PER_THREAD, RL_THREADS>(gbits, bits_fstride, wpr_g, W, H, lds_bits_words, min_len, kshift, tbits, segs, pool,
                                                pool_fstride, pool_cap, kept_out, kept_cap, kcap, tail_keys, tail_off, counts, hint, small_g, rstate,
                                                nullptr, 0, small_elsewhere, lut_g, f0, candq_g, candq_fstride)) {
        __syncthreads();
        relay_frame<true, RL_SLOTS_PER_THREAD, RL_THREADS>(gbits, bits_fstride, wpr_g, W, H, lds_bits_words, min_len, kshift, tbits, segs, pool,
                                               pool_fstride, pool_cap, kept_out, kept_cap, kcap, tail_keys, tail_off, counts, hint, small_g, rstate,
                                               nullptr, 0, small_elsewhere, lut_g, f0, candq_g, candq_fstride);
    }
}

// The same with eight table slots per thread (tbits = 13): large frames, whose workgroup owns a CU anyway (LDS), so the
// register budget of two resident workgroups does not apply.
__global__ __launch_bounds__(RL_THREADS_BIG) void k_contours_relay8(
    const uint32_t* __restrict__ gbits, size_t bits_fstride, int wpr_g, int W, int H, int lds_bits_words, int min_len,
    int kshift, int tbits, RelaySeg* __restrict__ segs, uint32_t* __restrict__ pool, size_t pool_fstride, int pool_cap,
    ArKept* __restrict__ kept_out, int kept_cap, int kcap, unsigned long long* __restrict__ tail_keys, int32_t* __restrict__ tail_off,
    int32_t* __restrict__ counts, int32_t* __restrict__ hint, uint4* __restrict__ small_g, int32_t* __restrict__ rstate, int small_elsewhere,
    const uint16_t* __restrict__ lut_g, int f0, uint32_t* __restrict__ candq_g, size_t candq_fstride)
{
    __builtin_amdgcn_s_setprio(2);
    if (relay_frame<false, 8192 / RL_THREADS_BIG, RL_THREADS_BIG>(gbits, bits_fstride, wpr_g, W, H, lds_bits_words, min_len, kshift, tbits, segs, pool, pool_fstride, pool_cap,
                              kept_out, kept_cap, kcap, tail_keys, tail_off, counts, hint, small_g, rstate, nullptr, 0, small_elsewhere, lut_g, f0, candq_g, candq_fstride)) {
        __syncthreads();
        relay_frame<true, 8192 / RL_THREADS_BIG, RL_THREADS_BIG>(gbits, bits_fstride, wpr_g, W, H, lds_bits_words, min_len, kshift, tbits, segs, pool, pool_fstride, pool_cap,
                             kept_out, kept_cap, kcap, tail_keys, tail_off, counts, hint, small_g, rstate, nullptr, 0, small_elsewhere, lut_g, f0, candq_g, candq_fstride);
    }
}

// Small batches (a single frame of the drop-in path, up to 32): twice the waves per frame on the same 4096-slot table.  On an empty
// chip the walks are bound by their dependent LDS round trips, and sixteen waves hide them better (phase clocks of one 640 x 480
// frame: 833 k -> 572 k cycles); a full batch is bound by instruction issue over all its workgroups, where the 8-wave version
// wins (387 against 446 us per 300 frames).
__global__ __launch_bounds__(RL_THREADS_BIG) void k_contours_relay_wide(
    const uint32_t* __restrict__ gbits, size_t bits_fstride, int wpr_g, int W, int H, int lds_bits_words, int min_len,
    int kshift, int tbits, RelaySeg* __restrict__ segs, uint32_t* __restrict__ pool, size_t pool_fstride, int pool_cap,
    ArKept* __restrict__ kept_out, int kept_cap, int kcap, unsigned long long* __restrict__ tail_keys, int32_t* __restrict__ tail_off,
    int32_t* __restrict__ counts, int32_t* __restrict__ hint, uint4* __restrict__ small_g, int32_t* __restrict__ rstate, int small_elsewhere,
    const uint16_t* __restrict__ lut_g, int f0, uint32_t* __restrict__ candq_g, size_t candq_fstride)
{
    __builtin_amdgcn_s_setprio(2);
    if (relay_frame<false, 4096 / RL_THREADS_BIG, RL_THREADS_BIG>(gbits, bits_fstride, wpr_g, W, H, lds_bits_words, min_len, kshift, tbits, segs, pool, pool_fstride, pool_cap,
                              kept_out, kept_cap, kcap, tail_keys, tail_off, counts, hint, small_g, rstate, nullptr, 0, small_elsewhere, lut_g, f0, candq_g, candq_fstride)) {
        __syncthreads();
        relay_frame<true, 4096 / RL_THREADS_BIG, RL_THREADS_BIG>(gbits, bits_fstride, wpr_g, W, H, lds_bits_words, min_len, kshift, tbits, segs, pool, pool_fstride, pool_cap,
                             kept_out, kept_cap, kcap, tail_keys, tail_off, counts, hint, small_g, rstate, nullptr, 0, small_elsewhere, lut_g, f0, candq_g, candq_fstride);
    }
}

// Frames whose bit image does not fit LDS (1920 x 1080: 264 KB): the same formulation with the padded bit image in HBM -- it stays
// in L2 -- and only the lists and the 8192-slot marker table in LDS.  A step costs an L2 round trip instead of an LDS one, but the
// segments are as short as ever (the single-walker kernel follows a 5000-point border in one lane), and the small borders are
// k_contours_small's, whose bands do fit LDS.
__global__ __launch_bounds__(RL_THREADS_BIG) void k_contours_relay8g(
    const uint32_t* __restrict__ gbits, size_t bits_fstride, int wpr_g, int W, int H, int lds_bits_words, int min_len,
    int kshift, int tbits, RelaySeg* __restrict__ segs, uint32_t* __restrict__ pool, size_t pool_fstride, int pool_cap,
    ArKept* __restrict__ kept_out, int kept_cap, int kcap, unsigned long long* __restrict__ tail_keys, int32_t* __restrict__ tail_off,
    int32_t* __restrict__ counts, int32_t* __restrict__ hint, uint4* __restrict__ small_g, int32_t* __restrict__ rstate,
    uint32_t* __restrict__ gpad, size_t gpad_fstride, const uint16_t* __restrict__ lut_g, int f0)
{
    __builtin_amdgcn_s_setprio(2);
    if (relay_frame<false, 8192 / RL_THREADS_BIG, RL_THREADS_BIG, true>(gbits, bits_fstride, wpr_g, W, H, lds_bits_words, min_len, kshift, tbits, segs, pool,
                              pool_fstride, pool_cap, kept_out, kept_cap, kcap, tail_keys, tail_off, counts, hint, small_g, rstate, gpad, gpad_fstride, 1, lut_g, f0)) {
        __syncthreads();
        relay_frame<true, 8192 / RL_THREADS_BIG, RL_THREADS_BIG, true>(gbits, bits_fstride, wpr_g, W, H, lds_bits_words, min_len, kshift, tbits, segs, pool,
                             pool_fstride, pool_cap, kept_out, kept_cap, kcap, tail_keys, tail_off, counts, hint, small_g, rstate, gpad, gpad_fstride, 1, lut_g, f0);
    }
}

// The step table of the relay walks (rl_lut_entry) in HBM, built once per detector: k_contours_small's workgroups are small and
// many, so they copy it instead of computing it.
__global__ void k_relay_lut(uint16_t* __restrict__ lut)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2048) lut[i] = rl_lut_entry((unsigned)i >> 3, i & 7);
}

// ---- (c) of the relay formulation for frames that have a grid, as its own launch between the relay kernel and k_contours_tail.
// A walk that starts at a border's start candidate stops at the first grid marker it meets, and a border cannot get past a relay
// row or column without one (aruco_trace.hpp), so the walk stays between the grid lines around its start.  One WAVE per block of
// K rows x RS_BLOCK_COLS columns of start pixels, four independent waves per workgroup, no workgroup barrier and no atomics on
// LDS: the block's part of the bit image in a wave-private LDS tile (K + 4 rows x 7 words), the start candidates of a round of
// rows compacted into a wave-private queue with ballots and scans, then every free lane takes the next candidate off the queue
// (lane prefix of the ballot of free lanes) and follows it.  Thousands of such waves fill the chip where phase (c) inside a
// relay kernel is one workgroup per frame waiting on dependent round trips -- to HBM / L2 when the bit image is not in LDS.
// The result is almost always "nothing": a border longer than min_len that touches no grid line is rare.  If there is one it is
// appended to the frame's kept list behind the relay kernel's borders (the tail sorts by discovery key).
__global__ __launch_bounds__(RS_THREADS) void k_contours_small(
    const uint32_t* __restrict__ gbits, size_t bits_fstride, int wpr_g, int W, int H, int min_len,
    const uint16_t* __restrict__ lut_g, int32_t* __restrict__ rstate, uint32_t* __restrict__ pool, size_t pool_fstride, int pool_cap,
    int kcap, unsigned long long* __restrict__ tail_keys, int32_t* __restrict__ tail_off, int32_t* __restrict__ counts)
{
    __shared__ __align__(16) uint16_t s_lut[2048];
    __shared__ uint32_t s_tile[RS_THREADS / 64][RS_TILE_ROWS * RS_TW + 2];
    __shared__ uint16_t s_q[RS_THREADS / 64][RS_QCAP];
    const int tid = threadIdx.x, lane = tid & 63, wid = wave_id(), f = blockIdx.y;
    if (counts[f * 4 + 2]) return;              // the relay kernel gave the frame up
    const int kshift = rstate[f * 2 + 0];
    if (kshift >= 30) return;                   // no grid (or phase (c) done inside the relay kernel)
    reinterpret_cast<uint4*>(s_lut)[tid] = reinterpret_cast<const uint4*>(lut_g)[tid]; // 2048 x 2 B = RS_THREADS x 16 B
    __syncthreads();                            // the only workgroup barrier: the step table
    const int kmask = (1 << kshift) - 1;
    // padded coordinates throughout (pixel (x, y) = bit x + 1 of row y + 1); relay rows / columns are the multiples of K
    const int ncb = (W >> RS_BLOCK_SHIFT) + 1, nbands = (H >> kshift) + 1;
    const int w = blockIdx.x * (RS_THREADS / 64) + wid;
    if (w >= ncb * nbands) return;
    const int band = w / ncb, cb = w - band * ncb;
    const int ys0 = max(1, band << kshift), ys1 = min(H, ((band + 1) << kshift) - 1);     // start rows of this block
    const int xs0 = max(1, cb << RS_BLOCK_SHIFT), xs1 = min(W, ((cb + 1) << RS_BLOCK_SHIFT) - 1); // start columns
    if (ys0 > ys1 || xs0 > xs1) return;
    const int j0 = cb << (RS_BLOCK_SHIFT - 5);  // first word of the block's columns; the tile holds words j0 - 1 .. j0 + RS_TW - 2
    const int ty0 = max(0, ys0 - 2), ty1 = min(H + 1, ys1 + 2);
    uint32_t* tile = s_tile[wid];
    uint16_t* queue = s_q[wid];
    const uint32_t* gb = gbits + (size_t)f * bits_fstride;
    uint32_t* pl = pool + (size_t)f * pool_fstride;
    {
        const int nw = RS_TW * (ty1 - ty0 + 1);
        for (int i = lane; i < nw; i += 64) {
            const int r = i / RS_TW, j = j0 - 1 + (i - r * RS_TW), py = ty0 + r; // word j of the padded image
            uint32_t v = 0;
            if (py >= 1 && py <= H && j >= 0) {
                const uint32_t* row = gb + (size_t)(py - 1) * wpr_g;
                const uint32_t cur = j < wpr_g ? row[j] : 0u;
                const uint32_t prv = (j >= 1 && j - 1 < wpr_g) ? row[j - 1] : 0u;
                v = (cur << 1) | (prv >> 31);
            }
            tile[i] = v;
        }
        if (lane < 2) tile[nw + lane] = 0; // spare words read by ring8()'s funnel loads
    }
    __builtin_amdgcn_wave_barrier();
    // the tile under the image's row and word numbers (nothing outside it is touched through this view, see ring_at)
    const BitImage im{tile - ty0 * RS_TW - (j0 - 1), RS_TW, W, H};
    // 3 x 3 neighbourhood of a padded pixel: from the tile, or -- a walk that left it: one that started on a relay column and went
    // more than a word to the left -- bit by bit from HBM
    auto ring_at = [&](int x, int y) -> unsigned {
        const int w0 = (x - 1) >> 5;
        if (y - 1 >= ty0 && y + 1 <= ty1 && w0 >= j0 - 1 && w0 + 1 <= j0 + RS_TW - 2) return ring8(im, x, y);
        auto px = [&](int qx, int qy) -> unsigned {
            if (qx < 1 || qx > W || qy < 1 || qy > H) return 0u;
            return (gb[(size_t)(qy - 1) * wpr_g + ((qx - 1) >> 5)] >> ((qx - 1) & 31)) & 1u;
        };
        return px(x + 1, y) | (px(x + 1, y - 1) << 1) | (px(x, y - 1) << 2) | (px(x - 1, y - 1) << 3) | (px(x - 1, y) << 4) |
               (px(x - 1, y + 1) << 5) | (px(x, y + 1) << 6) | (px(x + 1, y + 1) << 7);
    };
    auto step_from = [&](RelayWalk& wk, unsigned e) {
        wk.x += (int)((e >> 11) & 3u) - 1;
        wk.y += (int)((e >> 13) & 3u) - 1;
        wk.s = (int)((e + 4u) & 7u);
        wk.n++;
        wk.ring = ring_at(wk.x, wk.y);
    };
    const int stage0 = pool_cap >> 2; // the final part of the frame's pool (the relay kernel's staging arenas lie above it)
    // start candidates of word j0 + k of row wy that belong to this block: bit b = padded pixel (32 (j0 + k) + b, wy).  Candidates
    // are owned by their START pixel: a hole candidate's start is the pixel to its left, so the hole bits are the block's shifted by one
    constexpr int NWB = (1 << (RS_BLOCK_SHIFT - 5)) + 1; // words per row looked at: the block's and bit 0 of the next
    auto cand_masks = [&](int wy, int k, uint32_t& m_outer, uint32_t& m_hole) {
        const int wj = j0 + k;
        const uint32_t* row = im.bits + wy * RS_TW + wj;
        const uint32_t* up = row - RS_TW;
        const uint32_t cur = row[0], upw = up[0];
        const uint32_t cur_l = (cur << 1) | (row[-1] >> 31);
        const uint32_t up_l = (upw << 1) | (up[-1] >> 31);
        const uint32_t up_r = (upw >> 1) | (up[1] << 31);
        start_candidate_masks(cur, cur_l, upw, up_l, up_r, ORBFE_CAND_FILTER != 0, &m_outer, &m_hole);
        if (k == NWB - 1) m_outer = 0u;
        if (k == 0) m_hole &= ~1u;
        if (k == NWB - 1) m_hole &= 1u;
        if (wj == 0) { m_outer &= ~1u; m_hole &= ~3u; } // column 0 is the frame; a hole candidate at column 1 has no start pixel
    };
    int ncand_w = 0;
    // One loop for everything: when the queue is empty the wave (all lanes, walking or not) enumerates the next round of rows into
    // it; free lanes take candidates; busy lanes step.  Walks carry on across rounds, so a long walk delays nothing but its own lane.
    RelayWalk wk;
    bool busy = false;
    int head = 0, total = 0, q_r0 = ys0, r_next = ys0, R = RS_ROUND_ROWS;
    int sx = 0, sy = 0, s0 = 0, is_hole = 0, start_key = 0;
    for (;;) {
        if (head >= total && r_next <= ys1) {
            const int r0 = r_next, r1 = min(ys1, r0 + R - 1), nitems = (r1 - r0 + 1) * NWB;
            int cnt = 0;
            for (int i0 = 0; i0 < nitems; i0 += 64) {
                const int it = i0 + lane;
                uint32_t mo = 0, mh = 0;
                if (it < nitems) { const int r = it / NWB; cand_masks(r0 + r, it - r * NWB, mo, mh); }
                cnt += wave_sum(__popc(mo) + __popc(mh));
            }
            if (cnt > RS_QCAP) { R = max(1, R >> 1); continue; } // one row of a block has at most RS_BLOCK_COLS + 1 <= RS_QCAP candidates
            int qbase = 0;
            for (int i0 = 0; i0 < nitems; i0 += 64) {
                const int it = i0 + lane;
                uint32_t mo = 0, mh = 0;
                int r = 0, k = 0;
                if (it < nitems) { r = it / NWB; k = it - r * NWB; cand_masks(r0 + r, k, mo, mh); }
                const int c = __popc(mo) + __popc(mh);
                const int incl = wave_incl_scan_add(c);
                int q = qbase + incl - c;
                const uint32_t hi = ((uint32_t)r << 10) | ((uint32_t)k << 5); // row in the round (5 bits) | word (4) | bit (5) | hole << 15
                while (mo) { const int b = __ffs(mo) - 1; mo &= mo - 1; queue[q++] = (uint16_t)(hi | (uint32_t)b); }
                while (mh) { const int b = __ffs(mh) - 1; mh &= mh - 1; queue[q++] = (uint16_t)(hi | (uint32_t)b | 0x8000u); }
                qbase += __builtin_amdgcn_readlane(incl, 63);
            }
            __builtin_amdgcn_wave_barrier();
            ncand_w += cnt;
            head = 0; total = cnt; q_r0 = r0; r_next = r1 + 1;
        }
        if (head < total) {
            const unsigned long long fm = __ballot(!busy);
            const int q = head + a_lane_prefix(fm);
            if (!busy && q < total) {
                const uint32_t c = queue[q];
                is_hole = (int)(c >> 15);
                const int qx = ((j0 + (int)((c >> 5) & 15u)) << 5) + (int)(c & 31u), wy = q_r0 + (int)((c >> 10) & 31u);
                sx = qx - is_hole; sy = wy;
                start_key = wy * 65536 + qx;
                wk.x = sx; wk.y = sy; wk.n = 0;
                wk.ring = ring_at(sx, sy);
                s0 = relay_start_dir(wk.ring, is_hole);
                wk.s = s0;
                busy = s0 >= 0; // single-pixel borders are never kept
            }
            head = min(total, head + (int)__popcll(fm));
            __builtin_amdgcn_wave_barrier(); // reads of the queue stay in front of the next round's writes
        }
        if (!__any(busy)) {
            if (head >= total && r_next > ys1) break;
            continue;
        }
#pragma unroll
        for (int u = 0; u < RS_STEPS; u++) {
            if (busy) {
                const unsigned e = s_lut[(wk.ring << 3) | (unsigned)wk.s];
                const int key3 = wk.y * 65536 + wk.x;
                bool stop = rl_is_marker(e, wk.x, wk.y, kmask); // the border belongs to the segment walkers
                if (is_hole)                                      // relay_not_canonical() on the table's run bits
                    stop |= ((e & 0x080u) && key3 - 65536 < start_key) || ((e & 0x100u) && key3 - 1 < start_key) ||
                            ((e & 0x200u) && key3 + 1 < start_key) || ((e & 0x400u) && key3 + 65536 < start_key);
                else stop |= key3 < start_key;
                if (stop) busy = false;
                else {
                    step_from(wk, e);
                    if (wk.x == sx && wk.y == sy && wk.s == s0) {
                        busy = false;
                        if (wk.n > min_len) {
                            // rare: more than min_len points between grid lines.  The border is whole, so its final place
                            // is known -- walk it once more, straight into the pool
                            const int n = wk.n;
                            const int k = atomicAdd(&counts[f * 4 + 0], 1);
                            const int base = atomicAdd(&rstate[f * 2 + 1], n);
                            if (k >= kcap) { atomicSub(&counts[f * 4 + 0], 1); atomicOr(&counts[f * 4 + 2], 2); }
                            else if (base + n > stage0) atomicOr(&counts[f * 4 + 2], 4);
                            else {
                                RelayWalk w2;
                                w2.x = sx; w2.y = sy; w2.s = s0; w2.n = 0;
                                w2.ring = ring_at(sx, sy);
                                for (int o = 0; o < n; o++) {
                                    pl[base + o] = relay_point(w2);
                                    step_from(w2, s_lut[(w2.ring << 3) | (unsigned)w2.s]);
                                }
                                tail_keys[(size_t)f * kcap + k] =
                                    ((unsigned long long)(0xffffffffu - (uint32_t)start_key) << 32) |
                                    ((unsigned long long)(n & 0x7ffff) << 13) | ((unsigned)k << 1) | (unsigned)is_hole;
                                tail_off[(size_t)f * kcap + k] = base;
                            }
                        }
                    }
                }
            }
        }
    }
    if (lane == 0 && ncand_w) atomicAdd(&counts[f * 4 + 3], ncand_w);
}

// ---- (g) of the relay formulation: sort, approxPolyDP, rectangles for the borders the relay kernel (and k_contours_small) kept.
// Three launches, because approxPolyDP is one wave per border and a frame has 60 (640 x 480) to 2000 (1920 x 1080) of them: as one
// workgroup of four waves per frame it kept a fifth of the chip busy (and 1280 x 720 frames waited 0.8 ms for it).
//   k_tail_prep    (workgroup per frame)  sort the kept borders into findContours' order (reverse discovery), rank them by length,
//                                         append (frame, border) work items, longest first, to one list for the whole batch
//   k_tail_approx  (persistent waves)     a wave takes the next work item off the list: approxPolyDP(eps = 0.05 len) + convexity test
//   k_tail_finish  (wave per frame)       ordered compaction of the 4-gons into the frame's rectangle list, per-frame counts
// Frames the relay kernel gave up on (flags != 0) contribute nothing; the host entry points redo them.
__global__ __launch_bounds__(1024) void k_tail_prep(const unsigned long long* __restrict__ tail_keys, const int32_t* __restrict__ tail_off,
                                                    int kcap, const int32_t* __restrict__ counts, uint4* __restrict__ work,
                                                    size_t work_half, int32_t* __restrict__ ctr)
{
    __builtin_amdgcn_s_setprio(2);
    extern __shared__ __align__(16) unsigned char tp_smem[];
    __shared__ int s_base, s_base2, s_hist[RT_BUCKETS];
    const int tid = threadIdx.x, f = blockIdx.x, NT = (int)blockDim.x;
    const int nkept = counts[f * 4 + 2] ? 0 : min(counts[f * 4 + 0], kcap);
    if (nkept <= 0) return;
    unsigned long long* kkey = (unsigned long long*)tp_smem;
    int* off_u = (int*)(kkey + kcap);
    for (int k = tid; k < nkept; k += NT) { kkey[k] = tail_keys[(size_t)f * kcap + k]; off_u[k] = tail_off[(size_t)f * kcap + k]; }
    int Pn = 1;
    while (Pn < nkept) Pn <<= 1;
    for (int i = nkept + tid; i < Pn; i += NT) kkey[i] = ~0ull;
    for (int i = tid; i < RT_BUCKETS; i += NT) s_hist[i] = 0;
    __syncthreads();
    for (int k = 2; k <= Pn; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (Pn >> 1); t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const unsigned long long a = kkey[i], b = kkey[l];
                const bool up = ((i & k) == 0);
                if ((a > b) == up) { kkey[i] = b; kkey[l] = a; }
            }
            __syncthreads();
        }
    // the frame's part of the work list, long borders first: a counting sort on length classes of 8 points (the order only spreads
    // the work -- every border's result lands in its own slot -- so the order inside a class may vary from run to run)
    auto bucket_of = [](int len) { return RT_BUCKETS - 1 - (min(len, RT_BUCKETS * 8 - 1) >> 3); };
    for (int k = tid; k < nkept; k += NT) atomicAdd(&s_hist[bucket_of((int)((kkey[k] >> 13) & 0x7ffff))], 1);
    __syncthreads();
    if (tid < 64) { // exclusive scan of the class counts by one wave
        int v[RT_BUCKETS / 64], sum = 0;
        for (int i = 0; i < RT_BUCKETS / 64; i++) { v[i] = s_hist[tid * (RT_BUCKETS / 64) + i]; sum += v[i]; }
        int incl = sum;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (tid >= d) incl += o; }
        int run = incl - sum;
        for (int i = 0; i < RT_BUCKETS / 64; i++) { s_hist[tid * (RT_BUCKETS / 64) + i] = run; run += v[i]; }
    }
    __syncthreads();
    // borders of fewer than RT_QUAD_MAX points go to the second list (k_tail_approx does four of them at a time)
    const int nlong = s_hist[bucket_of(RT_QUAD_MAX - 1)];
    __syncthreads();
    if (tid == 0) { s_base = atomicAdd(&ctr[0], nlong); s_base2 = atomicAdd(&ctr[1], nkept - nlong); }
    __syncthreads();
    const int base = s_base, base2 = s_base2 - nlong;
    uint4* work2 = work + work_half;
    for (int k = tid; k < nkept; k += NT) {
        const unsigned long long key = kkey[k];
        const int len = (int)((key >> 13) & 0x7ffff);
        const int r = atomicAdd(&s_hist[bucket_of(len)], 1);
        const uint4 e = make_uint4(((uint32_t)f << 12) | (uint32_t)k, (uint32_t)len, (uint32_t)off_u[(int)((key >> 1) & 0xfff)], 0u);
        if (r < nlong) work[base + r] = e;
        else work2[base2 + r] = e;
    }
}

__global__ __launch_bounds__(256) void k_tail_approx(int kcap, const uint4* __restrict__ work, size_t work_half, const int32_t* __restrict__ ctr,
                                                     const uint32_t* __restrict__ pool, size_t pool_fstride, ArKept* __restrict__ kept_out,
                                                     int kept_cap, uint8_t* __restrict__ rectflag, int pts)
{
    if (ORBFE_PRIO_DET_TAIL) __builtin_amdgcn_s_setprio(ORBFE_PRIO_DET_TAIL);
    extern __shared__ __align__(16) unsigned char ta_smem[];
    const int lane = threadIdx.x & 63, wid = wave_id();
    ApPt* o = (ApPt*)ta_smem + wid * AP_OUT;
    int2* stack = (int2*)((ApPt*)ta_smem + 4 * AP_OUT) + wid * AP_STACK;
    uint32_t* b = (uint32_t*)((int2*)((ApPt*)ta_smem + 4 * AP_OUT) + 4 * AP_STACK) + (size_t)wid * pts;
    // The list is sorted by length frame by frame, so dealing it out round-robin balances the waves (a ticket counter shared by
    // 4096 waves was measured slower than the whole of approxPolyDP: 247 us against the 148 us of the per-frame kernel).  A border
    // costs a wave three dependent trips to memory (work item, points, result) next to a few microseconds of arithmetic: the item
    // after the next and the first 128 points of the next border are fetched while the current one is worked on.
    const int total = ctr[0], nw = (int)gridDim.x * 4;
    int t = (int)blockIdx.x + wid * (int)gridDim.x;
    const uint4 none = make_uint4(0u, 0u, 0u, 0u);
    auto src_of = [&](const uint4& e) { return pool + (size_t)(e.x >> 12) * pool_fstride + e.z; };
    uint4 e = t < total ? work[t] : none;
    uint4 e1 = t + nw < total ? work[t + nw] : none;
    uint32_t p0 = 0, p1 = 0;
    if (lane < (int)e.y) p0 = src_of(e)[lane];
    if (lane + 64 < (int)e.y) p1 = src_of(e)[lane + 64];
    for (; t < total; t += nw) {
        const uint4 e2 = t + 2 * nw < total ? work[t + 2 * nw] : none;
        uint32_t q0 = 0, q1 = 0;
        if (lane < (int)e1.y) q0 = src_of(e1)[lane];
        if (lane + 64 < (int)e1.y) q1 = src_of(e1)[lane + 64];
        const int f = (int)(e.x >> 12), k = (int)(e.x & 0xfffu), off = (int)e.z;
        const int n = (int)e.y;
        int ok = 0;
        if (n > 0) {
            const uint32_t* src = src_of(e);
            int nv;
            if (n <= pts) {
                b[lane] = p0;         // pts >= 128
                b[lane + 64] = p1;
                for (int i = lane + 128; i < n; i += 64) b[i] = src[i];
                __builtin_amdgcn_wave_barrier();
                nv = approx_poly_wave(b, n, o, stack, lane);
            } else {
                nv = approx_poly_wave(src, n, o, stack, lane);
            }
            __builtin_amdgcn_wave_barrier();
            ok = (nv == 4) && convex4(o);
        }
        if (lane == 0) {
            rectflag[(size_t)f * kcap + k] = (uint8_t)ok;
            if (ok && k < kept_cap) {
                ArKept kk;
                kk.off = off; kk.len = n;
                for (int j = 0; j < 4; j++) { kk.vx[j] = (short)o[j].x; kk.vy[j] = (short)o[j].y; }
                kept_out[(size_t)f * kept_cap + k] = kk;
            }
        }
        __builtin_amdgcn_wave_barrier();
        e = e1; e1 = e2; p0 = q0; p1 = q1;
    }
    // The short borders (most of them: a 1280 x 720 frame keeps hundreds of 70 .. 160 points), four per wave, a DPP row of 16 lanes
    // each: a slice of a short border has a handful of points, so what a border costs is the chain of dependent steps per slice
    // (pop, two point reads, the sweep, two reductions, the f64 test, push) -- four such chains now overlap in one wave.
    {
        const uint4* work2 = work + work_half;
        const int nshort = ctr[1], g = lane >> 4, gl = lane & 15;
        ApPt* og = o + g * 16;       // per group: 16 vertices and 16 stack entries, a quarter of the wave's
        int2* sg = stack + g * 16;
        uint32_t* bg = b + g * RT_QUAD_MAX;
        for (int q = (int)blockIdx.x + wid * (int)gridDim.x; q * 4 < nshort; q += nw) {
            const int idx = q * 4 + g;
            const uint4 it = idx < nshort ? work2[idx] : none;
            const int n = (int)it.y;
            const uint32_t* src = src_of(it);
            for (int i = gl; i < n; i += 16) bg[i] = src[i];
            __builtin_amdgcn_wave_barrier();
            if (n > 0) {
                const int nv = approx_poly_g<16, 16, 16>(bg, n, og, sg, gl);
                __builtin_amdgcn_wave_barrier();
                const int ok = (nv == 4) && convex4(og);
                if (gl == 0) {
                    const int f = (int)(it.x >> 12), k = (int)(it.x & 0xfffu);
                    rectflag[(size_t)f * kcap + k] = (uint8_t)ok;
                    if (ok && k < kept_cap) {
                        ArKept kk;
                        kk.off = (int)it.z; kk.len = n;
                        for (int j = 0; j < 4; j++) { kk.vx[j] = (short)og[j].x; kk.vy[j] = (short)og[j].y; }
                        kept_out[(size_t)f * kept_cap + k] = kk;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// the 4-gons of a frame's kept borders -> its rectangle list (one wave)
__device__ __forceinline__ void tail_finish_frame(int f, int lane, int kcap, const uint8_t* __restrict__ rectflag, const ArKept* __restrict__ kept_out,
                                                  int kept_cap, ArRect* __restrict__ rects_out, int rect_cap, int32_t* __restrict__ counts,
                                                  int32_t* __restrict__ ctr)
{
    if (f == 0 && lane == 0) { ctr[0] = 0; ctr[1] = 0; } // every consumer of this batch's work list is done: ready for the next batch
    const int flags = counts[f * 4 + 2];
    if (flags) return; // the relay kernel reported the frame as failed
    const int nkept = counts[f * 4 + 0];
    int nr = 0;
    for (int k0 = 0; k0 < nkept; k0 += 64) {
        const int k = k0 + lane;
        const bool is = k < nkept && k < kept_cap && rectflag[(size_t)f * kcap + k];
        const unsigned long long m = __ballot(is);
        if (is) {
            const int q = nr + a_lane_prefix(m);
            if (q < rect_cap) {
                const ArKept kk = kept_out[(size_t)f * kept_cap + k];
                ArRect r;
                for (int j = 0; j < 4; j++) { r.c[j][0] = (float)kk.vx[j]; r.c[j][1] = (float)kk.vy[j]; }
                r.off = kk.off; r.len = kk.len;
                rects_out[(size_t)f * rect_cap + q] = r;
            }
        }
        nr += (int)__popcll(m);
    }
    if (lane == 0) {
        counts[f * 4 + 1] = min(nr, rect_cap);
        if (nr > rect_cap) counts[f * 4 + 2] = flags | 8;
    }
}
__global__ __launch_bounds__(64) void k_tail_finish(int kcap, const uint8_t* __restrict__ rectflag, const ArKept* __restrict__ kept_out,
                                                    int kept_cap, ArRect* __restrict__ rects_out, int rect_cap, int32_t* __restrict__ counts,
                                                    int32_t* __restrict__ ctr)
{
    if (ORBFE_PRIO_DET_TAIL) __builtin_amdgcn_s_setprio(ORBFE_PRIO_DET_TAIL);
    tail_finish_frame(blockIdx.x, threadIdx.x, kcap, rectflag, kept_out, kept_cap, rects_out, rect_cap, counts, ctr);
}

// ---------------------------------------------------------------------------------------- prefilter -----------
__device__ __forceinline__ int ar_perimeter(const float c[4][2])
{
    int sum = 0;
    for (int i = 0; i < 4; i++) {
        const int i2 = (i + 1) & 3;
        const float dx = c[i][0] - c[i2][0], dy = c[i][1] - c[i2][1];
        sum += (int)__fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    }
    return sum;
}

// prefilterCandidates: one workgroup (256 threads) per frame
__global__ __launch_bounds__(256) void k_prefilter(ArRect* __restrict__ rects, int rect_cap, int32_t* __restrict__ counts,
                                                   int W, int H, int too_near, int32_t* __restrict__ cand_idx,
                                                   int32_t* __restrict__ ncand_out, uint32_t* __restrict__ work, int32_t* __restrict__ wctr, TailFinish tf)
{
    __builtin_amdgcn_s_setprio(2); // latency-bound: its few waves go first when a VALU-bound kernel shares the CU
    __shared__ int s_rm[AR_MAX_RECTS];
    __shared__ int s_per[AR_MAX_RECTS];
    const int f = blockIdx.x, tid = threadIdx.x;
    if (tf.rectflag) {   // the rectangle list first (k_tail_finish's work, one wave: a launch less on the detector's chain)
        if (tid < 64) tail_finish_frame(f, tid, tf.kcap, tf.rectflag, tf.kept, tf.kept_cap, rects, rect_cap, counts, tf.ctr);
        __threadfence_block();
        __syncthreads();
    }
    const int n = *(volatile int32_t*)&counts[f * 4 + 1];
    ArRect* R = rects + (size_t)f * rect_cap;
    for (int i = tid; i < n; i += 256) {
        ArRect r = R[i];
        const double dx1 = r.c[1][0] - r.c[0][0], dy1 = r.c[1][1] - r.c[0][1];
        const double dx2 = r.c[2][0] - r.c[0][0], dy2 = r.c[2][1] - r.c[0][1];
        const double o = (dx1 * dy2) - (dy1 * dx2);
        if (o < 0.0) {
            const float tx = r.c[1][0], ty = r.c[1][1];
            r.c[1][0] = r.c[3][0]; r.c[1][1] = r.c[3][1];
            r.c[3][0] = tx; r.c[3][1] = ty;
            R[i] = r;
        }
        s_per[i] = ar_perimeter(r.c);
        s_rm[i] = 0;
    }
    __threadfence_block();
    __syncthreads();
    const float tn = (float)too_near;
    for (int p = tid; p < n * n; p += 256) {
        const int i = p / n, j = p - i * n;
        if (j <= i) continue;
        bool near = true;
        for (int k = 0; k < 4; k++) {
            const float dx = R[i].c[k][0] - R[j].c[k][0], dy = R[i].c[k][1] - R[j].c[k][1];
            const float d = (float)sqrt((double)dx * dx + (double)dy * dy);
            near = near && (d < tn);
        }
        if (near) {
            if (s_per[i] > s_per[j]) atomicOr(&s_rm[j], 1);
            else atomicOr(&s_rm[i], 1);
        }
    }
    __syncthreads();
    const int bx = (int)(0.015f * (float)W), by = (int)(0.015f * (float)H);
    for (int i = tid; i < n; i += 256) {
        bool rm = false;
        for (int k = 0; k < 4; k++) {
            const float x = R[i].c[k][0], y = R[i].c[k][1];
            rm = rm || x < (float)bx || y < (float)by || x > (float)(W - bx) || y > (float)(H - by);
        }
        if (rm) s_rm[i] = 1;
    }
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int i = 0; i < n; i++)
            if (!s_rm[i]) cand_idx[(size_t)f * rect_cap + m++] = i;
        ncand_out[f] = m;
        // the batch's candidates as ONE work list for the decode kernels (frame << 16 | slot; any order: results go to
        // (frame, slot)); k_finalize leaves the counter at zero for the next batch
        if (m > 0) {
            const int base = atomicAdd(wctr, m);
            for (int i = 0; i < m; i++) work[base + i] = ((uint32_t)f << 16) | (uint32_t)i;
        }
    }
}

// ---------------------------------------------------------------------------------------- decode --------------
__device__ __forceinline__ int ar_sat_int(double v)
{
    if (v <= -2147483648.0) return (int)0x80000000;
    if (v >= 2147483647.0) return 2147483647;
    return orbfe_round_d(v);
}

__device__ __forceinline__ double shfl_d(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return __hiloint2double(hi, lo);
}

// 8x8 linear system by Gaussian elimination with partial pivoting, one row per lane (lanes 0..7), same operation
// order per element as the serial loop in the oracle (so the results are bit-identical).  Returns false if singular;
// x[k] (all lanes) = solution.
// value of lane `src` (wave-uniform) in every lane: v_readlane_b32 through a scalar register -- a few cycles, where ds_bpermute
// (__shfl) is an LDS round trip; the elimination below makes ~400 such broadcasts per candidate, one after the other
__device__ __forceinline__ double rl_d(double v, int src)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ bool solve8_wave(double a[8], double b, int lane, double x[8])
{
#pragma unroll
    for (int c = 0; c < 8; c++) {
        // pivot: first row r >= c with the largest |a[r][c]| (rows live in lanes 0..7; everything here is wave-uniform)
        const double mine = fabs(a[c]);
        double v = rl_d(mine, c);
        int piv = c;
#pragma unroll
        for (int r = c + 1; r < 8; r++) {
            const double ov = rl_d(mine, r);
            if (ov > v) { v = ov; piv = r; }
        }
        if (!(v > 0.0)) return false;
        piv = __builtin_amdgcn_readfirstlane(piv);
        // swap rows c and piv
        if (piv != c) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const double rc = rl_d(a[k], c), rp = rl_d(a[k], piv);
                a[k] = lane == c ? rp : lane == piv ? rc : a[k];
            }
            const double bc = rl_d(b, c), bp = rl_d(b, piv);
            b = lane == c ? bp : lane == piv ? bc : b;
        }
        // eliminate below
        const double pc = rl_d(a[c], c);
        const double pb = rl_d(b, c);
        const bool below = lane > c && lane < 8;
        const double f = below ? a[c] / pc : 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double pk = rl_d(a[k], c);
            if (k >= c && below && f != 0.0) a[k] -= f * pk;
        }
        if (below && f != 0.0) b -= f * pb;
    }
    // back substitution, r = 7..0; sums run over k = r+1..7 in ascending order like the serial loop
#pragma unroll
    for (int r = 7; r >= 0; r--) {
        double s = b;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k > r) s -= a[k] * x[k];
        const double xr = s / a[r];
        x[r] = rl_d(xr, r);
    }
    return true;
}

// Decoding the rectangle candidates: three launches over ONE work list for the whole batch (k_prefilter builds it).  Until round 3
// this was one workgroup of 8 waves and 76 KB of LDS per frame -- 136 us alone, 420 - 670 us next to the other engines' kernels,
// most of it eight waves parked on a CU while one of them ran the serial Otsu recurrence.
//   k_decode_warp  (wave per candidate, persistent waves): pyramid level, homography (wave-parallel 8x8 elimination), 1/32-px
//                  fixed-point warp -> the S x S patch and its 256-bin histogram, both to HBM (they stay in L2)
//   k_decode_otsu  (ONE LANE per candidate, 64 candidates of the batch side by side): getThreshVal_Otsu_8u.  Its running sums are
//                  serial by definition (every step rounds, and the gaps between the two clusters of a marker histogram are
//                  exact ties that the rounding noise decides): ~256 dependent double divisions per candidate
//   k_decode_vote  (wave per candidate): the patch against the threshold -> cell votes, border check, 4 rotations, dictionary
#define DC_PXCAP DC_PATCH_BYTES // bytes per kept patch: 35 x 35, the warp size of every configuration the detector is used in here

__device__ __forceinline__ int dc_warp_pixel(const double* Mi, int y, int xx, const uint8_t* img, int pitch, int LW, int LH)
{
    // warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0): 1/32-px coordinates, 15-bit bilinear weights
    const double X0 = Mi[0] * 0 + Mi[1] * y + Mi[2];
    const double Y0 = Mi[3] * 0 + Mi[4] * y + Mi[5];
    const double W0d = Mi[6] * 0 + Mi[7] * y + Mi[8];
    double Wd = W0d + Mi[6] * xx;
    Wd = Wd ? 32.0 / Wd : 0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + Mi[0] * xx) * Wd));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + Mi[3] * xx) * Wd));
    const int X = ar_sat_int(fX), Y = ar_sat_int(fY);
    int sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : sx > 32767 ? 32767 : sx;
    sy = sy < -32768 ? -32768 : sy > 32767 ? 32767 : sy;
    const int ax = X & 31, ay = Y & 31;
    const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
    // the four neighbours as unconditional loads (clamped coordinates, the value masked afterwards): under `if (inside)` each was a
    // branch with a wait for memory behind it -- four serial round trips a pixel, twenty pixels a lane
    const bool x0in = sx >= 0 && sx < LW, x1in = sx + 1 >= 0 && sx + 1 < LW, y0in = sy >= 0 && sy < LH, y1in = sy + 1 >= 0 && sy + 1 < LH;
    const int cx0 = min(max(sx, 0), LW - 1), cx1 = min(max(sx + 1, 0), LW - 1);
    const uint8_t* r0 = img + (size_t)min(max(sy, 0), LH - 1) * pitch;
    const uint8_t* r1 = img + (size_t)min(max(sy + 1, 0), LH - 1) * pitch;
    const int p00 = r0[cx0], p01 = r0[cx1], p10 = r1[cx0], p11 = r1[cx1];
    int v = (x0in && y0in ? p00 : 0) * w00 + (x1in && y0in ? p01 : 0) * w01 + (x0in && y1in ? p10 : 0) * w10 + (x1in && y1in ? p11 : 0) * w11;
    v = (v + (1 << 14)) >> 15;
    return v > 255 ? 255 : v;
}


__global__ __launch_bounds__(DC_WAVES * 64) void k_decode_warp(ImgView src0, ImgView pyr, const ArLevel* __restrict__ levels, int nlevels,
                                                               const ArRect* __restrict__ rects, int rect_cap,
                                                               const int32_t* __restrict__ cand_idx, int S, int W0,
                                                               const uint32_t* __restrict__ work, const int32_t* __restrict__ wctr,
                                                               DcItem* __restrict__ items, uint16_t* __restrict__ hist,
                                                               uint8_t* __restrict__ patch)
{
    if (ORBFE_PRIO_DET_TAIL) __builtin_amdgcn_s_setprio(ORBFE_PRIO_DET_TAIL);
    __shared__ uint32_t s_hist[DC_WAVES][256];
    const int lane = threadIdx.x & 63, wid = wave_id();
    const int nitems = wctr[0];
    const bool keep_px = S * S <= DC_PXCAP;
    for (int it = blockIdx.x * DC_WAVES + wid; it < nitems; it += gridDim.x * DC_WAVES) {
        const uint32_t wi = work[it];
        const int f = (int)(wi >> 16), slot = (int)(wi & 0xffffu);
        uint32_t* h = s_hist[wid];
        h[lane] = 0; h[lane + 64] = 0; h[lane + 128] = 0; h[lane + 192] = 0;
        const ArRect r = rects[(size_t)f * rect_cap + cand_idx[(size_t)f * rect_cap + slot]];
        // pyramid level: largest p with area / 4^p >= S^2 (markerdetector_impl.cpp:6507-6586)
        const float v01x = r.c[1][0] - r.c[0][0], v01y = r.c[1][1] - r.c[0][1];
        const float v03x = r.c[3][0] - r.c[0][0], v03y = r.c[3][1] - r.c[0][1];
        const float area1 = fabsf(__fsub_rn(__fmul_rn(v01x, v03y), __fmul_rn(v01y, v03x)));
        const float v21x = r.c[1][0] - r.c[2][0], v21y = r.c[1][1] - r.c[2][1];
        const float v23x = r.c[3][0] - r.c[2][0], v23y = r.c[3][1] - r.c[2][1];
        const float area2 = fabsf(__fsub_rn(__fmul_rn(v21x, v23y), __fmul_rn(v21y, v23x)));
        const float area = __fdiv_rn(__fadd_rn(area2, area1), 2.f);
        const float desired = __fmul_rn((float)S, (float)S);
        int lvl = 0;
        double p4 = 4.0;
        for (int p = 1; p < nlevels; p++, p4 *= 4.0) {
            if ((double)area / p4 >= (double)desired) lvl = p;
            else break;
        }
        const ArLevel L = levels[lvl];
        const float ratio = __fdiv_rn((float)L.w, (float)W0);
        // getPerspectiveTransform(quad -> (0,0),(S-1,0),(S-1,S-1),(0,S-1)): rows 0..3 = x equations, 4..7 = y equations
        double a[8], bb = 0.0, x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { a[k] = 0.0; x[k] = 0.0; }
        if (lane < 8) {
            const int i = lane & 3;
            const float qx = __fmul_rn(r.c[i][0], ratio), qy = __fmul_rn(r.c[i][1], ratio);
            const float dx = (i == 1 || i == 2) ? (float)(S - 1) : 0.f, dy = (i >= 2) ? (float)(S - 1) : 0.f;
            if (lane < 4) {
                a[0] = qx; a[1] = qy; a[2] = 1.0;
                a[6] = -(double)qx * dx; a[7] = -(double)qy * dx;
                bb = dx;
            } else {
                a[3] = qx; a[4] = qy; a[5] = 1.0;
                a[6] = -(double)qx * dy; a[7] = -(double)qy * dy;
                bb = dy;
            }
        }
        bool ok = solve8_wave(a, bb, lane, x);
        double Mi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) {
            const double M[9] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], 1.0};
            const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                               M[2] * (M[3] * M[7] - M[4] * M[6]);
            if (det == 0.0) ok = false;
            else {
                const double d = 1.0 / det;
                Mi[0] = (M[4] * M[8] - M[5] * M[7]) * d;
                Mi[1] = (M[2] * M[7] - M[1] * M[8]) * d;
                Mi[2] = (M[1] * M[5] - M[2] * M[4]) * d;
                Mi[3] = (M[5] * M[6] - M[3] * M[8]) * d;
                Mi[4] = (M[0] * M[8] - M[2] * M[6]) * d;
                Mi[5] = (M[2] * M[3] - M[0] * M[5]) * d;
                Mi[6] = (M[3] * M[7] - M[4] * M[6]) * d;
                Mi[7] = (M[1] * M[6] - M[0] * M[7]) * d;
                Mi[8] = (M[0] * M[4] - M[1] * M[3]) * d;
            }
        }
        int isum = 0, blo = 0, bhi = 0;
        if (ok) {
            const uint8_t* img = (lvl == 0) ? src0.base + (size_t)f * src0.fstride : pyr.base + (size_t)f * pyr.fstride + L.off;
            const int pitch = (lvl == 0) ? src0.pitch : L.pitch;
            uint8_t* po = patch + (size_t)it * DC_PXCAP;
            for (int i0 = 0; i0 < S * S; i0 += 4 * 64) {   // four pixels a lane and trip: sixteen loads in flight
                int vv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int i = min(i0 + 64 * k + lane, S * S - 1);
                    const int y = i / S, xx = i - y * S;
                    vv[k] = dc_warp_pixel(Mi, y, xx, img, pitch, L.w, L.h);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int i = i0 + 64 * k + lane;
                    if (i < S * S) {
                        atomicAdd(&h[vv[k]], 1u);
                        if (keep_px) po[i] = (uint8_t)vv[k];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // the histogram as u16 (a bin holds at most S * S pixels; the host refuses S > 255) and its first moment, which is a sum
            // of integers -- exact in double in any order, so the Otsu lane starts from it instead of walking the bins twice
            const uint32_t h0 = h[4 * lane], h1 = h[4 * lane + 1], h2 = h[4 * lane + 2], h3 = h[4 * lane + 3];
            reinterpret_cast<uint2*>(hist + (size_t)it * 256)[lane] = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
            isum = wave_sum((int)(h0 * (4 * lane) + h1 * (4 * lane + 1) + h2 * (4 * lane + 2) + h3 * (4 * lane + 3)));
            // first and last non-empty bin (in units of this lane's four bins): the Otsu recurrence does nothing outside them
            const unsigned long long ne = __ballot((h0 | h1 | h2 | h3) != 0u);
            blo = ne ? 4 * __builtin_ctzll(ne) : 0;
            bhi = ne ? 4 * (63 - __builtin_clzll(ne)) + 3 : 0;
        }
        if (lane == 0) {
            DcItem* o = items + it;
#pragma unroll
            for (int k = 0; k < 9; k++) o->Mi[k] = Mi[k];
            o->lvl = lvl; o->ok = ok ? 1 : 0; o->isum = isum; o->th = 0; o->bin_lo = blo; o->bin_hi = bhi;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// getThreshVal_Otsu_8u with exactly the reference's operation sequence, one lane per candidate of the batch
__global__ __launch_bounds__(64) void k_decode_otsu(const int32_t* __restrict__ wctr, DcItem* __restrict__ items,
                                                    const uint16_t* __restrict__ hist, int S)
{
    __builtin_amdgcn_s_setprio(2); // latency-bound: one wave of serial f64 chains goes first when a VALU-bound kernel shares the CU
    const int nitems = wctr[0];
    const int it = blockIdx.x * 64 + threadIdx.x;
    if (it >= nitems) return;
    int max_val = 0;
    if (items[it].ok) {
        const uint4* hp = reinterpret_cast<const uint4*>(hist + (size_t)it * 256);
        const int n = S * S;
        const double scale = 1. / n;
        const double mu = (double)items[it].isum * scale;
        // The recurrence mu1, q1 is the serial part; mu2 and sigma of a bin do not feed back.  They are evaluated ONE BIN LATE, from
        // the saved (q1, mu1) of the previous bin, and without branches, so that the two dependent chains of a trip -- this bin's
        // recurrence (multiply, add, divide) and the previous bin's variance (divide, four multiplies, compare) -- are independent
        // instruction streams the in-order wave can interleave.  Same operations on the same values in the same order of bins.
        double mu1 = 0, q1 = 0, max_sigma = 0;
        double pq1 = 0, pmu1 = 0;
        bool pvalid = false;
        auto variance_of_previous = [&](int i_prev) {
            const double q2p = 1. - pq1;
            const double mu2 = __ddiv_rn(mu - pq1 * pmu1, q2p);
            const double dm = pmu1 - mu2;
            double sigma = pq1 * q2p * dm * dm;
            sigma = pvalid ? sigma : -1.0; // a skipped bin never wins (max_sigma >= 0)
            const bool better = sigma > max_sigma;
            max_sigma = better ? sigma : max_sigma;
            max_val = better ? i_prev : max_val;
        };
        // Bins below the first non-empty one leave mu1 = q1 = 0, bins above the last one are all skipped (q2 = 1 - q1 is below the
        // tolerance once every pixel is counted): the loop runs over the groups of 32 bins in between -- same values, fewer trips
        const int i_begin = items[it].bin_lo & ~31, i_end = min(256, (items[it].bin_hi | 31) + 1);
        for (int i0 = i_begin; i0 < i_end; i0 += 32) { // four 16-byte loads (32 bins) in flight per lane
            uint4 hv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) hv[k] = hp[i0 / 8 + k];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t w[4] = {hv[k].x, hv[k].y, hv[k].z, hv[k].w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int i = i0 + k * 8 + j;
                    variance_of_previous(i - 1);
                    const uint32_t hi = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                    const double p_i = hi * scale;
                    mu1 *= q1;
                    q1 += p_i;
                    const double q2 = 1. - q1;
                    const bool skip = fmin(q1, q2) < 1.1920928955078125e-07 || fmax(q1, q2) > 1. - 1.1920928955078125e-07;
                    const double upd = __ddiv_rn(mu1 + i * p_i, q1);
                    mu1 = skip ? mu1 : upd;
                    pvalid = !skip; pq1 = q1; pmu1 = mu1;
                }
            }
        }
        variance_of_previous(i_end - 1);
    }
    items[it].th = max_val;
}

__global__ __launch_bounds__(DC_WAVES * 64) void k_decode_vote(ImgView src0, ImgView pyr, const ArLevel* __restrict__ levels, int rect_cap,
                                                               int S, int nb, const unsigned long long* __restrict__ codes, int ncodes,
                                                               const unsigned long long* __restrict__ scodes,
                                                               const int32_t* __restrict__ sids, int nsorted, int max_corr,
                                                               const uint32_t* __restrict__ work, const int32_t* __restrict__ wctr,
                                                               const DcItem* __restrict__ items, const uint8_t* __restrict__ patch,
                                                               int32_t* __restrict__ result /*per slot: id, nrot*/)
{
    if (ORBFE_PRIO_DET_TAIL) __builtin_amdgcn_s_setprio(ORBFE_PRIO_DET_TAIL);
    __shared__ int s_ones[DC_WAVES][64], s_tot[DC_WAVES][64];
    __shared__ uint8_t s_bits[DC_WAVES][64];
    __shared__ unsigned long long s_ids[DC_WAVES][4];
    const int lane = threadIdx.x & 63, wid = wave_id();
    const int nitems = wctr[0];
    const bool keep_px = S * S <= DC_PXCAP;
    for (int it = blockIdx.x * DC_WAVES + wid; it < nitems; it += gridDim.x * DC_WAVES) {
        const uint32_t wi = work[it];
        const int f = (int)(wi >> 16), slot = (int)(wi & 0xffffu);
        int32_t* res = result + ((size_t)f * rect_cap + slot) * 2;
        const DcItem* ci = items + it;
        if (!ci->ok) {
            if (lane == 0) { res[0] = -1; res[1] = 0; }
            continue;
        }
        double Mi[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Mi[k] = ci->Mi[k];
        const int lvl = ci->lvl;
        const ArLevel L = levels[lvl];
        const uint8_t* img = (lvl == 0) ? src0.base + (size_t)f * src0.fstride : pyr.base + (size_t)f * pyr.fstride + L.off;
        const int pitch = (lvl == 0) ? src0.pitch : L.pitch;
        const uint8_t* px = patch + (size_t)it * DC_PXCAP;
        s_ones[wid][lane] = 0;
        s_tot[wid][lane] = 0;
        __builtin_amdgcn_wave_barrier();
        const int th = ci->th, n = nb + 2;
        for (int i = lane; i < S * S; i += 64) {
            const int y = i / S, xx = i - y * S;
            const int my = (int)__fdiv_rn(__fmul_rn((float)n, (float)y), (float)S);
            const int mx = (int)__fdiv_rn(__fmul_rn((float)n, (float)xx), (float)S);
            // the patch k_decode_warp kept; a warp size whose patch does not fit is warped again (per pixel three f64 multiply-adds,
            // an f64 division and four byte loads)
            const int v = keep_px ? (int)px[i] : dc_warp_pixel(Mi, y, xx, img, pitch, L.w, L.h);
            if (v > th) atomicAdd(&s_ones[wid][my * n + mx], 1);
            atomicAdd(&s_tot[wid][my * n + mx], 1);
        }
        __builtin_amdgcn_wave_barrier();
        // cell bits (n*n <= 64: one lane per cell), border must be black
        const int cy = lane / n, cx = lane - cy * n;
        const bool incell = lane < n * n;
        const int bit = incell && (s_ones[wid][lane] > s_tot[wid][lane] / 2);
        const bool border = incell && (cy == 0 || cy == n - 1 || cx == 0 || cx == n - 1);
        s_bits[wid][lane] = (uint8_t)bit;
        const bool bad = __ballot(border && bit) != 0ull;
        __builtin_amdgcn_wave_barrier();
        if (lane < 4) {
            // code of the inner nb x nb matrix rotated `lane` times: rotate(out(i,j) = in(nb-1-j, i)) applied lane times
            unsigned long long v = 0;
            int bpos = 0;
            for (int y = nb - 1; y >= 0; y--)
                for (int xx = nb - 1; xx >= 0; xx--) {
                    int yy = y, xc = xx;
                    for (int t = 0; t < lane; t++) { const int ny = nb - 1 - xc, nx = yy; yy = ny; xc = nx; }
                    v |= (unsigned long long)s_bits[wid][(yy + 1) * n + (xc + 1)] << bpos++;
                }
            s_ids[wid][lane] = v;
        }
        __builtin_amdgcn_wave_barrier();
        int id = -1, nrot = 0;
        if (!bad && s_ids[wid][0] != 0) {
            // first rotation whose code is in the dictionary; id = first index holding that code (map.insert semantics)
            // one pass over the dictionary for all four rotations: key = rotation << 24 | index, the minimum is the answer
            {
                const unsigned long long w0 = s_ids[wid][0], w1 = s_ids[wid][1], w2 = s_ids[wid][2], w3 = s_ids[wid][3];
                int best = 0x7fffffff;
                for (int i = lane; i < ncodes; i += 64) {
                    const unsigned long long c = codes[i];
                    const int k = c == w0 ? i : c == w1 ? (1 << 24) | i : c == w2 ? (2 << 24) | i : c == w3 ? (3 << 24) | i : 0x7fffffff;
                    best = min(best, k);
                }
                best = wave_min(best);
                if (best != 0x7fffffff) { id = best & 0xffffff; nrot = best >> 24; }
            }
            if (id < 0 && max_corr > 0) {
                // error correction (dictionary_based.cpp:1423-1560): the dictionary's code map in ascending code order, the four
                // rotations inside, first entry closer than int(tau * error_correction_rate) bits wins.  scodes / sids = that map.
                int best = 0x7fffffff;
                for (int i = lane; i < nsorted && best == 0x7fffffff; i += 64) {
                    const unsigned long long c = scodes[i];
                    for (int rr = 0; rr < 4; rr++)
                        if (__popcll(c ^ s_ids[wid][rr]) < max_corr) { best = i * 4 + rr; break; }
                }
                best = wave_min(best);
                if (best != 0x7fffffff) { id = sids[best >> 2]; nrot = best & 3; }
            }
        }
        if (lane == 0) { res[0] = id; res[1] = nrot; }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------- finalize ------------
// interpolate2Dline + getCrossPoint in double (sums of integer coordinates are exact, so the reduction order is free)
__device__ void fit_line(double n, double su, double sv, double suu, double suv, bool xdom, float line[3])
{
    const double det = n * suu - su * su;
    double pa, qa;
    if (fabs(det) > 1e-9 * fmax(1.0, n * suu)) {
        pa = (n * suv - su * sv) / det;
        qa = (sv * suu - su * suv) / det;
    } else {
        const double um = su / n, vm = sv / n;
        pa = vm * um / (um * um + 1.0);
        qa = vm / (um * um + 1.0);
    }
    if (xdom) { line[0] = (float)pa; line[1] = -1.f; line[2] = (float)qa; }
    else { line[0] = -1.f; line[1] = (float)pa; line[2] = (float)qa; }
}

// One workgroup (256 threads) per frame.
__global__ __launch_bounds__(256) void k_finalize(const ArRect* __restrict__ rects, int rect_cap,
                                                  const int32_t* __restrict__ cand_idx,
                                                  const int32_t* __restrict__ ncand, const int32_t* __restrict__ result,
                                                  const uint32_t* __restrict__ pool, size_t pool_fstride,
                                                  orbfe_marker* __restrict__ out, int out_cap,
                                                  int32_t* __restrict__ n_out, int refine_lines,
                                                  int32_t* __restrict__ out_src /*per output slot: its rectangle (contour)*/,
                                                  int32_t* __restrict__ wctr)
{
    __builtin_amdgcn_s_setprio(2); // latency-bound: its few waves go first when a VALU-bound kernel shares the CU
    __shared__ int s_id[AR_MAX_RECTS], s_src[AR_MAX_RECTS], s_rot[AR_MAX_RECTS], s_per[AR_MAX_RECTS], s_rm[AR_MAX_RECTS];
    __shared__ float s_c[AR_MAX_RECTS][4][2];
    __shared__ int s_n, s_written;
    __shared__ int s_slot[AR_MAX_RECTS];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int nc = ncand[f];
    const ArRect* R = rects + (size_t)f * rect_cap;
    if (f == 0 && tid == 0) wctr[0] = 0; // every consumer of this batch's decode work list is done: ready for the next batch
    for (int i = tid; i < AR_MAX_RECTS; i += blockDim.x) out_src[(size_t)f * AR_MAX_RECTS + i] = -1; // slot holds no marker (the
    if (tid == 0) {                                                                                    // workgroup barriers below order it)
        // detected markers in candidate order, corners rotated by 4 - nRot (:6723-6823), then a stable sort by id
        int m = 0;
        for (int s = 0; s < nc; s++) {
            const int id = result[((size_t)f * rect_cap + s) * 2], rot = result[((size_t)f * rect_cap + s) * 2 + 1];
            if (id < 0) continue;
            s_id[m] = id; s_src[m] = cand_idx[(size_t)f * rect_cap + s]; s_rot[m] = rot;
            m++;
        }
        for (int i = 1; i < m; i++) { // insertion sort: stable
            const int id = s_id[i], sr = s_src[i], ro = s_rot[i];
            int j = i - 1;
            while (j >= 0 && s_id[j] > id) { s_id[j + 1] = s_id[j]; s_src[j + 1] = s_src[j]; s_rot[j + 1] = s_rot[j]; j--; }
            s_id[j + 1] = id; s_src[j + 1] = sr; s_rot[j + 1] = ro;
        }
        for (int i = 0; i < m; i++) {
            const ArRect& r = R[s_src[i]];
            // std::rotate(begin, begin + 4 - nRot, end): new[k] = old[(k + 4 - nRot) % 4]
            for (int k = 0; k < 4; k++) {
                const int o = (k + 4 - s_rot[i]) & 3;
                s_c[i][k][0] = r.c[o][0];
                s_c[i][k][1] = r.c[o][1];
            }
            s_per[i] = ar_perimeter(s_c[i]);
            s_rm[i] = 0;
        }
        for (int i = 0; i < m - 1; i++) // same id: keep the larger perimeter (:8153-8365)
            for (int j = i + 1; j < m && !s_rm[i]; j++)
                if (s_id[i] == s_id[j]) {
                    if (s_per[i] < s_per[j]) s_rm[i] = 1;
                    else s_rm[j] = 1;
                }
        int w = 0;
        for (int i = 0; i < m; i++) { s_slot[i] = w; w += s_rm[i] ? 0 : 1; } // output order = id order, removed ones skipped
        s_written = w;
        s_n = m;
    }
    __syncthreads();
    const int m = s_n;
    // ---- refineCornerWithContourLines (:8978-10044), one WAVE per marker.  The reference walks the four point runs
    // between the contour points nearest to the corners and accumulates sums of integer coordinates in double: those
    // sums are exact, so they are formed in parallel over the run and reduced on the DPP network.
    const int lane = tid & 63, wid = tid >> 6;
    for (int i = wid; i < m; i += 4) {
        if (s_rm[i]) continue;
        if (lane == 0 && s_slot[i] < AR_MAX_RECTS) out_src[(size_t)f * AR_MAX_RECTS + s_slot[i]] = s_src[i];
        if (!refine_lines) { // CORNER_NONE: the rotated approxPolyDP corners as they are (:8634-8701 does nothing)
            const int slot = s_slot[i];
            if (lane == 0 && slot < out_cap) {
                orbfe_marker mk;
                mk.id = s_id[i];
                for (int k = 0; k < 4; k++) { mk.corners[k][0] = s_c[i][k][0]; mk.corners[k][1] = s_c[i][k][1]; }
                out[(size_t)f * out_cap + slot] = mk;
            }
            continue;
        }
        const ArRect& r = R[s_src[i]];
        const uint32_t* P = pool + (size_t)f * pool_fstride + r.off;
        const int len = r.len;
        // nearest contour point to each corner, first minimum wins
        int ci[4];
        {
            unsigned long long b[4] = {~0ull, ~0ull, ~0ull, ~0ull};
            for (int j = lane; j < len; j += 64) {
                const uint32_t v = P[j];
                const float x = (float)(v & 0xffff), y = (float)(v >> 16);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float dx = x - s_c[i][k][0], dy = y - s_c[i][k][1];
                    const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
                    b[k] = key < b[k] ? key : b[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) ci[k] = (int)(wave_min_u64(b[k]) & 0xffffffffu);
        }
        bool inverse;
        if ((ci[1] > ci[0]) && (ci[2] > ci[1] || ci[2] < ci[0])) inverse = false;
        else if (ci[2] > ci[1] && ci[2] < ci[0]) inverse = false;
        else inverse = true;
        float line[4][3];
#pragma unroll
        for (int l = 0; l < 4; l++) {
            // The points the reference loop visits (for (j = ci[l]; j != target; j += inc) with its wrap rules, incl. the
            // quirks: going forward a wrap that lands on target == 0 still takes point 0; going backward index 0 is
            // replaced by len-1 before it is read, and a wrap that lands on target == len-1 still takes that point):
            // up to two index ranges [a0, a1) and [b0, b1) plus at most one extra point.
            const int j0 = ci[l], target = ci[(l + 1) & 3];
            int a0 = 0, a1 = 0, b0 = 0, b1 = 0, extra = -1;
            if (j0 != target) {
                if (!inverse) {
                    if (target > j0) { a0 = j0; a1 = target; }
                    else { a0 = j0; a1 = len; b0 = 0; b1 = target; if (target == 0) extra = 0; }
                } else {
                    if (target < j0) { a0 = target + 1; a1 = j0 + 1; }
                    else {
                        a0 = 1; a1 = j0 + 1; // (empty when j0 == 0)
                        if (target == len - 1) extra = len - 1;
                        else { b0 = target + 1; b1 = len; }
                    }
                }
            }
            double su = 0, sv = 0, sxx = 0, syy = 0, sxy = 0;
            int cnt = 0, minX = 0x7fffffff, maxX = -1, minY = 0x7fffffff, maxY = -1;
            auto take = [&](int j) {
                const uint32_t v = P[j];
                const int xi = (int)(v & 0xffff), yi = (int)(v >> 16);
                const double x = xi, y = yi;
                minX = min(minX, xi); maxX = max(maxX, xi); minY = min(minY, yi); maxY = max(maxY, yi);
                su += x; sv += y; sxx += x * x; syy += y * y; sxy += x * y;
                cnt++;
            };
            for (int j = a0 + lane; j < a1; j += 64) take(j);
            for (int j = b0 + lane; j < b1; j += 64) take(j);
            if (extra >= 0 && lane == 0) take(extra);
            cnt = wave_sum(cnt);
            if (cnt == 0) { line[l][0] = line[l][1] = line[l][2] = 0.f; }
            else {
                su = wave_sum_f64(su); sv = wave_sum_f64(sv); sxx = wave_sum_f64(sxx); syy = wave_sum_f64(syy); sxy = wave_sum_f64(sxy);
                minX = wave_min(minX); maxX = wave_max(maxX); minY = wave_min(minY); maxY = wave_max(maxY);
                const bool xdom = ((float)maxX - (float)minX > (float)maxY - (float)minY);
                if (xdom) fit_line((double)cnt, su, sv, sxx, sxy, true, line[l]);
                else fit_line((double)cnt, sv, su, syy, sxy, false, line[l]);
            }
        }
        const int slot = s_slot[i];
        if (lane == 0 && slot < out_cap) {
            orbfe_marker mk;
            mk.id = s_id[i];
#pragma unroll
            for (unsigned k = 0; k < 4; k++) {
                const float* l1 = line[(k - 1) % 4];
                const float* l2 = line[k];
                const double a = l1[0], b = l1[1], c = l2[0], d = l2[1], e = -(double)l1[2], g = -(double)l2[2];
                const double det = a * d - b * c;
                float x = 0.f, y = 0.f;
                if (det != 0.0) { x = (float)((e * d - b * g) / det); y = (float)((a * g - e * c) / det); }
                mk.corners[k][0] = x;
                mk.corners[k][1] = y;
            }
            out[(size_t)f * out_cap + slot] = mk;
        }
    }
    if (tid == 0) n_out[f] = s_written < out_cap ? s_written : out_cap;
}

} // namespace orbfe
