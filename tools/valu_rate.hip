// Issue cost of the VALU instructions the hot kernels are made of, on gfx950 (run ON the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -w -o build/valu_rate tools/valu_rate.hip && build/valu_rate
// 8 waves per SIMD, every wave runs 8 independent chains of the same instruction; prints cycles per wave-instruction and SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 2048
#define OPS(X)                                                                                                     \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %2")                                                                    \
    X(1, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %2")                                                              \
    X(2, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %2")                                                            \
    X(3, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %2")                                                              \
    X(4, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %2, %0")                                                        \
    X(5, "v_mad_u64_u32", "v_mad_u64_u32 %1, vcc, %0, %2, %1")                                                   \
    X(6, "v_pk_mul_f32", "v_pk_mul_f32 %1, %1, %1")                                                              \
    X(7, "v_mul_f32", "v_mul_f32 %0, %0, %0")                                                                    \
    X(8, "v_fma_f64", "v_fma_f64 %1, %1, %1, %1")                                                                \
    X(9, "v_perm_b32", "v_perm_b32 %0, %0, %2, %3")                                                              \
    X(10, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %2, 1")                                                    \
    X(11, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %2")                                                             \
    X(12, "v_pk_max_u16", "v_pk_max_u16 %0, %0, %2")                                                             \
    X(13, "v_pk_sub_u16 clamp", "v_pk_sub_u16 %0, %0, %2 clamp")                                                 \
    X(14, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %2")                                                             \
    X(15, "v_pk_minimum3_f16", "v_pk_minimum3_f16 %0, %0, %2, %3")                                               \
    X(16, "v_pk_maximum3_f16", "v_pk_maximum3_f16 %0, %0, %2, %3")                                               \
    X(17, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %2")                                                             \
    X(18, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %2")                                                             \
    X(19, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %0, %2, %0")                                                       \
    X(20, "v_dot2_u32_u16", "v_dot2_u32_u16 %0, %0, %2, %0")                                                     \
    X(21, "v_and_b32", "v_and_b32 %0, %0, %2")                                                                   \
    X(22, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0")                                                            \
    X(23, "v_bfe_u32", "v_bfe_u32 %0, %0, 1, 31")                                                                \
    X(24, "v_and_or_b32", "v_and_or_b32 %0, %0, %2, %3")                                                         \
    X(25, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 1, %2")                                                        \
    X(26, "v_add3_u32", "v_add3_u32 %0, %0, %2, %3")                                                             \
    X(27, "v_min_u32", "v_min_u32 %0, %0, %2")                                                                   \
    X(28, "v_min3_u32", "v_min3_u32 %0, %0, %2, %3")                                                             \
    X(29, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0")                                                               \
    X(30, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %0")                                                         \
    X(31, "v_rndne_f32", "v_rndne_f32 %0, %0")                                                                   \
    X(32, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0")                                                               \
    X(33, "v_fma_f32", "v_fma_f32 %0, %0, %0, %0")                                                               \
    X(34, "v_add_f32", "v_add_f32 %0, %0, %0")                                                                   \
    X(35, "v_add_f64", "v_add_f64 %1, %1, %1")                                                                   \
    X(36, "v_mul_f64", "v_mul_f64 %1, %1, %1")                                                                   \
    X(37, "v_cmp + v_cndmask", "v_cmp_lt_u32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc")                    \
    X(38, "v_min_u16", "v_min_u16 %0, %0, %2")                                                                   \
    X(39, "v_mov_b32 dpp", "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")                        \
    X(40, "v_max_i32 dpp", "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")                   \
    X(41, "v_bfi_b32", "v_bfi_b32 %0, %0, %2, %3")                                                               \
    X(42, "v_sub_u32", "v_sub_u32 %0, %0, %2")                                                                   \
    X(43, "v_mbcnt_lo", "v_mbcnt_lo_u32_b32 %0, %2, %0")                                                         \
    X(44, "v_ffbl/ctz", "v_ffbl_b32 %0, %0")                                                                     \
    X(45, "v_bcnt", "v_bcnt_u32_b32 %0, %0, %2")                                                                 \
    X(46, "v_sad_u8", "v_sad_u8 %0, %0, %2, %3")                                                                 \
    X(47, "v_lshrrev_b64", "v_lshrrev_b64 %1, 1, %1")                                                            \
    X(48, "v_xad_u32", "v_xad_u32 %0, %0, %2, %3")                                                               \
    X(49, "v_med3_i32", "v_med3_i32 %0, %0, %2, %3")

template <int OP> __global__ void k(unsigned* out, unsigned a0, unsigned b0, unsigned c0)
{
    unsigned x[8];
    unsigned long long y[8];
    for (int i = 0; i < 8; i++) { x[i] = a0 + threadIdx.x + i; y[i] = x[i]; }
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#define X(n, name, txt) if (OP == n) asm volatile(txt : "+v"(x[i]), "+v"(y[i]) : "v"(b0), "v"(c0) : "vcc");
            OPS(X)
#undef X
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + (unsigned)y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, unsigned* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d, 3u, 5u, 7u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d, 3u, 5u, 7u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)wgs * 4 / 1024.0 * REP * 8;
    printf("%-20s %6.2f cycles\n", name, ms * 1e-3 * 2.4e9 / insts_per_simd);
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
#define X(n, name, txt) run<n>(name, d);
    OPS(X)
#undef X
    return 0;
}
