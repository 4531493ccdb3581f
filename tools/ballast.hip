// tools/ballast.hip -- EXPERIMENT INFRASTRUCTURE, not part of the product: a kernel that occupies ONE resource of the chip for a known
// time, launched next to every step of bench.py (ORBFE_BENCH_BALLAST="<kind>:<iterations>"), to find which resource the pipeline's
// step time is made of: the kind whose added busy time shows up one to one in the step is the binding one.
//   valu  -- independent integer multiply-adds, no memory at all
//   ta    -- 16-byte global loads that always hit the CU's vector cache (one line per quad): the texture addresser / L1 path
//   l2    -- dword loads striding through a 32 MB block: misses in L1, hits in L2
//   lds   -- ds_read_b128 from a block's own 4 KB
//   hbm   -- streams <iterations> MB of a 2 GB block per launch
// Every kind runs 1024 workgroups of 64 lanes -- a wave per SIMD when the chip is otherwise empty -- at the lowest wave priority.
//     hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o build/libballast.so tools/ballast.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {
__global__ __launch_bounds__(64) void k_valu(int iters, uint32_t* sink)
{
    uint32_t a = threadIdx.x, b = blockIdx.x, c = 3, d = 5, e = 7, f = 11, g = 13, h = 17;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a = a * 3u + b; b = b * 5u + c; c = c * 7u + d; d = d * 9u + e; e = e * 11u + f; f = f * 13u + g; g = g * 15u + h; h = h * 17u + a;
        }
    }
    if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x12345u) sink[0] = a;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_ta(int iters, const u32x4* __restrict__ buf, uint32_t* sink)
{
    // lane l reads 16 bytes at (l / 4) * 128 + (l % 4) * 16 (+ a rotating line): a quad = one 64-byte sector, 16 sectors an instruction, 2 KB a wave
    const u32x4* p = buf + (blockIdx.x & 7) * 1024 + (threadIdx.x >> 2) * 8 + (threadIdx.x & 3);
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc += __builtin_nontemporal_load(p + ((i * 8 + k) & 3) * 128);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}
__global__ __launch_bounds__(64) void k_l2(int iters, const uint32_t* __restrict__ buf, uint32_t* sink)
{
    // every lane its own 128-byte line, a new set of lines every load: 8 M dwords = 32 MB
    uint32_t idx = (blockIdx.x * 64u + threadIdx.x) * 32u, acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { acc += buf[idx & ((8u << 20) - 1u)]; idx += 64u * 1024u * 32u + 32u; }
    }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void k_lds(int iters, uint32_t* sink)
{
    __shared__ u32x4 s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = u32x4{(uint32_t)i, 1, 2, 3};
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    uint32_t j = threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { acc += s[j & 255]; j += 64 + (acc.y & 1); }
    }
    if ((acc.x ^ acc.w) == 0x12345u) sink[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_hbm(const u32x4* __restrict__ buf, size_t n16, uint32_t* sink)
{
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(buf + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}
hipStream_t g_stream = nullptr;
void* g_buf = nullptr;
uint32_t* g_sink = nullptr;
hipEvent_t g_e0 = nullptr, g_e1 = nullptr;
} // namespace

extern "C" __attribute__((visibility("default"))) int ballast_launch(const char* kind, int iters)
{
    if (!g_stream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, lo) != hipSuccess) return 1;
        if (hipMalloc(&g_buf, (size_t)2 << 30) != hipSuccess || hipMemset(g_buf, 1, (size_t)2 << 30) != hipSuccess) return 2;
        if (hipMalloc(&g_sink, 64) != hipSuccess) return 3;
        (void)hipEventCreate(&g_e0); (void)hipEventCreate(&g_e1);
        (void)hipDeviceSynchronize();
    }
    (void)hipEventRecord(g_e0, g_stream);
    if (!strcmp(kind, "valu")) hipLaunchKernelGGL(k_valu, dim3(1024), dim3(64), 0, g_stream, iters, g_sink);
    else if (!strcmp(kind, "ta")) hipLaunchKernelGGL(k_ta, dim3(1024), dim3(64), 0, g_stream, iters, (const u32x4*)g_buf, g_sink);
    else if (!strcmp(kind, "l2")) hipLaunchKernelGGL(k_l2, dim3(1024), dim3(64), 0, g_stream, iters, (const uint32_t*)g_buf, g_sink);
    else if (!strcmp(kind, "lds")) hipLaunchKernelGGL(k_lds, dim3(1024), dim3(64), 0, g_stream, iters, g_sink);
    else if (!strcmp(kind, "hbm")) hipLaunchKernelGGL(k_hbm, dim3(2048), dim3(256), 0, g_stream, (const u32x4*)g_buf, ((size_t)(iters < 2048 ? iters : 2048)) << 16, g_sink);   // iterations = MB read, at most the 2 GB block
    else return 4;
    (void)hipEventRecord(g_e1, g_stream);
    return hipGetLastError() == hipSuccess ? 0 : 5;
}
// duration of the last launch in microseconds (waits for it)
extern "C" __attribute__((visibility("default"))) float ballast_last_us()
{
    float ms = 0.f;
    if (!g_e1 || hipEventSynchronize(g_e1) != hipSuccess || hipEventElapsedTime(&ms, g_e0, g_e1) != hipSuccess) return -1.f;
    return ms * 1000.f;
}
