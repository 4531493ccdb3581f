// tools/h2d_bw.hip -- what the host link of this box can do: page-locked host memory -> HBM, 1 GiB a transfer.
//   hipcc --offload-arch=gfx950 -O2 tools/h2d_bw.hip -o build/h2d_bw && build/h2d_bw [MiB]
// Variants: host memory from hipHostMalloc (default = coherent), hipHostMallocNonCoherent, hipHostMallocWriteCombined, and ordinary
// pages bound to each NUMA node of the box (mbind) and registered (hipHostRegister); copied by the DMA engines (hipMemcpyAsync, and
// hipMemcpy2DAsync with the row geometry orbfe_pipeline_step_host uses) and by a blit kernel that reads the mapped host pointer with
// 16-byte loads.  D2H the same way for the default allocation.  One line per variant: best and median of 7 transfers, GB/s (1e9).
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_blit(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static void report(const char* what, std::vector<double>& gbps)
{
    std::sort(gbps.begin(), gbps.end());
    printf("%-64s best %6.2f  median %6.2f GB/s\n", what, gbps.back(), gbps[gbps.size() / 2]);
    fflush(stdout);
}

template <class F>
static void timeit(const char* what, size_t bytes, hipStream_t s, F enqueue)
{
    std::vector<double> g;
    for (int it = 0; it < 8; it++) {
        HIP(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        enqueue();
        HIP(hipStreamSynchronize(s));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (it) g.push_back(bytes / dt / 1e9); // the first transfer warms up
    }
    report(what, g);
}

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    HIP(hipSetDevice(0));
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint8_t* d = nullptr;
    HIP(hipMalloc(&d, bytes));
    hipDeviceProp_t prop;
    HIP(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %zu MiB per transfer\n", prop.name, bytes >> 20);

    struct Alloc { const char* name; unsigned flags; };
    const Alloc allocs[] = {{"hipHostMalloc default (coherent)", hipHostMallocDefault},
                            {"hipHostMalloc non-coherent", hipHostMallocNonCoherent},
                            {"hipHostMalloc write-combined", hipHostMallocWriteCombined},
                            {"hipHostMalloc portable | mapped", hipHostMallocPortable | hipHostMallocMapped}};
    for (const Alloc& a : allocs) {
        uint8_t* h = nullptr;
        if (hipHostMalloc((void**)&h, bytes, a.flags) != hipSuccess) { printf("%-64s not available\n", a.name); (void)hipGetLastError(); continue; }
        memset(h, 1, bytes);
        std::string n = std::string("H2D hipMemcpyAsync, ") + a.name;
        timeit(n.c_str(), bytes, s, [&] { HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); });
        if (a.flags == hipHostMallocDefault) {
            // the geometry of orbfe_pipeline_step_host: rows of 640 bytes, source rows 640 apart, destination rows 640 apart
            timeit("H2D hipMemcpy2DAsync 640-byte rows (contiguous both sides)", bytes, s,
                   [&] { HIP(hipMemcpy2DAsync(d, 640, h, 640, 640, bytes / 640, hipMemcpyHostToDevice, s)); });
            timeit("H2D hipMemcpy2DAsync 1280-byte rows into a 1280-byte pitch", bytes, s,
                   [&] { HIP(hipMemcpy2DAsync(d, 1280, h, 1280, 1280, bytes / 1280, hipMemcpyHostToDevice, s)); });
            timeit("D2H hipMemcpyAsync, hipHostMalloc default", bytes, s, [&] { HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); });
            hipStream_t s2;
            HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            timeit("H2D two halves on two streams (two DMA engines)", bytes, s, [&] {
                HIP(hipMemcpyAsync(d, h, bytes / 2, hipMemcpyHostToDevice, s));
                HIP(hipMemcpyAsync(d + bytes / 2, h + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s2));
                HIP(hipStreamSynchronize(s2));
            });
            HIP(hipStreamDestroy(s2));
        }
        void* hd = nullptr;
        if (hipHostGetDevicePointer(&hd, h, 0) == hipSuccess && hd) {
            n = std::string("H2D blit kernel (16-byte loads of the mapped pointer), ") + a.name;
            timeit(n.c_str(), bytes, s, [&] { hipLaunchKernelGGL(k_blit, dim3(1024), dim3(256), 0, s, (const uint4*)hd, (uint4*)d, bytes / 16); });
        } else
            (void)hipGetLastError();
        HIP(hipHostFree(h));
    }
    // ordinary pages on a chosen NUMA node, registered
    for (int node = 0; node < 8; node++) {
        char path[96];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d", node);
        if (access(path, F_OK) != 0) break;
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) break;
        unsigned long mask = 1ul << node;
        const long rc = syscall(SYS_mbind, m, bytes, 2 /* MPOL_BIND */, &mask, sizeof(mask) * 8, 0);
        memset(m, 1, bytes);
        char what[128];
        if (hipHostRegister(m, bytes, hipHostRegisterDefault) == hipSuccess) {
            snprintf(what, sizeof(what), "H2D hipMemcpyAsync, pages on NUMA node %d%s, hipHostRegister", node, rc ? " (mbind refused)" : "");
            timeit(what, bytes, s, [&] { HIP(hipMemcpyAsync(d, m, bytes, hipMemcpyHostToDevice, s)); });
            HIP(hipHostUnregister(m));
        } else {
            (void)hipGetLastError();
            printf("pages on NUMA node %d: hipHostRegister failed\n", node);
        }
        munmap(m, bytes);
    }
    {   // pageable memory, for scale
        std::vector<uint8_t> p(bytes, 1);
        timeit("H2D hipMemcpyAsync, pageable std::vector", bytes, s, [&] { HIP(hipMemcpyAsync(d, p.data(), bytes, hipMemcpyHostToDevice, s)); });
    }
    HIP(hipFree(d));
    return 0;
}
