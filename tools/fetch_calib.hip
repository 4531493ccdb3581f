// FETCH_SIZE / WRITE_SIZE calibration (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reads half the bytes of a 16 B/lane
// streaming read; other widths and WRITE_SIZE are uncalibrated -- "calibrate on a known byte count in your own access pattern").
// Every kernel streams the same NBYTES (> the 256 MiB Infinity Cache) once, coalesced, at 1 / 4 / 8 / 16 bytes per lane; the
// counters of each kernel divided by NBYTES are the correction factors tools/make_profiles.py applies per access width.
//   hipcc --offload-arch=gfx950 -O3 -o build/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- build/fetch_calib     (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <class T> __global__ void calib_read(const T* __restrict__ p, size_t n, T* __restrict__ sink, unsigned flag)
{
    T acc{};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const unsigned char* a = (const unsigned char*)&v;
        unsigned char* b = (unsigned char*)&acc;
        for (unsigned k = 0; k < sizeof(T); k++) b[k] ^= a[k];
    }
    const unsigned char* b = (const unsigned char*)&acc;
    unsigned x = 0;
    for (unsigned k = 0; k < sizeof(T); k++) x |= b[k];
    if (x == flag) *sink = acc; // flag = 0x1ff can never equal an OR of bytes, but only the host knows: keeps the loads alive
}
template <class T> __global__ void calib_write(T* __restrict__ p, size_t n, T v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// the explicit instantiations give every width its own kernel name in the trace
template __global__ void calib_read<uint8_t>(const uint8_t*, size_t, uint8_t*, unsigned);
template __global__ void calib_read<uint32_t>(const uint32_t*, size_t, uint32_t*, unsigned);
template __global__ void calib_read<uint2>(const uint2*, size_t, uint2*, unsigned);
template __global__ void calib_read<uint4>(const uint4*, size_t, uint4*, unsigned);

int main()
{
    const size_t NBYTES = (size_t)1 << 30;
    void *buf, *sink;
    if (hipMalloc(&buf, NBYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(buf, 0x37, NBYTES);
    hipDeviceSynchronize();
    const int grid = 256 * 8, block = 256;
    volatile unsigned flag_v = 0x1ff;
    const unsigned flag = flag_v;
    for (int rep = 0; rep < 3; rep++) {
        calib_read<uint8_t><<<grid, block>>>((const uint8_t*)buf, NBYTES, (uint8_t*)sink, flag);
        calib_read<uint32_t><<<grid, block>>>((const uint32_t*)buf, NBYTES / 4, (uint32_t*)sink, flag);
        calib_read<uint2><<<grid, block>>>((const uint2*)buf, NBYTES / 8, (uint2*)sink, flag);
        calib_read<uint4><<<grid, block>>>((const uint4*)buf, NBYTES / 16, (uint4*)sink, flag);
        calib_write<uint8_t><<<grid, block>>>((uint8_t*)buf, NBYTES, (uint8_t)rep);
        calib_write<uint32_t><<<grid, block>>>((uint32_t*)buf, NBYTES / 4, (uint32_t)rep);
        calib_write<uint2><<<grid, block>>>((uint2*)buf, NBYTES / 8, make_uint2(rep, rep));
        calib_write<uint4><<<grid, block>>>((uint4*)buf, NBYTES / 16, make_uint4(rep, rep, rep, rep));
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("calibration bytes per kernel: %zu\n", NBYTES);
    return 0;
}
