// wave_dpp.hpp -- wave64 scans and reductions on the DPP network of gfx950.
//
// __shfl_xor / __shfl_up compile to ds_bpermute_b32: every butterfly step is a round trip through the LDS crossbar
// (~50+ cycles of latency, twice that for 64-bit values), which dominates the single-wave, latency-bound kernels here
// (quadtree, approxPolyDP, SearchForInitialization's serial phase).  The DPP row operations move data between lanes
// inside the VALU: a Kogge-Stone scan over each row of 16 lanes (row_shr 1, 2, 4, 8), then row_bcast15 carries the
// row totals into rows 1 and 3 and row_bcast31 the half-wave total into rows 2 and 3.  Six dependent VALU ops give
// an inclusive scan; lane 63 holds the reduction.
//
// EVERY LANE OF THE WAVE MUST BE ACTIVE at the call (wave-uniform control flow); a lane that should not take part
// passes the operation's identity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbfe {

// Index of the wave inside its workgroup as a SCALAR.  threadIdx.x >> 6 is the same in all lanes, but the compiler only knows
// that it derives from a per-lane value: everything computed from it (which cell / keypoint / candidate the wave owns, its
// geometry, loop bounds) is then kept in vector registers, addressed with vector arithmetic, and every loop and branch on it
// is compiled as a divergent one with exec-mask bookkeeping.  v_readfirstlane moves it -- and all that follows -- to the scalar unit.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

#define ORBFE_DPP(old_, v_, ctrl_, rmask_) __builtin_amdgcn_update_dpp((old_), (v_), (ctrl_), (rmask_), 0xf, false)
#define ORBFE_DPP_STEPS(STEP)                                                                  \
    STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) /* row_shr 1 2 4 8 */ \
    STEP(0x142, 0xa) STEP(0x143, 0xc)                                    /* row_bcast15, row_bcast31 */

__device__ __forceinline__ int wave_incl_scan_add(int v)
{
#define ORBFE_STEP(ctrl, rmask) v += ORBFE_DPP(0, v, ctrl, rmask);
    ORBFE_DPP_STEPS(ORBFE_STEP)
#undef ORBFE_STEP
    return v;
}
// inclusive scan inside each row of 16 lanes
__device__ __forceinline__ int row16_incl_scan_add(int v)
{
    v += ORBFE_DPP(0, v, 0x111, 0xf); v += ORBFE_DPP(0, v, 0x112, 0xf); v += ORBFE_DPP(0, v, 0x114, 0xf); v += ORBFE_DPP(0, v, 0x118, 0xf);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_add(v), 63); }

__device__ __forceinline__ int wave_max(int v)
{
#define ORBFE_STEP(ctrl, rmask) v = max(v, ORBFE_DPP((int)0x80000000, v, ctrl, rmask));
    ORBFE_DPP_STEPS(ORBFE_STEP)
#undef ORBFE_STEP
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min(int v)
{
#define ORBFE_STEP(ctrl, rmask) v = min(v, ORBFE_DPP(0x7fffffff, v, ctrl, rmask));
    ORBFE_DPP_STEPS(ORBFE_STEP)
#undef ORBFE_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#define ORBFE_STEP(ctrl, rmask)                                                                              \
    {                                                                                                        \
        const unsigned lo = (unsigned)ORBFE_DPP(0, (int)(unsigned)v, ctrl, rmask);                           \
        const unsigned hi = (unsigned)ORBFE_DPP(0, (int)(unsigned)(v >> 32), ctrl, rmask);                   \
        const unsigned long long t = ((unsigned long long)hi << 32) | lo;                                    \
        v = t > v ? t : v;                                                                                   \
    }
    ORBFE_DPP_STEPS(ORBFE_STEP)
#undef ORBFE_STEP
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) { return ~wave_max_u64(~v); }

// sum of doubles (exact, hence order-free, when every partial sum is an integer below 2^53)
__device__ __forceinline__ double wave_sum_f64(double v)
{
#define ORBFE_STEP(ctrl, rmask)                                                                              \
    {                                                                                                        \
        const int lo = ORBFE_DPP(0, __double2loint(v), ctrl, rmask);                                         \
        const int hi = ORBFE_DPP(0, __double2hiint(v), ctrl, rmask);                                         \
        v += __hiloint2double(hi, lo);                                                                       \
    }
    ORBFE_DPP_STEPS(ORBFE_STEP)
#undef ORBFE_STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

} // namespace orbfe
