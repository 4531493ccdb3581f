// orbfe_common.hpp -- shared host-side plumbing of liborbfe.so (error state, HIP checks, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbfe.h"
#include "../../include/orbfe_math.h"

namespace orbfe {

extern thread_local char g_err[512];
// Launch ablation ("what does this kernel cost the concurrent pipeline") exists only in the diagnosis build
// (-DORBFE_ABLATION, tools/ablate.sh -> build/liborbfe_ablate.so): bit masks of launches to leave out, set with
// orbfe_debug_control "orb_skip" / "aruco_skip"; results are garbage while a bit is set.  The shipped library has no
// switch that skips work: there the macros are the constant 0 and orbfe_debug_control rejects the keys.
#ifdef ORBFE_ABLATION
extern int g_orb_skip, g_aruco_skip;
#define ORBFE_SKIP_ORB(bit) (::orbfe::g_orb_skip & (bit))
#define ORBFE_SKIP_ARUCO(bit) (::orbfe::g_aruco_skip & (bit))
// sensitivity instead of ablation: mask bit << 8 launches the kernel TWICE (same results; removing a kernel also removes the work of
// everything downstream of it, doubling one does not) -- "how much does the step grow per microsecond of this kernel's work"
#define ORBFE_REPS_ORB(bit) (ORBFE_SKIP_ORB(bit) ? 0 : (::orbfe::g_orb_skip & ((bit) << 8)) ? 2 : 1)
#define ORBFE_REPS_ARUCO(bit) (ORBFE_SKIP_ARUCO(bit) ? 0 : (::orbfe::g_aruco_skip & ((bit) << 8)) ? 2 : 1)
#else
#define ORBFE_SKIP_ORB(bit) 0
#define ORBFE_SKIP_ARUCO(bit) 0
#define ORBFE_REPS_ORB(bit) 1
#define ORBFE_REPS_ARUCO(bit) 1
#endif

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define ORBFE_HIP(call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return ::orbfe::fail(ORBFE_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                                    \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, size) instead of before every launch: it is a driver call of
// a few microseconds, and it is not a stream operation (nothing for a captured graph to replay).
int ensure_dyn_lds(const void* fn, size_t bytes);

// Select the device, or report that there is none (no CPU fallback exists in this library).
int use_device(int device);

// Grow-only device allocation.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int ensure(size_t need)
    {
        if (need <= bytes) return ORBFE_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 8 + 256;
        ORBFE_HIP(hipMalloc(&p, want));
        bytes = want;
        return ORBFE_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

// Scratch of the entry points that have no handle (matching, poses, keyframe records): one workspace per calling thread,
// HIP device and stream.  A buffer allocated on one GPU is never handed to a kernel on another, two asynchronous calls
// of one thread on different streams never share scratch, and the buffers are released when the thread exits.
template <class W> class ThreadWorkspaces {
    struct Slot {
        int device;
        hipStream_t stream;
        W* w;
    };
    std::vector<Slot> slots;

public:
    // workspace of (this thread, the current device, stream); the caller has selected the device already
    W& get(hipStream_t stream = nullptr)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        for (auto& s : slots)
            if (s.device == dev && s.stream == stream) return *s.w;
        slots.push_back(Slot{dev, stream, new W()});
        return *slots.back().w;
    }
    ~ThreadWorkspaces()
    {
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        for (auto& s : slots) {
            (void)hipSetDevice(s.device);
            delete s.w;
        }
        if (have) (void)hipSetDevice(cur);
    }
};

// Event timing of individual launches (profiling aid; no events are recorded unless enabled).  A mark closes the
// interval that the previous mark ON THE SAME STREAM opened; `opens_only` marks (the first one of a stream) report nothing.
struct KernelTimer {
    bool enabled = false;
    std::vector<hipEvent_t> ev;
    std::vector<hipStream_t> streams;
    std::vector<char> opens;
    size_t used = 0;
    void begin() { used = 0; streams.clear(); opens.clear(); }
    void mark(hipStream_t s, const char* /*name: documentation at the call site*/, bool opens_only = false)
    {
        if (!enabled) return;
        if (used == ev.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            ev.push_back(e);
        }
        (void)hipEventRecord(ev[used++], s);
        streams.push_back(s);
        opens.push_back(opens_only || streams.size() == 1);
    }
    int collect(float* out_us, int capacity)
    {
        if (!enabled || used < 2) return 0;
        for (size_t i = 0; i < used; i++) (void)hipEventSynchronize(ev[i]);
        int n = 0;
        for (size_t i = 1; i < used && n < capacity; i++) {
            if (opens[i]) continue;
            size_t j = i;
            while (j-- > 0)
                if (streams[j] == streams[i]) break;
            float ms = 0;
            if (j < i) (void)hipEventElapsedTime(&ms, ev[j], ev[i]);
            out_us[n++] = ms * 1000.f;
        }
        return n;
    }
    ~KernelTimer()
    {
        for (auto e : ev) (void)hipEventDestroy(e);
    }
};

} // namespace orbfe
