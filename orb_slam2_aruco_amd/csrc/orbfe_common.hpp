// orbfe_common.hpp -- shared host-side plumbing of liborbfe.so (error state, HIP checks, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
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

// Grow-only page-locked host buffer: the staging area of the host-pointer entry points.  Their results come back as a few asynchronous
// copies into it behind the kernels and ONE stream synchronisation, instead of one blocking hipMemcpy (a round trip of 20 - 40 us)
// per output array.
struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { release(); }
    int ensure(size_t need)
    {
        if (need <= bytes) return ORBFE_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        const size_t want = need + need / 8 + 4096;
        ORBFE_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        bytes = want;
        return ORBFE_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

// The small results of a host-pointer call (counts, flags, a few KB of markers, 60 KB of keypoints) go to the caller's page-locked
// staging buffer by ONE launch that writes the mapped host memory, instead of one hipMemcpyAsync per array: the runtime turns every
// small device-to-host copy into a blit kernel of its own (rocprofv3: twelve __amd_rocclr_copyBuffer launches per frame of the drop-in
// path, ~5 us each plus the gaps between them).  Items are 4-byte aligned, sizes multiples of 4.
struct PackItem { const uint32_t* src; uint32_t* dst; uint32_t words; };
struct PackList { PackItem it[8]; int n; };
template <int TAG>
__global__ __launch_bounds__(256) void k_pack_out(PackList L)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, nt = gridDim.x * 256u;
    for (int k = 0; k < L.n; k++)
        for (uint32_t i = t; i < L.it[k].words; i += nt) L.it[k].dst[i] = L.it[k].src[i];
}
struct OutPack {
    PackList L{};
    size_t total = 0;
    bool bad = false;   // an item that does not fit the contract: flush() fails instead of dropping bytes or writing past the list
    void add(void* host_dst, const void* dev_src, size_t bytes)
    {
        if (!bytes) return;
        if (L.n >= (int)(sizeof(L.it) / sizeof(L.it[0])) || (bytes & 3) || ((uintptr_t)host_dst & 3) || ((uintptr_t)dev_src & 3) ||
            bytes / 4 > 0xffffffffull) { bad = true; return; }
        L.it[L.n++] = PackItem{(const uint32_t*)dev_src, (uint32_t*)host_dst, (uint32_t)(bytes / 4)};
        total += bytes;
    }
    // host_dst pointers are page-locked memory of this process (hipHostMalloc: the same address on the device)
    template <int TAG> int flush(hipStream_t s)
    {
        if (bad) { bad = false; L.n = 0; total = 0; return fail(ORBFE_ERR_INVALID, "OutPack: more than 8 items, or an item that is not 4-byte aligned / sized"); }
        if (!L.n) return ORBFE_OK;
        const int wgs = (int)std::min<size_t>(32, (total / 4 + 1023) / 1024 + 1);
        hipLaunchKernelGGL(k_pack_out<TAG>, dim3(wgs), dim3(256), 0, s, L);
        ORBFE_HIP(hipGetLastError());
        L.n = 0; total = 0;
        return ORBFE_OK;
    }
};

// ---- a detector paired with an extractor (orbfe_extractor_pair_detector): the drop-in path calls ORBextractor::operator() and
// MarkerDetector::detect on the SAME image one after the other (Frame.cc:91 -> :142); the extractor's call uploads the image once
// and starts the detector on it on the detector's own stream, next to its own launches; the detector's call finds its work done if
// it is handed the same image, else it runs as if nothing had happened.  "The same image" = byte for byte what the extractor
// staged in its page-locked buffer (`host_copy`, rows of `host_pitch` bytes: valid until the extractor's next call or its
// destruction, both of which end the speculation first).  Until the end of round 3 both calls hashed the frame instead (two passes
// of ~30 us over a 640 x 480 frame; one memcmp is ~12 us and stops at the first difference).
int aruco_speculate(orbfe_aruco* a, const uint8_t* d_img, size_t dframe, int rows, int cols, size_t dpitch, hipEvent_t uploaded,
                    const uint8_t* host_copy, size_t host_pitch);
void aruco_speculation_wait(orbfe_aruco* a); // until the detector no longer reads the extractor's copy of the image
void aruco_unpair_notice(orbfe_aruco* a);
int aruco_device_of(const orbfe_aruco* a);   // the HIP device the detector was created on
// The batched pipeline enqueues a batch's descriptor kernel itself, one step late: behind the NEXT batch's resize chain (gate / gate_stage
// as in orbfe_extractor_stage_wait), so that the two kernels that live on the vector memory path do not run next to each other.
void extractor_defer_describe(orbfe_extractor* h, bool on);
int extractor_describe_now(orbfe_extractor* h, orbfe_extractor* gate, int gate_stage);   // FAST of every batch of `h` behind aruco_contours_wait(det) (nullptr: off)

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
    // workspace of (this thread, the current device, stream); the caller has selected the device already.  Slots are kept in
    // most-recently-used order and capped: a thread that keeps creating streams does not grow scratch without bound (the least
    // recently used slot is freed -- hipFree waits for the device, so work still in flight on that stream is safe), and a
    // destroyed stream whose address is recycled should be dropped with release() so that it does not inherit stale state.
    static constexpr size_t MAX_SLOTS = 16;
    W& get(hipStream_t stream = nullptr)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        for (size_t i = 0; i < slots.size(); i++)
            if (slots[i].device == dev && slots[i].stream == stream) {
                if (i + 1 != slots.size()) std::rotate(slots.begin() + (long)i, slots.begin() + (long)i + 1, slots.end());
                return *slots.back().w;
            }
        if (slots.size() >= MAX_SLOTS) {
            int cur = dev;
            (void)hipSetDevice(slots.front().device);
            delete slots.front().w;
            (void)hipSetDevice(cur);
            slots.erase(slots.begin());
        }
        slots.push_back(Slot{dev, stream, new W()});
        return *slots.back().w;
    }
    // drop the workspace of (this thread, the current device, stream), if any; returns whether there was one
    bool release(hipStream_t stream)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        for (size_t i = 0; i < slots.size(); i++)
            if (slots[i].device == dev && slots[i].stream == stream) {
                delete slots[i].w;
                slots.erase(slots.begin() + (long)i);
                return true;
            }
        return false;
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
// The last HISTORY batches keep their own event sets, so a caller that times K batches without synchronising in between
// can ask for the per-interval MEDIAN over them afterwards (bench.py: one step's events are not the steady state).
struct KernelTimer {
    static constexpr int HISTORY = 64;
    struct Set {
        std::vector<hipEvent_t> ev;
        std::vector<hipStream_t> streams;
        std::vector<char> opens;
        size_t used = 0;
    };
    bool enabled = false;
    Set sets[HISTORY];
    int cur = 0;
    long batches = 0; // begin() calls while enabled
    void begin()
    {
        if (!enabled) return;
        cur = (int)(batches++ % HISTORY);
        Set& t = sets[cur];
        t.used = 0;
        t.streams.clear();
        t.opens.clear();
    }
    void reset_history()
    {
        batches = 0;
        for (auto& t : sets) t.used = 0;
    }
    void mark(hipStream_t s, const char* /*name: documentation at the call site*/, bool opens_only = false)
    {
        if (!enabled) return;
        Set& t = sets[cur];
        if (t.used == t.ev.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            t.ev.push_back(e);
        }
        (void)hipEventRecord(t.ev[t.used++], s);
        t.streams.push_back(s);
        t.opens.push_back(opens_only || t.streams.size() == 1);
    }
    static int collect_set(Set& t, float* out_us, int capacity)
    {
        if (t.used < 2) return 0;
        for (size_t i = 0; i < t.used; i++) (void)hipEventSynchronize(t.ev[i]);
        int n = 0;
        for (size_t i = 1; i < t.used && n < capacity; i++) {
            if (t.opens[i]) continue;
            size_t j = i;
            while (j-- > 0)
                if (t.streams[j] == t.streams[i]) break;
            float ms = 0;
            if (j < i) (void)hipEventElapsedTime(&ms, t.ev[j], t.ev[i]);
            out_us[n++] = ms * 1000.f;
        }
        return n;
    }
    // intervals of the most recent batch
    int collect(float* out_us, int capacity)
    {
        if (!enabled) return 0;
        return collect_set(sets[cur], out_us, capacity);
    }
    // per-interval median over the recorded batches (at most HISTORY, the newest ones); *nbatches = how many went in
    int collect_median(float* out_us, int capacity, int* nbatches)
    {
        if (nbatches) *nbatches = 0;
        if (!enabled || capacity <= 0) return 0;
        std::vector<std::vector<float>> cols;
        std::vector<float> row((size_t)capacity);
        int n0 = -1, nb = 0;
        const long have = batches < HISTORY ? batches : HISTORY;
        for (long k = 0; k < have; k++) {
            const int n = collect_set(sets[k], row.data(), capacity);
            if (n <= 0) continue;
            if (n0 < 0) { n0 = n; cols.assign((size_t)n, {}); }
            if (n != n0) continue; // a batch that took another launch path (fallback kernels) is not mixed in
            for (int i = 0; i < n; i++) cols[(size_t)i].push_back(row[(size_t)i]);
            nb++;
        }
        if (n0 < 0) return 0;
        for (int i = 0; i < n0; i++) {
            auto& c = cols[(size_t)i];
            std::sort(c.begin(), c.end());
            const size_t m = c.size();
            out_us[i] = (m & 1) ? c[m / 2] : 0.5f * (c[m / 2 - 1] + c[m / 2]);
        }
        if (nbatches) *nbatches = nb;
        return n0;
    }
    ~KernelTimer()
    {
        for (auto& t : sets)
            for (auto e : t.ev) (void)hipEventDestroy(e);
    }
};

} // namespace orbfe
