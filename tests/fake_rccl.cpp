// fake_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product: the eight librccl entry points csrc/pipeline.hip binds at run time
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclGetErrorString), between
// PROCESSES THAT SHARE ONE GPU.  The GPU box of this project has one device and RCCL refuses two ranks on it, so the N > 1 branch of the
// pipeline's gather (receive loop, own-block copy, rotation of the receive blocks, error paths) could only ever run with a peer on an
// 8-GPU node; with ORBFE_RCCL_LIB pointing here it runs as two processes on device 0 (tests/test_pipeline_gpu.py).
//
// Transport: one POSIX shared-memory segment named after the unique id, a mailbox per (source, destination) pair with a byte slot and
// two sequence numbers.  Everything happens in ncclGroupEnd, in the order sends, then receives:
//   send:  wait until the peer has taken the previous message; wait for the stream (the data is ready); device -> slot; publish;
//   recv:  wait for the message; slot -> device on the caller's stream; wait for that copy; hand the slot back.
// So a call blocks the host where RCCL would only enqueue -- the pipeline's schedule is not what is being tested, its bookkeeping is.
// Every wait is bounded (FAKE_RCCL_TIMEOUT_S, default 60): a lost peer is an error code, never a hang.
// FAKE_RCCL_FAIL_SEND_AT = k / FAKE_RCCL_FAIL_RECV_AT = k: the k-th ncclSend / ncclRecv of the process (1-based) fails.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int MAXW = 8;
struct Box {
    volatile uint64_t ready, done, nbytes;
    uint64_t pad[5];
};
struct Header {
    volatile uint32_t arrived[MAXW];
    volatile uint32_t left[MAXW];
    Box box[MAXW][MAXW]; // [src][dst]
};
struct Comm {
    char name[80];
    int rank, world;
    size_t slot, total;
    Header* h;
    uint8_t* data; // world x world slots
    uint64_t sent[MAXW], recvd[MAXW];
    uint8_t* slot_of(int src, int dst) const { return data + ((size_t)src * world + dst) * slot; }
};
struct Op {
    bool send;
    void* buf;
    size_t n;
    int peer;
    Comm* c;
    hipStream_t s;
};
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;
int g_nsend = 0, g_nrecv = 0;

double timeout_s() { const char* e = getenv("FAKE_RCCL_TIMEOUT_S"); return e && *e ? atof(e) : 60.0; }
template <class F>
bool wait_for(F cond)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (!cond()) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}

int run(const Op& o)
{
    Comm* c = o.c;
    if (o.n > c->slot) return 4; // ncclInvalidArgument: the message does not fit a slot (FAKE_RCCL_SLOT_MB)
    if (o.send) {
        Box& b = c->h->box[c->rank][o.peer];
        const uint64_t seq = ++c->sent[o.peer];
        if (!wait_for([&] { return b.done == seq - 1; })) return 6;
        if (hipStreamSynchronize(o.s) != hipSuccess) return 1;
        if (hipMemcpy(c->slot_of(c->rank, o.peer), o.buf, o.n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        b.nbytes = o.n;
        __sync_synchronize();
        b.ready = seq;
    } else {
        Box& b = c->h->box[o.peer][c->rank];
        const uint64_t seq = ++c->recvd[o.peer];
        if (!wait_for([&] { return b.ready == seq; })) return 6;
        __sync_synchronize();
        if (b.nbytes != o.n) return 4;
        if (hipMemcpyAsync(o.buf, c->slot_of(o.peer, c->rank), o.n, hipMemcpyHostToDevice, o.s) != hipSuccess) return 1;
        if (hipStreamSynchronize(o.s) != hipSuccess) return 1;
        __sync_synchronize();
        b.done = seq;
    }
    return 0;
}

int flush()
{
    int rc = 0;
    for (int pass = 0; pass < 2; pass++) // a rank's own message must be in its slot before it looks for it
        for (const Op& o : g_ops)
            if (o.send == (pass == 0) && !rc) rc = run(o);
    g_ops.clear();
    return rc;
}

} // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof(*id));
    unsigned long long r = (unsigned long long)getpid() * 2654435761ull ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%llx", r);
    return 0;
}

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank)
{
    if (!comm || world < 1 || world > MAXW || rank < 0 || rank >= world || id.internal[0] != '/') return 4;
    Comm* c = new Comm();
    memset(c, 0, sizeof(*c));
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->rank = rank; c->world = world;
    const char* mb = getenv("FAKE_RCCL_SLOT_MB");
    c->slot = (size_t)(mb && *mb ? atoi(mb) : 32) << 20;
    const size_t hdr = (sizeof(Header) + 4095) / 4096 * 4096;
    c->total = hdr + (size_t)world * world * c->slot;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->total) != 0) { if (fd >= 0) close(fd); delete c; return 2; }
    void* m = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return 2; }
    c->h = reinterpret_cast<Header*>(m);
    c->data = reinterpret_cast<uint8_t*>(m) + hdr;
    c->h->arrived[rank] = 1;
    __sync_synchronize();
    // the real call is collective: it returns when every rank has joined
    if (!wait_for([&] { for (int r = 0; r < world; r++) if (!c->h->arrived[r]) return false; return true; })) { munmap(m, c->total); delete c; return 6; }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return 4;
    c->h->left[c->rank] = 1;
    __sync_synchronize();
    bool last = true;
    for (int r = 0; r < c->world; r++) if (!c->h->left[r]) last = false;
    munmap(c->h, c->total);
    if (last || c->rank == 0) shm_unlink(c->name); // (the mapping of a peer that is still alive stays valid)
    delete c;
    return 0;
}

int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd()
{
    if (g_depth <= 0) return 4;
    if (--g_depth) return 0;
    return flush();
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !buf || dtype != 1 /* ncclUint8 */ || peer < 0 || peer >= c->world) return 4;
    const char* f = getenv("FAKE_RCCL_FAIL_SEND_AT");
    if (f && ++g_nsend == atoi(f)) return 5; // ncclRemoteError
    g_ops.push_back(Op{true, const_cast<void*>(buf), count, peer, c, s});
    return g_depth ? 0 : flush();
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || !buf || dtype != 1 || peer < 0 || peer >= c->world) return 4;
    const char* f = getenv("FAKE_RCCL_FAIL_RECV_AT");
    if (f && ++g_nrecv == atoi(f)) return 5;
    g_ops.push_back(Op{false, buf, count, peer, c, s});
    return g_depth ? 0 : flush();
}

const char* ncclGetErrorString(int e)
{
    switch (e) {
    case 0: return "no error";
    case 1: return "fake rccl: a HIP call failed";
    case 2: return "fake rccl: shared memory";
    case 4: return "fake rccl: invalid argument";
    case 5: return "fake rccl: injected failure";
    case 6: return "fake rccl: timed out waiting for a peer";
    default: return "fake rccl: error";
    }
}

} // extern "C"
