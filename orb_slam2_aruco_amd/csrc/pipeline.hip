// pipeline.hip -- host side of the batched-video mode (orbfe_pipeline_*, include/orbfe.h): the schedule of one stream of frames on
// one GPU, and the batch's gather over RCCL.
//
// Reference call sites this stands for: Frame::Frame (src/Frame.cc:91 -> :200-206 ORBextractor::operator(), :142
// MarkerDetector::detect with poses) once per frame, and Tracking's frame-to-frame matching (src/Tracking.cc:531-532).  Here a batch
// of B frames resident in HBM goes through the library's device-pointer entry points on HIP streams the pipeline owns:
//
//   extractor set d (batch i % D):  orbfe_extract_batch_device                                   stream ex[d]   (its blur lent st_match)
//   detector:                       orbfe_aruco_detect_batch_device + orbfe_marker_poses_...     stream st_det
//   matching of batch i - 1 (or i): orbfe_knn2_batch_device + orbfe_search_for_initialization_batch_device, then the gather of the
//                                   record set (ncclSend / ncclRecv), then the halo for the next batch       stream st_match
//
// What the measurements of rounds 2 - 3 settled is encoded as defaults: two extractor engine sets alternating
// batches, phase-locked behind each other's quadtree; the detector's batch behind the resize chain of the extractor's previous batch;
// four record sets in rotation; a batch's matching enqueued one step late (frames up to 640 x 480); the detector's /2 pyramid in
// line (the same); the gather on the matching stream (a fifth active stream costs 9 % of the step by itself).
#include <dlfcn.h>

#include <cstdlib>
#include <memory>
#include <mutex>

#include "orbfe_common.hpp"
#include "gather_plan.hpp"

using namespace orbfe;

namespace {

// ---- RCCL, opened at run time (the library has no link-time dependency on it; a process that already holds a copy -- PyTorch's --
// gets that one)
struct Id128 { char internal[128]; }; // ncclUniqueId
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(Id128*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // ORBFE_RCCL_LIB: the library to bind instead of the system's librccl (tests/fake_rccl.cpp: the N > 1 branch of the gather
        // between processes that share the one GPU of a test box)
        const char* over = getenv("ORBFE_RCCL_LIB");
        if (over && *over) r.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!r.lib && !(over && *over)) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* n : names)
            if (!r.lib && !(over && *over)) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) return;
        auto sym = [&](const char* n) { return dlsym(r.lib, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) r.lib = nullptr;
    });
    return r.lib ? &r : nullptr;
}
#define ORBFE_NCCL(call)                                                                                                             \
    do {                                                                                                                             \
        const int e_ = (call);                                                                                                       \
        if (e_ != 0) return fail(ORBFE_ERR_HIP, "%s failed: %s", #call, R->GetErrorString ? R->GetErrorString(e_) : "RCCL error");   \
    } while (0)
constexpr int NCCL_UINT8 = 1; // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

size_t up256(size_t v) { return (v + 255) / 256 * 256; }
int env_or(const char* name, int v) { const char* e = getenv(name); return e && *e ? atoi(e) : v; }

// the last frame of one record set -> the halo slot of another (keypoints, descriptors, count), one launch
__global__ void k_copy_halo(const uint32_t* __restrict__ src_kps, uint32_t* __restrict__ dst_kps, const uint32_t* __restrict__ src_desc,
                            uint32_t* __restrict__ dst_desc, const int32_t* __restrict__ src_n, int32_t* __restrict__ dst_n, int cap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *dst_n = *src_n;
    if (i < cap * 7) dst_kps[i] = src_kps[i];
    if (i < cap * 8) dst_desc[i] = src_desc[i];
}

} // namespace

struct orbfe_pipeline {
    orbfe_pipeline_config cfg{};
    int B = 0, rows = 0, cols = 0, cap = 0, mcap = 0, R = 0, D = 0;
    int phase_pin = 0, det_pin = 0;
    std::vector<long long> set_batch;      // per record set: the batch that was written to it last
    bool defer_post = false, det_nofork = false, use_orb = true, use_aruco = true;
    std::vector<orbfe_extractor*> ex;
    orbfe_aruco* det = nullptr;               // detector of engine set 0 (= dets[0])
    std::vector<orbfe_aruco*> dets;           // ORBFE_ENGINE_SETS_ARUCO detector sets alternate batches (measurement switch; default 1)
    std::vector<hipStream_t> st_ex, st_dets;
    hipStream_t st_det = nullptr, st_match = nullptr;
    orbfe_record_layout lay{};
    std::vector<uint8_t*> recs;
    int32_t *d_bidx = nullptr, *d_bdist = nullptr, *d_sdist = nullptr, *d_m12 = nullptr, *d_nm = nullptr;
    std::vector<hipEvent_t> ex_done, det_done, match_done, gather_done;
    bool timing = false;
    static constexpr int HIST = 64;
    hipEvent_t match_ev[HIST][3] = {}, gather_ev[HIST][2] = {};
    long match_steps = 0, gather_steps = 0;
    long step_no = 0;
    int pending = -1;
    // frames from host memory (orbfe_pipeline_step_host): a ring of device input buffers filled on a copy stream ahead of the engines,
    // every record set copied back to page-locked host memory on a second copy stream behind the batch's post-work
    static constexpr int NIN = 3;
    bool host_mode = false;
    int copy_stream_priority = 0;   // of the upload / read-back streams (the lowest the device offers; orbfe_pipeline_copy_stream_priority)
    hipStream_t st_h2d = nullptr, st_d2h = nullptr;
    hipEvent_t up_t0[NIN] = {}, up_t1[NIN] = {}, rb_t0 = nullptr, rb_t1 = nullptr;   // timing of the newest upload per slot / read-back
    uint8_t* d_in[NIN] = {};
    size_t in_pitch = 0;
    hipEvent_t in_ready[NIN] = {}, in_used_ex[NIN] = {}, in_used_det[NIN] = {};
    long in_uses[NIN] = {};
    std::vector<uint8_t*> h_recs;
    std::vector<hipEvent_t> rb_done;
    std::vector<char> rb_valid;
    // gather
    void* comm = nullptr;
    bool own_comm = false;
    int rank = 0, world = 0, dst = 0;
    uint8_t* blocks = nullptr; // on dst: R sets x world x lay.nbytes -- the batch written to record set s is received into block set s,
                               // so a consumer reads batch i (orbfe_pipeline_gathered_wait / _set) while batch i + 1 .. i + R - 1 arrive
    std::vector<hipEvent_t> gather_free;   // orbfe_pipeline_gathered_release: the consumer's reads of block set s (the next receive into it waits)
    std::vector<char> gather_free_valid;
    int last_gathered = -1;    // record set of the newest gather that was enqueued
    std::vector<long long> gather_batch;   // per record set: the batch (step number, from 0) whose gather was enqueued into it last; -1 = none yet
    bool failed = false;       // an enqueue failed half way: events of that step were never recorded; the handle refuses further steps

    ~orbfe_pipeline()
    {
        (void)hipSetDevice(cfg.device);
        (void)hipDeviceSynchronize();
        if (own_comm && comm) { if (Rccl* R_ = rccl()) (void)R_->CommDestroy(comm); }
        for (auto e : ex) if (e) orbfe_extractor_destroy(e);
        for (auto d : dets) if (d) orbfe_aruco_destroy(d);
        for (size_t k = 1; k < st_dets.size(); k++) if (st_dets[k]) (void)hipStreamDestroy(st_dets[k]);
        for (auto r : recs) if (r) (void)hipFree(r);
        for (void* p : {(void*)d_bidx, (void*)d_bdist, (void*)d_sdist, (void*)d_m12, (void*)d_nm, (void*)blocks}) if (p) (void)hipFree(p);
        for (auto* v : {&ex_done, &det_done, &match_done, &gather_done, &gather_free}) for (auto e : *v) if (e) (void)hipEventDestroy(e);
        for (auto& t : match_ev) for (auto e : t) if (e) (void)hipEventDestroy(e);
        for (auto& t : gather_ev) for (auto e : t) if (e) (void)hipEventDestroy(e);
        for (auto b : d_in) if (b) (void)hipFree(b);
        for (auto h : h_recs) if (h) (void)hipHostFree(h);
        for (auto e : rb_done) if (e) (void)hipEventDestroy(e);
        for (int k = 0; k < NIN; k++) for (hipEvent_t e : {in_ready[k], in_used_ex[k], in_used_det[k]}) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : {up_t0[0], up_t0[1], up_t0[2], up_t1[0], up_t1[1], up_t1[2], rb_t0, rb_t1}) if (e) (void)hipEventDestroy(e);
        if (st_h2d) (void)hipStreamDestroy(st_h2d);
        if (st_d2h) (void)hipStreamDestroy(st_d2h);
        for (auto s : st_ex) if (s) (void)hipStreamDestroy(s);
        if (st_det) (void)hipStreamDestroy(st_det);
        if (st_match) (void)hipStreamDestroy(st_match);
        if (st_gather) (void)hipStreamDestroy(st_gather);
        if (ev_gather_fork) (void)hipEventDestroy(ev_gather_fork);
    }

    uint8_t* slot_kps(int set, int slot) const { return recs[set] + lay.off_kps + (size_t)slot * cap * sizeof(orbfe_keypoint); }
    uint8_t* slot_desc(int set, int slot) const { return recs[set] + lay.off_desc + (size_t)slot * cap * 32; }
    int32_t* slot_n(int set, int slot) const { return reinterpret_cast<int32_t*>(recs[set] + lay.off_n) + slot; }

    int enqueue_matching(int cur)
    {
        hipEvent_t* e = timing ? match_ev[match_steps % HIST] : nullptr;
        if (timing) {
            for (int k = 0; k < 3; k++) if (!e[k]) ORBFE_HIP(hipEventCreate(&e[k]));
            match_steps++;
            ORBFE_HIP(hipEventRecord(e[0], st_match));
        }
        int rc;
        // pair p = slot p (F1: the halo for p = 0) against slot p + 1 (F2), B pairs
        if ((rc = orbfe_knn2_batch_device(slot_desc(cur, 0), slot_n(cur, 0), (size_t)cap * 32, cap, slot_desc(cur, 1), slot_n(cur, 1), (size_t)cap * 32, cap,
                                          B, 256, d_bidx, d_bdist, d_sdist, st_match)))
            return rc;
        if (timing) ORBFE_HIP(hipEventRecord(e[1], st_match));
        if ((rc = orbfe_search_for_initialization_batch_device(reinterpret_cast<const orbfe_keypoint*>(slot_kps(cur, 0)), slot_desc(cur, 0), slot_n(cur, 0), cap, B,
                                                               cols, rows, nullptr, cfg.window_size, cfg.nnratio, cfg.check_orientation, d_m12, d_nm, st_match)))
            return rc;
        if (timing) ORBFE_HIP(hipEventRecord(e[2], st_match));
        return ORBFE_OK;
    }

    // ORBFE_GATHER_STREAM=1: the gather's RCCL kernels on a stream of the LOWEST priority (another pool of hardware queues than the four
    // engine streams, like the host mode's copy streams) behind the batch's matching, instead of on the matching stream itself
    hipStream_t st_gather = nullptr;
    hipEvent_t ev_gather_fork = nullptr;
    int enqueue_gather(int cur)
    {
        hipStream_t sg = st_match;
        if (st_gather) {
            if (!ev_gather_fork) ORBFE_HIP(hipEventCreateWithFlags(&ev_gather_fork, hipEventDisableTiming));
            ORBFE_HIP(hipEventRecord(ev_gather_fork, st_match));
            ORBFE_HIP(hipStreamWaitEvent(st_gather, ev_gather_fork, 0));
            sg = st_gather;
        }
        Rccl* R = rccl();
        if (!R) return fail(ORBFE_ERR_HIP, "librccl is not available");
        hipEvent_t* e = timing ? gather_ev[gather_steps % HIST] : nullptr;
        if (timing) {
            for (int k = 0; k < 2; k++) if (!e[k]) ORBFE_HIP(hipEventCreate(&e[k]));
            gather_steps++;
            ORBFE_HIP(hipEventRecord(e[0], sg));
        }
        const size_t nb = (size_t)lay.nbytes;
                if (rank == dst && gather_free_valid[(size_t)cur]) {   // a consumer said when it is done with the batch this set held before
            ORBFE_HIP(hipStreamWaitEvent(sg, gather_free[(size_t)cur], 0));
            gather_free_valid[(size_t)cur] = 0;
        }
        // the batch's operations (csrc/gather_plan.hpp: the same function tests/test_multigpu_cpu.py drives over gloo): the messages inside
        // one group, the copy of the own block behind it
        std::vector<orbfe_gather_op> ops((size_t)world + 2);
        const int nops = gather_plan(rank, world, dst, cur, this->R, nb, ops.data(), (int)ops.size());
        if (nops < 0) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline: no gather plan for rank %d of %d", rank, world);
        ORBFE_NCCL(R->GroupStart());
        int ge = 0; // the first error inside the group: the group is closed whatever happens
        const char* gwhat = "";
        for (int k = 0; k < nops && !ge; k++) {
            if (ops[(size_t)k].kind == ORBFE_GATHER_RECV) { if ((ge = R->Recv(blocks + ops[(size_t)k].offset, nb, NCCL_UINT8, ops[(size_t)k].peer, comm, sg))) gwhat = "ncclRecv"; }
            else if (ops[(size_t)k].kind == ORBFE_GATHER_SEND) { if ((ge = R->Send(recs[cur], nb, NCCL_UINT8, ops[(size_t)k].peer, comm, sg))) gwhat = "ncclSend"; }
        }
        const int gend = R->GroupEnd();
        if (ge) return fail(ORBFE_ERR_HIP, "%s failed: %s", gwhat, R->GetErrorString ? R->GetErrorString(ge) : "RCCL error");
        if (gend) return fail(ORBFE_ERR_HIP, "ncclGroupEnd failed: %s", R->GetErrorString ? R->GetErrorString(gend) : "RCCL error");
        for (int k = 0; k < nops; k++)
            if (ops[(size_t)k].kind == ORBFE_GATHER_COPY_OWN) ORBFE_HIP(hipMemcpyAsync(blocks + ops[(size_t)k].offset, recs[cur], nb, hipMemcpyDeviceToDevice, sg));
        if (timing) ORBFE_HIP(hipEventRecord(e[1], sg));
        ORBFE_HIP(hipEventRecord(gather_done[cur], sg));
        last_gathered = cur;
        if (gather_batch.size() == (size_t)this->R) gather_batch[(size_t)cur] = set_batch[(size_t)cur];
        return ORBFE_OK;
    }

    // ORBFE_DESCRIBE_LATE: the descriptor kernel of the batch whose front part (pyramid, FAST, quadtree) was enqueued last, behind the
    // resize chain of the batch on engine set `gate_set` (-1: no gate -- the flush)
    bool describe_late = false;
    int late_set = -1, late_cur = -1, late_in_slot = -1;
    int finish_describe(int gate_set)
    {
        if (late_set < 0) return ORBFE_OK;
        int rc = extractor_describe_now(ex[(size_t)late_set], gate_set >= 0 ? ex[(size_t)gate_set] : nullptr, 4);
        if (rc) return rc;
        ORBFE_HIP(hipEventRecord(ex_done[late_cur], st_ex[(size_t)late_set]));
        if (late_in_slot >= 0) ORBFE_HIP(hipEventRecord(in_used_ex[late_in_slot], st_ex[(size_t)late_set]));
        late_set = -1;
        return ORBFE_OK;
    }

    // what follows a batch's engines: its matching, its gather, and the halo of the next record set
    int enqueue_post(int cur)
    {
        int rc;
        if (use_orb) {
            ORBFE_HIP(hipStreamWaitEvent(st_match, ex_done[cur], 0));
            if ((rc = enqueue_matching(cur))) return rc;
        }
        if (comm) {
            if (use_aruco) ORBFE_HIP(hipStreamWaitEvent(st_match, det_done[cur], 0));
            if ((rc = enqueue_gather(cur))) return rc;
        }
        if (use_orb) {
            // the stream goes on: the batch's last frame becomes the halo of the set the NEXT batch is written to (that set's halo was
            // last read by the matching of R batches ago, earlier on this stream).  Only then may this set be written again.
            const int nxt = (cur + 1) % R;
            // ... and in host mode the set's previous contents are still being copied to the host (the read-back stream reads the whole
            // set, halo slot included): the write waits for that copy like the engines do
            if (host_mode && rb_valid[(size_t)nxt]) ORBFE_HIP(hipStreamWaitEvent(st_match, rb_done[(size_t)nxt], 0));
            hipLaunchKernelGGL(k_copy_halo, dim3((cap * 8 + 255) / 256), dim3(256), 0, st_match, reinterpret_cast<const uint32_t*>(slot_kps(cur, B)),
                               reinterpret_cast<uint32_t*>(slot_kps(nxt, 0)), reinterpret_cast<const uint32_t*>(slot_desc(cur, B)),
                               reinterpret_cast<uint32_t*>(slot_desc(nxt, 0)), slot_n(cur, B), slot_n(nxt, 0), cap);
            ORBFE_HIP(hipEventRecord(match_done[cur], st_match));
        }
        if (host_mode) { // the record set back to the host, behind everything that writes or reads it on the device
            if (use_orb) ORBFE_HIP(hipStreamWaitEvent(st_d2h, match_done[cur], 0));
            if (use_aruco) ORBFE_HIP(hipStreamWaitEvent(st_d2h, det_done[cur], 0));
            if (comm) ORBFE_HIP(hipStreamWaitEvent(st_d2h, gather_done[cur], 0));
            if (!rb_t0) { ORBFE_HIP(hipEventCreate(&rb_t0)); ORBFE_HIP(hipEventCreate(&rb_t1)); }
            ORBFE_HIP(hipEventRecord(rb_t0, st_d2h));
            ORBFE_HIP(hipMemcpyAsync(h_recs[(size_t)cur], recs[(size_t)cur], lay.nbytes, hipMemcpyDeviceToHost, st_d2h));
            ORBFE_HIP(hipEventRecord(rb_t1, st_d2h));
            ORBFE_HIP(hipEventRecord(rb_done[(size_t)cur], st_d2h));
            rb_valid[(size_t)cur] = 1;
        }
        return ORBFE_OK;
    }
};

const char* orbfe_pipeline_env_defaults(void)
{
    // one list for the pipeline and the engines: bench.py marks a line as diagnostic when one of these is set to something else
    return "ORBFE_ENGINE_SETS=2;ORBFE_ENGINE_SETS_ARUCO=1;ORBFE_RECORD_SETS=4;ORBFE_PHASE_PIN=size;ORBFE_DET_PIN=4;ORBFE_DEFER_POST=size;ORBFE_DET_NOFORK=size;"
           "ORBFE_ARUCO_RELAY_WIDE=1;"
           "ORBFE_ARUCO_SPECKS=size;ORBFE_DESCRIBE_LATE=1;ORBFE_ARUCO_SMALL_SEPARATE=size;ORBFE_ARUCO_TILED=size;ORBFE_ARUCO_TILE_W=0;ORBFE_ARUCO_TPW=0;ORBFE_ARUCO_BANDED=size;ORBFE_ARUCO_BAND_ROWS=0;ORBFE_ARUCO_LCAP=0;ORBFE_GRAPH=0;"
           "ORBFE_GATHER_STREAM=0;ORBFE_CU_DET=0;ORBFE_CU_EX=0;ORBFE_NO_LEND=0;ORBFE_GRAPH_VERBOSE=0;ORBFE_RCCL_LIB=";
}

int orbfe_pipeline_config_default(orbfe_pipeline_config* c, int frames, int rows, int cols)
{
    if (!c || frames < 1 || rows < 1 || cols < 1) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_config_default: invalid argument");
    memset(c, 0, sizeof(*c));
    c->frames = frames; c->rows = rows; c->cols = cols;
    c->nfeatures = 1000; c->nlevels = 8; c->scale_factor = 1.2f; c->ini_th_fast = 20; c->min_th_fast = 7;
    snprintf(c->dictionary, sizeof(c->dictionary), "ARUCO");
    c->device = 0; c->marker_capacity = 64; c->use_orb = 1; c->use_aruco = 1;
    c->marker_size = 0.187f; // Frame.cc:131
    const float tum1[4] = {517.306408f, 516.469215f, 318.643040f, 255.313989f}; // Examples/Monocular/TUM1.yaml
    const float d5[5] = {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f};
    orbfe_camera_resize(tum1, 1280, 720, cols, rows, c->K); // the detector is handed CamSize 1280 x 720 (Frame.cc:132)
    for (int i = 0; i < 5; i++) c->dist[i] = d5[i];
    c->ndist = 5;
    c->window_size = 100; c->nnratio = 0.9f; c->check_orientation = 1;
    c->engine_sets = c->record_sets = c->phase_pin = c->det_pin = c->defer_post = c->det_nofork = -1;
    return ORBFE_OK;
}

orbfe_pipeline* orbfe_pipeline_create(const orbfe_pipeline_config* cfg)
{
    if (!cfg || cfg->frames < 1 || cfg->rows < 1 || cfg->cols < 1 || (!cfg->use_orb && !cfg->use_aruco)) {
        fail(ORBFE_ERR_INVALID, "orbfe_pipeline_create: invalid configuration");
        return nullptr;
    }
    if (use_device(cfg->device)) return nullptr;
    std::unique_ptr<orbfe_pipeline> p(new orbfe_pipeline());
    p->cfg = *cfg;
    p->cfg.dictionary[sizeof(p->cfg.dictionary) - 1] = 0;
    const int B = p->B = cfg->frames, rows = p->rows = cfg->rows, cols = p->cols = cfg->cols;
    p->use_orb = cfg->use_orb != 0; p->use_aruco = cfg->use_aruco != 0;
    const bool vga = (size_t)rows * cols <= (size_t)640 * 480;
    // defaults by frame size, each overridable by the configuration and, for measurements, by the environment
    auto pick = [&](int cfgv, const char* env, int dflt) { return env_or(env, cfgv >= 0 ? cfgv : dflt); };
    // two extractor sets (1.4955 against 1.5288 ms per C2 step in round 3; 1920 x 1080 lost then, 4.84 -> 5.00, while its contour
    // stage held whole CUs -- with the banded contour kernels it gains: 3.58 against 3.64 ms, three interleaved runs each)
    p->D = std::max(1, pick(cfg->engine_sets, "ORBFE_ENGINE_SETS", 2));
    if (!p->use_orb) p->D = 1;
    p->R = std::max(2, pick(cfg->record_sets, "ORBFE_RECORD_SETS", 4));
    // the extractor sets' lock: behind the other set's quadtree up to 1280 x 720; behind its FAST above (1920 x 1080 with the banded contour
    // kernels, four runs each: 3.15 - 3.19 ms per step, no lock at all 3.16 - 3.18, behind the quadtree 3.33 - 3.37, tools/r04_pins35b.sh)
    const bool above_720p = (size_t)rows * cols > (size_t)1280 * 720;
    p->phase_pin = pick(cfg->phase_pin, "ORBFE_PHASE_PIN", above_720p ? 1 : 2);
    p->det_pin = pick(cfg->det_pin, "ORBFE_DET_PIN", 4);
    p->defer_post = pick(cfg->defer_post, "ORBFE_DEFER_POST", vga ? 1 : 0) != 0;
    p->det_nofork = pick(cfg->det_nofork, "ORBFE_DET_NOFORK", vga ? 1 : 0) != 0;
    auto bail = [&](const char* what) -> orbfe_pipeline* {
        if (what) { std::string m = g_err; fail(ORBFE_ERR_HIP, "orbfe_pipeline_create: %s (%s)", what, m.c_str()); }
        return nullptr;
    };
    auto mkstream = [&](hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess; };
    // CU partition (measurement switch, round 6): ORBFE_CU_DET = K gives the detector's stream the first K compute units of the
    // enumeration (the driver deals mask bits round-robin over the eight XCDs, so K / 8 of every XCD); ORBFE_CU_EX = 1 gives the
    // extractor sets' streams the complement, 2 also the matching stream.  Off by default: see tools/sweeps.md.
    const int cu_det = env_or("ORBFE_CU_DET", 0), cu_ex = env_or("ORBFE_CU_EX", 0);
    auto mkmasked = [&](hipStream_t* s, int first, int last) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = first; i < last && i < 256; i++) mask[i >> 5] |= 1u << (i & 31);
        return hipExtStreamCreateWithCUMask(s, 8, mask) == hipSuccess;
    };
    if (cu_det > 0 ? !mkmasked(&p->st_det, 0, cu_det) : !mkstream(&p->st_det)) return bail("stream");
    if (cu_det > 0 && cu_ex >= 2 ? !mkmasked(&p->st_match, cu_det, 256) : !mkstream(&p->st_match)) return bail("stream");
    if (env_or("ORBFE_GATHER_STREAM", 0)) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        if (hipStreamCreateWithPriority(&p->st_gather, hipStreamNonBlocking, lo) != hipSuccess) return bail("stream");
    }
    p->st_ex.assign((size_t)p->D, nullptr);
    for (auto& s : p->st_ex) if (cu_det > 0 && cu_ex ? !mkmasked(&s, cu_det, 256) : !mkstream(&s)) return bail("stream");
    if (p->use_orb) {
        for (int d = 0; d < p->D; d++) {
            orbfe_extractor* e = orbfe_extractor_create(cfg->nfeatures, cfg->scale_factor, cfg->nlevels, cfg->ini_th_fast, cfg->min_th_fast, cfg->device);
            if (!e) return nullptr;
            p->ex.push_back(e);
            // every set's blur on the matching stream (1.4466 against 1.4873 ms with the handles' own fork streams, which share hardware
            // queues with the busy ones)
            // (round 5: the blur of all sets on ONE extra stream with a hardware queue of its own -- lowest / highest stream priority --
            // 1.59 / 1.69 ms against 1.35: what the lending buys is that blur and matching do NOT run next to each other)
            if (!env_or("ORBFE_NO_LEND", 0)) orbfe_extractor_set_aux_stream(e, p->st_match);
        }
        if (p->phase_pin && p->D > 1)
            for (int d = 0; d < p->D; d++) orbfe_extractor_follow(p->ex[d], p->ex[(d + p->D - 1) % p->D], p->phase_pin);
        // A batch's descriptor kernel one step late, behind the NEXT batch's resize chain (round 6).  Both live on the CU's vector memory
        // path -- unaligned 8- and 16-byte lane loads -- and next to each other the resize chain, which is on the step's critical chain,
        // took 400 - 450 us (200 alone); next to FAST, which is VALU-bound, the descriptors cost less than they gave back at 640 x 480:
        // 1.308 against 1.338 ms per C2 step (twelve interleaved runs each; resize 294 - 336 us, FAST 810 - 890 instead of 610 - 690).
        // With the blur on the matrix cores (k_blur7_mfma) FAST has the vector ALUs more to itself and every size gains: C2 1.265
        // against 1.327, 1280 x 720 3.70 against 3.74, 1920 x 1080 3.13 against 3.21 ms.
        p->describe_late = pick(-1, "ORBFE_DESCRIBE_LATE", 1) != 0 && p->D > 1;
        if (p->describe_late) p->defer_post = true;
        p->cap = orbfe_extractor_max_keypoints(p->ex[0]);
    } else
        p->cap = 1;
    if (p->use_aruco) {
        const int DA = std::max(1, env_or("ORBFE_ENGINE_SETS_ARUCO", 1));
        for (int d = 0; d < DA; d++) {
            orbfe_aruco* a = orbfe_aruco_create(p->cfg.dictionary, cfg->device);
            if (!a) return nullptr;
            p->dets.push_back(a);
            hipStream_t sd = p->st_det;
            if (d > 0 && !mkstream(&sd)) return bail("stream");
            p->st_dets.push_back(sd);
            if (p->det_nofork) orbfe_aruco_set_aux_stream(a, sd);
        }
        p->det = p->dets[0];
        p->mcap = std::min(orbfe_aruco_max_markers(p->det), std::max(1, cfg->marker_capacity));
    }
    orbfe_record_layout& L = p->lay;
    L.frames = B; L.capacity = p->cap; L.marker_capacity = p->mcap; L.halo = 1;
    L.off_kps = 0;
    L.off_desc = up256((size_t)(B + 1) * p->cap * sizeof(orbfe_keypoint));
    L.off_n = L.off_desc + up256((size_t)(B + 1) * p->cap * 32);
    L.off_markers = L.off_n + up256((size_t)(B + 1) * 4);
    L.off_nmarkers = L.off_markers + up256((size_t)B * p->mcap * sizeof(orbfe_marker));
    L.off_poses = L.off_nmarkers + up256((size_t)B * 4);
    L.nbytes = L.off_poses + up256((size_t)B * p->mcap * sizeof(orbfe_marker_pose));
    p->recs.assign((size_t)p->R, nullptr);
    for (auto& r : p->recs) {
        if (hipMalloc(&r, L.nbytes) != hipSuccess || hipMemset(r, 0, L.nbytes) != hipSuccess) return bail("record sets");
    }
    const size_t mb = (size_t)B * p->cap * 4;
    if (hipMalloc(&p->d_bidx, mb) != hipSuccess || hipMalloc(&p->d_bdist, mb) != hipSuccess || hipMalloc(&p->d_sdist, mb) != hipSuccess ||
        hipMalloc(&p->d_m12, mb) != hipSuccess || hipMalloc(&p->d_nm, (size_t)B * 4) != hipSuccess)
        return bail("matching outputs");
    (void)hipMemset(p->d_bidx, 0, mb); (void)hipMemset(p->d_bdist, 0, mb); (void)hipMemset(p->d_sdist, 0, mb); (void)hipMemset(p->d_m12, 0, mb);
    (void)hipMemset(p->d_nm, 0, (size_t)B * 4);
    for (auto* v : {&p->ex_done, &p->det_done, &p->match_done, &p->gather_done}) {
        v->assign((size_t)p->R, nullptr);
        for (auto& e : *v) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail("events");
    }
    if (hipDeviceSynchronize() != hipSuccess) return bail("synchronize");
    return p.release();
}

void orbfe_pipeline_destroy(orbfe_pipeline* p) { delete p; }

int orbfe_pipeline_layout(const orbfe_pipeline* p, orbfe_record_layout* out)
{
    if (!p || !out) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_layout: null argument");
    *out = p->lay;
    return ORBFE_OK;
}

static int step_impl(orbfe_pipeline* p, const uint8_t* d_imgs, size_t pitch, int32_t* record_set, int in_slot);

int orbfe_pipeline_step(orbfe_pipeline* p, const uint8_t* d_imgs, size_t pitch, int32_t* record_set)
{
    if (!p || !d_imgs || pitch < (size_t)p->cols) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_step: invalid argument");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    return step_impl(p, d_imgs, pitch, record_set, -1);
}

static int host_mode_init(orbfe_pipeline* p)
{
    if (p->host_mode) return ORBFE_OK;
    // every resource is created once: a call that failed half way (host_mode still false) is continued, not repeated, by the next one
    p->in_pitch = ((size_t)p->cols + 63) / 64 * 64;
    // The copy streams must not share a hardware queue with an engine's stream: the runtime multiplexes the streams of one priority
    // onto GPU_MAX_HW_QUEUES (4) queues, the pipeline has four engine streams already, and an upload queued behind an extractor chain
    // waited for it -- the copy engine idle 1.1 ms of every 2.8 ms step.  Streams of another priority come from a pool of their own:
    // C2 from host memory 107 k frames/s with plain copy streams, 152 k with lowest-priority ones (137 k with highest; 156 k with plain
    // streams and GPU_MAX_HW_QUEUES=8), round 5.
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess || prio_lo == prio_hi) {
        // no second priority: the copy streams come out of the engines' pool of hardware queues again (107 k instead of 152 k frames/s)
        (void)hipGetLastError();
        fprintf(stderr, "orbfe_pipeline_step_host: this device offers one stream priority only; uploads will share hardware queues with the engines\n");
        prio_lo = prio_hi = 0;
    }
    p->copy_stream_priority = prio_lo;
    for (hipStream_t* st : {&p->st_h2d, &p->st_d2h})
        if (!*st && hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_lo) != hipSuccess) return fail(ORBFE_ERR_HIP, "orbfe_pipeline_step_host: streams");
    for (int k = 0; k < orbfe_pipeline::NIN; k++) {
        if (!p->d_in[k]) {
            ORBFE_HIP(hipMalloc(&p->d_in[k], (size_t)p->B * p->rows * p->in_pitch));
            ORBFE_HIP(hipMemset(p->d_in[k], 0, (size_t)p->B * p->rows * p->in_pitch));
        }
        for (hipEvent_t* e : {&p->in_ready[k], &p->in_used_ex[k], &p->in_used_det[k]})
            if (!*e) ORBFE_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if (p->h_recs.empty()) {
        p->h_recs.assign((size_t)p->R, nullptr);
        p->rb_done.assign((size_t)p->R, nullptr);
        p->rb_valid.assign((size_t)p->R, 0);
    }
    for (int k = 0; k < p->R; k++) {
        if (!p->h_recs[(size_t)k]) ORBFE_HIP(hipHostMalloc((void**)&p->h_recs[(size_t)k], p->lay.nbytes, hipHostMallocDefault));
        if (!p->rb_done[(size_t)k]) ORBFE_HIP(hipEventCreateWithFlags(&p->rb_done[(size_t)k], hipEventDisableTiming));
    }
    p->host_mode = true;
    return ORBFE_OK;
}

int orbfe_pipeline_step_host(orbfe_pipeline* p, const uint8_t* h_imgs, size_t step, int32_t* record_set)
{
    if (!p || !h_imgs || step < (size_t)p->cols) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_step_host: invalid argument");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    if ((rc = host_mode_init(p))) return rc;
    const int slot = (int)(p->step_no % orbfe_pipeline::NIN);
    // the slot's previous batch (NIN steps ago) must have been read by its engines
    if (p->in_uses[slot]) {
        if (p->use_orb) ORBFE_HIP(hipStreamWaitEvent(p->st_h2d, p->in_used_ex[slot], 0));
        if (p->use_aruco) ORBFE_HIP(hipStreamWaitEvent(p->st_h2d, p->in_used_det[slot], 0));
    }
    const size_t nrows = (size_t)p->B * p->rows;
    if (!p->up_t0[slot]) { ORBFE_HIP(hipEventCreate(&p->up_t0[slot])); ORBFE_HIP(hipEventCreate(&p->up_t1[slot])); }
    ORBFE_HIP(hipEventRecord(p->up_t0[slot], p->st_h2d));
    // rows that lie back to back on both sides are one linear copy.  (Measured and gone: the batch in two halves on two copy streams,
    // 34.2 against 34.3 GB/s while the copy stream still shared a hardware queue; a copy kernel of the pipeline's own reading the
    // mapped host pointer -- 57 GB/s alone like hipMemcpyAsync, tools/h2d_bw.hip -- 30.5 against 34.3 GB/s.)
    if (step == (size_t)p->cols && p->in_pitch == (size_t)p->cols) ORBFE_HIP(hipMemcpyAsync(p->d_in[slot], h_imgs, nrows * step, hipMemcpyHostToDevice, p->st_h2d));
    else ORBFE_HIP(hipMemcpy2DAsync(p->d_in[slot], p->in_pitch, h_imgs, step, (size_t)p->cols, nrows, hipMemcpyHostToDevice, p->st_h2d));
    ORBFE_HIP(hipEventRecord(p->up_t1[slot], p->st_h2d));
    ORBFE_HIP(hipEventRecord(p->in_ready[slot], p->st_h2d));
    p->in_uses[slot]++;
    return step_impl(p, p->d_in[slot], p->in_pitch, record_set, slot);
}

int orbfe_pipeline_host_records(orbfe_pipeline* p, int set, const uint8_t** h_records)
{
    if (!p || !h_records || !p->host_mode || set < 0 || set >= p->R || !p->rb_valid[(size_t)set])
        return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_host_records: no host copy of that record set (orbfe_pipeline_step_host, then flush)");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    ORBFE_HIP(hipEventSynchronize(p->rb_done[(size_t)set]));
    *h_records = p->h_recs[(size_t)set];
    return ORBFE_OK;
}

int orbfe_pipeline_copy_stream_priority(orbfe_pipeline* p, int* priority, int* lowest, int* highest)
{
    if (!p || !priority) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_copy_stream_priority: null argument");
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
    *priority = p->host_mode ? p->copy_stream_priority : lo;
    if (lowest) *lowest = lo;
    if (highest) *highest = hi;
    return ORBFE_OK;
}

int orbfe_pipeline_host_copy_us(orbfe_pipeline* p, float out[2])
{
    if (!p || !out || !p->host_mode) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_host_copy_us: no step from host memory yet");
    int rc = orbfe_pipeline_synchronize(p);
    if (rc) return rc;
    out[0] = out[1] = 0.f;
    int n = 0;
    for (int k = 0; k < orbfe_pipeline::NIN; k++) {
        float ms = 0.f;
        if (p->up_t0[k] && hipEventElapsedTime(&ms, p->up_t0[k], p->up_t1[k]) == hipSuccess) { out[0] += ms * 1000.f; n++; }
    }
    if (n) out[0] /= (float)n;
    float ms = 0.f;
    if (p->rb_t0 && hipEventElapsedTime(&ms, p->rb_t0, p->rb_t1) == hipSuccess) out[1] = ms * 1000.f;
    (void)hipGetLastError();
    return ORBFE_OK;
}

void* orbfe_host_alloc(size_t bytes)
{
    void* q = nullptr;
    if (hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) { fail(ORBFE_ERR_HIP, "orbfe_host_alloc: %zu bytes of page-locked memory", bytes); return nullptr; }
    return q;
}
void orbfe_host_free(void* q) { if (q) (void)hipHostFree(q); }

void* orbfe_device_alloc(int device, size_t bytes)
{
    void* q = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&q, bytes ? bytes : 1) != hipSuccess || hipMemset(q, 0, bytes) != hipSuccess) {
        if (q) (void)hipFree(q);
        fail(ORBFE_ERR_HIP, "orbfe_device_alloc: %zu bytes on device %d", bytes, device);
        return nullptr;
    }
    return q;
}
void orbfe_device_free(void* d) { if (d) (void)hipFree(d); }
int orbfe_device_upload_rows(void* d_dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t nrows)
{
    if (!d_dst || !src || dpitch < width || spitch < width) return fail(ORBFE_ERR_INVALID, "orbfe_device_upload_rows: invalid argument");
    ORBFE_HIP(hipMemcpy2D(d_dst, dpitch, src, spitch, width, nrows, hipMemcpyHostToDevice));
    return ORBFE_OK;
}
int orbfe_device_download(void* dst, const void* d_src, size_t bytes)
{
    if (!dst || !d_src) return fail(ORBFE_ERR_INVALID, "orbfe_device_download: null argument");
    ORBFE_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

static int step_body(orbfe_pipeline* p, const uint8_t* d_imgs, size_t pitch, int32_t* record_set, int in_slot);

// A step that fails half way leaves events of that batch unrecorded which later steps would wait for (and step_no already counts
// it): the handle is marked and refuses further steps -- destroy it.
static int step_impl(orbfe_pipeline* p, const uint8_t* d_imgs, size_t pitch, int32_t* record_set, int in_slot)
{
    if (p->failed) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_step: an earlier step of this pipeline failed while it was being enqueued; destroy the handle");
    const int rc = step_body(p, d_imgs, pitch, record_set, in_slot);
    if (rc) p->failed = true;
    return rc;
}

static int step_body(orbfe_pipeline* p, const uint8_t* d_imgs, size_t pitch, int32_t* record_set, int in_slot)
{
    int rc;
    const long i = p->step_no++;
    const int cur = (int)(i % p->R), eset = (int)(i % p->D), B = p->B, rows = p->rows, cols = p->cols;
    const size_t fstride = (size_t)rows * pitch;
    uint8_t* base = p->recs[cur];
    if (record_set) *record_set = cur;
    if (p->set_batch.size() != (size_t)p->R) { p->set_batch.assign((size_t)p->R, -1); p->gather_batch.assign((size_t)p->R, -1); }
    p->set_batch[(size_t)cur] = i;
    auto enqueue_detector = [&]() -> int {
        if (!p->use_aruco) return ORBFE_OK;
        const size_t aset = (size_t)(i % (long)p->dets.size());
        orbfe_aruco* det_i = p->dets[aset];
        hipStream_t st_det_i = p->st_dets[aset];
        // the detector only depends on the resident frames and on its own previous batch: it is not joined with the extractor per step
        // (several detector sets -- a measurement switch -- alternate streams: the batch that used this record set R steps ago ran on
        // another one when the number of sets does not divide R)
        if (p->dets.size() > 1 && i >= p->R) ORBFE_HIP(hipStreamWaitEvent(st_det_i, p->det_done[cur], 0));
        if (p->comm && i >= p->R) ORBFE_HIP(hipStreamWaitEvent(st_det_i, p->gather_done[cur], 0)); // batch i - R has left this record set
        if (p->host_mode && p->rb_valid[(size_t)cur]) ORBFE_HIP(hipStreamWaitEvent(st_det_i, p->rb_done[(size_t)cur], 0)); // ... and has been copied to the host
        if (in_slot >= 0) { ORBFE_HIP(hipStreamWaitEvent(st_det_i, p->in_ready[in_slot], 0)); }
        if (p->det_pin && p->use_orb) {
            const long j = p->det_pin >= 10 ? i : i - 1; // + 10: a stage of THIS batch's extractor, which is then enqueued first
            if (j >= 0 && (rc = orbfe_extractor_stage_wait(p->ex[(size_t)(j % p->D)], p->det_pin % 10, st_det_i))) return rc;
        }
        orbfe_marker* mk = reinterpret_cast<orbfe_marker*>(base + p->lay.off_markers);
        int32_t* nmk = reinterpret_cast<int32_t*>(base + p->lay.off_nmarkers);
        if ((rc = orbfe_aruco_detect_batch_device(det_i, d_imgs, B, fstride, rows, cols, pitch, mk, p->mcap, nmk, st_det_i))) return rc;
        // detect(image, CameraParameters, 0.187): every marker gets its IPPE pose (markerdetector_impl.cpp:8720-8780)
        if ((rc = orbfe_marker_poses_batch_device(mk, nmk, p->mcap, B, p->cfg.marker_size, p->cfg.K, p->cfg.dist, p->cfg.ndist,
                                                  reinterpret_cast<orbfe_marker_pose*>(base + p->lay.off_poses), st_det_i)))
            return rc;
        ORBFE_HIP(hipEventRecord(p->det_done[cur], st_det_i));
        if (in_slot >= 0) ORBFE_HIP(hipEventRecord(p->in_used_det[in_slot], st_det_i));
        return ORBFE_OK;
    };
    auto enqueue_extractor = [&]() -> int {
        if (!p->use_orb) return ORBFE_OK;
        hipStream_t st = p->st_ex[(size_t)eset];
        if (i >= p->R) {
            ORBFE_HIP(hipStreamWaitEvent(st, p->match_done[cur], 0)); // the matching of batch i - R has read this record set
            if (p->comm) ORBFE_HIP(hipStreamWaitEvent(st, p->gather_done[cur], 0));
        }
        if (p->host_mode && p->rb_valid[(size_t)cur]) ORBFE_HIP(hipStreamWaitEvent(st, p->rb_done[(size_t)cur], 0));
        if (in_slot >= 0) { ORBFE_HIP(hipStreamWaitEvent(st, p->in_ready[in_slot], 0)); }
        // (the deferral is the pipeline's, call by call: anybody else who calls the handle's entry points gets the whole extraction)
        extractor_defer_describe(p->ex[(size_t)eset], p->describe_late);
        rc = orbfe_extract_batch_device(p->ex[(size_t)eset], d_imgs, B, fstride, rows, cols, pitch, reinterpret_cast<orbfe_keypoint*>(p->slot_kps(cur, 1)),
                                        p->slot_desc(cur, 1), p->cap, p->slot_n(cur, 1), st);
        extractor_defer_describe(p->ex[(size_t)eset], false);
        if (rc) return rc;
        if (p->describe_late) {
            // the PREVIOUS batch's descriptors now, behind this batch's resize chain; this batch's at the next step (or the flush)
            if ((rc = p->finish_describe(eset))) return rc;
            p->late_set = eset; p->late_cur = cur; p->late_in_slot = in_slot;
            return ORBFE_OK;
        }
        ORBFE_HIP(hipEventRecord(p->ex_done[cur], st));
        if (in_slot >= 0) ORBFE_HIP(hipEventRecord(p->in_used_ex[in_slot], st));
        return ORBFE_OK;
    };
    if (p->det_pin >= 10) { if ((rc = enqueue_extractor()) || (rc = enqueue_detector())) return rc; }
    else if ((rc = enqueue_detector()) || (rc = enqueue_extractor())) return rc;
    // What follows a batch's engines goes onto the matching stream, which also carries the extractors' blur (lent).  Enqueued right
    // away, the matching of batch i (which waits for the whole extractor chain of batch i) would sit IN FRONT of the blur of batch
    // i + 1 on that stream, and the descriptors of batch i + 1 wait for that blur.  So the post-work of batch i is enqueued one step
    // late, behind the blur of batch i + 1 (C2 1.4924 -> 1.4759 ms; 1280 x 720 loses 1.5 %: off there).
    if (p->defer_post) {
        if (p->pending >= 0 && (rc = p->enqueue_post(p->pending))) return rc;
        p->pending = cur;
    } else if ((rc = p->enqueue_post(cur)))
        return rc;
    return ORBFE_OK;
}

int orbfe_pipeline_flush(orbfe_pipeline* p)
{
    if (!p) return fail(ORBFE_ERR_INVALID, "null handle");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    if ((rc = p->finish_describe(-1))) { p->failed = true; return rc; }
    if (p->pending >= 0) {
        const int cur = p->pending;
        p->pending = -1;
        if ((rc = p->enqueue_post(cur))) { p->failed = true; return rc; }
    }
    return ORBFE_OK;
}

int orbfe_pipeline_synchronize(orbfe_pipeline* p)
{
    int rc = orbfe_pipeline_flush(p);
    if (rc) return rc;
    for (auto s : p->st_ex) ORBFE_HIP(hipStreamSynchronize(s));
    ORBFE_HIP(hipStreamSynchronize(p->st_det));
    for (auto sd : p->st_dets) ORBFE_HIP(hipStreamSynchronize(sd));
    ORBFE_HIP(hipStreamSynchronize(p->st_match));
    if (p->st_gather) ORBFE_HIP(hipStreamSynchronize(p->st_gather));
    if (p->host_mode) { ORBFE_HIP(hipStreamSynchronize(p->st_h2d)); ORBFE_HIP(hipStreamSynchronize(p->st_d2h)); }
    return ORBFE_OK;
}

int orbfe_pipeline_input_done(orbfe_pipeline* p, int set)
{
    if (!p || set < 0 || set >= p->R) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_input_done: invalid argument");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    if (p->use_orb) ORBFE_HIP(hipEventSynchronize(p->ex_done[(size_t)set]));
    if (p->use_aruco) ORBFE_HIP(hipEventSynchronize(p->det_done[(size_t)set]));
    return ORBFE_OK;
}

int orbfe_pipeline_status(orbfe_pipeline* p, int32_t out[4])
{
    if (!p || !out) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_status: null argument");
    int rc = orbfe_pipeline_synchronize(p);
    if (rc) return rc;
    out[0] = out[1] = out[2] = out[3] = 0;
    for (auto e : p->ex) {
        int32_t o = 0;
        if ((rc = orbfe_extractor_batch_status(e, &o))) return rc;
        out[0] = std::max(out[0], o);
    }
    if (p->use_orb) {
        int32_t o = 0;
        rc = orbfe_search_for_initialization_batch_status(p->st_match, &o);
        if (rc && rc != ORBFE_ERR_CAPACITY) return rc;
        out[1] = o;
    }
    for (auto d : p->dets) {
        int32_t n = 0, fl = 0;
        if ((rc = orbfe_aruco_batch_status(d, &n, &fl))) return rc;
        out[2] += n; out[3] |= fl;
    }
    return ORBFE_OK;
}

int orbfe_pipeline_set_big_frames(orbfe_pipeline* p, int on)
{
    if (!p) return fail(ORBFE_ERR_INVALID, "null handle");
    for (auto d : p->dets) { int rc = orbfe_aruco_set_big_frames(d, on); if (rc) return rc; }
    return ORBFE_OK;
}

int orbfe_pipeline_records(orbfe_pipeline* p, int set, uint8_t** d_records)
{
    if (!p || !d_records || set < 0 || set >= p->R) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_records: invalid argument");
    *d_records = p->recs[(size_t)set];
    return ORBFE_OK;
}

int orbfe_pipeline_matches(orbfe_pipeline* p, int32_t** bi, int32_t** bd, int32_t** sd, int32_t** m12, int32_t** nm)
{
    if (!p) return fail(ORBFE_ERR_INVALID, "null handle");
    if (bi) *bi = p->d_bidx;
    if (bd) *bd = p->d_bdist;
    if (sd) *sd = p->d_sdist;
    if (m12) *m12 = p->d_m12;
    if (nm) *nm = p->d_nm;
    return ORBFE_OK;
}

int orbfe_pipeline_reset_stream(orbfe_pipeline* p)
{
    int rc = orbfe_pipeline_synchronize(p);
    if (rc) return rc;
    for (int s = 0; s < p->R; s++) ORBFE_HIP(hipMemset(p->slot_n(s, 0), 0, 4));
    return ORBFE_OK;
}

orbfe_extractor* orbfe_pipeline_extractor(orbfe_pipeline* p, int set) { return p && set >= 0 && set < (int)p->ex.size() ? p->ex[(size_t)set] : nullptr; }
orbfe_aruco* orbfe_pipeline_detector(orbfe_pipeline* p) { return p ? p->det : nullptr; }

int orbfe_pipeline_engine_sets(const orbfe_pipeline* p, int32_t* engine_sets, int32_t* record_sets, int32_t* phase_pin, int32_t* det_pin,
                               int32_t* defer_post, int32_t* det_nofork)
{
    if (!p) return fail(ORBFE_ERR_INVALID, "null handle");
    if (engine_sets) *engine_sets = p->D;
    if (record_sets) *record_sets = p->R;
    if (phase_pin) *phase_pin = p->D > 1 ? p->phase_pin : 0;
    if (det_pin) *det_pin = p->det_pin;
    if (defer_post) *defer_post = p->defer_post;
    if (det_nofork) *det_nofork = p->det_nofork;
    return ORBFE_OK;
}

int orbfe_pipeline_enable_timing(orbfe_pipeline* p, int on)
{
    if (!p) return fail(ORBFE_ERR_INVALID, "null handle");
    p->timing = on != 0;
    p->match_steps = p->gather_steps = 0;
    return ORBFE_OK;
}

int orbfe_pipeline_timing_us(orbfe_pipeline* p, int last, float out[3])
{
    if (!p || !out) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_timing_us: null argument");
    int rc = orbfe_pipeline_synchronize(p);
    if (rc) return rc;
    out[0] = out[1] = out[2] = 0.f;
    auto med = [](std::vector<float>& v) {
        if (v.empty()) return 0.f;
        std::sort(v.begin(), v.end());
        const size_t m = v.size();
        return (m & 1) ? v[m / 2] : 0.5f * (v[m / 2 - 1] + v[m / 2]);
    };
    constexpr long H = orbfe_pipeline::HIST;
    std::vector<float> a, b, g;
    const long nm = std::min(p->match_steps, H), ng = std::min(p->gather_steps, H);
    for (long k = last ? std::max(0L, nm - 1) : 0; k < nm; k++) {
        hipEvent_t* e = p->match_ev[last ? (p->match_steps - 1) % H : k];
        float t0 = 0, t1 = 0;
        if (hipEventElapsedTime(&t0, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&t1, e[1], e[2]) == hipSuccess) { a.push_back(t0 * 1000.f); b.push_back(t1 * 1000.f); }
    }
    for (long k = last ? std::max(0L, ng - 1) : 0; k < ng; k++) {
        hipEvent_t* e = p->gather_ev[last ? (p->gather_steps - 1) % H : k];
        float t0 = 0;
        if (hipEventElapsedTime(&t0, e[0], e[1]) == hipSuccess) g.push_back(t0 * 1000.f);
    }
    (void)hipGetLastError();
    out[0] = med(a); out[1] = med(b); out[2] = med(g);
    return ORBFE_OK;
}

// ---- multi-GPU
int orbfe_pipeline_comm_unique_id(uint8_t id[128])
{
    Rccl* R = rccl();
    if (!R || !id) return fail(ORBFE_ERR_HIP, "orbfe_pipeline_comm_unique_id: librccl is not available");
    Id128 u{};
    ORBFE_NCCL(R->GetUniqueId(&u));
    memcpy(id, u.internal, 128);
    return ORBFE_OK;
}

static int attach_comm(orbfe_pipeline* p, void* comm, bool own, int rank, int world, int dst)
{
    if (p->step_no) { int rc = orbfe_pipeline_synchronize(p); if (rc) return rc; }
    p->comm = comm; p->own_comm = own; p->rank = rank; p->world = world; p->dst = dst;
    if (rank == dst && !p->blocks) {
        ORBFE_HIP(hipMalloc(&p->blocks, (size_t)p->R * world * p->lay.nbytes));
        ORBFE_HIP(hipMemset(p->blocks, 0, (size_t)p->R * world * p->lay.nbytes));
    }
    if (p->gather_free.empty()) {
        p->gather_free.assign((size_t)p->R, nullptr);
        p->gather_free_valid.assign((size_t)p->R, 0);
        for (auto& e : p->gather_free) ORBFE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return ORBFE_OK;
}

int orbfe_pipeline_comm_init(orbfe_pipeline* p, const uint8_t id[128], int rank, int world, int dst)
{
    Rccl* R = rccl();
    if (!R) return fail(ORBFE_ERR_HIP, "orbfe_pipeline_comm_init: librccl is not available");
    if (!p || !id || world < 1 || rank < 0 || rank >= world || dst < 0 || dst >= world || p->comm) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_comm_init: invalid argument");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    Id128 u{};
    memcpy(u.internal, id, 128);
    void* comm = nullptr;
    ORBFE_NCCL(R->CommInitRank(&comm, world, u, rank));
    return attach_comm(p, comm, true, rank, world, dst);
}

int orbfe_pipeline_set_comm(orbfe_pipeline* p, void* nccl_comm, int rank, int world, int dst)
{
    if (!rccl()) return fail(ORBFE_ERR_HIP, "orbfe_pipeline_set_comm: librccl is not available");
    if (!p || !nccl_comm || world < 1 || rank < 0 || rank >= world || dst < 0 || dst >= world || p->comm) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_set_comm: invalid argument");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    return attach_comm(p, nccl_comm, false, rank, world, dst);
}

int orbfe_pipeline_gathered_set(orbfe_pipeline* p, int set, int rank, uint8_t** d_block)
{
    if (!p || !d_block || !p->comm || p->rank != p->dst || !p->blocks || rank < 0 || rank >= p->world || set < 0 || set >= p->R)
        return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_gathered_set: not the destination rank, no communicator, or no such set / rank");
    *d_block = p->blocks + ((size_t)set * p->world + (size_t)rank) * p->lay.nbytes;
    return ORBFE_OK;
}

int orbfe_pipeline_gathered(orbfe_pipeline* p, int rank, uint8_t** d_block)
{
    if (!p || p->last_gathered < 0) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_gathered: nothing has been gathered yet (a batch's gather is enqueued with its matching: step, or flush)");
    return orbfe_pipeline_gathered_set(p, p->last_gathered, rank, d_block);
}

int orbfe_pipeline_gathered_wait(orbfe_pipeline* p, int set)
{
    if (!p || !p->comm || set < 0 || set >= p->R) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_gathered_wait: no communicator, or no such set");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    if (p->pending == set && (rc = orbfe_pipeline_flush(p))) return rc;   // the set's post-work (matching, gather) was still held back
    if (p->gather_batch.size() != (size_t)p->R || p->gather_batch[(size_t)set] < 0)
        return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_gathered_wait: no batch has been gathered into record set %d yet", set);
    ORBFE_HIP(hipEventSynchronize(p->gather_done[(size_t)set]));
    return ORBFE_OK;
}

int orbfe_pipeline_gather_plan(int rank, int world, int dst, int record_set, int record_sets, size_t nbytes, orbfe_gather_op* ops, int capacity)
{
    const int n = gather_plan(rank, world, dst, record_set, record_sets, nbytes, ops, capacity);
    if (n < 0) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_gather_plan: invalid rank / world / dst / record set, or capacity below world + 1");
    return n;
}

int orbfe_pipeline_gathered_batch(orbfe_pipeline* p, int set, long long* batch)
{
    if (!p || !batch || !p->comm || set < 0 || set >= p->R) return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_gathered_batch: no communicator, or no such set");
    *batch = p->gather_batch.size() == (size_t)p->R ? p->gather_batch[(size_t)set] : -1;
    return ORBFE_OK;
}

int orbfe_pipeline_gathered_release(orbfe_pipeline* p, int set, void* stream)
{
    if (!p || !p->comm || p->rank != p->dst || set < 0 || set >= p->R || p->gather_free.empty())
        return fail(ORBFE_ERR_INVALID, "orbfe_pipeline_gathered_release: not the destination rank, no communicator, or no such set");
    int rc = use_device(p->cfg.device);
    if (rc) return rc;
    ORBFE_HIP(hipEventRecord(p->gather_free[(size_t)set], reinterpret_cast<hipStream_t>(stream)));
    p->gather_free_valid[(size_t)set] = 1;
    return ORBFE_OK;
}
