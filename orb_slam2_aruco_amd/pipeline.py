"""Host side of the batched-video mode (BASELINE north_star): one B-frame batch resident in HBM goes through

    ORBextractor::operator()          (Frame.cc:200-206)  -> orbfe_extract_batch_device
    MarkerDetector::detect(img, cam, 0.187) (Frame.cc:142) -> orbfe_aruco_detect_batch_device + orbfe_marker_poses_batch_device
    frame t vs t-1 descriptor matching (ORBmatcher)        -> orbfe_knn2_batch_device + orbfe_search_for_initialization_batch_device

on three HIP streams, into fixed-capacity result records.  bench.py times exactly this class and the GPU tests check exactly
this class against the oracle, so the tested code is the benchmarked code.  torch is plumbing here (device buffers, streams,
events, torch.distributed); every computation is a call into liborbfe.so.
"""
import ctypes
import os

import numpy as np

from . import binding

TUM1_K = [517.306408, 516.469215, 318.643040, 255.313989]       # Examples/Monocular/TUM1.yaml
TUM1_DIST = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
MARKER_SIZE = 0.187                                             # Frame.cc:131


def _up(v):
    return (v + 255) // 256 * 256


def _stream(torch, dev, priority=0):
    """A non-blocking HIP stream of the given priority (-1 high, 0 normal, 1 low; torch's pool only has 0 and -1) as a torch stream."""
    if priority == 0:
        return torch.cuda.Stream(dev)
    hip = ctypes.CDLL("libamdhip64.so")
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(priority))   # hipStreamNonBlocking
    if rc != 0:
        raise RuntimeError("hipStreamCreateWithPriority: %d" % rc)
    return torch.cuda.ExternalStream(h.value, device=dev)


class RecordLayout:
    """One result set = ONE contiguous buffer, the record SURVEY 8e gathers, array by array over the B frames:
    {kp[B][cap] x 28 B | desc[B][cap] x 32 B | n_kp[B] | markers[B][mcap] x 36 B | n_mk[B] | poses[B][mcap] x 56 B}."""

    def __init__(self, B, cap, mcap):
        self.B, self.cap, self.mcap = B, cap, mcap
        self.kps = 0
        self.desc = _up(B * cap * 28)
        self.n = self.desc + _up(B * cap * 32)
        self.mk = self.n + _up(B * 4)
        self.nmk = self.mk + _up(B * mcap * 36)
        self.pose = self.nmk + _up(B * 4)
        self.nbytes = self.pose + _up(B * mcap * 56)

    def unpack(self, buf):
        """bytes of one record set (numpy uint8) -> dict of per-frame arrays (views)."""
        B, cap, mcap = self.B, self.cap, self.mcap
        out = {"n": buf[self.n:self.n + B * 4].view(np.int32),
               "kps": buf[self.kps:self.kps + B * cap * 28].view(binding.KP_DTYPE).reshape(B, cap),
               "desc": buf[self.desc:self.desc + B * cap * 32].reshape(B, cap, 32)}
        if mcap:
            out["nmk"] = buf[self.nmk:self.nmk + B * 4].view(np.int32)
            out["markers"] = buf[self.mk:self.mk + B * mcap * 36].view(binding.MARKER_DTYPE).reshape(B, mcap)
            out["poses"] = buf[self.pose:self.pose + B * mcap * 56].view(binding.POSE_DTYPE).reshape(B, mcap)
        return out


def valid_records(rec, use_orb=True):
    """The defined part of an unpacked record set (entries past a frame's count are unspecified): a list of per-frame tuples."""
    out = []
    for f in range(len(rec["n"])):
        n = int(rec["n"][f]) if use_orb else 0
        item = [n, rec["kps"][f, :n].tobytes(), rec["desc"][f, :n].tobytes()]
        if "nmk" in rec:
            m = min(int(rec["nmk"][f]), rec["markers"].shape[1])
            item += [int(rec["nmk"][f]), rec["markers"][f, :m].tobytes(), rec["poses"][f, :m].tobytes()]
        out.append(tuple(item))
    return out


class FrontEndPipeline:
    """Extractor, detector and matching of one stream of frames on one GPU.

    step(d_imgs) enqueues one batch (B frames, rows x pitch bytes each, resident on the device) and returns at once; result
    set i % R (R = 4) receives batch i, so the matching of batch i (third stream) -- and on N > 1 its gather (communication
    stream) -- overlap with the engines of batch i + 1.  The engines are joined where their results meet: before the
    gather and in synchronize()."""

    def __init__(self, frames, rows, cols, nfeatures=1000, nlevels=8, dictionary="ARUCO", device=0, marker_capacity=64,
                 use_orb=True, use_aruco=True, splits=1, gather=None, lend_aux_stream=True, engine_sets=None, record_sets=4, gather_stream="match", phase_pin=2):
        import torch
        self.torch = torch
        self.L = binding.load()
        self.B, self.rows, self.cols = frames, rows, cols
        self.pitch = (cols + 63) // 64 * 64
        self.use_orb, self.use_aruco = use_orb, use_aruco
        self.device = device
        self.dev = dev = torch.device("cuda", device)
        B = frames
        S = self.S = max(1, min(splits, B // 2))
        self.bounds = [B * k // S for k in range(S + 1)]          # sub-batch k = frames bounds[k] .. bounds[k+1]
        # Engine sets: consecutive batches alternate between D sets of extractor handles and streams (a set owns its pyramid, candidate
        # and keypoint workspaces), so that batch i + 1's resize / FAST run next to batch i's quadtree / descriptors -- the tail of the
        # extractor chain is latency-bound (quadtree 116 us + descriptors 350 us in the pipeline for 140 us of VALU issue) and leaves
        # issue slots free.  Round 2 measured two sets as a loss (2.02 against 1.89 ms, when the detector chain with its 540 us
        # k_decode and the 640 us k_search_init set the step); with those split up (round 3) two EXTRACTOR sets win: 1.4955 against
        # 1.5288 ms per C2 step (four interleaved runs each), a second detector set still loses (1.628).  Default: 2 / 1.
        # Frames of more than a megapixel (1920 x 1080: 4.84 against 5.00 ms per 100-frame step) keep one set: a single batch of
        # them keeps the chip busy through the extractor's tail, and the second set's workspace traffic costs more than it hides.
        if engine_sets is None:
            engine_sets = 2 if rows * cols <= 1280 * 720 else 1
        D = self.D = max(1, int(os.environ.get("ORBFE_ENGINE_SETS", engine_sets)))
        DA = self.DA = max(1, int(os.environ.get("ORBFE_ENGINE_SETS_ARUCO", 1)))
        self.ex_sets = [[binding.ORBextractor(nfeatures, 1.2, nlevels, 20, 7, device=device) for _ in range(S)] for _ in range(D)]
        self.exs = [e for es in self.ex_sets for e in es]
        self.ex = self.exs[0]
        # Phase lock of the engine sets (ORBFE_PHASE_PIN = stage 1 .. 3, 0 = free running): set d's batches start behind that stage of
        # set d - 1's latest batch, round the ring.  Free running, the two sets' chains drift into whatever phase the contention of
        # the moment leaves them in: the step time was bimodal from run to run (1.38 / 1.52 ms with the blur on the sets' own
        # streams) and got LONGER when kernels got cheaper (the address-arithmetic rewrite of round 3 -- fewer instructions, k_blur7
        # at 64 registers -- took the free-running step 1.42 -> 1.49 ms).  Behind the other set's QUADTREE (stage 2) batch i + 1's
        # resize / FAST run next to batch i's blur join and descriptors, every step: C2 1.49 -> 1.355 ms (1.33 - 1.39 over eight
        # interleaved runs; the old kernels under the same lock: 1.40), C3 4.19 -> 3.95; behind FAST (1) 1.45, behind the
        # descriptors (3) 1.47.
        self.phase_pin = int(os.environ.get("ORBFE_PHASE_PIN", phase_pin))
        # The detector's phase (ORBFE_DET_PIN = stage of the extractor's PREVIOUS batch its batch starts behind; + 10: of the current
        # batch; 0 = free running).  Free running the C2 step spread over 1.31 - 1.39 ms from run to run (two attractors); behind the
        # previous batch's RESIZE CHAIN (4) 1.329 - 1.367 with the mean a little lower (1.3435 against 1.3495, twelve interleaved runs
        # each; C3 3.98 against 4.00, gather branch 1.375 against 1.392).  Behind its FAST (1) 1.357, behind the current batch's
        # stages (11 / 12 / 14) 1.348 / 1.381 / 1.365.
        self.det_pin = int(os.environ.get("ORBFE_DET_PIN", 4))
        if self.phase_pin and D > 1 and S == 1:
            for d in range(D):
                self.ex_sets[d][0].follow(self.ex_sets[(d - 1) % D][0], self.phase_pin)
        if os.environ.get("ORBFE_BLUR_PLACE"):            # A/B of the blur's fork point: 0 after FAST, 1 before FAST, 2 no fork
            for e in self.exs:
                e.L.orbfe_extractor_debug_kernel_times(e.h, None, 20 + int(os.environ["ORBFE_BLUR_PLACE"]))
        if os.environ.get("ORBFE_NO_LEND"):
            lend_aux_stream = False
        self.cap = cap = self.ex.capacity
        self.det_sets = [[binding.MarkerDetector(dictionary, device=device) for _ in range(S)] for _ in range(DA)] if use_aruco else []
        self.dets = [d for ds in self.det_sets for d in ds]
        self.det = self.dets[0] if use_aruco else None
        # marker records per frame in the result set (the detector clamps a frame's count to it; its own limit is 256 candidates)
        self.mcap = mcap = min(self.det.capacity, marker_capacity) if use_aruco else 0
        self.layout = lay = RecordLayout(B, cap, mcap)
        # camera of the reference's monocular example; the detector is handed CamSize 1280x720 (Frame.cc:132), so the matrix
        # is rescaled to the frame size before the marker poses (markerdetector_impl.cpp:1110-1172)
        self.cam_K = binding.camera_resize(np.array(TUM1_K, np.float32), (1280, 720), (cols, rows)) if use_aruco else None
        self.cam_D = np.array(TUM1_DIST, np.float32)
        # Result sets in rotation: batch i writes set i % R.  A set is reused only when the matching (and, on N > 1, the gather) of the
        # batch that last wrote it is done, which ties every engine to the slowest one R batches back.  With two sets the gather
        # branch cost 12 % (the detector, 1.05 ms per batch, may run ahead of the extractor chain, 1.47 ms, by less than two
        # batches); four sets (80 MB per rank) take the coupling out: 1.65 -> see profiles/r03_record_sets.txt.
        R = self.R = max(2, int(os.environ.get("ORBFE_RECORD_SETS", record_sets)))
        self.recs = [torch.zeros(lay.nbytes, dtype=torch.uint8, device=dev) for _ in range(R)]
        self.rec_ptr = [r.data_ptr() for r in self.recs]
        z = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
        # matching outputs of the newest batch: pair p = frame p (queries / F1) against frame p + 1 (train / F2)
        self.d_bidx, self.d_bdist, self.d_sdist, self.d_m12 = z(B - 1, cap), z(B - 1, cap), z(B - 1, cap), z(B - 1, cap)
        self.d_nm = z(B - 1)
        # three HIP streams: the ORB extractor, the ArUco detector, the matching.  The first two only read the resident
        # frames; the matching of batch i reads result set i % 2 while batch i+1 is extracted into the other set.
        # So the latency-bound kernels (contours, quadtree, SearchForInitialization) overlap with the VALU-bound ones.
        # None of them is the null stream: work on the legacy default stream synchronises implicitly with every blocking
        # stream of the process (measured: 2.30 ms per C2 step with the extractor on the null stream, 2.01 ms on its own).
        prio = [int(v) for v in os.environ.get("ORBFE_STREAM_PRIO", "0,0,0").split(",")]   # experiment: extractor, detector, matching
        self.stream, self.stream2, self.stream3 = (_stream(torch, dev, p) for p in prio)
        self.sp3 = ctypes.c_void_p(self.stream3.cuda_stream)
        self.orb_stream_sets = [[self.stream] + [torch.cuda.Stream(dev) for _ in range(S - 1)]] + \
                               [[torch.cuda.Stream(dev) for _ in range(S)] for _ in range(D - 1)]
        self.aru_stream_sets = [[self.stream2] + [torch.cuda.Stream(dev) for _ in range(S - 1)]] + \
                               [[torch.cuda.Stream(dev) for _ in range(S)] for _ in range(DA - 1)]
        self.orb_streams, self.aru_streams = self.orb_stream_sets[0], self.aru_stream_sets[0]
        self.last_set = self.last_aset = 0
        if S == 1 and D > 1 and lend_aux_stream and os.environ.get("ORBFE_LEND_ALL", "1") != "0":
            # every set's blur on the matching stream (1.4466 against 1.4873 ms with the handles' own fork streams, which share
            # hardware queues with the busy ones)
            # (experiment ORBFE_BLUR_LEND = det / other: the blur on the detector's stream / on the OTHER extractor set's stream)
            lend = os.environ.get("ORBFE_BLUR_LEND", "match")
            for d, es in enumerate(self.ex_sets):
                for e in es:
                    if lend == "det" and use_aruco:
                        e.set_aux_stream(ctypes.c_void_p(self.aru_stream_sets[0][0].cuda_stream))
                    elif lend == "other":
                        e.set_aux_stream(ctypes.c_void_p(self.orb_stream_sets[(d + 1) % D][0].cuda_stream))
                    else:
                        e.set_aux_stream(self.sp3)
        if S == 1 and D == 1 and lend_aux_stream:
            # ROCm maps streams onto 4 hardware queues, and two busy streams on one queue serialise.  The extractor's forked
            # launch (the blur) is lent the matching stream; measured against the handle's own fork stream and against one
            # shared fork stream for both engines: 2.02 vs 2.12 vs 2.14 ms per step.
            self.ex.set_aux_stream(self.sp3)
            # The fourth queue: the detector's /2 pyramid (~100 us per batch) and the extractor's FAST of level 0 (~120 us), both
            # wanted at the start of a batch and both independent of everything else, share one stream -- a fifth stream would
            # share a hardware queue with a busy one (measured: FAST of level 0 ran behind the whole detector chain).
            # Experiment (ORBFE_FAST0=1 ORBFE_EARLY_SHARED=1): no gain, off by default.
            if use_aruco and use_orb and os.environ.get("ORBFE_EARLY_SHARED", "0") != "0":
                self.stream4 = torch.cuda.Stream(dev)
                self.sp4 = ctypes.c_void_p(self.stream4.cuda_stream)
                self.det.set_aux_stream(self.sp4)
                self.ex.set_early_stream(self.sp4)
        # The detector's /2 pyramid in line on the detector's stream instead of forked onto the handle's second stream: one active
        # stream fewer.  Measured as "no gain" while the engine sets were free-running; under the phase lock, frames up to VGA size:
        # C2 1.360 -> 1.320 ms (six interleaved runs each, 1.305 - 1.335), the gather branch 1.393 -> 1.364; 1280 x 720 loses (3.99 ->
        # 4.05: the pyramid there is 0.3 ms of HBM-bound work worth hiding), 1920 x 1080 does not care.
        self.det_nofork = os.environ.get("ORBFE_DET_NOFORK", "1" if rows * cols <= 640 * 480 else "0") != "0"
        if use_aruco and S == 1 and self.det_nofork:
            for dset, sset in zip(self.det_sets, self.aru_stream_sets):
                dset[0].set_aux_stream(ctypes.c_void_p(sset[0].cuda_stream))
        ev = lambda **kw: torch.cuda.Event(**kw)
        self.ex_done = [[ev() for _ in range(S)] for _ in range(R)]
        self.det_done = [[ev() for _ in range(S)] for _ in range(R)]
        self.match_done = [ev() for _ in range(R)]
        self.gather_done = [ev() for _ in range(R)]
        # around the matching launches (their stream); one event triple per step, the newest 64 steps kept for the median
        self.match_evs = [[ev(enable_timing=True) for _ in range(3)] for _ in range(64)]
        self.match_ev = self.match_evs[0]
        self.match_steps = 0
        self.gather_evs = [[ev(enable_timing=True) for _ in range(2)] for _ in range(64)]
        self.gather_steps = 0
        # The batch's gather needs a stream.  ROCm runs the process's streams on four hardware queues and the engines use four
        # (extractor sets, detector, matching + blur, the detector's /2 pyramid): a stream of its own for the collective is a fifth
        # ACTIVE one, and that alone -- one event record per batch on it, the collective replaced by a no-op, no waits -- costs the
        # C2 step 9 % (1.50 -> 1.63 ms; profiles/r03_gather_stream.txt).  So the gather is issued on a stream that exists anyway:
        # "match" (default): behind this batch's matching, which has waited for the extractor already; "det": behind the
        # detector's poses; "comm": the separate stream (the round-2 arrangement).
        which = os.environ.get("ORBFE_GATHER_STREAM", gather_stream)
        self.comm_stream = {"match": self.stream3, "det": self.stream2}.get(which) or torch.cuda.Stream(dev)
        self.gather_stream_name = which
        self.gather = gather            # sharding.RecordGather or None (single GPU)
        self.step_no = 0
        self.pending = None
        # (measured: C2 1.4924 -> 1.4759 ms, five interleaved runs each; 1280 x 720: 4.40 -> 4.46, so only for frames up to VGA size)
        self.defer_post = os.environ.get("ORBFE_DEFER_POST", "1" if rows * cols <= 640 * 480 else "0") != "0"
        self.big_frames = False

    # ------------------------------------------------------------------------------------------------------------
    def upload(self, frames_u8):
        """(B, rows, cols) uint8 host frames -> resident device batch with 64-byte aligned rows."""
        torch = self.torch
        d = torch.zeros((len(frames_u8), self.rows, self.pitch), dtype=torch.uint8, device=self.dev)
        d[:, :, :self.cols] = torch.from_numpy(np.ascontiguousarray(frames_u8)).to(self.dev)
        torch.cuda.synchronize(self.dev)     # the engines run on their own streams, not on the one that filled the batch
        return d

    def step(self, d_imgs):
        """Enqueue one batch; returns the result-set index (0 / 1) it writes."""
        L, lay, S, B = self.L, self.layout, self.S, self.B
        rows, cols, pitch, cap, mcap = self.rows, self.cols, self.pitch, self.cap, self.mcap
        i = self.step_no
        self.step_no += 1
        cur = i % self.R
        eset = self.last_set = i % self.D
        aset = self.last_aset = i % self.DA
        exs, dets = self.ex_sets[eset], (self.det_sets[aset] if self.use_aruco else [])
        orb_streams, aru_streams = self.orb_stream_sets[eset], self.aru_stream_sets[aset]
        base = self.rec_ptr[cur]
        img0 = d_imgs.data_ptr()
        multi = self.gather is not None
        def enqueue_detector():
            if self.use_aruco:
                # the detector streams only depend on the (resident) input frames and on their own previous batch, so they are
                # not joined with the ORB streams per step: consecutive batches of the two engines pipeline freely.
                for k in range(S):
                    f0, nf = self.bounds[k], self.bounds[k + 1] - self.bounds[k]
                    st = aru_streams[k]
                    if multi and i >= self.R:
                        st.wait_event(self.gather_done[cur])         # batch i-R has left this record set
                    sp = ctypes.c_void_p(st.cuda_stream)
                    if self.det_pin and self.use_orb:
                        # experiment (ORBFE_DET_PIN = stage, + 10: of THIS batch's extractor, which is then enqueued first): the detector's
                        # batch starts behind a stage of the extractor's previous / current batch
                        j = i if self.det_pin >= 10 else i - 1
                        if j >= 0:
                            binding._check(L, L.orbfe_extractor_stage_wait(self.ex_sets[j % self.D][k].h, self.det_pin % 10, sp), "stage_wait")
                    dets[k].detect_batch_device(img0 + f0 * rows * pitch, nf, rows * pitch, rows, cols, pitch,
                                                     base + lay.mk + f0 * mcap * 36, mcap, base + lay.nmk + f0 * 4, sp)
                    # detect(image, CameraParameters, 0.187): every marker gets its IPPE pose (markerdetector_impl.cpp:8720-8780)
                    binding._check(L, L.orbfe_marker_poses_batch_device(
                        base + lay.mk + f0 * mcap * 36, base + lay.nmk + f0 * 4, mcap, nf, MARKER_SIZE,
                        self.cam_K.ctypes.data_as(ctypes.c_void_p), self.cam_D.ctypes.data_as(ctypes.c_void_p), len(self.cam_D),
                        base + lay.pose + f0 * mcap * 56, sp), "orbfe_marker_poses_batch_device")
                    self.det_done[cur][k].record(st)

        def enqueue_extractor():
            if self.use_orb:
                for k in range(S):
                    f0, nf = self.bounds[k], self.bounds[k + 1] - self.bounds[k]
                    st = orb_streams[k]
                    if i >= self.R:
                        st.wait_event(self.match_done[cur])          # the matching of batch i-2 has read this record set
                        if multi:
                            st.wait_event(self.gather_done[cur])
                    exs[k].extract_batch_device(img0 + f0 * rows * pitch, nf, rows * pitch, rows, cols, pitch,
                                                     base + lay.kps + f0 * cap * 28, base + lay.desc + f0 * cap * 32, cap,
                                                     base + lay.n + f0 * 4, ctypes.c_void_p(st.cuda_stream))
                    self.ex_done[cur][k].record(st)

        if self.det_pin >= 10:
            enqueue_extractor(); enqueue_detector()
        else:
            enqueue_detector(); enqueue_extractor()
        # What follows a batch's engines -- its matching and, on N > 1, its gather -- goes onto the matching stream, which also
        # carries the extractor's blur (lent).  Enqueued right away, the matching of batch i (which waits for the whole extractor
        # chain of batch i) would sit IN FRONT of the blur of batch i + 1 on that stream, and the descriptors of batch i + 1 wait
        # for that blur: descriptors(i) -> matching(i) -> blur(i+1) -> descriptors(i+1), one after the other, although the second
        # extractor set has long been ready.  So the post-work of batch i is enqueued one step late, behind the blur of batch i + 1.
        if self.defer_post:
            if self.pending is not None:
                self._enqueue_post(self.pending)
            self.pending = cur
        else:
            self._enqueue_post(cur)
        return cur

    def flush(self):
        """Enqueue the post-work (matching, gather) of the newest batch if it is still held back; call before synchronising."""
        if self.pending is not None:
            self._enqueue_post(self.pending)
            self.pending = None

    def _enqueue_post(self, cur):
        S = self.S
        multi = self.gather is not None
        if self.use_orb:
            for k in range(S):
                self.stream3.wait_event(self.ex_done[cur][k])
            self.enqueue_matching(cur)
            if os.environ.get("ORBFE_MATCH_TWICE"):      # sensitivity study only (tools/sensitivity.sh): the matching launched twice
                self.enqueue_matching(cur)
        if multi:
            # the batch's one collective: it waits for the engines of THIS batch and runs while the next batches are
            # computed into the other record sets (on which stream: see __init__)
            with self.torch.cuda.stream(self.comm_stream):
                for k in range(S):
                    if self.use_orb:
                        self.comm_stream.wait_event(self.ex_done[cur][k])
                    if self.use_aruco:
                        self.comm_stream.wait_event(self.det_done[cur][k])
                ge = self.gather_evs[self.gather_steps % len(self.gather_evs)]
                self.gather_steps += 1
                ge[0].record(self.comm_stream)
                self.gather(self.recs[cur])
                ge[1].record(self.comm_stream)
                self.gather_done[cur].record(self.comm_stream)
        return cur

    def enqueue_matching(self, cur):
        """Frame t vs t-1 over result set `cur` on the matching stream: all-pairs knn2 + one SearchForInitialization-style
        windowed pass."""
        L, lay, B, cap = self.L, self.layout, self.B, self.cap
        base = self.rec_ptr[cur]
        e = self.match_ev = self.match_evs[self.match_steps % len(self.match_evs)]
        self.match_steps += 1
        e[0].record(self.stream3)
        binding._check(L, L.orbfe_knn2_batch_device(base + lay.desc, base + lay.n, cap * 32, cap,
                                                    base + lay.desc + cap * 32, base + lay.n + 4, cap * 32, cap,
                                                    B - 1, 256, self.d_bidx.data_ptr(), self.d_bdist.data_ptr(),
                                                    self.d_sdist.data_ptr(), self.sp3), "orbfe_knn2_batch_device")
        e[1].record(self.stream3)
        binding._check(L, L.orbfe_search_for_initialization_batch_device(
            base + lay.kps, base + lay.desc, base + lay.n, cap, B - 1, self.cols, self.rows, None, 100, 0.9, 1,
            self.d_m12.data_ptr(), self.d_nm.data_ptr(), self.sp3), "orbfe_search_for_initialization_batch_device")
        e[2].record(self.stream3)
        self.match_done[cur].record(self.stream3)

    def last_engines(self):
        """(extractor, detector) handles that ran the most recent step (their launch timers describe that step)."""
        return self.ex_sets[self.last_set][0], (self.det_sets[self.last_aset][0] if self.use_aruco else None)

    def synchronize(self):
        self.flush()
        self.torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------------------------------------------------
    def status(self):
        """Capacity flags of the last batch of every engine (synchronises): dict, all zero = results complete."""
        out = {"extractor_overflow": 0, "search_init_overflow": 0, "aruco_flagged_frames": 0, "aruco_flags": 0}
        self.flush()                 # the newest batch's matching may still be held back (defer_post)
        if self.use_orb:
            out["extractor_overflow"] = max(e.batch_status() for e in self.exs)
            ovf = ctypes.c_int32(0)
            binding._check(self.L, self.L.orbfe_search_for_initialization_batch_status(self.sp3, ctypes.byref(ovf)),
                           "orbfe_search_for_initialization_batch_status")
            out["search_init_overflow"] = ovf.value
        for d in self.dets:
            n, fl = d.batch_status()
            out["aruco_flagged_frames"] += n
            out["aruco_flags"] |= fl
        return out

    def warmup(self, d_imgs, steps):
        """Untimed steps; afterwards the capacity flags are asked once: frames with more long contours than the LDS-resident
        contour kernels hold (large, busy images) switch the detector to its big-frame kernel, and a SearchForInitialization
        candidate overflow has grown the scratch, so the steps are repeated once."""
        for _ in range(max(steps, 1)):
            self.step(d_imgs)
        self.synchronize()
        st = self.status()
        again = False
        if st["aruco_flagged_frames"]:
            for d in self.dets:
                d.set_big_frames(True)
            self.big_frames = again = True
        if st["search_init_overflow"]:
            again = True
        if again:
            for _ in range(max(steps, 1)):
                self.step(d_imgs)
            self.synchronize()
            st = self.status()
        if any(st.values()):
            raise binding.OrbfeError("front-end capacity exceeded at this frame size: %r" % (st,))

    # ------------------------------------------------------------------------------------------------------------
    def read_records(self, cur):
        """Result set `cur` as host arrays (flushes the held-back post-work and synchronises)."""
        self.synchronize()
        return self.layout.unpack(self.recs[cur].cpu().numpy())

    def read_matches(self):
        """Matching outputs of the newest batch: dict of (B-1, cap) arrays + nmatches (B-1)."""
        self.synchronize()           # the newest batch's matching is enqueued one step late (defer_post): flush before reading
        g = lambda t: t.cpu().numpy()
        return {"best_idx": g(self.d_bidx), "best_dist": g(self.d_bdist), "second_dist": g(self.d_sdist),
                "matches12": g(self.d_m12), "nmatches": g(self.d_nm)}

    def reset_timing_history(self):
        self.match_steps = 0
        self.gather_steps = 0

    def matching_times_us(self, median=False):
        """(knn2, SearchForInitialization) launch times of the newest step, or their medians over the steps since
        reset_timing_history() (the newest 64); after synchronize()."""
        if not median:
            e = self.match_ev
            return e[0].elapsed_time(e[1]) * 1000.0, e[1].elapsed_time(e[2]) * 1000.0
        n = min(self.match_steps, len(self.match_evs))
        t = np.array([[e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])] for e in self.match_evs[:n]]) * 1000.0
        return float(np.median(t[:, 0])), float(np.median(t[:, 1]))

    def gather_times_us(self):
        """Median duration of the batch gather on the communication stream over the steps since reset_timing_history()."""
        n = min(self.gather_steps, len(self.gather_evs))
        if n == 0:
            return None
        return float(np.median([e[0].elapsed_time(e[1]) for e in self.gather_evs[:n]]) * 1000.0)
