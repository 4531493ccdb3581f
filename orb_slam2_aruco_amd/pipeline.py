"""Python face of the batched-video mode (BASELINE north_star).  The mode itself -- engines, HIP streams, events,
record sets in rotation, the phase locks, the deferred matching, the RCCL gather -- is C++ inside liborbfe.so
(csrc/pipeline.hip: orbfe_pipeline_*); this class only marshals arguments for bench.py and the GPU tests, so the tested code is
the benchmarked code and a C++ application gets the same mode from include/orbfe.h (tests/pipeline_driver.cpp runs it without
Python).  No PyTorch here: device memory comes from the HIP runtime through ctypes."""
import ctypes as C

import numpy as np

from . import binding

TUM1_K = [517.306408, 516.469215, 318.643040, 255.313989]       # Examples/Monocular/TUM1.yaml
TUM1_DIST = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
MARKER_SIZE = 0.187                                             # Frame.cc:131


class PipelineConfig(C.Structure):
    """orbfe_pipeline_config (include/orbfe.h)."""
    _fields_ = [("frames", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("nfeatures", C.c_int32), ("nlevels", C.c_int32),
                ("scale_factor", C.c_float), ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("dictionary", C.c_char * 32),
                ("device", C.c_int32), ("marker_capacity", C.c_int32), ("use_orb", C.c_int32), ("use_aruco", C.c_int32),
                ("marker_size", C.c_float), ("K", C.c_float * 4), ("dist", C.c_float * 12), ("ndist", C.c_int32),
                ("window_size", C.c_int32), ("nnratio", C.c_float), ("check_orientation", C.c_int32),
                ("engine_sets", C.c_int32), ("record_sets", C.c_int32), ("phase_pin", C.c_int32), ("det_pin", C.c_int32),
                ("defer_post", C.c_int32), ("det_nofork", C.c_int32)]


class RecordLayoutC(C.Structure):
    """orbfe_record_layout (include/orbfe.h)."""
    _fields_ = [("frames", C.c_int32), ("capacity", C.c_int32), ("marker_capacity", C.c_int32), ("halo", C.c_int32),
                ("off_kps", C.c_uint64), ("off_desc", C.c_uint64), ("off_n", C.c_uint64), ("off_markers", C.c_uint64),
                ("off_nmarkers", C.c_uint64), ("off_poses", C.c_uint64), ("nbytes", C.c_uint64)]


class RecordLayout:
    """One result set = ONE contiguous buffer, the record SURVEY 8e gathers, array by array over the frames:
    {kp[B + 1][cap] x 28 B | desc[B + 1][cap] x 32 B | n_kp[B + 1] | markers[B][mcap] x 36 B | n_mk[B] | poses[B][mcap] x 56 B};
    slot 0 of the keypoint arrays is the halo: the last frame of the previous batch of the stream."""

    def __init__(self, c):
        self.B, self.cap, self.mcap, self.halo = c.frames, c.capacity, c.marker_capacity, c.halo
        self.kps, self.desc, self.n = int(c.off_kps), int(c.off_desc), int(c.off_n)
        self.mk, self.nmk, self.pose = int(c.off_markers), int(c.off_nmarkers), int(c.off_poses)
        self.nbytes = int(c.nbytes)

    def unpack(self, buf):
        """bytes of one record set (numpy uint8) -> dict of per-frame arrays (views); "halo_*" = the slot in front of the batch."""
        B, cap, mcap, h = self.B, self.cap, self.mcap, self.halo
        n = buf[self.n:self.n + (B + h) * 4].view(np.int32)
        kps = buf[self.kps:self.kps + (B + h) * cap * 28].view(binding.KP_DTYPE).reshape(B + h, cap)
        desc = buf[self.desc:self.desc + (B + h) * cap * 32].reshape(B + h, cap, 32)
        out = {"n": n[h:], "kps": kps[h:], "desc": desc[h:], "halo_n": n[:h], "halo_kps": kps[:h], "halo_desc": desc[:h]}
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


class DeviceBatch:
    """A resident input batch: frames in device memory with rows `pitch` bytes apart (orbfe_device_alloc of the library -- the runtime the
    pipeline itself uses; freed with the object)."""

    def __init__(self, frames_u8, pitch, device=0):
        from . import binding
        self.L = binding.load()
        _setup(self.L)
        f = np.ascontiguousarray(frames_u8, np.uint8)
        B, rows, cols = f.shape
        self.shape, self.pitch, self.nbytes = (B, rows, pitch), pitch, B * rows * pitch
        self.ptr = self.L.orbfe_device_alloc(device, self.nbytes)
        if not self.ptr:
            raise RuntimeError("orbfe_device_alloc of %d bytes failed: %s" % (self.nbytes, self.L.orbfe_last_error().decode()))
        rc = self.L.orbfe_device_upload_rows(self.ptr, pitch, f.ctypes.data_as(C.c_void_p), cols, cols, B * rows)
        if rc != 0:
            raise RuntimeError("orbfe_device_upload_rows: %s" % self.L.orbfe_last_error().decode())

    def data_ptr(self):
        return self.ptr

    def __del__(self):
        if getattr(self, "ptr", None):
            self.L.orbfe_device_free(self.ptr)
            self.ptr = None


def device_bytes(ptr, nbytes):
    """nbytes at device address ptr -> numpy uint8 (blocking copy, device to host)."""
    from . import binding
    L = binding.load()
    _setup(L)
    out = np.empty(nbytes, np.uint8)
    if L.orbfe_device_download(out.ctypes.data_as(C.c_void_p), ptr, nbytes) != 0:
        raise RuntimeError("orbfe_device_download: %s" % L.orbfe_last_error().decode())
    return out


def _setup(L):
    if getattr(L, "_pipeline_ready", False):
        return
    vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
    L.orbfe_pipeline_config_default.argtypes = [C.POINTER(PipelineConfig), C.c_int, C.c_int, C.c_int]
    L.orbfe_pipeline_create.argtypes = [C.POINTER(PipelineConfig)]
    L.orbfe_pipeline_create.restype = vp
    L.orbfe_pipeline_destroy.argtypes = [vp]
    L.orbfe_pipeline_destroy.restype = None
    L.orbfe_pipeline_layout.argtypes = [vp, C.POINTER(RecordLayoutC)]
    L.orbfe_pipeline_step.argtypes = [vp, vp, C.c_size_t, i32p]
    for f in ("flush", "synchronize", "reset_stream"):
        getattr(L, "orbfe_pipeline_" + f).argtypes = [vp]
    L.orbfe_pipeline_input_done.argtypes = [vp, C.c_int]
    L.orbfe_pipeline_status.argtypes = [vp, C.POINTER(C.c_int32 * 4)]
    L.orbfe_pipeline_set_big_frames.argtypes = [vp, C.c_int]
    L.orbfe_pipeline_records.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.orbfe_pipeline_matches.argtypes = [vp] + [C.POINTER(vp)] * 5
    L.orbfe_pipeline_extractor.argtypes = [vp, C.c_int]
    L.orbfe_pipeline_extractor.restype = vp
    L.orbfe_pipeline_detector.argtypes = [vp]
    L.orbfe_pipeline_detector.restype = vp
    L.orbfe_pipeline_engine_sets.argtypes = [vp] + [i32p] * 6
    L.orbfe_pipeline_enable_timing.argtypes = [vp, C.c_int]
    L.orbfe_pipeline_timing_us.argtypes = [vp, C.c_int, C.POINTER(C.c_float * 3)]
    L.orbfe_pipeline_env_defaults.restype = C.c_char_p
    L.orbfe_pipeline_comm_unique_id.argtypes = [C.POINTER(C.c_uint8 * 128)]
    L.orbfe_pipeline_comm_init.argtypes = [vp, C.POINTER(C.c_uint8 * 128), C.c_int, C.c_int, C.c_int]
    L.orbfe_pipeline_set_comm.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.orbfe_pipeline_gathered.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.orbfe_pipeline_gathered_set.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.orbfe_pipeline_gathered_wait.argtypes = [vp, C.c_int]
    L.orbfe_pipeline_gathered_release.argtypes = [vp, C.c_int, vp]
    L.orbfe_pipeline_gathered_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_longlong)]
    L.orbfe_pipeline_copy_stream_priority.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orbfe_pipeline_step_host.argtypes = [vp, vp, C.c_size_t, i32p]
    L.orbfe_pipeline_host_records.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.orbfe_pipeline_host_copy_us.argtypes = [vp, C.POINTER(C.c_float * 2)]
    L.orbfe_host_alloc.argtypes = [C.c_size_t]
    L.orbfe_host_alloc.restype = vp
    L.orbfe_host_free.argtypes = [vp]
    L.orbfe_host_free.restype = None
    L.orbfe_device_alloc.argtypes = [C.c_int, C.c_size_t]
    L.orbfe_device_alloc.restype = vp
    L.orbfe_device_free.argtypes = [vp]
    L.orbfe_device_free.restype = None
    L.orbfe_device_upload_rows.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t]
    L.orbfe_device_download.argtypes = [vp, vp, C.c_size_t]
    L._pipeline_ready = True


class PinnedFrames:
    """A batch of frames in page-locked host memory (orbfe_host_alloc): .array is the (B, rows, cols) uint8 view."""

    def __init__(self, frames_u8):
        self.L = binding.load()
        _setup(self.L)
        f = np.ascontiguousarray(frames_u8, np.uint8)
        self.ptr = self.L.orbfe_host_alloc(f.nbytes)
        if not self.ptr:
            raise binding.OrbfeError("orbfe_host_alloc: " + self.L.orbfe_last_error().decode())
        self.array = np.ctypeslib.as_array((C.c_uint8 * f.nbytes).from_address(self.ptr)).reshape(f.shape)
        self.array[...] = f

    def __del__(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.L.orbfe_host_free(self.ptr)
            self.ptr = None


def env_defaults():
    """{ORBFE_* variable: default} as the library states them (orbfe_pipeline_env_defaults): "size" = decided by the frame size."""
    L = binding.load()
    _setup(L)
    return dict(kv.split("=", 1) for kv in L.orbfe_pipeline_env_defaults().decode().split(";") if kv)


def comm_unique_id():
    """ncclGetUniqueId through the library (128 bytes): rank 0 calls it and hands the bytes to every rank."""
    L = binding.load()
    _setup(L)
    buf = (C.c_uint8 * 128)()
    binding._check(L, L.orbfe_pipeline_comm_unique_id(C.byref(buf)), "orbfe_pipeline_comm_unique_id")
    return bytes(buf)


class FrontEndPipeline:
    """Extractor, detector and matching of one stream of frames on one GPU (orbfe_pipeline_*).

    step(d_imgs) enqueues one batch (B frames, rows x pitch bytes each, resident on the device) and returns at once with the
    record set the batch is written to; flush() / synchronize() before reading the newest batch (its matching is held back one
    step for frames up to 640 x 480)."""

    def __init__(self, frames, rows, cols, nfeatures=1000, nlevels=8, dictionary="ARUCO", device=0, marker_capacity=64,
                 use_orb=True, use_aruco=True, engine_sets=None, record_sets=None, phase_pin=None, det_pin=None, defer_post=None,
                 det_nofork=None):
        self.L = L = binding.load()
        _setup(L)
        self.B, self.rows, self.cols = frames, rows, cols
        self.pitch = (cols + 63) // 64 * 64
        self.use_orb, self.use_aruco = use_orb, use_aruco
        self.device = device
        cfg = PipelineConfig()
        binding._check(L, L.orbfe_pipeline_config_default(C.byref(cfg), frames, rows, cols), "orbfe_pipeline_config_default")
        cfg.nfeatures, cfg.nlevels, cfg.device, cfg.marker_capacity = nfeatures, nlevels, device, marker_capacity
        cfg.dictionary = dictionary.encode()
        cfg.use_orb, cfg.use_aruco = int(use_orb), int(use_aruco)
        for name, v in (("engine_sets", engine_sets), ("record_sets", record_sets), ("phase_pin", phase_pin), ("det_pin", det_pin),
                        ("defer_post", defer_post), ("det_nofork", det_nofork)):
            if v is not None:
                setattr(cfg, name, int(v))
        self.cfg = cfg
        self.cam_K = np.array(list(cfg.K), np.float32)
        self.cam_D = np.array(list(cfg.dist)[:cfg.ndist], np.float32)
        self.h = L.orbfe_pipeline_create(C.byref(cfg))
        if not self.h:
            raise binding.OrbfeError("orbfe_pipeline_create: " + L.orbfe_last_error().decode())
        lc = RecordLayoutC()
        binding._check(L, L.orbfe_pipeline_layout(self.h, C.byref(lc)), "orbfe_pipeline_layout")
        self.layout = RecordLayout(lc)
        self.cap, self.mcap = lc.capacity, lc.marker_capacity
        v = [C.c_int32(0) for _ in range(6)]
        binding._check(L, L.orbfe_pipeline_engine_sets(self.h, *[C.byref(x) for x in v]), "orbfe_pipeline_engine_sets")
        self.D, self.R, self.phase_pin, self.det_pin = v[0].value, v[1].value, v[2].value, v[3].value
        self.defer_post, self.det_nofork = bool(v[4].value), bool(v[5].value)
        # the engines, for their kernel timers and debug entry points (owned by the pipeline)
        self.exs = [binding.ORBextractor.wrap(L.orbfe_pipeline_extractor(self.h, d), nlevels) for d in range(self.D)] if use_orb else []
        self.ex = self.exs[0] if self.exs else None
        dh = L.orbfe_pipeline_detector(self.h)
        self.det = binding.MarkerDetector.wrap(dh) if dh else None
        self.dets = [self.det] if self.det else []
        self.rec_ptr = []
        for s in range(self.R):
            p = C.c_void_p()
            binding._check(L, L.orbfe_pipeline_records(self.h, s, C.byref(p)), "orbfe_pipeline_records")
            self.rec_ptr.append(p.value)
        self.step_no = 0
        self.big_frames = False
        self.world = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orbfe_pipeline_destroy(self.h)
            self.h = None

    # ------------------------------------------------------------------------------------------------------------
    def upload(self, frames_u8):
        """(B, rows, cols) uint8 host frames -> resident device batch with 64-byte aligned rows (keep the object alive while it is used)."""
        return DeviceBatch(frames_u8, self.pitch, self.device)

    def step_ptr(self, ptr, pitch=None):
        cur = C.c_int32(0)
        binding._check(self.L, self.L.orbfe_pipeline_step(self.h, C.c_void_p(ptr), pitch or self.pitch, C.byref(cur)), "orbfe_pipeline_step")
        self.step_no += 1
        return cur.value

    def step(self, d_imgs):
        """Enqueue one batch; returns the record set it writes."""
        return self.step_ptr(d_imgs.data_ptr())

    def step_host(self, pinned):
        """Enqueue one batch whose frames are in host memory (a PinnedFrames, or any C-contiguous (B, rows, cols) uint8 array): the
        pipeline uploads it on its own copy stream and copies the record set back (host_records)."""
        arr = pinned.array if isinstance(pinned, PinnedFrames) else pinned
        cur = C.c_int32(0)
        binding._check(self.L, self.L.orbfe_pipeline_step_host(self.h, C.c_void_p(arr.ctypes.data), arr.strides[1], C.byref(cur)), "orbfe_pipeline_step_host")
        self.step_no += 1
        return cur.value

    def host_records(self, cur):
        """The host copy of record set `cur` (after step_host + flush; waits for its copy), unpacked."""
        p = C.c_void_p()
        binding._check(self.L, self.L.orbfe_pipeline_host_records(self.h, cur, C.byref(p)), "orbfe_pipeline_host_records")
        buf = np.ctypeslib.as_array((C.c_uint8 * self.layout.nbytes).from_address(p.value))
        return self.layout.unpack(buf.copy())

    def flush(self):
        binding._check(self.L, self.L.orbfe_pipeline_flush(self.h), "orbfe_pipeline_flush")

    def synchronize(self):
        binding._check(self.L, self.L.orbfe_pipeline_synchronize(self.h), "orbfe_pipeline_synchronize")

    def input_done(self, cur):
        binding._check(self.L, self.L.orbfe_pipeline_input_done(self.h, cur), "orbfe_pipeline_input_done")

    def reset_stream(self):
        binding._check(self.L, self.L.orbfe_pipeline_reset_stream(self.h), "orbfe_pipeline_reset_stream")

    def last_engines(self):
        """(extractor, detector) handles that ran the most recent step (their launch timers describe that step)."""
        return (self.exs[(self.step_no - 1) % self.D] if self.exs else None), self.det

    # ------------------------------------------------------------------------------------------------------------
    def comm_init(self, unique_id, rank, world, dst=0):
        """The batch's gather over RCCL: every rank calls this with rank 0's comm_unique_id() bytes (ncclCommInitRank inside)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        binding._check(self.L, self.L.orbfe_pipeline_comm_init(self.h, C.byref(buf), rank, world, dst), "orbfe_pipeline_comm_init")
        self.world, self.rank, self.dst = world, rank, dst

    def gathered(self, rank):
        """On the destination rank, after synchronize(): the record set rank `rank` sent with the newest batch, unpacked."""
        p = C.c_void_p()
        binding._check(self.L, self.L.orbfe_pipeline_gathered(self.h, rank, C.byref(p)), "orbfe_pipeline_gathered")
        return self.layout.unpack(device_bytes(p.value, self.layout.nbytes))

    def host_copy_us(self):
        """(upload, read-back) durations in microseconds of the newest steps from host memory, measured on the copy streams."""
        out = (C.c_float * 2)()
        binding._check(self.L, self.L.orbfe_pipeline_host_copy_us(self.h, C.byref(out)), "orbfe_pipeline_host_copy_us")
        return float(out[0]), float(out[1])

    def gathered_set(self, record_set, rank, wait=True):
        """On the destination rank: the block rank `rank` sent with the batch that was written to `record_set` (every record set has
        receive blocks of its own, so this batch stays readable while the following ones arrive); wait: until that gather is done."""
        if wait:
            binding._check(self.L, self.L.orbfe_pipeline_gathered_wait(self.h, record_set), "orbfe_pipeline_gathered_wait")
        p = C.c_void_p()
        binding._check(self.L, self.L.orbfe_pipeline_gathered_set(self.h, record_set, rank, C.byref(p)), "orbfe_pipeline_gathered_set")
        return self.layout.unpack(device_bytes(p.value, self.layout.nbytes))

    def gathered_batch(self, record_set):
        """The batch (step number, from 0) the newest gather into this record set carries; -1 before the first."""
        b = C.c_longlong(-1)
        binding._check(self.L, self.L.orbfe_pipeline_gathered_batch(self.h, record_set, C.byref(b)), "orbfe_pipeline_gathered_batch")
        return int(b.value)

    def copy_stream_priority(self):
        """(priority of the host-mode copy streams, lowest, highest) of the device's stream priority range."""
        a, lo, hi = C.c_int(0), C.c_int(0), C.c_int(0)
        binding._check(self.L, self.L.orbfe_pipeline_copy_stream_priority(self.h, C.byref(a), C.byref(lo), C.byref(hi)), "orbfe_pipeline_copy_stream_priority")
        return a.value, lo.value, hi.value

    def gathered_release(self, record_set, stream=None):
        """The consumer's reads of `record_set`'s blocks enqueued on `stream` so far must finish before the set is received into again."""
        binding._check(self.L, self.L.orbfe_pipeline_gathered_release(self.h, record_set, stream), "orbfe_pipeline_gathered_release")

    # ------------------------------------------------------------------------------------------------------------
    def status(self):
        """Capacity flags since the last call (flushes and synchronises): dict, all zero = results complete."""
        out = (C.c_int32 * 4)()
        binding._check(self.L, self.L.orbfe_pipeline_status(self.h, C.byref(out)), "orbfe_pipeline_status")
        return {"extractor_overflow": out[0], "search_init_overflow": out[1], "aruco_flagged_frames": out[2], "aruco_flags": out[3]}

    def warmup(self, d_imgs, steps):
        """Untimed steps; afterwards the capacity flags are asked once: frames with more long contours than the contour kernels
        hold (large, busy images) switch the detector to its big-frame kernel, and a SearchForInitialization candidate overflow
        has grown the scratch, so the steps are repeated once."""
        for _ in range(max(steps, 1)):
            self.step(d_imgs)
        st = self.status()
        again = False
        if st["aruco_flagged_frames"]:
            binding._check(self.L, self.L.orbfe_pipeline_set_big_frames(self.h, 1), "orbfe_pipeline_set_big_frames")
            self.big_frames = again = True
        if st["search_init_overflow"]:
            again = True
        if again:
            for _ in range(max(steps, 1)):
                self.step(d_imgs)
            st = self.status()
        if any(st.values()):
            raise binding.OrbfeError("front-end capacity exceeded at this frame size: %r" % (st,))

    # ------------------------------------------------------------------------------------------------------------
    def record_bytes(self, cur):
        """Result set `cur` as raw bytes (flushes and synchronises)."""
        self.synchronize()
        return device_bytes(self.rec_ptr[cur], self.layout.nbytes)

    def read_records(self, cur):
        """Result set `cur` as host arrays (flushes the held-back post-work and synchronises)."""
        return self.layout.unpack(self.record_bytes(cur))

    def read_matches(self):
        """Matching outputs of the newest batch: dict of (B, cap) arrays + nmatches (B); row p = slot p (F1) against slot p + 1
        (F2) of the record set, i.e. row 0 = the last frame of the previous batch against frame 0, row p = frame p - 1 against p."""
        self.synchronize()
        ptrs = [C.c_void_p() for _ in range(5)]
        binding._check(self.L, self.L.orbfe_pipeline_matches(self.h, *[C.byref(p) for p in ptrs]), "orbfe_pipeline_matches")
        B, cap = self.B, self.cap
        g = lambda p, n, shape: device_bytes(p.value, n * 4).view(np.int32).reshape(shape)
        return {"best_idx": g(ptrs[0], B * cap, (B, cap)), "best_dist": g(ptrs[1], B * cap, (B, cap)), "second_dist": g(ptrs[2], B * cap, (B, cap)),
                "matches12": g(ptrs[3], B * cap, (B, cap)), "nmatches": g(ptrs[4], B, (B,))}

    def enable_timing(self, on=True):
        binding._check(self.L, self.L.orbfe_pipeline_enable_timing(self.h, int(on)), "orbfe_pipeline_enable_timing")

    def reset_timing_history(self):
        self.enable_timing(True)

    def _timing(self, last):
        out = (C.c_float * 3)()
        binding._check(self.L, self.L.orbfe_pipeline_timing_us(self.h, int(last), C.byref(out)), "orbfe_pipeline_timing_us")
        return float(out[0]), float(out[1]), float(out[2])

    def matching_times_us(self, median=False):
        """(knn2, SearchForInitialization) launch times of the newest step, or their medians over the steps since timing was
        switched on (the newest 64); synchronises."""
        t = self._timing(not median)
        return t[0], t[1]

    def gather_times_us(self):
        """Median duration of the batch gather on the matching stream over the steps since timing was switched on."""
        return self._timing(False)[2] or None
