"""Seeded synthetic inputs for tests and bench.

The reference ships no images or videos (README.md:7 points at an external
drive), so every workload is generated: a textured, FAST-rich background
(three octaves of box-blurred uniform noise plus random rectangles), K fiducial
markers rendered the way ``Dictionary::getMarkerImage_id`` does
(Thirdparty/aruco/aruco/dictionary.cpp:254-342, bit order :277-284) with a
one-bit white quiet zone, pasted under a random homography (tilt <= 35 deg), and
sigma=2 Gaussian pixel noise.  Pure numpy; independent of both the oracle and
the HIP library.
"""
import os
import re

import numpy as np

_TABLES = os.path.join(os.path.dirname(__file__), "csrc", "orbfe_tables.inc")
_dict_cache = {}


def dictionary_codes(name):
    """Codes of a predefined dictionary, parsed from the generated data tables."""
    if not _dict_cache:
        txt = open(_TABLES).read()
        meta = {m.group(1): (int(m.group(2)), int(m.group(4)))
                for m in re.finditer(r'\{"(\w+)", (\d+), (\d+), (\d+), ORBFE_DICT_\w+\}', txt)}
        for m in re.finditer(r"ORBFE_DICT_(\w+)\[\d+\] = \{(.*?)\};", txt, re.S):
            nm = m.group(1)
            codes = [int(c, 16) for c in re.findall(r"0x[0-9a-fA-F]+", m.group(2))]
            nbits, n = meta[nm]
            _dict_cache[nm] = (nbits, codes[:n])
    return _dict_cache[name]


def marker_bits(name, marker_id):
    """n x n bit matrix (row-major, top-left = MSB) of a marker, n = sqrt(nbits)."""
    nbits, codes = dictionary_codes(name)
    n = int(round(nbits ** 0.5))
    code = codes[marker_id]
    bits = np.zeros((n, n), np.uint8)
    b = 0
    for y in range(n - 1, -1, -1):
        for x in range(n - 1, -1, -1):
            bits[y, x] = (code >> b) & 1
            b += 1
    return bits


def render_marker(name, marker_id, bit_size, quiet=1):
    """Marker image: black border of one bit, inner code, `quiet` bits of white around it."""
    bits = marker_bits(name, marker_id)
    n = bits.shape[0]
    cells = np.zeros((n + 2, n + 2), np.uint8)
    cells[1:-1, 1:-1] = bits
    cells = np.pad(cells, quiet, constant_values=1)
    return np.kron(cells, np.ones((bit_size, bit_size), np.uint8)) * 255


def _box_blur(a, r):
    if r <= 0:
        return a
    k = 2 * r + 1
    p = np.pad(a, r, mode="wrap")
    c = np.cumsum(p, axis=0)
    c = np.concatenate([np.zeros((1, c.shape[1])), c], 0)
    a = (c[k:] - c[:-k]) / k
    c = np.cumsum(a, axis=1)
    c = np.concatenate([np.zeros((c.shape[0], 1)), c], 1)
    return (c[:, k:] - c[:, :-k]) / k


def background(h, w, rng, n_rect=24):
    img = np.zeros((h, w), np.float64)
    for r, amp in ((1, 10.0), (3, 50.0), (8, 130.0)):
        img += amp * (_box_blur(rng.random((h, w)), r) - 0.5) * (2 * r + 1) ** 0.8
    img = 128 + img
    for _ in range(n_rect):
        x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
        rw, rh = int(rng.integers(8, max(9, w // 6))), int(rng.integers(8, max(9, h // 6)))
        if rng.random() < 0.25:  # flat patch: exercises the per-cell minThFAST fallback
            img[y0:y0 + rh, x0:x0 + rw] = float(rng.uniform(40, 215))
        else:
            img[y0:y0 + rh, x0:x0 + rw] += float(rng.choice([-70.0, 70.0]))
    return np.clip(img, 0, 255)


def _homography(src, dst):
    """3x3 H with H*src = dst for four point pairs (float64)."""
    A, b = [], []
    for (x, y), (u, v) in zip(src, dst):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.array(A, np.float64), np.array(b, np.float64))
    return np.append(h, 1.0).reshape(3, 3)


def _warp_into(canvas, src, H_src_to_dst):
    """Paste `src` into `canvas` under H (bilinear, inverse mapping over the bounding box)."""
    sh, sw = src.shape
    corners = np.array([[0, 0, 1], [sw, 0, 1], [sw, sh, 1], [0, sh, 1]], np.float64).T
    d = H_src_to_dst @ corners
    d = d[:2] / d[2]
    x0, x1 = int(np.floor(d[0].min())), int(np.ceil(d[0].max()))
    y0, y1 = int(np.floor(d[1].min())), int(np.ceil(d[1].max()))
    x0, y0 = max(x0, 0), max(y0, 0)
    x1, y1 = min(x1, canvas.shape[1] - 1), min(y1, canvas.shape[0] - 1)
    if x1 <= x0 or y1 <= y0:
        return
    Hi = np.linalg.inv(H_src_to_dst)
    xs, ys = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1))
    p = Hi @ np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5, np.ones(xs.size)])
    u, v = p[0] / p[2] - 0.5, p[1] / p[2] - 0.5
    ok = (u >= 0) & (u <= sw - 1) & (v >= 0) & (v <= sh - 1)
    u, v = np.clip(u, 0, sw - 1.001), np.clip(v, 0, sh - 1.001)
    iu, iv = u.astype(int), v.astype(int)
    fu, fv = u - iu, v - iv
    s = src.astype(np.float64)
    val = (s[iv, iu] * (1 - fu) * (1 - fv) + s[iv, iu + 1] * fu * (1 - fv)
           + s[iv + 1, iu] * (1 - fu) * fv + s[iv + 1, iu + 1] * fu * fv)
    sub = canvas[y0:y1 + 1, x0:x1 + 1].ravel()
    sub[ok] = val[ok]
    canvas[y0:y1 + 1, x0:x1 + 1] = sub.reshape(y1 - y0 + 1, x1 - x0 + 1)


def _tilted_quad(cx, cy, side, rng, max_tilt_deg=35.0):
    """Image corners (TL, TR, BR, BL) of a square of `side` px seen under a random tilt."""
    tilt = np.deg2rad(rng.uniform(0, max_tilt_deg))
    axis = rng.uniform(0, 2 * np.pi)
    roll = rng.uniform(0, 2 * np.pi)
    f = 4.0 * side
    k = np.array([np.cos(axis), np.sin(axis), 0.0])
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(tilt) * K + (1 - np.cos(tilt)) * (K @ K)
    c, s = np.cos(roll), np.sin(roll)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    pts = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float64) * side / 2
    P = (R @ Rz @ pts.T).T + np.array([0, 0, f])
    return [(cx + f * p[0] / p[2], cy + f * p[1] / p[2]) for p in P]


def scene(h, w, seed, dictionary="ARUCO", n_markers=4, side_range=(40, 120), noise_sigma=2.0):
    """One frame.  Returns (uint8 image h x w, list of (id, 4x2 corner array TL,TR,BR,BL of the black border))."""
    rng = np.random.default_rng(seed)
    img = background(h, w, rng)
    nbits, codes = dictionary_codes(dictionary)
    n = int(round(nbits ** 0.5))
    ids = rng.choice(len(codes), size=n_markers, replace=False)
    truth = []
    placed = []
    for mid in ids:
        for _ in range(50):
            side = float(rng.uniform(*side_range))
            half = side * 0.95
            cx = float(rng.uniform(half + 0.03 * w, w - half - 0.03 * w))
            cy = float(rng.uniform(half + 0.03 * h, h - half - 0.03 * h))
            if all((cx - px) ** 2 + (cy - py) ** 2 > (half + ph) ** 2 for px, py, ph in placed):
                break
        else:
            continue
        placed.append((cx, cy, half))
        bit = 8
        m = render_marker(dictionary, int(mid), bit, quiet=1)
        S = m.shape[0]  # (n + 4) * bit
        q = bit  # quiet zone in px
        # quad of the black border; the quiet zone extends it by one bit
        quad = _tilted_quad(cx, cy, side, rng)
        src_border = [(q, q), (S - q, q), (S - q, S - q), (q, S - q)]
        H = _homography(src_border, quad)
        _warp_into(img, m, H)
        truth.append((int(mid), np.array(quad, np.float64)))
    if noise_sigma > 0:
        img = img + rng.normal(0.0, noise_sigma, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), truth


def paste_markers(image_u8, seed, dictionary="ARUCO", n_markers=3, side_range=(40, 110)):
    """Markers of `dictionary` warped into a given grey image (a photograph: tests/natural_cases.py) the way scene() places them on
    its synthetic background.  Returns (uint8 image, list of (id, 4x2 corner array))."""
    rng = np.random.default_rng(seed)
    img = np.asarray(image_u8, np.float64).copy()
    h, w = img.shape
    nbits, codes = dictionary_codes(dictionary)
    ids = rng.choice(len(codes), size=n_markers, replace=False)
    truth, placed = [], []
    for mid in ids:
        for _ in range(50):
            side = float(rng.uniform(*side_range))
            half = side * 0.95
            cx = float(rng.uniform(half + 0.03 * w, w - half - 0.03 * w))
            cy = float(rng.uniform(half + 0.03 * h, h - half - 0.03 * h))
            if all((cx - px) ** 2 + (cy - py) ** 2 > (half + ph) ** 2 for px, py, ph in placed):
                break
        else:
            continue
        placed.append((cx, cy, half))
        m = render_marker(dictionary, int(mid), 8, quiet=1)
        S = m.shape[0]
        quad = _tilted_quad(cx, cy, side, rng)
        _warp_into(img, m, _homography([(8, 8), (S - 8, 8), (S - 8, S - 8), (8, S - 8)], quad))
        truth.append((int(mid), np.array(quad, np.float64)))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), truth


def _stream_chunk(args):
    """Worker of stream(workers > 1): frames [i0, i1) of the stream, rendered in a process of their own."""
    h, w, n_frames, seed_base, dictionary, n_markers, noise_sigma, i0, i1 = args
    return i0, stream(h, w, n_frames, seed_base, dictionary, n_markers, noise_sigma, only=range(i0, i1))[i0:i1].copy()


def stream(h, w, n_frames, seed_base, dictionary="ARUCO", n_markers=4, noise_sigma=2.0, workers=1, only=None):
    """A video-like stream: one larger scene viewed under a smoothly drifting homography.

    Frame i adds noise seeded with seed_base + i.  Returns uint8 array (n_frames, h, w).
    workers > 1: the frames are rendered by that many processes (forkserver: safe next to an initialised HIP runtime; only for
    callers whose main module is import-safe, e.g. bench.py) -- every frame is a function of (seed_base, i, n_frames) alone, so the
    result is the same bytes.  only = the frame indices to render (the rest of the array is left unset).
    """
    if workers > 1 and n_frames >= 2 * workers:
        import multiprocessing as mp
        chunks = [(h, w, n_frames, seed_base, dictionary, n_markers, noise_sigma, n_frames * k // workers, n_frames * (k + 1) // workers)
                  for k in range(workers)]
        out = np.empty((n_frames, h, w), np.uint8)
        with mp.get_context("forkserver").Pool(workers) as pool:
            for i0, part in pool.imap_unordered(_stream_chunk, chunks):
                out[i0:i0 + len(part)] = part
        return out
    margin = 0.25
    H0, W0 = int(h * (1 + 2 * margin)), int(w * (1 + 2 * margin))
    base, _ = scene(H0, W0, seed_base, dictionary, n_markers=n_markers * 2,
                    side_range=(0.085 * w, 0.22 * w), noise_sigma=0.0)
    base = base.astype(np.float64)
    out = np.empty((n_frames, h, w), np.uint8)
    ys, xs = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")

    def frame(i):   # a function of (seed_base, i, n_frames) alone
        t = i / max(1, n_frames - 1)
        ang = np.deg2rad(6.0 * np.sin(2 * np.pi * t))
        sc = 1.0 + 0.06 * np.sin(2 * np.pi * t * 0.7)
        tx = margin * w * (1 + 0.8 * np.sin(2 * np.pi * t * 0.9))
        ty = margin * h * (1 + 0.8 * np.cos(2 * np.pi * t * 1.1))
        px, py = 1.2e-4 * np.sin(2 * np.pi * t), 0.8e-4 * np.cos(2 * np.pi * t * 1.3)
        c, s = np.cos(ang) * sc, np.sin(ang) * sc
        cx, cy = w / 2, h / 2
        u0, v0 = xs - cx, ys - cy
        den = 1.0 + px * u0 + py * v0
        u = (c * u0 - s * v0) / den + cx + tx
        v = (s * u0 + c * v0) / den + cy + ty
        u = np.clip(u - 0.5, 0, W0 - 1.001)
        v = np.clip(v - 0.5, 0, H0 - 1.001)
        iu, iv = u.astype(int), v.astype(int)
        fu, fv = u - iu, v - iv
        val = (base[iv, iu] * (1 - fu) * (1 - fv) + base[iv, iu + 1] * fu * (1 - fv)
               + base[iv + 1, iu] * (1 - fu) * fv + base[iv + 1, iu + 1] * fu * fv)
        rng = np.random.default_rng(seed_base + 1000 + i)
        if noise_sigma > 0:
            val = val + rng.normal(0.0, noise_sigma, val.shape)
        out[i] = np.clip(np.rint(val), 0, 255).astype(np.uint8)

    for i in (range(n_frames) if only is None else only):
        frame(i)
    return out


def random_descriptors(n, seed):
    """n x 32 uint8, i.i.d. uniform bits (config C5)."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
