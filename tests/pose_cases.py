"""Seeded marker-pose cases shared by the CPU (oracle) and GPU (parity) tests: square markers of side `size` placed at
random poses in front of a TUM1-like camera (the reference's Examples/Monocular/TUM1.yaml), corners projected through the
forward Brown model in float64 and rounded to float like detected corners."""
import numpy as np

K4 = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


def rodrigues(r):
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def object_points(size):
    h = np.float32(size) / np.float32(2)
    return np.array([[-h, h, 0], [h, h, 0], [h, -h, 0], [-h, -h, 0]], np.float64)


def project(P, R, t, K=K4, d=DIST):
    d = np.concatenate([np.asarray(d, np.float64), np.zeros(12 - len(d))])
    X = (R @ P.T).T + t
    x = X[:, 0] / X[:, 2]; y = X[:, 1] / X[:, 2]
    r2 = x * x + y * y
    cd = (1 + d[0] * r2 + d[1] * r2 ** 2 + d[4] * r2 ** 3) / (1 + d[5] * r2 + d[6] * r2 ** 2 + d[7] * r2 ** 3)
    xd = x * cd + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
    yd = y * cd + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
    return np.stack([xd * float(K[0]) + float(K[2]), yd * float(K[1]) + float(K[3])], 1)


def random_cases(n, seed, size=0.187, noise=0.0):
    """-> list of (R, t, corners float32 (4, 2)); markers face the camera within ~60 degrees."""
    rng = np.random.default_rng(seed)
    P = object_points(size)
    out = []
    while len(out) < n:
        tilt = rng.uniform(0.05, 1.05)
        axis = rng.normal(size=3); axis[2] *= 0.3; axis /= np.linalg.norm(axis)
        R = rodrigues(axis * tilt) @ rodrigues([np.pi, 0, 0]) @ rodrigues([0, 0, rng.uniform(-np.pi, np.pi)])
        t = np.array([rng.uniform(-0.35, 0.35), rng.uniform(-0.25, 0.25), rng.uniform(0.5, 2.5)])
        c = project(P, R, t)
        if c.min() < 5 or c[:, 0].max() > 635 or c[:, 1].max() > 475:
            continue
        c = c + rng.normal(0, noise, c.shape) if noise else c
        out.append((R, t, c.astype(np.float32)))
    return out
