"""Deterministic synthetic clouds shared by the tests, the oracle legs and bench.py.

The reference ships no sample data, so every workload is synthetic:
counter-based splitmix64 -> uniform (z >> 11) * 2**-53, Gaussians by Box-Muller computed
once on the host (the same float64 arrays are fed to the CPU oracle and to the GPU, so libm
differences cannot matter).  Coordinates are snapped to multiples of 2**-10 so that
differences of coordinates are exact in binary64 (SURVEY.md section 7 "exact thresholding").

Per point the generator yields the two coordinate sets the reference's Point3D carries
(BaseClass/DataModel.cs:121-126): motor = (motor_x, motor_y), the 2-D scan coordinates the
live DBSCAN clusters on (BaseClass/DBImproved.cs:16-21), and xyz = (X, Y, Z).  Both share
the same blob membership.  The motor cloud is laid out so that the uniform background has
10 points per unit area (noise at eps 0.1 / minPts 10) and blobs have sigma 2.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
QUANT = 1024.0


def splitmix64(seed, start, count):
    """Values start..start+count-1 of the splitmix64 stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, start, count):
    return (splitmix64(seed, start, count) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def normal(seed, start, count):
    """Box-Muller on two independent streams (seed, seed ^ golden)."""
    u1 = uniform01(seed, start, count)
    u2 = uniform01(seed ^ 0x5851F42D4C957F2D, start, count)
    u1 = np.maximum(u1, 2.0 ** -53)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def snap(a):
    return np.round(a * QUANT) / QUANT


def permutation(seed, n):
    return np.argsort(splitmix64(seed ^ 0xA5A5A5A5, 0, n), kind="stable")


def make_cloud(n, seed, n_blobs, blob_pts, xyz_extent, xyz_sigma, motor_sigma=2.0,
               motor_bg_density=10.0, centers=None, shuffle=True, xyz_lo=0.0, motor_centers=None,
               motor_lo=0.0):
    """n points = n_blobs*blob_pts blob points + the rest uniform background.

    Returns dict(xyz [n,3], motor [n,2], blob [n] int32 (-1 = background), motor_extent).
    """
    n_blob_total = n_blobs * blob_pts
    n_bg = n - n_blob_total
    assert n_bg >= 0
    motor_extent = float(np.sqrt(max(n_bg, 1) / motor_bg_density))
    xyz = np.empty((n, 3))
    motor = np.empty((n, 2))
    blob = np.full(n, -1, np.int32)
    # background
    for a in range(3):
        xyz[:n_bg, a] = xyz_lo + xyz_extent * uniform01(seed + 11 + a, 0, n_bg)
    for a in range(2):
        motor[:n_bg, a] = motor_lo + motor_extent * uniform01(seed + 21 + a, 0, n_bg)
    # blob centres
    if centers is None:
        c3 = xyz_lo + xyz_extent * (0.1 + 0.8 * uniform01(seed + 31, 0, 3 * max(n_blobs, 1)).reshape(-1, 3))
    else:
        c3 = np.asarray(centers, np.float64)
    if motor_centers is None:
        c2 = motor_lo + motor_extent * (0.1 + 0.8 * uniform01(seed + 32, 0, 2 * max(n_blobs, 1)).reshape(-1, 2))
    else:
        c2 = np.asarray(motor_centers, np.float64)
    for b in range(n_blobs):
        s = n_bg + b * blob_pts
        for a in range(3):
            xyz[s:s + blob_pts, a] = c3[b, a] + xyz_sigma * normal(seed + 41 + a, s, blob_pts)
        for a in range(2):
            motor[s:s + blob_pts, a] = c2[b, a] + motor_sigma * normal(seed + 51 + a, s, blob_pts)
        blob[s:s + blob_pts] = b
    xyz = snap(xyz)
    motor = snap(motor)
    if shuffle:
        p = permutation(seed, n)
        xyz, motor, blob = xyz[p], motor[p], blob[p]
    return dict(xyz=np.ascontiguousarray(xyz), motor=np.ascontiguousarray(motor), blob=blob,
                motor_extent=motor_extent)


# ---- the BASELINE.json configs ----------------------------------------------------------
def config_c1():
    """C1: 10k pts = 3 blobs x 3000 + 1000 uniform in [-4,12]^3, eps 0.5, minPts 10."""
    d = make_cloud(10_000, 1, 3, 3000, 16.0, 0.6, motor_sigma=0.6, motor_bg_density=1000.0 / 256.0,
                   centers=[(0, 0, 0), (8, 0, 0), (0, 8, 0)], xyz_lo=-4.0,
                   motor_centers=[(0, 0), (8, 0), (0, 8)], motor_lo=-4.0)
    d.update(eps_l1=0.5, min_pts=10, eps_l2=0.5)
    return d


def config_cloud(n, seed=None):
    """C2 (1M, seed 2) / C4 (10M, seed 4) family: half uniform background, 25k-pt blobs, constant density."""
    n_blobs = max(1, n // 50_000)
    blob_pts = (n // 2) // n_blobs
    if seed is None:
        seed = {1_000_000: 2, 10_000_000: 4}.get(n, 7)
    ext = 100.0 * (n / 1_000_000.0) ** (1.0 / 3.0)
    d = make_cloud(n, seed, n_blobs, blob_pts, ext, 2.0)
    d.update(eps_l1=0.1, min_pts=10, eps_l2=0.5)
    return d


def config_c5(n=50_000_000, seed=5):
    """C5: scan-like, 70 % ground band z in [0,0.3], 30 % in object blobs sigma 0.5."""
    n_obj = int(n * 0.3)
    n_blobs = max(1, n_obj // 7500)
    blob_pts = n_obj // n_blobs
    ext = 400.0 * (n / 50_000_000.0) ** 0.5
    d = make_cloud(n, seed, n_blobs, blob_pts, ext, 0.5, motor_sigma=0.5)
    bg = d["blob"] < 0
    d["xyz"][bg, 2] = snap(0.3 * (d["xyz"][bg, 2] / ext))
    d.update(eps_l1=0.05, min_pts=10, eps_l2=0.2)
    return d


def rotation_about(axis, deg):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    t = np.deg2rad(deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * (K @ K)


def config_icp(nd=1_000_000, nm=100, seed=3, jitter=0.05, deg=5.0, shift=(0.3, -0.2, 0.1)):
    """C3: model = nm pts uniform in [0,50]^3 with pairwise separation >= 2; data = model points
    (cyclic) + N(0, jitter^2), then rotated `deg` about (1,1,1)/sqrt3 and shifted."""
    pts = []
    k = 0
    while len(pts) < nm:
        cand = 50.0 * uniform01(seed + 61, 3 * k, 3)
        k += 1
        if all(np.linalg.norm(cand - q) >= 2.0 for q in pts):
            pts.append(cand)
    model = snap(np.array(pts))
    base = model[np.arange(nd) % nm]
    if jitter > 0:
        noise = np.stack([normal(seed + 71 + a, 0, nd) for a in range(3)], 1) * jitter
        base = base + noise
    Rt = rotation_about((1, 1, 1), deg)
    data = base @ Rt.T + np.asarray(shift)
    # data = Rt * base + shift  =>  the transform that maps data onto the model is
    # R = Rt^T, T = -Rt^T shift
    R_true = Rt.T
    T_true = -Rt.T @ np.asarray(shift)
    return dict(model=np.ascontiguousarray(model), data=np.ascontiguousarray(data), R_true=R_true,
                T_true=T_true)
