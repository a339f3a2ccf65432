"""Seeded synthetic scenes / maps / sweeps for the parity tests and bench.py.

Piecewise-planar world: ground plane z = -1.7 m plus 4 m x 4 m x 6 m boxes on a 10 m lattice (vertical
faces x = const / y = const, so all six pose directions are observable).  Surface samples on a jittered 0.16 m grid with sigma = 0.02 m noise along the normal.
The MAP is not produced here: callers push `map_candidates()` through addPointsToMap (the oracle's or
the product's) so that it is a legal reference map (voxel 1.0 m, cap 20, min-distance 0.15 m, keys
by truncation of the FP32 position).  No arithmetic of the hot path lives in this file.
"""
import math

import numpy as np

GROUND_Z = -1.7
WALL_PITCH = 10.0
WALL_HEIGHT = 6.0
GRID = 0.16
SIGMA = 0.02


BOX_HALF = 2.0          # 4 m x 4 m x 6 m boxes centred on a 10 m lattice
BOX_OFF = 5.37          # lattice offset: faces never sit on integer voxel faces


def _box_centres(L):
    ks = np.arange(math.floor((-L - BOX_OFF) / WALL_PITCH), math.ceil((L - BOX_OFF) / WALL_PITCH) + 1)
    c = ks * WALL_PITCH + BOX_OFF
    c = c[(c - BOX_HALF > -L) & (c + BOX_HALF < L)]
    return c


def _faces(L):
    """(axis, plane coordinate, lo, hi) of every vertical box face inside [-L, L]^2."""
    out = []
    cs = _box_centres(L)
    for cx in cs:
        for cy in cs:
            out.append((0, cx - BOX_HALF, cy - BOX_HALF, cy + BOX_HALF))
            out.append((0, cx + BOX_HALF, cy - BOX_HALF, cy + BOX_HALF))
            out.append((1, cy - BOX_HALF, cx - BOX_HALF, cx + BOX_HALF))
            out.append((1, cy + BOX_HALF, cx - BOX_HALF, cx + BOX_HALF))
    return out


def _surface_samples(rng, half_extent):
    """Jittered-grid samples of ground + box faces inside [-L, L]^2, with normal noise.  Returns (M,3)."""
    L = float(half_extent)
    n1 = int(2 * L / GRID)
    gx, gy = np.meshgrid(np.arange(n1), np.arange(n1), indexing="ij")
    g = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float64) * GRID - L
    g += rng.uniform(0.0, GRID, g.shape)
    ground = np.column_stack([g, GROUND_Z + rng.normal(0.0, SIGMA, len(g))])
    parts = [ground]
    nu = int(2 * BOX_HALF / GRID)
    nz = int(WALL_HEIGHT / GRID)
    su, sz = np.meshgrid(np.arange(nu), np.arange(nz), indexing="ij")
    base = np.stack([su.ravel(), sz.ravel()], 1).astype(np.float64) * GRID
    for axis, c, lo, _hi in _faces(L):
        uz = base + rng.uniform(0.0, GRID, base.shape)
        u = lo + uz[:, 0]
        z = GROUND_Z + uz[:, 1]
        off = c + rng.normal(0.0, SIGMA, len(u))
        parts.append(np.column_stack([off, u, z]) if axis == 0 else np.column_stack([u, off, z]))
    return np.concatenate(parts, 0)


def map_candidates(seed, target_points):
    """Shuffled candidate points for a map of about `target_points` points.  Insert ALL of them through
    addPointsToMap: the extent is chosen so that the saturated map (cap 20 per 1 m voxel, as in a
    long-running session) lands near the target."""
    rng = np.random.default_rng(seed)
    # saturated: ~18.4 kept points per voxel, ~0.9 voxel per m^2 of surface, ~1.96 m^2 surface per m^2 footprint
    area = target_points / 20.0 / 1.65     # calibrated: saturated maps land within ~3 % of the target
    L = max(16.0, 0.5 * math.sqrt(area))
    pts = _surface_samples(rng, L)
    rng.shuffle(pts, axis=0)
    return pts, L


def _raycast(o, d, L, max_range):
    """Nearest hit of rays o + s d with the ground plane and the box faces.  Returns s (inf = miss)."""
    n = len(d)
    best = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = (GROUND_Z - o[2]) / d[:, 2]
    ok = (d[:, 2] < 0) & (s > 0.5) & (np.abs(o[0] + s * d[:, 0]) < L) & (np.abs(o[1] + s * d[:, 1]) < L)
    best = np.where(ok & (s < best), s, best)
    z_lo, z_hi = GROUND_Z, GROUND_Z + WALL_HEIGHT
    for axis, c, lo, hi in _faces(L):
        if abs(c - o[axis]) > max_range:
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (c - o[axis]) / d[:, axis]
        z = o[2] + s * d[:, 2]
        u = o[1 - axis] + s * d[:, 1 - axis]
        ok = np.isfinite(s) & (s > 0.5) & (z > z_lo) & (z < z_hi) & (u > lo) & (u < hi)
        best = np.where(ok & (s < best), s, best)
    best[best > max_range] = np.inf
    return best


def quat_from_rotvec(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.array([1.0, 0.0, 0.0, 0.0])
    a = w / th
    return np.concatenate([[math.cos(th / 2)], a * math.sin(th / 2)])


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def make_sweep(seed, n, L, pattern="livox", max_range=50.0):
    """N lidar-frame points (extrinsic identity) seen from a ground-truth pose, plus a perturbed
    predicted pose.  pattern: 'livox' (70 deg forward cone, random) or 'ouster16' (16 rings x 360, ring-major).
    Returns dict(raw (N,3), q_gt, t_gt, q_pred, t_pred, t_last)."""
    rng = np.random.default_rng(seed)
    rng_lim = min(max_range, 0.85 * L)
    yaw = rng.uniform(-math.pi, math.pi)
    q_gt = quat_mul(quat_from_rotvec([0, 0, yaw]), quat_from_rotvec(rng.normal(0, 0.02, 3)))
    while True:
        t_gt = np.array([rng.uniform(-3, 3) + 2.3, rng.uniform(-3, 3) + 1.7, rng.uniform(-0.1, 0.1)])
        # keep the sensor out of (and 0.8 m away from) the boxes; re-draw otherwise (seeds that were fine are unchanged)
        cs = _box_centres(L)
        dx = np.maximum(np.abs(t_gt[0] - cs) - BOX_HALF, 0.0)
        dy = np.maximum(np.abs(t_gt[1] - cs) - BOX_HALF, 0.0)
        if np.min(np.hypot(dx[:, None], dy[None, :])) > 0.8:
            break
    R = quat_to_rot(q_gt)
    pts = np.zeros((0, 3))
    while len(pts) < n:
        m = int((n - len(pts)) * 1.6) + 1024
        if pattern == "livox":
            half = math.radians(35.0)
            cosmin = math.cos(half)
            cz = rng.uniform(cosmin, 1.0, m)
            ph = rng.uniform(0, 2 * math.pi, m)
            sz = np.sqrt(1 - cz * cz)
            dl = np.column_stack([cz, sz * np.cos(ph), sz * np.sin(ph)])    # forward = +x of the lidar
        else:
            rings = np.radians(np.linspace(-15.0, 15.0, 16))
            per = m // 16 + 1
            az = rng.uniform(0, 2 * math.pi, (16, per))
            el = np.repeat(rings[:, None], per, 1)
            dl = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1).reshape(-1, 3)
        dw = dl @ R.T
        s = _raycast(t_gt, dw, L, rng_lim)
        ok = np.isfinite(s)
        p_l = dl[ok] * s[ok, None]
        # noise along the ray (approximately along the normal for frontal hits), sigma 0.02
        p_l = p_l + dl[ok] * rng.normal(0.0, SIGMA, (ok.sum(), 1))
        pts = np.concatenate([pts, p_l], 0)
    raw = pts[:n].copy()
    if pattern != "livox":
        # ring-major order: sort by elevation ring then azimuth
        el = np.degrees(np.arcsin(raw[:, 2] / np.linalg.norm(raw, axis=1)))
        ring = np.clip(np.round((el + 15.0) / 2.0), 0, 15).astype(int)
        az = np.arctan2(raw[:, 1], raw[:, 0])
        raw = raw[np.lexsort((az, ring))]
    # predicted pose = gt (+) (0.05 m random direction, 0.5 deg random axis)
    dv = rng.normal(size=3); dv *= 0.05 / np.linalg.norm(dv)
    ax = rng.normal(size=3); ax *= math.radians(0.5) / np.linalg.norm(ax)
    q_pred = quat_mul(q_gt, quat_from_rotvec(ax))
    t_pred = t_gt + dv
    vel = rng.normal(0, 0.5, 3)
    t_last = t_pred - 0.1 * vel
    return dict(raw=raw, q_gt=q_gt, t_gt=t_gt, q_pred=q_pred, t_pred=t_pred, t_last=t_last, vel=vel)


def eskf_prior(eskf_like, q_pred, t_pred, vel):
    """ESKF prior of SURVEY 8(d): ctor state (eskfEstimator.cpp:3-21), tryInit covariance scaling
    (:74-76), noise of config/r3live.yaml:20-23, 10 stationary predict steps (dt 0.01), then the
    predicted pose written into p, q, v.  `eskf_like` exposes set_noise / init_imu / scale_init_cov /
    predict / get_state / set_state (oracle or product handle)."""
    eskf_like.set_noise(0.1, 0.1, 0.0001, 0.0001)
    eskf_like.scale_init_cov()
    acc = np.array([0.0, 0.0, 9.81]); gyr = np.zeros(3)
    eskf_like.init_imu(acc, gyr)
    for _ in range(10):
        eskf_like.predict(0.01, acc, gyr)
    s = eskf_like.get_state()
    s[0:3] = t_pred
    s[3:7] = q_pred
    s[7:10] = vel
    eskf_like.set_state(s)
    return s


def lattice_scene(seed, n_keypoints, K_hint=20):
    """A scene built to TIE: every map point sits on a 0.25 m lattice (exact in FP32 and FP64) and most keypoints on
    lattice-symmetric positions (cell centres, edge midpoints, lattice points), so that many candidate distances are
    exactly equal -- inside the K nearest and across the K-th / (K+1)-th cut.  Which of the tied points the reference
    keeps, and in which order, is decided by libstdc++'s heap (optimize.cpp:394-404); this scene is what the
    tie-faithful selection is tested on.  The pose is the identity rotation with a lattice translation, so the world
    points are exact too.  Returns (map_points in insertion order, sweep dict like make_sweep's)."""
    rng = np.random.default_rng(seed)
    h = 0.25
    ax = np.arange(-32, 33) * h                                    # -8 .. 8
    gx, gy = np.meshgrid(ax, ax, indexing="ij")
    ground = np.column_stack([gx.ravel(), gy.ravel(), np.full(gx.size, -1.0)])
    sel = gx.ravel() >= 0.0
    upper = np.column_stack([gx.ravel()[sel], gy.ravel()[sel], np.full(sel.sum(), -0.75)])
    wz = np.arange(-3, 5) * h                                      # -0.75 .. 1.0
    wy, wzz = np.meshgrid(ax, wz, indexing="ij")
    wall = np.column_stack([np.full(wy.size, 3.0), wy.ravel(), wzz.ravel()])
    wall2 = np.column_stack([wy.ravel(), np.full(wy.size, -2.5), wzz.ravel()])
    pts = np.concatenate([ground, upper, wall, wall2], 0)
    pts = np.unique(pts, axis=0)
    rng.shuffle(pts, axis=0)                                       # irregular per-voxel insertion order

    t = np.array([0.5, 0.25, 0.125])
    q = np.array([1.0, 0.0, 0.0, 0.0])
    n = int(n_keypoints)
    base = np.column_stack([rng.integers(-26, 27, n), rng.integers(-26, 27, n)]) * h
    kind = rng.integers(0, 6, n)
    off = np.zeros((n, 3))
    off[kind == 0] = [0.125, 0.125, 0.0]                           # cell centre above / on a layer
    off[kind == 1] = [0.125, 0.0, 0.0]                             # edge midpoint
    off[kind == 2] = [0.0, 0.0, 0.0]                               # on a lattice point
    off[kind == 3] = [0.125, 0.125, 0.125]                         # cell centre between the two layers
    off[kind == 4] = [0.0, 0.125, 0.0625]
    jitter = kind == 5                                             # generic positions: tie-free keypoints in between
    z0 = np.where(rng.random(n) < 0.5, -1.0, -0.875)
    pw = np.column_stack([base, z0]) + off
    pw[jitter] += rng.uniform(-0.11, 0.11, (int(jitter.sum()), 3))
    near_wall = rng.random(n) < 0.2                                # some keypoints next to the x = 3 wall, at wall heights
    pw[near_wall, 0] = 3.0 - 0.125 * rng.integers(0, 3, int(near_wall.sum()))
    pw[near_wall, 2] = -0.75 + 0.125 * rng.integers(0, 9, int(near_wall.sum()))
    raw = pw - t                                                   # identity extrinsics and rotation: p_w = raw + t exactly
    assert np.array_equal((raw + t)[~jitter], pw[~jitter])
    vel = np.zeros(3)
    return pts, dict(raw=raw, q_gt=q, t_gt=t, q_pred=q, t_pred=t, t_last=t + np.array([0.0, 0.0, 0.5]), vel=vel)


def cable_scene(seed, n_keypoints=6000, pole_share=0.2):
    """A scene with ILL-POSED neighbourhoods: a noisy ground plane and two walls (well-posed planes) plus thin straight cables high above
    the ground, along the space diagonal, with a point every 0.16 m (the map keeps points at least min_distance_points = 0.15 m apart and
    20 per voxel: a 3 x 3 x 3-voxel search box holds ~30 points of a diagonal line and nothing else up there).  A keypoint next to a cable
    finds its 20 neighbours ON the cable: collinear, the two small eigenvalues of the neighbourhood (almost) equal, the plane normal
    undetermined (computeNeighborhoodDistribution, optimize.cpp:316-353: a2D ~ 0)."""
    rng = np.random.default_rng(seed)
    g = np.column_stack([rng.uniform(-12, 12, 60_000), rng.uniform(-12, 12, 60_000), rng.normal(0.0, 0.01, 60_000)])
    w1 = np.column_stack([np.full(20_000, 11.0) + rng.normal(0, 0.01, 20_000), rng.uniform(-12, 12, 20_000), rng.uniform(0, 3, 20_000)])
    w2 = np.column_stack([rng.uniform(-12, 12, 20_000), np.full(20_000, -11.0) + rng.normal(0, 0.01, 20_000), rng.uniform(0, 3, 20_000)])
    d = np.ones(3) / np.sqrt(3.0)
    gx, gy = np.meshgrid([-9.0, -3.0, 3.0], [-9.0, -3.0, 3.0], indexing="ij")          # nine cables, 4.9 m apart: no neighbourhood sees two of them
    starts = np.column_stack([gx.ravel(), gy.ravel(), np.full(9, 6.0) + rng.uniform(0, 0.5, 9)])
    s_along = np.arange(0.0, 9.0, 0.16)
    cables = [st + s_along[:, None] * d + rng.normal(0, 2e-4, (len(s_along), 3)) for st in starts]
    pts = np.vstack([g, w1, w2] + cables)
    rng.shuffle(pts, axis=0)
    # keypoints: on the planes (well posed) and next to the cables (ill posed)
    n_pole = int(n_keypoints * pole_share)
    kg = np.column_stack([rng.uniform(-10, 10, n_keypoints - n_pole), rng.uniform(-10, 10, n_keypoints - n_pole), rng.normal(0, 0.01, n_keypoints - n_pole)])
    third = len(kg) // 3
    kg[:third] = np.column_stack([np.full(third, 11.0) + rng.normal(0, 0.01, third), rng.uniform(-10, 10, third), rng.uniform(0.3, 2.7, third)])
    which = rng.integers(0, len(starts), n_pole)
    side = np.cross(d, [0.0, 0.0, 1.0]); side /= np.linalg.norm(side)
    kp = starts[which] + rng.uniform(2.5, 6.5, n_pole)[:, None] * d + 0.03 * side + rng.normal(0, 0.003, (n_pole, 3))
    world = np.vstack([kg, kp])
    world = world[rng.permutation(len(world))]
    q_gt = quat_from_rotvec([0.01, -0.02, 0.3]); t_gt = np.array([0.4, -0.3, 1.2])
    R = quat_to_rot(q_gt)
    raw = (world - t_gt) @ R                                       # world = R raw + t (identity extrinsics)
    q_pred = quat_mul(quat_from_rotvec([0.003, -0.002, 0.004]), q_gt); t_pred = t_gt + np.array([0.03, -0.02, 0.02])
    return pts, dict(raw=raw, q_gt=q_gt, t_gt=t_gt, q_pred=q_pred, t_pred=t_pred, t_last=t_gt - np.array([0.1, 0.0, 0.0]), vel=np.zeros(3))


CONFIGS = {
    # name: (n_keypoints, map_points, pattern, seed)   -- SURVEY.md 8(d)
    "C1": (4096, 100_000, "livox", 20250304 + 1),
    "C2": (24576, 1_000_000, "livox", 20250304 + 2),
    "C3": (16384, 2_000_000, "ouster16", 20250304 + 3),
    "C4": (262144, 10_000_000, "livox", 20250304 + 4),
    "HEADLINE": (65536, 1_000_000, "livox", 20250304 + 5),
    # not a BASELINE configuration: the off-cache aux workload (make_spread_sweep over C4's map)
    "SPREAD": (65536, 10_000_000, "livox", 20250304 + 4),
}


def sweep_from_pose(rng, n, L, q, t, max_range=50.0):
    """n lidar-frame points (livox cone) ray-cast from the scene pose (q, t); same noise model as make_sweep."""
    rng_lim = min(max_range, 0.85 * L)
    R = quat_to_rot(q)
    pts = np.zeros((0, 3))
    while len(pts) < n:
        m = int((n - len(pts)) * 1.6) + 1024
        cosmin = math.cos(math.radians(35.0))
        cz = rng.uniform(cosmin, 1.0, m)
        ph = rng.uniform(0, 2 * math.pi, m)
        sz = np.sqrt(1 - cz * cz)
        dl = np.column_stack([cz, sz * np.cos(ph), sz * np.sin(ph)])
        s = _raycast(np.asarray(t, float), dl @ R.T, L, rng_lim)
        ok = np.isfinite(s)
        pts = np.concatenate([pts, dl[ok] * s[ok, None] + dl[ok] * rng.normal(0.0, SIGMA, (ok.sum(), 1))], 0)
    return pts[:n].copy()


def make_sequence(seed, n_moving, n_pts, L, imu_rate=200.0, sweep_dt=0.1, rest_time=3.3, acc_x=0.6, yaw_rate=0.15):
    """A ROS-free `Measurements` sequence for the replay driver: the sensor rests for rest_time seconds (IMU
    initialisation), then accelerates along its x axis while yawing.  Returns a list of dicts
    (time_frame, imu_t, imu_acc, imu_gyr, pts_raw, pts_timestamp, time_sweep_begin, time_sweep_offset) and the
    ground-truth odometry poses (relative to the first sensor pose) at every sweep end."""
    rng = np.random.default_rng(seed)
    t0 = 1000.0
    q0 = quat_from_rotvec([0.0, 0.0, 0.4]); p0 = np.array([2.1, 1.3, 0.05])        # scene pose of the odometry origin
    R0 = quat_to_rot(q0)
    g = np.array([0.0, 0.0, 9.81])
    n_rest = int(round(rest_time / sweep_dt)) + 1
    per = int(round(imu_rate * sweep_dt))
    t_move0 = t0 + n_rest * sweep_dt

    def pose(tt):               # odometry-frame pose at absolute time tt
        s = max(tt - t_move0, 0.0)
        yaw = yaw_rate * s
        # body-x acceleration integrated numerically on a fine grid (smooth, deterministic)
        k = max(int(s / 1e-3), 0)
        ts = (np.arange(k + 1) + 0.5) * 1e-3 if k else np.zeros(0)
        vel = np.zeros(3); pos = np.zeros(3)
        if k:
            a = acc_x * np.column_stack([np.cos(yaw_rate * ts), np.sin(yaw_rate * ts), np.zeros_like(ts)])
            v = np.cumsum(a, 0) * 1e-3
            pos = np.sum(v, 0) * 1e-3
            vel = v[-1]
        return quat_from_rotvec([0.0, 0.0, yaw]), pos, vel

    meas, gt = [], []
    for f in range(n_rest + n_moving):
        tb = t0 + f * sweep_dt
        te = tb + sweep_dt
        it = tb + (np.arange(per) + 1) / imu_rate
        moving = it > t_move0
        yaw = yaw_rate * np.maximum(it - t_move0, 0.0)
        acc_w = np.where(moving[:, None], acc_x * np.column_stack([np.cos(yaw), np.sin(yaw), np.zeros_like(yaw)]), 0.0) + g
        acc_b = np.stack([quat_to_rot(quat_from_rotvec([0, 0, y])).T @ a for y, a in zip(yaw, acc_w)])
        gyr_b = np.where(moving[:, None], np.array([0.0, 0.0, yaw_rate]), 0.0)
        acc_b = acc_b + rng.normal(0, 0.02, acc_b.shape)
        gyr_b = gyr_b + rng.normal(0, 0.002, gyr_b.shape) + np.array([0.001, -0.0005, 0.0008])
        q, p, _ = pose(te)
        qs = quat_mul(q0, q); ps = p0 + R0 @ p
        raw = sweep_from_pose(rng, n_pts, L, qs, ps)
        ts = np.sort(rng.uniform(tb, te, n_pts)); ts[-1] = te
        meas.append(dict(time_frame=te, imu_t=it, imu_acc=acc_b, imu_gyr=gyr_b, pts_raw=raw, pts_timestamp=ts,
                         time_sweep_begin=tb, time_sweep_offset=sweep_dt))
        gt.append((q, p))
    return meas, gt, (q0, p0)


def make_spread_sweep(seed, n, cands, L, radius=None):
    """An OFF-CACHE workload (never a lidar pattern): n keypoints drawn area-uniformly from the scene's surfaces within `radius` of the
    sensor (default: the whole map) in RANDOM order -- about one keypoint per voxel, consecutive keypoints far apart, so that a pass probes
    tens of thousands of distinct voxels (tens of MB of slabs) instead of the ~2 000 a 70-degree cone touches.  `cands` = map_candidates()
    of the scene (area-uniform surface samples).  Same return as make_sweep."""
    rng = np.random.default_rng(seed)
    yaw = rng.uniform(-math.pi, math.pi)
    q_gt = quat_mul(quat_from_rotvec([0, 0, yaw]), quat_from_rotvec(rng.normal(0, 0.02, 3)))
    t_gt = np.array([rng.uniform(-3, 3) + 2.3, rng.uniform(-3, 3) + 1.7, rng.uniform(-0.1, 0.1)])
    pool = cands
    if radius is not None:
        pool = cands[np.hypot(cands[:, 0] - t_gt[0], cands[:, 1] - t_gt[1]) < radius]
    world = pool[rng.choice(len(pool), n, replace=len(pool) < n)] + rng.normal(0.0, SIGMA, (n, 3))
    raw = (world - t_gt) @ quat_to_rot(q_gt)                          # world = R raw + t (identity extrinsics)
    dv = rng.normal(size=3); dv *= 0.05 / np.linalg.norm(dv)
    ax = rng.normal(size=3); ax *= math.radians(0.5) / np.linalg.norm(ax)
    q_pred = quat_mul(q_gt, quat_from_rotvec(ax))
    t_pred = t_gt + dv
    vel = rng.normal(0, 0.5, 3)
    return dict(raw=raw, q_gt=q_gt, t_gt=t_gt, q_pred=q_pred, t_pred=t_pred, t_last=t_pred - 0.1 * vel, vel=vel)
