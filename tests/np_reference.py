"""Independent NumPy/SciPy re-implementation of the hot path, written from the reference's formulas
(src/optimize.cpp, src/eskfEstimator.cpp, include/utility.h) with library linear algebra
(numpy.linalg.eigh / inv, brute-force sorting) instead of the oracle's hand-written Jacobi / LU / heap.
Used only by tests/test_oracle.py to pin the C++ oracle.  Small inputs only (pure Python loops).
"""
import numpy as np

THETA_THRESHOLD = 1e-4


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def quat_inv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]]) / (q @ q)


def rot_to_quat(R):
    # any correct conversion; sign fixed to w >= 0 like Shepperd's first branch for small rotations
    from scipy.spatial.transform import Rotation
    x, y, z, w = Rotation.from_matrix(R).as_quat()
    q = np.array([w, x, y, z])
    return q if w >= 0 else -q


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def normalize_R(R):
    # numType::normalizeR: quaternion round trip = nearest rotation for an almost-orthonormal matrix
    U, _, Vt = np.linalg.svd(R)
    return U @ Vt


def rotation_to_so3(R_in):
    R = normalize_R(R_in)
    theta = np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0))
    a = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if theta < THETA_THRESHOLD:
        return a / 2.0
    return theta * a / (2.0 * np.sin(theta))


def so3_to_rotation(w):
    theta = np.linalg.norm(w)
    if theta < THETA_THRESHOLD:
        u = skew(w)
        return np.eye(3) + u + 0.5 * u @ u
    u = skew(w / theta)
    return np.eye(3) + np.sin(theta) * u + (1 - np.cos(theta)) * u @ u


def so3_to_quat(w):
    theta = np.linalg.norm(w)
    if theta < THETA_THRESHOLD:
        q = np.array([1.0, w[0] / 2, w[1] / 2, w[2] / 2])
    else:
        u = w / theta
        q = np.concatenate([[np.cos(theta / 2)], u * np.sin(theta / 2)])
    return q / np.linalg.norm(q)


def derivative_s2(g_in):
    g = g_in / np.linalg.norm(g_in)
    B = np.zeros((3, 2))
    B[0, 0] = 1.0 - g[0] * g[0] / (1.0 + g[2])
    B[0, 1] = -g[0] * g[1] / (1.0 + g[2])
    B[1, 0] = B[0, 1]
    B[1, 1] = 1.0 - g[1] * g[1] / (1.0 + g[2])
    B[2, 0] = -g[0]
    B[2, 1] = -g[1]
    return B


def angular_distance(w):
    R = so3_to_rotation(w)
    return np.degrees(np.arccos((np.trace(R) - 1) / 2))


def trunc_key(p, size):
    return tuple(int(np.trunc(c / size)) for c in p)   # static_cast<short>: toward zero


def map_dict(keys, counts, xyz):
    return {tuple(int(v) for v in keys[i]): (i, xyz[i, : counts[i]].astype(np.float64)) for i in range(len(counts))}


def search_neighbors(md, p, nb, size, K, thr, cap=20):
    k = trunc_key(p, size)
    cand = []
    for dx in range(-nb, nb + 1):
        for dy in range(-nb, nb + 1):
            for dz in range(-nb, nb + 1):
                v = md.get((k[0] + dx, k[1] + dy, k[2] + dz))
                if v is None or len(v[1]) < thr:
                    continue
                vi, pts = v
                d = np.sqrt(((pts - p) ** 2).sum(1))
                for s in range(len(pts)):
                    cand.append((d[s], vi * cap + s, pts[s]))
    cand.sort(key=lambda t: t[0])          # tie-free inputs: any order-preserving sort
    cand = cand[:K]
    return cand, None


def build_plane_residuals(keys, counts, xyz, raw, q, t, t_last, R_il, t_il, frame_id=100, max_num_residuals=600,
                          size=1.0, K=20, min_nb=20, init_frames=20, nb_default=1, thr_default=1, max_dist=0.3,
                          alpha=0.9, beta=0.1, power=2.0):
    md = map_dict(keys, counts, xyz)
    nb = 2 if frame_id < init_frames else nb_default
    thr = 1 if frame_id < init_frames else thr_default
    q = np.asarray(q, float)
    Rn = quat_to_rot(q / np.linalg.norm(q))
    R = quat_to_rot(q)
    lw, ln = abs(alpha) / (abs(alpha) + abs(beta)), abs(beta) / (abs(alpha) + abs(beta))
    n = len(raw)
    out = dict(status=np.full(n, 3, np.uint8), ids=np.full((n, K), -1, np.int32), normal=np.zeros((n, 3)), a2D=np.zeros(n),
               weight=np.zeros(n), norm_offset=np.zeros(n), distance=np.zeros(n), jacobian=np.zeros((n, 6)))
    H, h, loss, nres = [], [], 0.0, 0
    for k in range(n):
        p_imu = R_il @ raw[k] + t_il
        p_w = Rn @ p_imu + t
        cand, _ = search_neighbors(md, p_w, nb, size, K, thr)
        out["status"][k] = 0
        out["ids"][k, : len(cand)] = [c[1] for c in cand]
        if len(cand) >= min_nb:
            P = np.array([c[2] for c in cand])
            b = P.mean(0)
            C = (P - b).T @ (P - b)
            w, V = np.linalg.eigh(C)
            normal = V[:, 0] / np.linalg.norm(V[:, 0])
            s1, s2, s3 = np.sqrt(abs(w[2])), np.sqrt(abs(w[1])), np.sqrt(abs(w[0]))
            a2D = (s2 - s3) / s1
            if normal @ (t_last - p_imu) < 0:
                normal = -normal
            weight = lw * a2D ** power + ln * np.exp(-np.linalg.norm(P[0] - p_w) / (max_dist * min_nb))
            off = -normal @ P[0]
            dist = normal @ (R @ p_imu + t) + off
            out["status"][k] = 1
            out["normal"][k] = normal; out["a2D"][k] = a2D; out["weight"][k] = weight
            out["norm_offset"][k] = off; out["distance"][k] = dist
            if dist < max_dist:
                J = np.concatenate([normal * weight, -(normal @ R @ skew(p_imu)) * weight])
                out["status"][k] = 2
                out["jacobian"][k] = J
                H.append(J); h.append(dist * weight); loss += dist * dist; nres += 1
        if nres >= max_num_residuals:
            break
    H = np.array(H).reshape(-1, 6); h = np.array(h)
    out.update(HtH=H.T @ H, Hth=H.T @ h, loss=loss, num_residuals=nres)
    return out


def eskf_observe(s, dx):
    s = s.copy()
    s[0:3] += dx[0:3]
    qn = quat_mul(s[3:7], so3_to_quat(dx[3:6]))
    s[3:7] = qn / np.linalg.norm(qn)
    s[7:10] += dx[6:9]; s[10:13] += dx[9:12]; s[13:16] += dx[12:15]
    B = derivative_s2(s[16:19])
    s[16:19] = so3_to_rotation(B @ dx[15:17]) @ s[16:19]
    return s


def eskf_predict(s, P, dt, acc0, gyr0, acc1, gyr1, noise):
    s = s.copy()
    p, q, v, ba, bg, g = s[0:3], s[3:7], s[7:10], s[10:13], s[13:16], s[16:19]
    un_gyr = 0.5 * (gyr0 + gyr1) - bg
    un_acc = 0.5 * (acc0 + acc1) - ba
    Rb = quat_to_rot(q)
    s[3:7] = quat_mul(q, so3_to_quat(un_gyr * dt))
    s[0:3] = p + v * dt
    s[7:10] = v + Rb @ un_acc * dt - g * dt
    B = derivative_s2(g)
    F = np.zeros((17, 17)); I = np.eye(3)
    F[0:3, 0:3] = I; F[0:3, 6:9] = I * dt
    F[3:6, 3:6] = I - skew(un_gyr) * dt; F[3:6, 12:15] = -I * dt
    F[6:9, 3:6] = -Rb @ skew(un_acc) * dt; F[6:9, 6:9] = I; F[6:9, 9:12] = -Rb * dt
    F[6:9, 15:17] = skew(g) @ B * dt
    F[9:12, 9:12] = I; F[12:15, 12:15] = I
    F[15:17, 15:17] = -1.0 / (g @ g) * B.T @ skew(g) @ skew(g) @ B
    Fw = np.zeros((17, 12))
    Fw[6:9, 0:3] = -Rb * dt; Fw[3:6, 3:6] = -I * dt; Fw[9:12, 6:9] = -I * dt; Fw[12:15, 9:12] = -I * dt
    return s, F @ P @ F.T + Fw @ noise @ Fw.T


def update_iekf(keys, counts, xyz, raw, eskf_state, eskf_cov, state16, t_last, max_iter=5, frame_id=100, laser_cov=0.001,
                thr_t=0.01, thr_r=0.1, max_num_residuals=2**31 - 1):
    es = eskf_state.copy(); P = eskf_cov.copy()
    q = state16[0:4].copy(); t = state16[4:7].copy(); vel = state16[7:10].copy(); ba = state16[10:13].copy(); bg = state16[13:16].copy()
    pred = es.copy()
    dxs, iters = [], 0
    I3 = np.eye(3)
    for i in range(-1, max_iter):
        r = build_plane_residuals(keys, counts, xyz, raw, q, t, t_last, I3, np.zeros(3), frame_id=frame_id, max_num_residuals=max_num_residuals)
        if r["num_residuals"] < 20:
            return dict(iters=-1)
        iters += 1
        d_p = es[0:3] - pred[0:3]
        d_so3 = rotation_to_so3(quat_to_rot(quat_mul(quat_inv(pred[3:7]), es[3:7])))
        d_v = es[7:10] - pred[7:10]; d_ba = es[10:13] - pred[10:13]; d_bg = es[13:16] - pred[13:16]
        g, gp = es[16:19], pred[16:19]
        gpn, gn = gp / np.linalg.norm(gp), g / np.linalg.norm(g)
        cr, dt_ = np.cross(gpn, gn), gpn @ gn
        if abs(1.0 - dt_) < 1e-6:
            R_dg = np.eye(3)
        else:
            sk = skew(cr)
            R_dg = np.eye(3) + sk + sk @ sk * (1.0 - dt_) / (cr @ cr)
        so3_dg = rotation_to_so3(R_dg)
        B = derivative_s2(gp)
        d_g = B.T @ so3_dg
        d_x = np.concatenate([d_p, d_so3, d_v, d_ba, d_bg, d_g])
        J3 = np.eye(3) - 0.5 * skew(d_so3)
        J2 = np.eye(2) + 0.5 * B.T @ skew(so3_dg) @ B
        d_x_new = d_x.copy(); d_x_new[3:6] = J3 @ d_so3; d_x_new[15:17] = J2 @ d_g
        L = np.eye(17); L[3:6, 3:6] = J3; L[15:17, 15:17] = J2
        cov = L @ P @ L.T
        temp = np.linalg.inv(cov / laser_cov)
        temp[0:6, 0:6] += r["HtH"]
        S = np.linalg.inv(temp)
        K_h = S[:, 0:6] @ r["Hth"]
        K_x = np.zeros((17, 17)); K_x[:, 0:6] = S[:, 0:6] @ r["HtH"]
        d_x = -K_h + (K_x - np.eye(17)) @ d_x_new
        dxs.append(d_x.copy())
        g_before = es[16:19].copy()
        if np.linalg.norm(d_x[0:3]) > 100 or angular_distance(d_x[3:6]) > 100:
            continue
        es = eskf_observe(es, d_x)
        t, q, vel, ba, bg = es[0:3].copy(), es[3:7].copy(), es[7:10].copy(), es[10:13].copy(), es[13:16].copy()
        conv = frame_id > 1 and np.linalg.norm(d_x[0:3]) < thr_t and angular_distance(d_x[3:6]) < thr_r
        if conv or i == max_iter - 1:
            Bb = derivative_s2(g_before)
            J3 = np.eye(3) - 0.5 * skew(d_x[3:6])
            J2 = np.eye(2) + 0.5 * Bb.T @ skew(Bb @ d_x[15:17]) @ Bb
            # optimize.cpp:274-305 literally: row ops into covariance_new, column ops from the ORIGINAL covariance
            cov_new = cov.copy()
            cov_new[3:6, :] = J3 @ cov[3:6, :]
            cov_new[15:17, :] = J2 @ cov[15:17, :]
            cov_new[:, 3:6] = cov[:, 3:6] @ J3.T
            cov2 = cov.copy(); cov2[:, 3:6] = cov[:, 3:6] @ J3.T
            cov_new[:, 15:17] = cov2[:, 15:17] @ J2.T
            cov2[:, 15:17] = cov2[:, 15:17] @ J2.T
            K_x[3:6, 0:6] = J3 @ K_x[3:6, 0:6]
            K_x[15:17, 0:6] = J2 @ K_x[15:17, 0:6]
            P = cov_new - K_x[:, 0:6] @ cov2[0:6, :]
            break
    state = np.concatenate([q, t, vel, ba, bg])
    return dict(iters=iters, dx=dxs, state=state, eskf_state=es, eskf_cov=P)


# ---------------------------------------------------------------------------------------------
# eskfEstimator::tryInit statistics (eskfEstimator.cpp:93-118), written independently: running mean and the
# biased running variance recurrence, then the acceptance test of :47-63.
def try_init_stats(batches, g_norm=9.81):
    """batches: list of (t, gyr, acc). Returns dict with the statistics after the last batch and the batch index
    at which initialisation happened (or None)."""
    n = 1
    mean_g = mean_a = None
    var_g = np.zeros(3); var_a = np.zeros(3)
    t0 = None
    done_at = None
    for bi, (t, gyr, acc) in enumerate(batches):
        if t0 is None:
            t0 = t[0]; mean_g = np.array(gyr[0], float); mean_a = np.array(acc[0], float)
        for w, a in zip(gyr, acc):
            mean_g = mean_g + (w - mean_g) / n
            mean_a = mean_a + (a - mean_a) / n
            var_g = var_g * (n - 1.0) / n + (w - mean_g) ** 2 * (n - 1.0) / (n * n)
            var_a = var_a * (n - 1.0) / n + (a - mean_a) ** 2 * (n - 1.0) / (n * n)
            n += 1
        if n > 10 and t[-1] - t0 > 3.0:
            var_a = var_a * (g_norm / np.linalg.norm(mean_a)) ** 2
            if np.linalg.norm(var_g) > 0.5:
                return dict(code=-1, at=bi, mean_gyr=mean_g, mean_acc=mean_a, gyr_cov=var_g, acc_cov=var_a, n=n)
            if np.linalg.norm(var_a) > 0.6:
                return dict(code=-2, at=bi, mean_gyr=mean_g, mean_acc=mean_a, gyr_cov=var_g, acc_cov=var_a, n=n)
            done_at = bi
            break
    return dict(code=1 if done_at is not None else 0, at=done_at, mean_gyr=mean_g, mean_acc=mean_a, gyr_cov=var_g, acc_cov=var_a,
                n=n, bg=mean_g, gravity=mean_a / np.linalg.norm(mean_a) * g_norm)


def state_initialization_const_velocity(q2, t2, q1, t1):
    """constant-velocity prior (lioOptimization.cpp:906-917) with rotation matrices (unit quaternions)."""
    R1, R2 = quat_to_rot(q1), quat_to_rot(q2)
    D = R1 @ R2.T
    return D @ R1, t1 + D @ (t1 - t2)
