"""workloads/s2.py -- synthetic workload S2: one Estimator sliding window as optimization() sees it
(KITTI-05 intrinsics / extrinsic / weights, <= 150 tracks per frame).  Input plumbing for tests and bench.py."""
import numpy as np


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


# laser_to_camera0 of kitti_config_05.yaml (mono_lidar_mapping/config/kitti_config_05.yaml:27-30), digits as in the file.  The 3x3
# block is a rotation to ~1e-8 only (8 printed digits); the Estimator goes through Eigen::Quaterniond(TLC.block<3,3>) and
# q.normalized().toRotationMatrix() (matrix2Double / double2Matrix), which this module mirrors where an exact rotation is needed.
LASER_TO_CAM0 = np.array([[-0.00185773, -0.00648143, 0.99997727, 0.3308678],
                          [-0.99996595, 0.00805187, -0.00180552, 0.05505896],
                          [-0.00803998, -0.99994658, -0.00649617, -0.07543742],
                          [0.0, 0.0, 0.0, 1.0]])


def kitti_extrinsic():
    return LASER_TO_CAM0.copy()


def R_to_q(m):
    """Eigen Quaternion(Matrix3) constructor: xyzw."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


FX, FY, CX, CY, W_IMG, H_IMG = 707.0912, 707.0912, 601.8873, 183.1104, 1241, 376


def make_window(seed=0, n_frames=11, n_landmarks=4000, max_tracks=150, pix_sigma=0.5, odo_sigma_t=0.01, odo_sigma_r=np.deg2rad(0.05),
                track_cnt=3, use_prior=True, perturb=True, min_dist=30):
    """One Estimator window as optimization() sees it (Estimator.cc:1124-1215): state, feature tracks, LiDAR increments."""
    rng = np.random.default_rng(20241 + seed)
    T = kitti_extrinsic()                       # laser <- camera
    q = R_to_q(T[:3, :3])
    T[:3, :3] = quat_R(q / np.linalg.norm(q))   # the exact rotation the Estimator works with (Quaterniond(TLC).normalized(), matrix2Double)
    Rlc, tlc = T[:3, :3], T[:3, 3]
    # LiDAR ground-truth poses in the LiDAR odometry frame: forward 0.8 m / frame, gentle yaw
    L0_R, L0_P = [], []
    yaw, pos = 0.0, np.zeros(3)
    for k in range(n_frames):
        c, s = np.cos(yaw), np.sin(yaw)
        L0_R.append(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])); L0_P.append(pos.copy())
        pos = pos + L0_R[-1] @ np.array([0.8, 0, 0]); yaw += rng.normal(0.01, 0.005)
    # Estimator world: Rs = Rlc^T L0_R, Ps = Rlc^T (L0_P - tlc)   (Estimator.cc:995-996)
    Rs = [Rlc.T @ R for R in L0_R]; Ps = [Rlc.T @ (P - tlc) for P in L0_P]
    cam_R = [Rs[k] @ Rlc for k in range(n_frames)]; cam_P = [Ps[k] + Rs[k] @ tlc for k in range(n_frames)]
    # landmarks in a corridor ahead of the path, expressed in the Estimator world (camera-aligned: z forward, y down)
    lm_l = np.stack([rng.uniform(2, 120, n_landmarks), rng.uniform(-20, 20, n_landmarks), rng.uniform(-1.5, 6.5, n_landmarks)], 1)
    lm = (Rlc.T @ (lm_l - tlc).T).T
    tracks = {}
    alive = []
    for k in range(n_frames):
        pc = (cam_R[k].T @ (lm - cam_P[k]).T).T
        z = pc[:, 2]
        u = FX * pc[:, 0] / z + CX; v = FY * pc[:, 1] / z + CY
        vis = (z > 1.0) & (u > 0) & (u < W_IMG) & (v > 0) & (v < H_IMG)
        alive = [t for t in alive if vis[t] and rng.uniform() > 0.1]
        cand = [t for t in np.nonzero(vis)[0] if t not in tracks]
        rng.shuffle(cand)
        taken = [(u[t], v[t]) for t in alive]
        for t in cand:
            if len(alive) >= max_tracks:
                break
            if all((u[t] - a) ** 2 + (v[t] - b) ** 2 > min_dist ** 2 for a, b in taken):
                alive.append(t); taken.append((u[t], v[t])); tracks[t] = (k, [])
        for t in alive:
            un = (u[t] + rng.normal(0, pix_sigma) - CX) / FX; vn = (v[t] + rng.normal(0, pix_sigma) - CY) / FY
            tracks[t][1].append((un, vn))
    feat_depth, obs_feat, obs_i, obs_j, obs_pts = [], [], [], [], []
    trk_start, trk_off, trk_pts, trk_true = [], [0], [], []
    for t, (k0, obs) in sorted(tracks.items(), key=lambda kv: (kv[1][0], kv[0])):
        if len(obs) < track_cnt:
            continue
        f = len(feat_depth)
        trk_start.append(k0); trk_pts.extend(obs); trk_off.append(trk_off[-1] + len(obs))
        trk_true.append((cam_R[k0].T @ (lm[t] - cam_P[k0]))[2])
        z0 = (cam_R[k0].T @ (lm[t] - cam_P[k0]))[2]
        feat_depth.append(z0 * (1 + rng.normal(0, 0.05)))
        for d, (un, vn) in enumerate(obs):
            if d == 0:
                continue
            obs_feat.append(f); obs_i.append(k0); obs_j.append(k0 + d); obs_pts.append([obs[0][0], obs[0][1], un, vn])
    # LiDAR odometry input: ground truth + noise; LASERFactor consts are (L0_Ri, L0_Rj, L0_Pi, L0_Pj)
    nR = [R @ quat_R(np.concatenate([rng.normal(0, odo_sigma_r / 2, 3), [1.0]]) / 1.0) for R in L0_R]
    nP = [P + rng.normal(0, odo_sigma_t, 3) for P in L0_P]
    laser = np.array([np.concatenate([nR[k].ravel(), nR[k + 1].ravel(), nP[k], nP[k + 1]]) for k in range(n_frames - 1)])
    # window state: ground truth, newest frame one motion behind (slideWindow keeps the previous newest pose), small noise
    poses = []
    for k in range(n_frames):
        kk = k if (k < n_frames - 1 or not perturb) else k - 1
        q = R_to_q(Rs[kk]); p = Ps[kk].copy()
        if perturb:
            p = p + rng.normal(0, 0.01, 3)
        poses.append(np.concatenate([p, q]))
    ex = np.concatenate([tlc, R_to_q(Rlc)])
    return dict(poses=np.array(poses), ex=ex, inv_depth=1.0 / np.array(feat_depth),
                obs_feat=np.array(obs_feat, np.int32), obs_i=np.array(obs_i, np.int32), obs_j=np.array(obs_j, np.int32),
                obs_pts=np.array(obs_pts), laser_consts=laser, laser_info=(3.0 * 1500.0) * np.eye(6), mono_info=1500.0 * np.eye(2),
                prior_T=T.copy(), prior_w=np.array([1000.0, 1000.0]), use_prior=use_prior, ex_constant=False, use_mono=True,
                gt_Rs=np.array(Rs), gt_Ps=np.array(Ps), tlc=T.copy(),
                trk_start=np.array(trk_start, np.int32), trk_off=np.array(trk_off, np.int32), trk_pts=np.array(trk_pts),
                trk_true_depth=np.array(trk_true))


def make_stream(n_frames=120, seed=0, n_landmarks=None, max_tracks=150, pix_sigma=0.5, odo_sigma_t=0.01, odo_sigma_r=np.deg2rad(0.05),
                death=0.1, min_dist=30.0, speed=0.8, stops=(), lidar_gt=None, lidar_meas=None):
    """BASELINE configs[2] as a frame STREAM (what Estimator::processEstimation consumes frame by frame, Estimator.cc:528-553):
    per frame the LiDAR odometry pose (ground truth + noise; topic /aft_mapped_to_init) and the tracker's output
    {feature id: (x_n, y_n, u, v)} (FeatureTracker::trackImage, <= max_tracks features, min_dist pixels apart, a track dies
    with probability `death` per frame).  Landmarks fill a corridor along the whole path.  `stops`: frame indices at which the
    vehicle stands still for that frame (static_status, Estimator.cc:259-265).  lidar_gt = (R [n,3,3], P [n,3]): ground-truth LiDAR
    poses to use instead of the built-in path (e.g. the S1 trajectory); lidar_meas [n,4,4]: the LiDAR odometry to hand to the
    Estimator instead of ground truth + noise (e.g. the output of the GPU laserOdometry / laserMapping on S1 scans: configs[0]).
    Returns dict(headers [n], L0 [n,4,4], feats (list of dicts), gt_R [n,3,3], gt_P [n,3] in the Estimator world, tlc)."""
    rng = np.random.default_rng(20241 + 7919 * seed)
    T = kitti_extrinsic()
    Rlc, tlc = quat_R(R_to_q(T[:3, :3]) / np.linalg.norm(R_to_q(T[:3, :3]))), T[:3, 3]
    L0_R, L0_P = [], []
    yaw, pos = 0.0, np.zeros(3)
    for k in range(n_frames):
        c, s = np.cos(yaw), np.sin(yaw)
        L0_R.append(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])); L0_P.append(pos.copy())
        if k + 1 not in stops:
            pos = pos + L0_R[-1] @ np.array([speed, 0, 0]); yaw += rng.normal(0.01, 0.005)
    if lidar_gt is not None:
        L0_R = [np.asarray(R, np.float64) for R in lidar_gt[0][:n_frames]]; L0_P = [np.asarray(P, np.float64) for P in lidar_gt[1][:n_frames]]
    Rs = [Rlc.T @ R for R in L0_R]; Ps = [Rlc.T @ (P - tlc) for P in L0_P]
    cam_R = [Rs[k] @ Rlc for k in range(n_frames)]; cam_P = [Ps[k] + Rs[k] @ tlc for k in range(n_frames)]
    # landmarks around the whole path (LiDAR frame: x forward, y left, z up), ~33 per metre of path as in make_window
    path = np.array(L0_P)
    if n_landmarks is None:
        n_landmarks = int(33 * (np.linalg.norm(np.diff(path, axis=0), axis=1).sum() + 120))
    a = rng.uniform(0, 1, n_landmarks)
    idx = np.minimum((a * (n_frames - 1)).astype(int), n_frames - 1)
    base = path[idx]; head = np.array([L0_R[i][:, 0] for i in idx]); left = np.array([L0_R[i][:, 1] for i in idx])
    lm_l = base + head * rng.uniform(2, 120, n_landmarks)[:, None] + left * rng.uniform(-20, 20, n_landmarks)[:, None]
    lm_l[:, 2] = rng.uniform(-1.5, 6.5, n_landmarks)
    lm = (Rlc.T @ (lm_l - tlc).T).T
    # landmarks sorted by the path index they hang on: a frame only looks at those within [-20, +200] frames of itself
    # (up to ~160 m ahead), which keeps the generator linear in the number of frames
    order = np.argsort(idx, kind="stable")
    lm = lm[order]; idx = idx[order]
    feats, alive, seen = [], [], set()
    for k in range(n_frames):
        lo, hi = np.searchsorted(idx, k - 20, "left"), np.searchsorted(idx, k + 200, "right")
        pc = (cam_R[k].T @ (lm[lo:hi] - cam_P[k]).T).T
        z = pc[:, 2]
        zs = np.where(z > 1e-6, z, 1.0)
        u = np.full(len(lm), -1.0); v = np.full(len(lm), -1.0); vis = np.zeros(len(lm), bool)
        u[lo:hi] = FX * pc[:, 0] / zs + CX; v[lo:hi] = FY * pc[:, 1] / zs + CY
        vis[lo:hi] = (z > 1.0) & (u[lo:hi] > 0) & (u[lo:hi] < W_IMG) & (v[lo:hi] > 0) & (v[lo:hi] < H_IMG)
        alive = [t for t in alive if vis[t] and rng.uniform() > death]
        cand = [t for t in (lo + np.nonzero(vis[lo:hi])[0]) if t not in seen]
        rng.shuffle(cand)
        taken = np.array([(u[t], v[t]) for t in alive]).reshape(-1, 2)
        for t in cand[:500]:                      # goodFeaturesToTrack looks at a bounded number of corners
            if len(alive) >= max_tracks:
                break
            if len(taken) == 0 or ((taken[:, 0] - u[t]) ** 2 + (taken[:, 1] - v[t]) ** 2).min() > min_dist ** 2:
                alive.append(t); seen.add(t); taken = np.concatenate([taken, [[u[t], v[t]]]])
        fr = {}
        for t in alive:
            uu = u[t] + rng.normal(0, pix_sigma); vv = v[t] + rng.normal(0, pix_sigma)
            fr[int(t)] = ((uu - CX) / FX, (vv - CY) / FY, uu, vv)
        feats.append(fr)
    L0 = np.zeros((n_frames, 4, 4))
    for k in range(n_frames):
        L0[k] = np.eye(4)
        L0[k, :3, :3] = L0_R[k] @ quat_R(np.concatenate([rng.normal(0, odo_sigma_r / 2, 3), [1.0]]))
        L0[k, :3, 3] = L0_P[k] + rng.normal(0, odo_sigma_t, 3)
    if lidar_meas is not None:
        L0 = np.asarray(lidar_meas, np.float64)[:n_frames].copy()
    return dict(headers=0.1 * np.arange(n_frames), L0=L0, feats=feats, gt_R=np.array(Rs), gt_P=np.array(Ps), tlc=T)


def write_stream(path, st, loops=()):
    """The frame stream in the binary layout lmono_amd/host/estimator_seq reads (doubles): n_frames, TLC[16], then per frame: header,
    L0_Pos[16], n_loop (0/1) [stamp, old_T[3], old_Q[4] w x y z, correct_T[3], correct_Q[4] w x y z], n_features, n x (id, x_n, y_n, u, v)."""
    by_frame = {e["frame"]: e for e in loops}
    buf = [float(len(st["headers"]))] + list(np.asarray(st["tlc"], np.float64).ravel())
    for k in range(len(st["headers"])):
        buf += [float(st["headers"][k])] + list(st["L0"][k].ravel())
        e = by_frame.get(k)
        if e is None:
            buf.append(0.0)
        else:
            buf.append(1.0)
            buf += [e["stamp"]] + list(e["old_T"]) + list(e["old_Q"]) + list(e["correct_T"]) + list(e["correct_Q"])
        fr = st["feats"][k]
        buf.append(float(len(fr)))
        for fid in sorted(fr):
            buf += [float(fid)] + list(fr[fid])
    np.array(buf, np.float64).tofile(str(path))
