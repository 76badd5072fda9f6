"""oracle/estimator_ref.py -- TEST INFRASTRUCTURE (CPU oracle).  PARITY UNPINNED (the reference holds no tests or fixtures).

The Estimator's frame loop restated on the CPU: the window / track bookkeeping of
/root/reference/mono_lidar_mapping/src/image_process/Estimator.cc and FeatureManager.cc in plain Python (list surgery, small
cases), every numeric step through the C oracle (oracle/lo_ba*.c, lo_marg.c via oracle.py).  One `process(header, L0_pose,
features)` call = one pass of Estimator::processEstimation (Estimator.cc:528-553) without ROS, the image tracker (its output is the
input here) and the visualiser:

    processCompactData   Estimator.cc:236-273    static_status from the LiDAR translation
    processImage         :367-499                 featureCheck -> keyframe?, state machine NOT_INITED -> INITED
    featureCheck         FeatureManager.cc:315-400, computeParallax :279-313
    runInitialization    Estimator.cc:852-1017    window poses from the LiDAR poses, clearDepth, triangulate, reject > 100
    loopCorrection       :309-365                 rigid re-anchoring of the window on a loop frame
    optimization         :1124-1305               through oracle.ba_solve + double2Matrix (:1059-1122), then margin()
    margin               :1307-1470               MARGIN_OLD and MARGIN_SECOND_NEW; the prior is never consumed (`valid` stays false)
    outliersRejection    :134-190, slideWindow :700-771, removeBackShiftDepth / removeBack / removeFront FeatureManager.cc:497-590
    trajectory of record :634-645                 Ps / Rs[WINDOW_SIZE] after the slide, one row per INITED frame

Only tests/ and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

from . import oracle as O
from . import ba_numpy as B

WINDOW_SIZE = 10
INIT_DEPTH = -1.0
NOT_INITED, INITED = 0, 1
MARGIN_OLD, MARGIN_SECOND_NEW = 0, 1


class Params:
    """kitti_config_05.yaml values used on the path."""
    FACTOR_WEIGHT = 1500.0
    LASER_W = 3.0
    PRIOR_T = 1000.0
    PRIOR_R = 1000.0
    OUTLIER_T = 5.0
    TRACK_CNT = 3
    FINE_TIMES = 1
    NUM_ITERATIONS = 30
    ESTIMATE_LASER = 1
    FEATURE_THRESHOLD = 10.0


class Feature:
    __slots__ = ("feature_id", "start_frame", "obs", "estimated_depth", "solve_flag")

    def __init__(self, fid, start):
        self.feature_id = fid; self.start_frame = start; self.obs = []; self.estimated_depth = INIT_DEPTH; self.solve_flag = 0

    def end_frame(self):
        return self.start_frame + len(self.obs) - 1


def _R_to_q(R):
    return B.R_to_q(np.asarray(R, np.float64))


class EstimatorRef:
    def __init__(self, TLC, params=None):
        self.p = params or Params()
        self.Rs = [np.eye(3) for _ in range(WINDOW_SIZE + 1)]
        self.Ps = [np.zeros(3) for _ in range(WINDOW_SIZE + 1)]
        self.Header = [0.0] * (WINDOW_SIZE + 1)
        self.TLC = np.array(TLC, np.float64).copy()
        self.frames = []                      # all_image_frame: (header, L0_R, L0_T)
        self.feature = []
        self.frame_count = 0
        self.first_refine = 0
        self.stage_flag = NOT_INITED
        self.marginalization_flag = MARGIN_OLD
        self.static_status = False
        self.last_laser_t = np.zeros(3)
        self.loop_buf = []
        self.loop_closure = False
        self.last_marg = None                 # dict(J, r, x0 [nb,7], blocks [names]) -- last_marginalization_info + parameter blocks
        self.marg_log = []                    # one entry per margin() call: (flag, n_blocks_after)
        self.trajectory = []                  # new_odometry rows: header, P (3), q xyzw (4)
        self.solve_log = []                   # (iterations, termination, final_cost) per optimization()
        self.back_R0 = np.eye(3); self.back_P0 = np.zeros(3)

    # ---- Estimator.cc:236-273
    def processCompactData(self, L0):
        t = L0[:3, 3]
        self.static_status = bool(np.linalg.norm(t - self.last_laser_t) < 0.1)
        self.last_laser_t = t.copy()

    # ---- FeatureManager.cc:315-400
    def featureCheck(self, frame_count, image):
        last_track_num = new_feature_num = 0
        by_id = {f.feature_id: f for f in self.feature}
        for fid in sorted(image):                  # std::map iteration: ascending feature id
            ob = image[fid]
            f = by_id.get(fid)
            if f is None:
                f = Feature(fid, frame_count)
                f.obs.append(tuple(ob))
                self.feature.append(f); by_id[fid] = f
                new_feature_num += 1
            else:
                f.obs.append(tuple(ob))
                last_track_num += 1
        if frame_count < 2 or last_track_num < 20 or new_feature_num > 0.5 * last_track_num:
            return True
        parallax_sum, parallax_num = 0.0, 0
        for f in self.feature:
            if f.start_frame <= frame_count - 2 and f.start_frame + len(f.obs) - 1 >= frame_count - 1:
                a = f.obs[frame_count - 2 - f.start_frame]; b = f.obs[frame_count - 1 - f.start_frame]
                parallax_sum += max(0.0, float(np.sqrt((a[2] - b[2]) ** 2 + (a[3] - b[3]) ** 2)))      # computeParallax: pixel (uv) distance
                parallax_num += 1
        if parallax_num == 0:
            return True
        return parallax_sum / parallax_num >= self.p.FEATURE_THRESHOLD

    # ---- packing shared by triangulate / outliersRejection (all tracks; the C side skips those below TRACK_CNT)
    def _pack_tracks(self):
        start = np.array([f.start_frame for f in self.feature], np.int32)
        off = np.concatenate([[0], np.cumsum([len(f.obs) for f in self.feature])]).astype(np.int32)
        pts = np.array([[o[0], o[1]] for f in self.feature for o in f.obs], np.float64).reshape(-1, 2)
        depth = np.array([f.estimated_depth for f in self.feature], np.float64)
        return start, off, pts, depth

    def _RsPs(self):
        return np.array(self.Rs).reshape(WINDOW_SIZE + 1, 9), np.array(self.Ps)

    # ---- FeatureManager.cc:75-255
    def triangulate(self):
        if not self.feature:
            return
        start, off, pts, depth = self._pack_tracks()
        R, P = self._RsPs()
        _, d1, flag = O.triangulate(R, P, self.TLC, start, off, pts, depth, track_cnt=self.p.TRACK_CNT, window_size=WINDOW_SIZE,
                                    weight=self.p.FACTOR_WEIGHT)
        for k, f in enumerate(self.feature):
            f.estimated_depth = float(d1[k])
            if len(f.obs) >= self.p.TRACK_CNT:
                f.solve_flag = int(flag[k])

    # ---- Estimator.cc:134-190
    def outliersRejection(self, error):
        if not self.feature:
            return set()
        start, off, pts, depth = self._pack_tracks()
        R, P = self._RsPs()
        sc = O.outlier_scores(R, P, self.TLC, start, off, pts, depth, track_cnt=self.p.TRACK_CNT, weight=self.p.FACTOR_WEIGHT)
        return {f.feature_id for k, f in enumerate(self.feature) if sc[k] >= 0 and sc[k] > error}

    def removeOutlier(self, ids):
        self.feature = [f for f in self.feature if f.feature_id not in ids]

    # ---- Estimator.cc:852-1017 (the live part: :986-1012)
    def runInitialization(self):
        rlc = self.TLC[:3, :3]; tlc = self.TLC[:3, 3]
        for i in range(self.frame_count + 1):
            _, L0_R, L0_T = self.frames[i]
            self.Rs[i] = rlc.T @ L0_R
            self.Ps[i] = rlc.T @ (L0_T - tlc)
        for f in self.feature:                     # clearDepth
            f.solve_flag = 0; f.estimated_depth = INIT_DEPTH
        self.triangulate()
        self.removeOutlier(self.outliersRejection(100.0))
        return True

    # ---- Estimator.cc:309-365
    def setLoopFrame(self, loop_time_stamp, old_T, old_Q_wxyz, correct_T, correct_Q_wxyz):
        self.loop_buf.append(dict(stamp=loop_time_stamp, old_T=np.array(old_T, float), old_Q=np.array(old_Q_wxyz, float),
                                  correct_T=np.array(correct_T, float), correct_Q=np.array(correct_Q_wxyz, float)))

    def loopCorrection(self):
        if not self.loop_buf:
            return
        lf = self.loop_buf[-1]                     # the while loop keeps the last one
        self.loop_buf = []
        idx = -1
        for i in range(WINDOW_SIZE):
            if lf["stamp"] == self.Header[i]:
                idx = i
        if idx < 0:
            return
        self.loop_closure = True
        w, x, y, z = lf["correct_Q"]
        Rc = B.q_to_R(np.array([x, y, z, w]))      # Eigen::Quaterniond(w, x, y, z).toRotationMatrix(): the message's quaternion as it is
        Ri, Pi = self.Rs[idx].copy(), self.Ps[idx].copy()
        for i in range(WINDOW_SIZE + 1):
            if i != idx:
                rel_r = Ri.T @ self.Rs[i]
                rel_t = Ri.T @ (Pi - self.Ps[i])
                self.Rs[i] = Rc @ rel_r
                self.Ps[i] = lf["correct_T"] - Rc @ rel_t
        self.Rs[idx] = Rc; self.Ps[idx] = lf["correct_T"].copy()

    # ---- Estimator.cc:1019-1057 / :1124-1305
    def _window(self):
        """The window as oracle.ba_solve takes it (the layout of workloads/s2.make_window)."""
        n = WINDOW_SIZE + 1
        poses = np.array([np.concatenate([self.Ps[i], _R_to_q(self.Rs[i])]) for i in range(n)])
        ex = np.concatenate([self.TLC[:3, 3], _R_to_q(self.TLC[:3, :3])])
        obs_feat, obs_i, obs_j, obs_pts, invd, used = [], [], [], [], [], []
        for f in self.feature:
            if len(f.obs) < self.p.TRACK_CNT:
                continue
            fi = len(invd)
            invd.append(1.0 / f.estimated_depth); used.append(f)
            for d, o in enumerate(f.obs):
                if d == 0:
                    continue
                obs_feat.append(fi); obs_i.append(f.start_frame); obs_j.append(f.start_frame + d)
                obs_pts.append([f.obs[0][0], f.obs[0][1], o[0], o[1]])
        laser = np.zeros((WINDOW_SIZE, 24))
        for i in range(self.frame_count):
            _, Ri, Ti = self.frames[i]; _, Rj, Tj = self.frames[i + 1]
            laser[i] = np.concatenate([Ri.ravel(), Rj.ravel(), Ti, Tj])
        w = dict(poses=poses, ex=ex, inv_depth=np.array(invd, np.float64), obs_feat=np.array(obs_feat, np.int32), obs_i=np.array(obs_i, np.int32),
                 obs_j=np.array(obs_j, np.int32), obs_pts=np.array(obs_pts, np.float64).reshape(-1, 4), laser_consts=laser,
                 laser_info=(self.p.LASER_W * self.p.FACTOR_WEIGHT) * np.eye(6), mono_info=self.p.FACTOR_WEIGHT * np.eye(2),
                 prior_T=self.TLC.copy(), prior_w=np.array([self.p.PRIOR_T, self.p.PRIOR_R]))
        return w, used

    def optimization(self):
        w, used = self._window()
        use_prior = self.first_refine >= self.p.FINE_TIMES
        if not use_prior:
            self.first_refine += 1
        w["use_prior"] = use_prior
        w["ex_constant"] = self.p.ESTIMATE_LASER == 0
        w["use_mono"] = bool(self.p.ESTIMATE_LASER) and not self.static_status
        poses, ex, invd, sm = O.ba_solve(w, max_iter=self.p.NUM_ITERATIONS)
        self.solve_log.append((sm.iterations, sm.termination, sm.final_cost))
        if getattr(self, "capture", None) is not None:
            # test hook: the window problem exactly as it went into the solve, and what came out (teacher-forced parity of a frame loop)
            import copy
            self.capture.append(dict(window=copy.deepcopy(w), poses=np.array(poses), ex=np.array(ex), inv_depth=np.array(invd),
                                     iterations=sm.iterations, termination=sm.termination, initial_cost=sm.initial_cost, final_cost=sm.final_cost,
                                     R0=self.Rs[0].copy(), P0=self.Ps[0].copy()))
        # double2Matrix
        R_new, P_new = O.ba_reanchor(poses, self.Rs[0], self.Ps[0])
        for i in range(WINDOW_SIZE + 1):
            self.Rs[i] = R_new[i].copy(); self.Ps[i] = P_new[i].copy()
        self.TLC[:3, 3] = ex[:3]
        self.TLC[:3, :3] = B.q_to_R(B.q_normalized(ex[3:]))
        inv_out = invd if w["use_mono"] else w["inv_depth"]
        for k, f in enumerate(used):                 # setDepth
            f.estimated_depth = 1.0 / inv_out[k]
            f.solve_flag = 2 if (f.estimated_depth < 0.1 or f.estimated_depth > 300) else 1
        self.feature = [f for f in self.feature if f.solve_flag != 2]      # removeFailures
        self.loop_closure = False
        if self.frame_count < WINDOW_SIZE:
            return False
        if self.p.ESTIMATE_LASER:
            self.margin()
        return sm.termination == 0 or sm.final_cost < 5e-3

    # ---- Estimator.cc:1307-1470
    def margin(self):
        if self.marginalization_flag == MARGIN_OLD:
            w, used = self._window()
            sel = np.nonzero(w["obs_i"] == 0)[0]
            feats = sorted(set(int(f) for f in w["obs_feat"][sel]))
            remap = {f: k for k, f in enumerate(feats)}
            ww = dict(w)
            ww["obs_feat"] = w["obs_feat"]; ww["obs_i"] = w["obs_i"]
            J, r, m, x0, _ = O.marginalize(ww)
            # kept blocks [ex, pose1..pose10]; the address shift (:1390-1396) renames pose i -> pose i-1
            self.last_marg = dict(J=J, r=r, x0=x0.copy(), blocks=["ex"] + ["pose%d" % i for i in range(WINDOW_SIZE)], m=m, n_f0=len(feats))
            self.marg_log.append((MARGIN_OLD, len(self.last_marg["blocks"])))
        else:
            lm = self.last_marg
            name = "pose%d" % (WINDOW_SIZE - 1)
            if lm is not None and name in lm["blocks"]:
                drop = lm["blocks"].index(name)
                cur = self._block_values(lm["blocks"])
                J, r = O.marg_second_new(lm["J"], lm["r"], lm["x0"], cur, drop)
                blocks = [b for b in lm["blocks"] if b != name]      # pose10 -> pose9 is not among them; all others keep their name
                self.last_marg = dict(J=J, r=r, x0=np.delete(cur, drop, 0), blocks=blocks, m=6, n_f0=0)
                self.marg_log.append((MARGIN_SECOND_NEW, len(blocks)))
            else:
                self.marg_log.append((MARGIN_SECOND_NEW, -1))

    def _block_values(self, blocks):
        """para_* of the named blocks after matrix2Double."""
        out = []
        for b in blocks:
            if b == "ex":
                out.append(np.concatenate([self.TLC[:3, 3], _R_to_q(self.TLC[:3, :3])]))
            else:
                i = int(b[4:])
                out.append(np.concatenate([self.Ps[i], _R_to_q(self.Rs[i])]))
        return np.array(out)

    # ---- Estimator.cc:700-771, FeatureManager.cc:497-590
    def slideWindow(self):
        if self.marginalization_flag == MARGIN_OLD:
            self.back_R0 = self.Rs[0].copy(); self.back_P0 = self.Ps[0].copy()
            if self.frame_count == WINDOW_SIZE:
                for i in range(self.frame_count):
                    self.Header[i] = self.Header[i + 1]
                    self.Rs[i], self.Rs[i + 1] = self.Rs[i + 1], self.Rs[i]
                    self.Ps[i], self.Ps[i + 1] = self.Ps[i + 1], self.Ps[i]
                self.Rs[WINDOW_SIZE] = self.Rs[WINDOW_SIZE - 1].copy()
                self.Ps[WINDOW_SIZE] = self.Ps[WINDOW_SIZE - 1].copy()
                self.Header[WINDOW_SIZE] = self.Header[WINDOW_SIZE - 1]
                self.frames.pop(0)
                if self.stage_flag == NOT_INITED:
                    self._removeBack()
                else:
                    self._removeBackShiftDepth()
        elif self.frame_count == WINDOW_SIZE:
            self.Header[self.frame_count - 1] = self.Header[self.frame_count]
            self.Ps[self.frame_count - 1] = self.Ps[self.frame_count].copy()
            self.Rs[self.frame_count - 1] = self.Rs[self.frame_count].copy()
            # all_image_frame.erase(all_image_frame.end() - 1) (:735) removes the LAST element, i.e. the NEWEST frame's LiDAR pose, while
            # slot WINDOW_SIZE - 1 takes the newest Ps / Rs / Header: reproduced as written
            self.frames.pop()
            self._removeFront(self.frame_count)

    def _removeBack(self):
        out = []
        for f in self.feature:
            if f.start_frame != 0:
                f.start_frame -= 1; out.append(f)
            else:
                f.obs.pop(0)
                if f.obs:
                    out.append(f)
        self.feature = out

    def _removeBackShiftDepth(self):
        rlc = self.TLC[:3, :3]; tlc = self.TLC[:3, 3]
        R0 = self.back_R0 @ rlc; P0 = self.back_P0 + self.back_R0 @ tlc
        R1 = self.Rs[0] @ rlc; P1 = self.Ps[0] + self.Rs[0] @ tlc
        del R0, P0, R1, P1          # the C function composes them itself from (back_R0, back_P0, Rs[0], Ps[0], TLC)
        sel = [f for f in self.feature if f.start_frame == 0 and len(f.obs) >= 3]
        new_d = O.shift_depth(self.back_R0, self.back_P0, self.Rs[0], self.Ps[0], self.TLC,
                              np.array([[f.obs[0][0], f.obs[0][1]] for f in sel]).reshape(-1, 2), np.array([f.estimated_depth for f in sel])) if sel else []
        nd = {id(f): new_d[k] for k, f in enumerate(sel)}
        out = []
        for f in self.feature:
            if f.start_frame != 0:
                f.start_frame -= 1; out.append(f); continue
            keep = len(f.obs) >= 3
            f.obs.pop(0)
            if not keep:
                continue
            f.estimated_depth = float(nd[id(f)])
            out.append(f)
        self.feature = out

    def _removeFront(self, frame_count):
        out = []
        for f in self.feature:
            if f.start_frame == frame_count:
                f.start_frame -= 1; out.append(f); continue
            j = WINDOW_SIZE - 1 - f.start_frame
            if f.end_frame() < frame_count - 1:
                out.append(f); continue
            f.obs.pop(j)
            if f.obs:
                out.append(f)
        self.feature = out

    # ---- Estimator.cc:367-499 + the trajectory of record (:634-645)
    def process(self, header, L0, image):
        L0 = np.asarray(L0, np.float64)
        self.processCompactData(L0)
        keyframe = self.featureCheck(self.frame_count, image)
        self.marginalization_flag = MARGIN_OLD if keyframe else MARGIN_SECOND_NEW
        self.Header[self.frame_count] = header
        self.frames.append((header, L0[:3, :3].copy(), L0[:3, 3].copy()))
        if self.stage_flag == NOT_INITED:
            if self.frame_count == WINDOW_SIZE:
                if self.p.ESTIMATE_LASER != 2 and self.runInitialization():
                    self.optimization()
                    self.stage_flag = INITED
                    self.removeOutlier(self.outliersRejection(3))
                    self.slideWindow()
                else:
                    self.slideWindow()
            if self.frame_count < WINDOW_SIZE:
                self.frame_count += 1
                self.Ps[self.frame_count] = self.Ps[self.frame_count - 1].copy()
                self.Rs[self.frame_count] = self.Rs[self.frame_count - 1].copy()
                self.Header[self.frame_count] = self.Header[self.frame_count - 1]
        else:
            self.loopCorrection()
            self.triangulate()
            self.optimization()
            self.removeOutlier(self.outliersRejection(self.p.OUTLIER_T))
            self.slideWindow()
        if self.stage_flag == INITED:
            self.trajectory.append(np.concatenate([[self.Header[WINDOW_SIZE]], self.Ps[WINDOW_SIZE], _R_to_q(self.Rs[WINDOW_SIZE])]))
        return keyframe
