"""Shared by the Estimator frame-loop tests: the S2 frame stream in the binary layout lmono_amd/host/estimator_seq reads, and the
CPU oracle's replay of the same stream."""
import numpy as np


def loop_event(st, frame, old_T=(1.0, 2.0, 3.0), shift=(0.05, -0.02, 0.03), yaw=0.002):
    """A loop-closure message arriving with frame `frame`, matched to the frame two slots back: layout of KeyFrame.cc:666-682 as
    Estimator.h:89-93 reads it (old_T, old_Q w x y z, correct_T, correct_Q w x y z, stamp)."""
    return dict(frame=frame, stamp=float(st["headers"][frame - 2]), old_T=np.array(old_T), old_Q=np.array([1.0, 0, 0, 0]),
                shift=np.array(shift), yaw=yaw)


from workloads.s2 import write_stream  # noqa: E402,F401  (the stream file writer is input plumbing)


def replay_oracle(st, loops=(), on_frame=None, capture=False):
    """-> (EstimatorRef after the stream, per-frame log rows (keyframe, stage, static, iterations, termination, final_cost, n_old, n_new, n_feat))."""
    from oracle import estimator_ref as E
    est = E.EstimatorRef(st["tlc"])
    if capture:
        est.capture = []
    by_frame = {e["frame"]: e for e in loops}
    log = []
    n_old = n_new = 0
    for k in range(len(st["headers"])):
        e = by_frame.get(k)
        if e is not None:
            if "correct_T" not in e:
                # the corrected pose of the matched window frame: its current estimate, nudged (needs the live window -> filled here)
                idx = [i for i in range(E.WINDOW_SIZE) if est.Header[i] == e["stamp"]]
                base_P = est.Ps[idx[0]] if idx else np.zeros(3)
                base_q = E._R_to_q(est.Rs[idx[0]]) if idx else np.array([0, 0, 0, 1.0])
                e["correct_T"] = base_P + e["shift"]
                from oracle import ba_numpy as B
                dq = np.array([0.0, np.sin(e["yaw"] / 2), 0.0, np.cos(e["yaw"] / 2)])       # about the camera-aligned world's vertical (y) axis
                q = B.q_mul(base_q, dq)
                e["correct_Q"] = np.array([q[3], q[0], q[1], q[2]])                          # w x y z
            est.setLoopFrame(e["stamp"], e["old_T"], e["old_Q"], e["correct_T"], e["correct_Q"])
        n_marg = len(est.marg_log)
        kf = est.process(st["headers"][k], st["L0"][k], st["feats"][k])
        for flag, nb in est.marg_log[n_marg:]:
            if flag == E.MARGIN_OLD:
                n_old += 1
            elif nb > 0:
                n_new += 1
        it, term, fc = est.solve_log[-1] if est.solve_log else (0, 0, 0.0)
        log.append((int(kf), est.stage_flag, int(est.static_status), it, term, fc, n_old, n_new, len(est.feature)))
        if on_frame:
            on_frame(k, est)
    return est, log
