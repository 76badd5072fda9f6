"""Triangulation / depth refinement / outlier score / depth shift: oracle vs numpy (CPU) and HIP vs oracle (GPU)."""
import numpy as np
import pytest

from tests import ba_cases as K


def _win(seed, noise=0.5):
    w = K.make_window(seed, pix_sigma=noise, perturb=False)
    return dict(Rs=w["gt_Rs"], Ps=w["gt_Ps"], tlc=w["tlc"], trk_start=w["trk_start"], trk_off=w["trk_off"], trk_pts=w["trk_pts"],
                true=w["trk_true_depth"])


def test_linear_triangulation_matches_numpy_svd(oracle):
    w = _win(0)
    n = len(w["trk_start"])
    d0, _, _ = oracle.triangulate(w["Rs"], w["Ps"], w["tlc"], w["trk_start"], w["trk_off"], w["trk_pts"], -np.ones(n), refine_iters=-1)
    Rlc, tlc = w["tlc"][:3, :3], w["tlc"][:3, 3]
    for f in range(0, n, 7):
        i = w["trk_start"][f]; pts = w["trk_pts"][w["trk_off"][f]:w["trk_off"][f + 1]]
        R0 = w["Rs"][i] @ Rlc; t0 = w["Ps"][i] + w["Rs"][i] @ tlc
        rows = []
        for o, (u, v) in enumerate(pts):
            j = i + o
            R1 = w["Rs"][j] @ Rlc; t1 = w["Ps"][j] + w["Rs"][j] @ tlc
            t = R0.T @ (t1 - t0); R = R0.T @ R1
            P = np.concatenate([R.T, (-R.T @ t)[:, None]], 1)
            fv = np.array([u, v, 1.0]); fv /= np.linalg.norm(fv)
            rows += [fv[0] * P[2] - fv[2] * P[0], fv[1] * P[2] - fv[2] * P[1]]
        V = np.linalg.svd(np.array(rows))[2][-1]
        z = V[2] / V[3]
        ref = -1.0 if z < 0.1 else z
        assert abs(d0[f] - ref) < 1e-7 * max(1.0, abs(ref))


def test_triangulation_recovers_depth(oracle):
    w = _win(1, noise=0.3)
    n = len(w["trk_start"])
    d0, d1, flag = oracle.triangulate(w["Rs"], w["Ps"], w["tlc"], w["trk_start"], w["trk_off"], w["trk_pts"], -np.ones(n))
    ok = (flag == 1) & (w["true"] < 60)
    assert ok.sum() > 50
    e0 = np.abs(d0[ok] - w["true"][ok]) / w["true"][ok]; e1 = np.abs(d1[ok] - w["true"][ok]) / w["true"][ok]
    assert np.median(e0) < 0.05 and np.median(e1) < 0.15      # the robust refinement need not beat the linear estimate
    # exact data: zero pixel noise -> exact depths
    w = _win(2, noise=0.0)
    n = len(w["trk_start"])
    d0, d1, flag = oracle.triangulate(w["Rs"], w["Ps"], w["tlc"], w["trk_start"], w["trk_off"], w["trk_pts"], -np.ones(n))
    good = w["true"] < 80
    assert np.abs(d0[good] - w["true"][good]).max() < 1e-6 * 80 and np.abs(d1[good] - w["true"][good]).max() < 1e-5 * 80
    sc = oracle.outlier_scores(w["Rs"], w["Ps"], w["tlc"], w["trk_start"], w["trk_off"], w["trk_pts"], w["true"])
    assert np.abs(sc).max() < 1e-6        # reprojection error of the true depth on exact data


def test_shift_depth_known_answer(oracle):
    w = _win(3, noise=0.0)
    f = np.nonzero(w["trk_start"] == 0)[0]
    pt_i = np.array([w["trk_pts"][w["trk_off"][k]] for k in f])
    out = oracle.shift_depth(w["Rs"][0], w["Ps"][0], w["Rs"][1], w["Ps"][1], w["tlc"], pt_i, w["true"][f])
    # the landmark's depth in the next frame can be computed directly from the second observation's geometry
    Rlc, tlc = w["tlc"][:3, :3], w["tlc"][:3, 3]
    for n, k in enumerate(f[:20]):
        X = (w["Rs"][0] @ Rlc) @ (np.array([pt_i[n][0], pt_i[n][1], 1.0]) * w["true"][k]) + w["Ps"][0] + w["Rs"][0] @ tlc
        z1 = ((w["Rs"][1] @ Rlc).T @ (X - (w["Ps"][1] + w["Rs"][1] @ tlc)))[2]
        assert abs(out[n] - z1) < 1e-9
    assert (oracle.shift_depth(w["Rs"][0], w["Ps"][0], w["Rs"][1], w["Ps"][1], w["tlc"], pt_i[:1], [-5.0]) == -1.0).all()


@pytest.mark.gpu
def test_gpu_feature_kernels_match_oracle(oracle, gpu_ctx):
    wins = [_win(s) for s in (4, 5, 6)]
    F = sum(len(w["trk_start"]) for w in wins)
    d_gpu0, _ = gpu_ctx.triangulate(wins, -np.ones(F), refine_iters=-1)
    d_gpu, flag_gpu = gpu_ctx.triangulate(wins, -np.ones(F))
    sc_gpu = gpu_ctx.outlier_scores(wins, d_gpu)
    o = 0
    for w in wins:
        n = len(w["trk_start"])
        d0, d1, flag = oracle.triangulate(w["Rs"], w["Ps"], w["tlc"], w["trk_start"], w["trk_off"], w["trk_pts"], -np.ones(n))
        assert np.abs(d_gpu0[o:o + n] - d0).max() < 1e-9 * np.abs(d0).max()
        assert np.array_equal(flag_gpu[o:o + n], flag)
        assert np.abs(1.0 / d_gpu[o:o + n] - 1.0 / d1).max() < 1e-9      # inverse depths (the optimised quantity)
        sc = oracle.outlier_scores(w["Rs"], w["Ps"], w["tlc"], w["trk_start"], w["trk_off"], w["trk_pts"], d1)
        assert np.abs(sc_gpu[o:o + n] - sc).max() < 1e-6 * (np.abs(sc).max() + 1)
        o += n
    w = wins[0]
    f = np.nonzero(w["trk_start"] == 0)[0]
    pt_i = np.array([w["trk_pts"][w["trk_off"][k]] for k in f])
    a = gpu_ctx.shift_depth(w["Rs"][0], w["Ps"][0], w["Rs"][1], w["Ps"][1], w["tlc"], pt_i, w["true"][f])
    b = oracle.shift_depth(w["Rs"][0], w["Ps"][0], w["Rs"][1], w["Ps"][1], w["tlc"], pt_i, w["true"][f])
    assert np.abs(a - b).max() < 1e-12 * np.abs(b).max()
    # the batched entry (EstimatorBatch: one call for N windows) gives every window the single call's bytes, an empty window in the middle included
    frames, pts, deps, want = [], [], [], []
    for k, w in enumerate(wins):
        f = np.nonzero(w["trk_start"] == 0)[0] if k != 1 else np.zeros(0, int)
        pt_i = np.array([w["trk_pts"][w["trk_off"][j]] for j in f]).reshape(-1, 2)
        frames.append(np.concatenate([np.ravel(w["Rs"][0]), np.ravel(w["Ps"][0]), np.ravel(w["Rs"][1]), np.ravel(w["Ps"][1]), np.ravel(w["tlc"])]))
        pts.append(pt_i); deps.append(w["true"][f])
        want.append(gpu_ctx.shift_depth(w["Rs"][0], w["Ps"][0], w["Rs"][1], w["Ps"][1], w["tlc"], pt_i, w["true"][f]) if len(f) else np.zeros(0))
    got = gpu_ctx.shift_depth_batch(frames, pts, deps)
    for g, wv in zip(got, want):
        assert g.tobytes() == wv.tobytes()


@pytest.mark.gpu
def test_gpu_feature_kernels_at_the_trackers_ceiling(oracle, gpu_ctx):
    """LMONO_BA_MAX_FEATURES = 1664 tracks per window (11 frames x the tracker's MAX_CNT = 150, FeatureTracker.cc:21): a window of ~1600 tracks goes
    through the track-per-thread refinement (k_depth_refine: above the 1024 threads of the item kernel) and matches the oracle; a larger one is refused."""
    import lmono_amd
    w0 = K.make_window(54, n_landmarks=40000, max_tracks=1000, min_dist=10, pix_sigma=0.5, perturb=False)
    w = dict(Rs=w0["gt_Rs"], Ps=w0["gt_Ps"], tlc=w0["tlc"], trk_start=w0["trk_start"], trk_off=w0["trk_off"], trk_pts=w0["trk_pts"], true=w0["trk_true_depth"])
    n = len(w["trk_start"])
    assert 1024 < n <= 1664
    d_gpu, flag_gpu = gpu_ctx.triangulate([w], -np.ones(n))
    d0, d1, flag = oracle.triangulate(w["Rs"], w["Ps"], w["tlc"], w["trk_start"], w["trk_off"], w["trk_pts"], -np.ones(n))
    assert np.array_equal(flag_gpu, flag)
    # inverse depths (the optimised quantity), relative: two tracks without a usable linear start run away to 1 / d ~ 6e7 and are flagged 2 on both sides
    assert (np.abs(1.0 / d_gpu - 1.0 / d1) <= 1e-9 * np.maximum(1.0, np.abs(1.0 / d1))).all()
    sc = oracle.outlier_scores(w["Rs"], w["Ps"], w["tlc"], w["trk_start"], w["trk_off"], w["trk_pts"], d1)
    assert np.abs(gpu_ctx.outlier_scores([w], d_gpu) - sc).max() < 1e-6 * (np.abs(sc).max() + 1)
    w1 = K.make_window(54, n_landmarks=60000, max_tracks=1100, min_dist=9, pix_sigma=0.5, perturb=False)
    big = dict(Rs=w1["gt_Rs"], Ps=w1["gt_Ps"], tlc=w1["tlc"], trk_start=w1["trk_start"], trk_off=w1["trk_off"], trk_pts=w1["trk_pts"])
    assert len(big["trk_start"]) > 1664
    with pytest.raises(lmono_amd.LmonoError):
        gpu_ctx.triangulate([big], -np.ones(len(big["trk_start"])))
