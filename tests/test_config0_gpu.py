"""BASELINE configs[0] end to end ("KITTI seq 00 HDL-64 scan 0-100, aloam_velodyne_HDL_64 + kitti_estimator CPU reference"): 101
full-resolution S1 scans through BOTH halves of the path on the GPU -- scanRegistration -> laserOdometry -> laserMapping (the
/aft_mapped_to_init topic the Estimator consumes, kitti_config_00.yaml:7-8) -> the Estimator frame loop fed with those LiDAR poses
and a tracker stream rendered along the same ground-truth trajectory -- against the same chain on the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from tests import estimator_stream as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "lmono_amd", "host", "estimator_seq")
pytestmark = pytest.mark.gpu


def _pose44(q_xyzw, t):
    from oracle import ba_numpy as B
    T = np.eye(4)
    T[:3, :3] = B.q_to_R(np.asarray(q_xyzw, np.float64) / np.linalg.norm(q_xyzw)); T[:3, 3] = t
    return T


def test_101_scans_through_both_halves(oracle, gpu_ctx, tmp_path):
    import torch
    import lmono_amd
    from lmono_amd import trajectory
    from workloads import s1 as S1, s2
    n = 101
    world = S1.S1World(n_az=2000)
    traj = world.trajectory(n)
    xyzi, off = world.scans(traj)
    # ---- LiDAR half on the GPU: scanRegistration + laserOdometry (the reference's sequential schedule) + laserMapping
    xd = torch.from_numpy(xyzi).cuda()
    batch = lmono_amd.ScanBatch(gpu_ctx, n, len(xyzi))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    _, odo = batch.odometry(1, 0)
    mapper = lmono_amd.Mapper(gpu_ctx)
    mapped = np.zeros((n, 7))
    for k in range(n):
        q, t, _ = mapper.process(batch, k, odo[k, :4], odo[k, 4:])
        mapped[k, :4] = q; mapped[k, 4:] = t
    # ---- the same half on the CPU oracle
    ref = oracle.run_sequence(xyzi, off)
    assert np.abs(odo - ref["poses"]).max() < 1e-7
    ref_map = oracle.run_mapping(xyzi, off, ref["poses"])
    assert np.abs(mapped - ref_map["poses"]).max() < 1e-6
    gt = oracle.gt_relative(traj)
    ate_odo, ate_map = trajectory.ate(odo, gt), trajectory.ate(mapped, gt)
    assert ate_map < ate_odo and ate_map < 0.15                    # the mapped poses are the better LiDAR odometry
    # ---- Estimator half: tracker stream along the ground-truth path, LiDAR odometry = the GPU's mapped poses
    gt_R = np.array([_pose44(g[:4], g[4:])[:3, :3] for g in gt]); gt_P = gt[:, 4:]
    L0 = np.array([_pose44(m[:4], m[4:]) for m in mapped])
    st = s2.make_stream(n, seed=5, lidar_gt=(gt_R, gt_P), lidar_meas=L0)
    est, log = S.replay_oracle(st)
    fx = tmp_path / "config0.bin"
    s2.write_stream(fx, st)
    out = subprocess.run([EXE, str(fx), str(tmp_path / "new_odometry.txt")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    odo_e = np.array([[float(v) for v in ln.split()[1:]] for ln in out.stdout.splitlines() if ln.startswith("ODO")])
    ref_e = np.array(est.trajectory)
    assert odo_e.shape == ref_e.shape == (n - 10, 8)
    d = np.abs(odo_e[:, 1:4] - ref_e[:, 1:4]).max()
    costs = np.array([r[5] for r in log[10:]])
    frm = [ln.split()[1:] for ln in out.stdout.splitlines() if ln.startswith("FRM")]
    assert len(frm) == n
    dump = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(dump):                       # diagnostic table (GPU / oracle per frame), read after the run
        with open(os.path.join(dump, "config0_frames.txt"), "w") as fh:
            for k, (row, r) in enumerate(zip(frm, log)):
                dp = np.abs(odo_e[k - 10, 1:4] - ref_e[k - 10, 1:4]).max() if k >= 10 else 0.0
                fh.write("%d kf %s/%d st %s/%d it %s/%d term %s/%d cost %.9g/%.9g marg %s,%s/%d,%d feat %s/%d dP %.2e\n" %
                         (k, row[1], r[0], row[3], r[2], row[4], r[3], row[5], r[4], float(row[6]), r[5], row[7], row[8], r[6], r[7], row[9], r[8], dp))
    print("configs[0]: LiDAR ATE odometry %.3f m -> mapped %.3f m; Estimator GPU vs oracle max |dP| %.2e m (max final cost %.3g)" % (ate_odo, ate_map, d, costs.max()))
    # Frame k starts from frame k - 1's result and a solve is 30 unconverged dogleg iterations, so the rounding-level difference of the two
    # implementations (1e-9 at the first window) grows along the sequence and jumps where a termination test falls on a knife edge
    # (parity is defined on converged states, not on traces); k_ba_solve accumulates with fp64 atomics, so the size of
    # the late difference also varies from run to run (4.8, 5.5, 6.0, 10.2 and 25.8 mm seen over five runs of the same binary).  Bars: 1e-6 m
    # over the first 40 windows, 20 cm overall (a fifth of the fused trajectory's own 1.2 m error against the truth), identical keyframe /
    # marginalisation decisions throughout, and the same distance from the truth as the CPU path (below).
    assert np.abs(odo_e[:40, 1:4] - ref_e[:40, 1:4]).max() < 1e-6
    assert d < 0.2
    for k, (row, r) in enumerate(zip(frm, log)):
        assert (int(row[1]), int(row[2]), int(row[3])) == (r[0], r[1], r[2]) and (int(row[7]), int(row[8])) == (r[6], r[7]), "frame %d" % k
    # the fused trajectory (camera-aligned Estimator world) follows the ground truth
    fused_err = np.linalg.norm(odo_e[:, 1:4] - st["gt_P"][10:], axis=1)
    fused_ref = np.linalg.norm(ref_e[:, 1:4] - st["gt_P"][10:], axis=1)
    assert fused_err.max() < 2.0 and abs(fused_err.max() - fused_ref.max()) < 0.1        # as far from the truth as the CPU path is
    # trajectory of record in the reference's file format (Estimator.cc:642-644)
    txt = np.loadtxt(str(tmp_path / "new_odometry.txt"))
    assert txt.shape == (n - 10, 8)
