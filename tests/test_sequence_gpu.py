"""BASELINE configs[0] plumbing on the GPU: a KITTI-layout sequence on disk -> host-buffer entry of the C ABI
(lmono_scanreg_batch_h) -> laserOdometry -> trajectory file in the reference's format, through the Python wrapper and
through the C++ host mirror (lmono_amd/host/run_sequence), both against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from lmono_amd import kitti_io as IO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "lmono_amd", "host")


@pytest.fixture(scope="module")
def sequence_on_disk(oracle, tmp_path_factory):
    root = tmp_path_factory.mktemp("kitti_seq")
    w = oracle.S1World(n_az=500)
    xyzi, off = w.scans(w.trajectory(6))
    os.makedirs(root / "velodyne")
    for k in range(6):
        IO.write_velodyne_bin(IO.velodyne_path(str(root), k), xyzi[off[k]:off[k + 1]])
    np.savetxt(root / "times.txt", 100.0 + 0.1 * np.arange(6), fmt="%.6f")
    ref = oracle.run_sequence(xyzi, off, threads=1)
    return str(root), xyzi, off, ref


def test_host_buffer_entry_matches_device_pointer_entry_and_oracle(sequence_on_disk, gpu_ctx):
    import torch
    import lmono_amd
    root, xyzi, off, ref = sequence_on_disk
    x2, o2, stamps = IO.load_scans(root)
    assert np.array_equal(x2, xyzi) and np.array_equal(o2, off)
    b1 = lmono_amd.ScanBatch(gpu_ctx, 6, len(xyzi))
    b1.scanreg_host(x2, o2, 64, 5.0)
    _, p1 = b1.odometry(n_chains=1, lead=0)
    xd = torch.from_numpy(xyzi).cuda()
    b2 = lmono_amd.ScanBatch(gpu_ctx, 6, len(xyzi))
    b2.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    _, p2 = b2.odometry(n_chains=1, lead=0)
    assert np.array_equal(p1, p2)                       # same kernels, same data: bit-identical
    assert np.abs(p1 - ref["poses"]).max() < 1e-6       # and the CPU oracle's trajectory


def test_cpp_run_sequence_writes_the_reference_format(sequence_on_disk, tmp_path):
    root, xyzi, off, ref = sequence_on_disk
    subprocess.check_call(["make", "-s", "-C", HOST, "run_sequence"])
    out = tmp_path / "loam_odometry.txt"
    txt = subprocess.check_output([os.path.join(HOST, "run_sequence"), root, str(out)], text=True)
    assert txt.startswith("DONE 6 scans")
    lines = out.read_text().split("\n")
    assert len(lines) == 7 and all(line.endswith(" ") for line in lines[:6])      # Estimator.cc:270: blank before \n
    traj = IO.read_trajectory(str(out))
    assert np.allclose(traj[:, 0], 100.0 + 0.1 * np.arange(6))
    # six printed decimals of the oracle's poses (t then q)
    assert np.abs(traj[:, 1:4] - ref["poses"][:, 4:7]).max() < 2e-6 and np.abs(traj[:, 4:8] - ref["poses"][:, 0:4]).max() < 2e-6


def test_cpp_laser_mapping_matches_the_oracle_trajectory(oracle, sequence_on_disk, tmp_path):
    """Host mirror LaserMapping (device-resident cube map behind lmono_mapper_*) over the sequence on disk against
    oracle.run_mapping."""
    root, xyzi, off, ref = sequence_on_disk
    subprocess.check_call(["make", "-s", "-C", HOST, "run_sequence"])
    out = tmp_path / "loam_odometry.txt"; mapped = tmp_path / "aft_mapped.txt"
    txt = subprocess.check_output([os.path.join(HOST, "run_sequence"), root, str(out), "0", "-1", "1", "0", str(mapped)], text=True)
    want = oracle.run_mapping(xyzi, off, ref["poses"])
    lines = [line.split() for line in txt.split("\n") if line.startswith("MAP ")]
    assert len(lines) == 6
    for k, line in enumerate(lines):
        st = want["stats"][k]
        assert [int(v) for v in (line[3], line[4], line[6], line[7], line[9], line[10])] == \
            [st.n_edge[0], st.n_edge[1], st.n_plane[0], st.n_plane[1], st.lm_iters[0], st.lm_iters[1]], k
    assert want["stats"][-1].n_plane[1] > 500
    traj = IO.read_trajectory(str(mapped))
    assert np.abs(traj[:, 1:4] - want["poses"][:, 4:7]).max() < 3e-6 and np.abs(traj[:, 4:8] - want["poses"][:, 0:4]).max() < 3e-6
    # and the refinement did something: it differs from the odometry trajectory
    odo = IO.read_trajectory(str(out))
    assert np.abs(traj[:, 1:4] - odo[:, 1:4]).max() > 1e-4


def test_cpp_online_nodes_write_the_same_trajectories(sequence_on_disk, tmp_path):
    """run_sequence with n_chains = 0: LaserOdometryNode::laserCloudHandler per scan (lmono_odom_step) and laserMapping behind it per scan,
    against the batch form of the same program: the same files, byte for byte."""
    root, xyzi, off, ref = sequence_on_disk
    subprocess.check_call(["make", "-s", "-C", HOST, "run_sequence"])
    exe = os.path.join(HOST, "run_sequence")
    a, am, b, bm = (tmp_path / n for n in ("odo_batch.txt", "map_batch.txt", "odo_online.txt", "map_online.txt"))
    subprocess.check_output([exe, root, str(a), "0", "-1", "1", "0", str(am)], text=True)
    txt = subprocess.check_output([exe, root, str(b), "0", "-1", "0", "0", str(bm)], text=True)
    assert a.read_text() == b.read_text() and am.read_text() == bm.read_text()
    lat = [float(line.split()[2]) for line in txt.split("\n") if line.startswith("LAT ")]
    assert len(lat) == 6 and all(v > 0 for v in lat)


def test_bench_reads_a_kitti_directory(sequence_on_disk):
    """bench.py --kitti-dir (VERDICT r2 Missing #5): the headline workload over a KITTI-layout sequence on disk instead of generated scans."""
    import json
    import sys
    root, xyzi, off, ref = sequence_on_disk
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--kitti-dir", root, "--steps", "2", "--warmup", "1", "--chains", "2", "--lead", "1",
                          "--no-extras", "--cpu-sample", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["config"]["scans_total"] == 6 and d["data"].startswith("KITTI-layout sequence") and d["value"] > 0
    assert d["config"]["points_per_scan"] == round(len(xyzi) / 6) and d["roofline"]["frac"] > 0 and d["boundary_validation"]["unresolved"] == 0
