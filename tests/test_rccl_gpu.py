"""The RCCL path on the one GPU a test box has: a world-1 "nccl" process group in a fresh child
process, bench.py's MULTI-rank step (LMONO_BENCH_FORCE_COLLECTIVES=1: lmono_odom_shard_main_d, sharding.validate_rank_boundaries,
sharding.gather_bases and the reductions of time and tolerance on DEVICE tensors) and the pose graph's all-reduce of the normal
equations.  A sum / gather over one rank is the identity, so the poses must be the no-collective run's bytes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, tmp_path, name, force, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LMONO_BENCH_REHEARSE", None)
    if force:
        env["LMONO_BENCH_FORCE_COLLECTIVES"] = "1"
    else:
        env.pop("LMONO_BENCH_FORCE_COLLECTIVES", None)
    dump = str(tmp_path / (name + ".npz"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--dump-poses", dump] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    return line, np.load(dump)


@pytest.mark.gpu
def test_lidar_step_through_a_world_1_rccl_group_is_the_same_bytes(tmp_path):
    extra = ["--scans", "192", "--chains", "16", "--no-extras", "--cpu-sample", "0"]
    a, pa = _bench(extra, tmp_path, "plain", False, 29581)
    b, pb = _bench(extra, tmp_path, "rccl", True, 29583)
    assert a["config"]["parallelism"] == "no collective (1 rank)" and a["config"]["collective_ranks"] == 0
    assert b["config"]["collective_backend"] == "nccl" and b["config"]["collective_ranks"] == 1
    assert b["config"]["collective_lib"] and "librccl" in b["config"]["collective_lib"], b["config"]
    assert b["boundary_validation"]["rank_boundary_rounds"] >= 1 and b["boundary_validation"]["unresolved"] == 0
    assert pa["poses"].tobytes() == pb["poses"].tobytes() and pa["incr"].tobytes() == pb["incr"].tobytes()
    assert b["ate_vs_cpu_m"] == a["ate_vs_cpu_m"] and b["ate_vs_cpu_m"] < 1e-3
    print("world-1 RCCL step: poses byte-equal to the no-collective run over %d scans (%s), ATE vs CPU %.2e m"
          % (len(pa["poses"]), b["config"]["collective_lib"], b["ate_vs_cpu_m"]))


@pytest.mark.gpu
def test_pose_graph_all_reduce_through_rccl_is_the_same_bytes(tmp_path):
    extra = ["--workload", "posegraph", "--keyframes", "600"]
    a, pa = _bench(extra, tmp_path, "pg_plain", False, 29585)
    b, pb = _bench(extra, tmp_path, "pg_rccl", True, 29587)
    assert a["config"]["collective_backend"] is None
    assert b["config"]["collective_backend"] == "nccl" and "librccl" in (b["config"]["collective_lib"] or "")
    assert pa["poses"].tobytes() == pb["poses"].tobytes()
    assert b["config"]["rounds"] == a["config"]["rounds"]


@pytest.mark.gpu
def test_two_nccl_ranks_on_the_one_card(tmp_path):
    """the first multi-rank `nccl` initialisation, device binding and fp64 all-gather should not wait for the day of the first SCALE
    run.  Two ranks on the ONE card of the test box (LMONO_BENCH_SAME_DEVICE=1: both bind cuda:0, the collectives go through RCCL) run
    `bench.py --gpus 2 --scaling strong --scans 512`; the poses must be the bytes of the same two-rank run over gloo (LMONO_BENCH_REHEARSE=1).  RCCL, like
    NCCL, may refuse two ranks on one device ("Duplicate GPU detected"): then the library's own message is recorded (gpurun_out/) and the
    test is skipped -- world 1 stays the widest RCCL group a one-card box can form."""
    def run(env_extra, name, port):
        dump = str(tmp_path / (name + ".npz"))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN", **env_extra)
        for k in ("LMONO_BENCH_REHEARSE", "LMONO_BENCH_SAME_DEVICE", "LMONO_BENCH_FORCE_COLLECTIVES"):
            if k not in env_extra:
                env.pop(k, None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--scans", "512", "--chains", "32", "--steps", "2", "--warmup", "1",
               "--no-extras", "--cpu-sample", "0", "--dump-poses", dump]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        return out, dump
    ref, ref_dump = run({"LMONO_BENCH_REHEARSE": "1"}, "gloo", 29591)
    assert ref.returncode == 0, ref.stdout[-2000:] + ref.stderr[-3000:]
    out, dump = run({"LMONO_BENCH_SAME_DEVICE": "1"}, "rccl2", 29593)
    if out.returncode != 0:
        text = out.stdout + out.stderr
        why = [ln.strip() for ln in text.splitlines() if "uplicate GPU" in ln] or \
              [ln.strip() for ln in text.splitlines() if "ncclInvalidUsage" in ln or "invalid usage" in ln or "ncclUnhandled" in ln or "ncclSystemError" in ln]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rccl_two_ranks_one_card.txt"), "w") as fh:
            fh.write("two nccl ranks on one device were refused:\n" + "\n".join(why[:12]) + "\n---- tail of the launcher's output\n" + text[-3000:])
        assert why, "two ranks on one card failed for a reason that is not the library refusing the placement:\n" + text[-3000:]
        pytest.skip("RCCL refuses two ranks on one device: " + why[0][:300])
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["collective_backend"] == "nccl" and d["config"]["collective_ranks"] == 2
    assert "librccl" in (d["config"]["collective_lib"] or "")
    a, b = np.load(ref_dump), np.load(dump)
    assert a["poses"].tobytes() == b["poses"].tobytes() and a["incr"].tobytes() == b["incr"].tobytes()
    with open(os.path.join(ROOT, "gpurun_out", "rccl_two_ranks_one_card.txt"), "w") as fh:
        fh.write("two nccl ranks on one device: accepted; rank 0's poses equal the gloo rehearsal's bytes over %d scans (%s)\n" % (len(b["poses"]), d["config"]["collective_lib"]))
    print("two RCCL ranks on one card: poses byte-equal to the gloo rehearsal (%s)" % d["config"]["collective_lib"])
