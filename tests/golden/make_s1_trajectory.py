#!/usr/bin/env python3
"""Generator of tests/golden/s1_seq00_oracle.npz: the strictly sequential CPU oracle (n_chains = 1, lead = 0; A-LOAM's
laserOdometry warm-starts every scan pair from the previous increment) run ONCE over the whole configs[1] workload --
4541 synthetic S1 HDL-64 scans, exactly the sequence bench.py generates (S1World(n_az=2000).trajectory(4541), scan ids
from 0).  bench.py and tests/test_full_sequence_gpu.py compare the timed chain-sharded GPU run against it over ALL scans
(ATE and RPE), not over a prefix.

Contents: poses [4541,7] f64 (q xyzw, t; laserOdometry's q_w_curr / t_w_curr), incr [4541,7] f64 (q_last_curr, t_last_curr),
feat_counts [4541,4] i32 (sharp, less sharp, flat, less flat), n_points [4541] i32 (raw returns per scan: pins the
generator), meta.  Takes ~3 min on 8 cores (front end in parallel over scans, odometry sequential).

    python tests/golden/make_s1_trajectory.py [n_scans] [seq]

seq 1 = the HELD-OUT sequence tests/golden/s1_seq01_oracle.npz: another world (S1World(seed=777): other boxes and poles) and another
trajectory (S1World.trajectory_clover: three-leaf clover at 4 .. 10 m/s).  The chain schedule's parameters were never tuned on it;
tests/test_chain_validation_gpu.py runs every chain layout on both sequences.

seq 2 = the STRESS sequence tests/golden/s1_seq02_oracle.npz: S1World(seed=4242, clutter=True) -- 200 small boxes on top of the 40 + 60,
20 % stray returns at random ranges, eight moving cylinders, 15 % of the rings with a dropped azimuth sector -- along the figure-8.  Nothing
was tuned against it; it says whether the lead-in, the repair tolerance and the search's round budgets are properties of the tidy world.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from workloads import s1 as S1          # noqa: E402
from oracle import oracle as O          # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
    seq = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    S1.build(); O.build()
    w = S1.S1World(n_az=2000) if seq == 0 else (S1.S1World(seed=777, n_az=2000) if seq == 1 else S1.S1World(seed=4242, n_az=2000, clutter=True))
    traj = w.trajectory_clover(n) if seq == 1 else w.trajectory(n)
    t0 = time.time()
    xyzi, off = w.scans(traj)
    print("generated %d scans, %d points in %.1f s" % (n, off[-1], time.time() - t0), flush=True)
    t0 = time.time()
    ref = O.run_sequence(xyzi, off, n_chains=1, lead=0, threads=len(os.sched_getaffinity(0)))
    print("oracle: %.1f s (scanreg %.0f ms, odometry %.0f ms)" % (time.time() - t0, ref["stage_ms"][0], ref["stage_ms"][1]), flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), ("s1_seq%02d_oracle.npz" % seq) if n == 4541 else "s1_seq%02d_oracle_%d.npz" % (seq, n))
    np.savez_compressed(out, poses=ref["poses"], incr=ref["incr"], feat_counts=ref["feat_counts"],
                        n_points=np.diff(off).astype(np.int32),
                        meta=np.array(("S1World(seed=20240, n_az=2000, n_rings=64), trajectory(%d), n_lines=64, min_range=5.0, "
                                       "oracle n_chains=1 lead=0 kd-tree" if seq == 0 else
                                       "S1World(seed=777, n_az=2000, n_rings=64), trajectory_clover(%d), n_lines=64, min_range=5.0, "
                                       "oracle n_chains=1 lead=0 kd-tree" if seq == 1 else
                                       "S1World(seed=4242, n_az=2000, n_rings=64, clutter=True), trajectory(%d), n_lines=64, min_range=5.0, "
                                       "oracle n_chains=1 lead=0 kd-tree") % n))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
