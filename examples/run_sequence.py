#!/usr/bin/env python3
"""KITTI-layout sequence -> scanRegistration + laserOdometry on one MI355X -> trajectory file in the reference's
"loam_odometry" format (BASELINE configs[0] plumbing).

    python examples/run_sequence.py --sequence /data/kitti/sequences/00 --out traj.txt [--first 0 --count 101]
    python examples/run_sequence.py --synthetic 32 --out traj.txt        # writes an S1 sequence to a temp dir first

With --poses the ground-truth file (12 numbers per line) is read and the RMS translation difference is printed."""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_synthetic_sequence(n, root):
    from workloads import s1 as S1
    from lmono_amd import kitti_io as IO
    w = S1.S1World()
    traj = w.trajectory(n)
    xyzi, off = w.scans(traj)
    os.makedirs(os.path.join(root, "velodyne"), exist_ok=True)
    for k in range(n):
        IO.write_velodyne_bin(IO.velodyne_path(root, k), xyzi[off[k]:off[k + 1]])
    np.savetxt(os.path.join(root, "times.txt"), 0.1 * np.arange(n), fmt="%.6e")
    return root


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequence", help="directory with velodyne/%%06d.bin and times.txt")
    ap.add_argument("--synthetic", type=int, default=0, help="generate this many S1 scans instead of reading a sequence")
    ap.add_argument("--out", required=True)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=None)
    ap.add_argument("--chains", type=int, default=1, help="1 = strictly sequential (reference behaviour)")
    ap.add_argument("--lead", type=int, default=0)
    ap.add_argument("--poses", help="KITTI ground-truth poses file (optional)")
    args = ap.parse_args()
    import lmono_amd
    from lmono_amd import kitti_io as IO
    tmp = None
    if args.synthetic > 0:
        tmp = tempfile.TemporaryDirectory()
        args.sequence = write_synthetic_sequence(args.synthetic, tmp.name)
    if not args.sequence:
        ap.error("--sequence or --synthetic is required")
    xyzi, off, stamps = IO.load_scans(args.sequence, args.first, args.count)
    ctx = lmono_amd.Context(0)
    batch = lmono_amd.ScanBatch(ctx, len(stamps), len(xyzi))
    batch.scanreg_host(xyzi, off, 64, 5.0)
    incr, poses = batch.odometry(n_chains=args.chains, lead=args.lead)
    IO.write_trajectory(args.out, stamps, poses, loam_style=True)
    print("wrote %d poses to %s" % (len(stamps), args.out))
    if args.poses:
        gt = IO.read_kitti_poses(args.poses)[args.first:args.first + len(stamps)]
        rel = np.einsum("ij,njk->nik", np.linalg.inv(np.vstack([gt[0], [0, 0, 0, 1]]))[:3, :3], gt[:, :, 3:4] - gt[0][:, 3:4])[:, :, 0]
        print("RMS translation difference to ground truth (no alignment, camera-frame poses need the calibration): %.3f m"
              % float(np.sqrt(((rel - poses[:, 4:7]) ** 2).sum(1).mean())))


if __name__ == "__main__":
    main()
