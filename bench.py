#!/usr/bin/env python3
"""bench.py -- headline benchmark: HDL-64 scans/sec through the MI355X LiDAR hot path.

Workload (BASELINE.json configs[1]): KITTI seq-00-shaped sequence, 4541 HDL-64 scans (synthetic S1 world,
~118 k returns/scan), scan registration + scan-to-scan laserOdometry only.  One "step" = one pass
over the whole resident sequence: lmono_scanreg_batch + lmono_odom_batch_d (+ the RCCL pose exchange when N > 1).
Inputs are resident in HBM before the timed region.  Weak scaling: every rank processes its own 4541-scan range of
one long trajectory; value = scans of all ranks * steps / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector / matrix peak
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def bench_ba(args):
    """Secondary workload (BASELINE configs[2] shape): batched Estimator::optimization() solves, S2 synthetic windows
    (KITTI-05 intrinsics / extrinsic / weights, <=150 tracks per frame).  One step = every window solved once from its
    initial state (<= 30 dogleg iterations).  Reported separately from the headline scans/s."""
    import torch
    import lmono_amd
    from workloads import s2 as K        # synthetic S2 windows (input plumbing)
    assert torch.cuda.is_available()
    ctx = lmono_amd.Context(0)
    base = [K.make_window(s) for s in range(16)]
    windows = [base[k % len(base)] for k in range(args.windows)]
    b = lmono_amd.BaBatch(ctx, windows)
    for _ in range(args.warmup):
        b.reset(); b.solve(30)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b.reset(); b.solve(30)
    ctx.synchronize()
    el = time.perf_counter() - t0
    poses, ex, invd, sm = b.read()
    # ---- cpu_baseline leg: the only place the oracle is touched
    from oracle import oracle as O
    t0 = time.time()
    ref = [O.ba_solve(w) for w in base]
    cpu_s = (time.time() - t0) / len(base)
    n_obs = float(np.mean([len(w["obs_feat"]) for w in base])); n_f = float(np.mean([len(w["inv_depth"]) for w in base]))
    iters = float(sm[:, 2].mean())
    # per iteration O x 2.0 kflop (residuals, Jacobians, J^T J) + 72^3 / 3 (Cholesky) fp64 flops
    flops = (2.0e3 * n_obs + 72.0 ** 3 / 3.0) * iters
    ms = el / args.steps * 1e3
    tflops = flops * args.windows / (ms * 1e-3) / 1e12
    out = {"metric": "BA window solves/sec (Estimator::optimization, S2 synthetic)", "value": round(args.windows * args.steps / el, 1),
           "unit": "windows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "S2 11-frame windows, %d independent windows" % args.windows, "mean_obs": n_obs, "mean_features": n_f,
                      "mean_iterations": iters},
           "roofline": {"bound": "mfma", "kernel": "k_ba_solve", "achieved": round(tflops, 3), "peak": FP64_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tflops / FP64_PEAK_TFLOPS, 5), "traffic": _ba_traffic(args.windows),
                        "note": "fp64; J^T J and the Schur complement run on v_mfma_f64_16x16x4_f64, the rest of an iteration "
                                "(72x72 Cholesky, triangular solves, reductions) is dependent-latency bound inside one workgroup per window"},
           "cpu_baseline": {"value": round(1.0 / cpu_s, 2), "unit": "windows/s", "cores": 1, "kind": "port",
                            "sample": "16 windows, oracle/lo_ba_solve.c (-O3), 1 thread"},
           "final_cost_rel_diff_vs_cpu": float(max(abs(sm[k, 1] - ref[k][3].final_cost) / ref[k][3].final_cost for k in range(min(len(base), args.windows))))}
    if out["roofline"]["traffic"]:       # the counters' HBM traffic over this run's launch time (the scratch of many windows is not L2-resident)
        gbps = out["roofline"]["traffic"] / (ms * 1e-3) / 1e9
        out["roofline"]["hbm_traffic_GBps"] = round(gbps, 1)
        out["roofline"]["hbm_frac"] = round(gbps / HBM_PEAK_GBS, 4)
    print(json.dumps(out), flush=True)


def _make_seq_stream(job):
    """(worker of bench_ba_seq_streams' process pool: one S2 frame stream written to a file; no GPU in here)"""
    from workloads import s2 as K
    n, seed, path = job
    st = K.make_stream(n, seed=seed, stops=(400, 401, 1500) if n > 1500 else ())
    K.write_stream(path, st)
    return path


def bench_ba_seq_streams(args):
    """N independent sequences through the host mirror's EstimatorBatch (the BA loop is parallel only across independent
    sequences): N Estimators stepped in lock-step, every numeric step of a frame one batched C-ABI call over the N windows (lmono_triangulate,
    lmono_ba_batch_update / _solve / _read, lmono_marginalize + lmono_marg_second_new, lmono_outlier_scores, lmono_shift_depth_batch).  Stream s
    replays file s mod 8 of eight different S2 streams; every stream's output digest must equal the digest of the single-stream run of its file."""
    import subprocess
    import tempfile
    from concurrent.futures import ProcessPoolExecutor
    n, N = args.frames_seq, args.seq_streams
    n_files = min(N, 8)
    d = tempfile.mkdtemp()
    t0 = time.time()
    jobs = [(n, 2 + k, os.path.join(d, "stream%d.bin" % k)) for k in range(n_files)]
    with ProcessPoolExecutor(max_workers=min(n_files, os.cpu_count() or 1)) as ex:
        files = list(ex.map(_make_seq_stream, jobs))
    gen_s = time.time() - t0
    exe = os.path.join(ROOT, "lmono_amd", "host", "estimator_seq")
    runs = {}
    for mode in ("sync", "async"):
        t0 = time.perf_counter()
        out = subprocess.run([exe, files[0], "-", mode, "streams=%d" % N, "groups=%d" % args.seq_groups, "digest"] + files[1:], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if out.returncode != 0:
            raise RuntimeError("estimator_seq streams=%d failed: %s" % (N, out.stderr[-1000:]))
        for ln in out.stderr.splitlines():          # LMONO_HOST_TIMING=1: the lock-step frame's phase clocks
            if ln.startswith("BATCHTIM"):
                print(mode, ln, file=sys.stderr)
        lines = out.stdout.splitlines()
        tim = [ln for ln in lines if ln.startswith("TIM")][0].split()
        flp = [ln for ln in lines if ln.startswith("FLP")][0].split()
        runs[mode] = {"dig": {int(ln.split()[1]): ln.split()[2] for ln in lines if ln.startswith("DIG")}, "n_inited": int(tim[1]), "ms_step": float(tim[2]),
                      "wall": wall, "flops": float(flp[1]), "obs": int(flp[2])}
    # every stream against the single-stream run of its file
    verified, single_ms = 0, []
    same = runs["sync"]["dig"] == runs["async"]["dig"]
    for k in range(n_files if not args.no_extras else min(n_files, 1)):
        out = subprocess.run([exe, files[k], "-", "async"], capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError("estimator_seq failed: " + out.stderr[-1000:])
        dig = [ln for ln in out.stdout.splitlines() if ln.startswith("DIG")][0].split()[2]
        single_ms.append(float([ln for ln in out.stdout.splitlines() if ln.startswith("TIM")][0].split()[2]))
        for s in range(k, N, n_files):
            same = same and runs["async"]["dig"][s] == dig
        verified += 1
    r = runs["async"]
    fps = N * 1e3 / r["ms_step"]
    tfl = r["flops"] / (r["ms_step"] * 1e-3 * r["n_inited"]) / 1e12
    res = {"metric": "Estimator frames/sec over N independent sequences in lock-step (sliding-window BA frame loop, S2 synthetic streams)", "value": round(fps, 1),
           "unit": "frames/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": round(r["ms_step"] * r["n_inited"], 1), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "S2 frame streams, KITTI-05 shape (configs[2]): %d streams x %d frames (%d different stream files), <= 150 tracks / frame, "
                                  "EstimatorBatch: one batched C-ABI call per numeric step" % (N, n, n_files),
                      "streams": N, "groups": args.seq_groups, "frames": n, "inited_frames_per_stream": r["n_inited"], "ms_per_lockstep_frame": round(r["ms_step"], 3), "wall_s": round(r["wall"], 1),
                      "gen_s": round(gen_s, 1), "marginalisation": "overlapped with the next frame (second context, own HIP stream, host thread)",
                      "inline_marginalisation": {"ms_per_lockstep_frame": round(runs["sync"]["ms_step"], 3), "frames_per_s": round(N * 1e3 / runs["sync"]["ms_step"], 1)},
                      "every_stream_equals_its_single_stream_run": bool(same), "files_verified": verified,
                      "single_stream_frames_per_s": (round(1e3 / float(np.mean(single_ms)), 1) if single_ms else None)},
           "roofline": {"bound": "mfma", "kernel": "k_ba_solve (%d windows per launch)" % N, "achieved": round(tfl, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tfl / FP64_PEAK_TFLOPS, 6), "traffic": None, "algorithmic_flops": r["flops"], "projection_blocks": r["obs"],
                        "note": "algorithmic flops of all window solves (SURVEY 8d: iterations x (2000 per projection block + 72^3 / 3)) over the WHOLE lock-step frame time "
                                "(host halves, uploads and read-backs included)"}}
    print(json.dumps(res), flush=True)


def bench_ba_seq(args):
    if args.seq_streams > 1:
        return bench_ba_seq_streams(args)
    return bench_ba_seq_single(args)


def bench_ba_seq_single(args):
    """BASELINE configs[2] as a SEQUENCE: KITTI-05-shaped S2 frame stream (2761 frames: LiDAR odometry pose +
    tracker output per frame) through the host mirror's Estimator::processImage loop (lmono_amd/host/estimator_seq: featureCheck ->
    triangulate -> optimization + margin -> outliersRejection -> slideWindow, every numeric step on the GPU through the C ABI), one
    frame after the other as the reference runs (batch 1: window k depends on window k - 1).  Reports frames/s of the INITED
    frames, the fused trajectory's ATE against ground truth and against the CPU oracle's replay of a bounded prefix."""
    import subprocess
    import tempfile
    from lmono_amd import trajectory
    from workloads import s2 as K
    n = args.frames_seq
    t0 = time.time()
    st = K.make_stream(n, seed=2, stops=(400, 401, 1500) if n > 1500 else ())
    gen_s = time.time() - t0
    d = tempfile.mkdtemp()
    fx = os.path.join(d, "stream.bin")
    K.write_stream(fx, st)
    exe = os.path.join(ROOT, "lmono_amd", "host", "estimator_seq")
    # two replays: marginalisation inline (the reference's order) and overlapped with the next frame on a second context / stream /
    # host thread (Estimator::setAsyncMargin -- the timed configuration); every printed quantity must be the same bytes
    runs = {}
    for mode in ("sync", "async"):
        t0 = time.perf_counter()
        out = subprocess.run([exe, fx, "-", mode], capture_output=True, text=True)  # the child owns the GPU; this process never touches it
        wall = time.perf_counter() - t0
        if out.returncode != 0:
            raise RuntimeError("estimator_seq failed: " + out.stderr[-1000:])
        for ln in out.stderr.splitlines():          # LMONO_HOST_TIMING=1: the mirror's phase clocks
            if ln.startswith("HOSTTIM"):
                print(mode, ln, file=sys.stderr)
        lines = out.stdout.splitlines()
        tim = [ln for ln in lines if ln.startswith("TIM")][0].split()
        flp = [ln for ln in lines if ln.startswith("FLP")]
        runs[mode] = {"lines": [ln for ln in lines if not ln.startswith(("TIM", "FLP"))], "n_inited": int(tim[1]), "ms_frame": float(tim[2]), "wall": wall,
                      "flops": float(flp[0].split()[1]) if flp else None, "obs": int(flp[0].split()[2]) if flp else None}
    identical = runs["sync"]["lines"] == runs["async"]["lines"]
    odo = np.array([[float(v) for v in ln.split()[1:]] for ln in runs["async"]["lines"] if ln.startswith("ODO")])
    n_inited, ms_frame, wall = runs["async"]["n_inited"], runs["async"]["ms_frame"], runs["async"]["wall"]
    gt = np.concatenate([np.zeros((len(odo), 4)), st["gt_P"][10:10 + len(odo)]], 1)
    est = np.concatenate([np.zeros((len(odo), 4)), odo[:, 1:4]], 1)
    # ---- cpu_baseline leg: the only place the oracle is touched
    from oracle import estimator_ref as E
    m = n if args.cpu_frames < 0 else min(args.cpu_frames, n)            # default: the whole stream (2761 frames: ~20 s on one core)
    res = {"metric": "Estimator frames/sec (sliding-window BA frame loop, S2 synthetic stream)", "value": round(1e3 / ms_frame, 2), "unit": "frames/s",
           "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": round(ms_frame * n_inited, 1), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "S2 frame stream, KITTI-05 shape (configs[2]): %d frames, <= 150 tracks / frame, batch 1 (one sequence)" % n,
                      "frames": n, "inited_frames": n_inited, "ms_per_frame": round(ms_frame, 3), "wall_s": round(wall, 1), "gen_s": round(gen_s, 1),
                      "marginalisation": "overlapped with the next frame (second context, own HIP stream, host thread)",
                      "inline_marginalisation": {"ms_per_frame": round(runs["sync"]["ms_frame"], 3), "frames_per_s": round(1e3 / runs["sync"]["ms_frame"], 2),
                                                 "output_identical_to_overlapped": identical},
                      "note": "a frame = processImage of the C++ mirror: triangulate + <= 30 dogleg iterations + marginalisation + outlier "
                              "rejection + window slide; each numeric step is one C-ABI call with its own upload / download (PCIe-inclusive; scratch from the context arena)"},
           "roofline": {"bound": "mfma", "kernel": "k_ba_solve (one window per launch: up to 8 of 256 CUs busy)",
                        "achieved": (round(runs["async"]["flops"] / (ms_frame * 1e-3 * n_inited) / 1e12, 5) if runs["async"]["flops"] else None), "peak": FP64_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": (round(runs["async"]["flops"] / (ms_frame * 1e-3 * n_inited) / 1e12 / FP64_PEAK_TFLOPS, 7) if runs["async"]["flops"] else None),
                        "traffic": None, "algorithmic_flops": runs["async"]["flops"], "projection_blocks": runs["async"]["obs"],
                        "note": "algorithmic flops of all window solves (SURVEY 8d: iterations x (2000 per projection block + 72^3 / 3)) over the WHOLE frame time; "
                                "batch 1 is latency bound by construction: the reference's own operating point; the batched rate is the `ba` workload"},
           "ate_vs_truth_m": round(trajectory.ate(est, gt), 4)}
    if m > 20:
        sub = {k: (v[:m] if k != "tlc" else v) for k, v in st.items()}
        t0 = time.time()
        ref_est = E.EstimatorRef(sub["tlc"])
        for k in range(m):
            ref_est.process(sub["headers"][k], sub["L0"][k], sub["feats"][k])
        cpu_s = time.time() - t0
        ref = np.array(ref_est.trajectory)
        res["cpu_baseline_python_replay"] = {"value": round((m - 10) / cpu_s, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                             "sample": "first %d frames of the same stream, oracle/estimator_ref.py over the C oracle (-O3), 1 thread" % m}
        # the baseline proper: the SAME C++ frame loop (lmono_amd/host/lmono_host.cpp) linked against the C oracle instead of the HIP library
        # (oracle/estimator_seq_cpu, oracle/cpu_shim.cpp: test infrastructure) -- no interpreter time in it
        cexe = os.path.join(ROOT, "oracle", "estimator_seq_cpu")
        if not os.path.exists(cexe):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "estimator_seq_cpu"], check=False)
        fxm = fx
        if m < n:
            fxm = os.path.join(d, "stream_prefix.bin")
            K.write_stream(fxm, sub)
        if os.path.exists(cexe):
            cout = subprocess.run([cexe, fxm, "-"], capture_output=True, text=True)
            ctim = [ln for ln in cout.stdout.splitlines() if ln.startswith("TIM")]
            if cout.returncode == 0 and ctim:
                c_inited, c_ms = int(ctim[0].split()[1]), float(ctim[0].split()[2])
                codo = np.array([[float(v) for v in ln.split()[1:]] for ln in cout.stdout.splitlines() if ln.startswith("ODO")])
                res["cpu_baseline"] = {"value": round(1e3 / c_ms, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                       "sample": "first %d frames of the same stream (%d INITED), the C++ host mirror over the C oracle (oracle/estimator_seq_cpu, -O3), 1 thread" % (m, c_inited)}
                kk = min(len(codo), len(odo))
                res["max_pos_diff_vs_cpu_cxx_m"] = float(np.abs(odo[:kk, 1:4] - codo[:kk, 1:4]).max())
        if "cpu_baseline" not in res:
            res["cpu_baseline"] = res["cpu_baseline_python_replay"]
        res["max_pos_diff_vs_cpu_m"] = float(np.abs(odo[:len(ref), 1:4] - ref[:, 1:4]).max())
        k = min(len(ref), len(odo))
        res["ate_vs_cpu_m"] = round(float(np.sqrt(np.mean(np.sum((odo[:k, 1:4] - ref[:k, 1:4]) ** 2, axis=1)))), 6)
        res["frames_compared_vs_cpu"] = k
    print(json.dumps(res), flush=True)


def _ba_traffic(windows):
    """HBM bytes of one k_ba_solve launch from the newest committed counter summary (scripts/profile_round6.sh ba: 1 and 1024 windows), or None."""
    for rnd in ("r6", "r5"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_k_ba_solve.json")
        try:
            with open(path) as fh:
                v = json.load(fh).get("hbm_bytes_per_launch", {}).get(str(windows))
            if v:
                return v
        except (OSError, ValueError, AttributeError):          # no summary, or an unreadable one: try the older one, else the line carries null
            pass
    return None


def _map_traffic(streams):
    """HBM bytes per single-stream laserMapping frame from the committed counter summary (scripts/profile_round5.sh), or None."""
    if streams != 1 or os.environ.get("LMONO_MAP_HOST_TABLES"):      # (the counters are the device-table frame's)
        return None
    for rnd in ("r6", "r5"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "pmc_map_frame.json")) as fh:
                v = json.load(fh).get("hbm_bytes_per_frame")
            if v:
                return v
        except (OSError, ValueError, AttributeError):
            pass
    return None


def bench_map(args):
    """Tertiary workload: laserMapping with the device-resident cube map over a synthetic S1 sequence, one
    stream.  One step = every scan of the sequence through lmono_mapper_process once (fresh map per step)."""
    import torch
    import lmono_amd
    from workloads import s1 as S1
    assert torch.cuda.is_available()
    n = min(args.scans, 256)
    w = S1.S1World(n_az=args.az)
    traj = w.trajectory(n)
    x, off = w.scans(traj)
    ctx = lmono_amd.Context(0)
    xd = torch.from_numpy(x).cuda()
    batch = lmono_amd.ScanBatch(ctx, n, len(x))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    _, odo = batch.odometry(n_chains=1, lead=0)

    B = max(1, args.streams)

    # B independent streams (here: the same sequence B times, every stream with its own map) advanced in lock-step
    mappers = [lmono_amd.Mapper(ctx) for _ in range(B)]
    batches = [batch] * B
    qs = [np.tile(odo[k, :4], (B, 1)) for k in range(n)]; ts = [np.tile(odo[k, 4:], (B, 1)) for k in range(n)]

    sizes = np.zeros((n, 2), np.int64)     # per frame: map points of the neighbourhood, points of the cubes the update rebuilt

    def run():
        for m in mappers:
            m.reset()                      # fresh map per step
        out = np.zeros((n, 7))
        for k in range(n):
            if B == 1:       # one stream: lmono_mapper_process (cube table on the device, one wait per frame)
                q1, t1, st1 = mappers[0].process(batch, k, odo[k, :4], odo[k, 4:])
                out[k, :4] = q1; out[k, 4:] = t1
                sizes[k] = st1[6:8]
                continue
            q, t, st = lmono_amd.Mapper.process_batch(ctx, mappers, batches, [k] * B, qs[k], ts[k])
            out[k, :4] = q[-1]; out[k, 4:] = t[-1]
            sizes[k] = st[-1][6:8]
        return out
    for _ in range(args.warmup):
        run()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = run()
    ctx.synchronize()
    el = time.perf_counter() - t0
    host_tables = None
    if B == 1 and not args.no_extras:       # the same stream through lmono_mapper_process_batch (cube table on the host, two waits per frame: round 4's frame), untimed extra
        hm = lmono_amd.Mapper(ctx)
        def run_host():
            hm.reset()
            last = None
            for k in range(n):
                q, t, _ = lmono_amd.Mapper.process_batch(ctx, [hm], [batch], [k], odo[k:k + 1, :4], odo[k:k + 1, 4:])
                last = np.concatenate([q[0], t[0]])
            return last
        run_host()
        ctx.synchronize()
        th = time.perf_counter()
        for _ in range(max(1, args.steps // 4)):
            last_h = run_host()
        ctx.synchronize()
        host_tables = {"frames_per_s": round(n * max(1, args.steps // 4) / (time.perf_counter() - th), 1), "last_pose_identical": bool(np.array_equal(last_h, got[-1]))}
    # ---- cpu_baseline leg: the only place the oracle is touched
    from oracle import oracle as O
    ref = O.run_mapping(x, off, odo)
    gt = O.gt_relative(traj)
    # frame-level algorithmic bytes: every cloud element a frame has to move once -- scan clouds into the filter (16 B), neighbourhood
    # gathered (16 B in + 16 B out), touched cubes rebuilt (16 B in + 16 B out); whole chain, not one kernel
    cnt = batch.counts()
    alg = 16.0 * (cnt[:n, 2].sum() + cnt[:n, 4].sum()) + 32.0 * sizes[:, 0].sum() + 32.0 * sizes[:, 1].sum()
    gbs = alg * B * args.steps / el / 1e9
    out = {"metric": "laserMapping frames/sec (scan-to-map refinement, device-resident cube maps, independent streams in lock-step)",
           "value": round(n * B * args.steps / el, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 clouds / f64 solve", "data": "synthetic",
           "config": {"workload": "S1 HDL-64 sequence, %d scans, laserMapping after laserOdometry (SURVEY 8f-1)" % n, "streams": B,
                      "table": ("device (lmono_mapper_process: one wait per frame)" if B == 1 else "host (lmono_mapper_process_batch: two waits per frame)"),
                      "same_stream_with_host_table": host_tables},
           "roofline": {"bound": "hbm", "kernel": "per-frame kernel chain (k_vox_* filter chain, k_grid_*, k_map_correspond, k_map_factor, k_map_solve)",
                        "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": _map_traffic(B),
                        "algorithmic_bytes_per_frame": round(alg / n),
                        "note": "whole-chain figure (bytes every cloud element must move once per frame / frame time), not one kernel's: a frame is a chain of ~35 short dependent launches (profiles/r5/map_kernel_stats_1stream.csv); one stream (lmono_mapper_process) keeps the cube table on the device and waits once per frame, several streams (lmono_mapper_process_batch) plan on the host and wait twice; latency of that chain, not a roofline measurement; traffic = FETCH_SIZE x 2 + WRITE_SIZE of all kernels of a single-stream frame (profiles/r5/pmc_map_frame.json)"},
           "cpu_baseline": {"value": round(n / (ref["stage_ms"][1] * 1e-3), 2), "unit": "frames/s", "cores": 1, "kind": "port",
                            "sample": "the same %d scans, oracle/lo_mapping.c (-O3), 1 thread, mapping stage only" % n},
           "max_pose_diff_vs_cpu": float(np.abs(got - ref["poses"]).max()),
           "ate_vs_truth_m": {"odometry": round(O.ate(odo, gt), 4), "mapped": round(O.ate(got, gt), 4)}}
    print(json.dumps(out), flush=True)


def bench_colour(args):
    """Workload of SURVEY config 5 / row 8f-3: colour projection (MapBuilder::associateToMap + depthFill + rgb_map
    accumulation) of S3 scans (32 rings x 1800 azimuth steps) into 1241 x 376 noise images, `--streams` independent map
    builders advanced in lock-step.  One step = `--frames` frames per stream; rgb_map is cleared every 10 frames as
    processMapping does.  Scans and images are resident in HBM before the timed region."""
    import torch
    import lmono_amd
    from workloads import s1 as S1
    assert torch.cuda.is_available()
    W, H = 1241, 376
    n_distinct = 16
    w = S1.S1World(n_rings=32, n_az=1800)
    traj = w.trajectory(n_distinct)
    x, off = w.scans(traj)
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(n_distinct)]
    ctx = lmono_amd.Context(0)
    cam = lmono_amd.Camera(W, H, 718.856, 718.856, 607.1928, 185.2157, 0, 0, 0, 0, 5, 0, 0)    # kitti00_cam.yaml + kitti_map_config_00.yaml
    M = lmono_amd.lidar_to_camera([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], [0.27, 0.0, -0.08])
    B = max(1, args.streams)
    F = max(1, args.frames)
    mbs = [lmono_amd.MapBuilder(ctx, cam, max_cloud_points=16) for _ in range(B)]
    dx = [torch.from_numpy(x[off[k]:off[k + 1]]).cuda() for k in range(n_distinct)]
    di = [torch.from_numpy(im).cuda() for im in imgs]
    npts = [int(off[k + 1] - off[k]) for k in range(n_distinct)]
    yaw = traj[:, 3]
    qs = np.stack([np.zeros(n_distinct), -np.sin(yaw / 2), np.zeros(n_distinct), np.cos(yaw / 2)], 1)     # camera y axis points down
    ts = np.stack([-traj[:, 1], np.zeros(n_distinct), traj[:, 0]], 1)
    sizes = np.zeros(n_distinct, np.int64)

    def run():
        for f in range(F):
            if f % 10 == 0:
                for m in mbs:
                    m.clear()
            ks = [(s + f) % n_distinct for s in range(B)]
            n = lmono_amd.MapBuilder.associate_batch(ctx, mbs, [dx[k].data_ptr() for k in ks], [npts[k] for k in ks], [M] * B,
                                                     [di[k].data_ptr() for k in ks], qs[ks], ts[ks])
            for s, k in enumerate(ks):
                sizes[k] = n[s]
    for _ in range(args.warmup):
        run()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    ctx.synchronize()
    el = time.perf_counter() - t0
    frames = B * F * args.steps
    # algorithmic bytes of one frame: scan in (16 B / point), image in (3 B / pixel), depth map out (1 B / pixel), camera-frame
    # and world-frame coloured clouds out (16 B / point each)
    ks_all = [(s + f) % n_distinct for f in range(F) for s in range(B)]
    alg = float(np.mean([16 * npts[k] + 4 * W * H + 32 * sizes[k] for k in ks_all]))
    # ---- cpu_baseline leg: the only place the oracle is touched
    from oracle import oracle as O
    oc = O.kitti00_cam()
    t1 = time.perf_counter()
    n_cpu = 0
    while n_cpu < n_distinct and (n_cpu < 4 or time.perf_counter() - t1 < 10.0):
        k = n_cpu
        d, a, b = O.associate_to_map(oc, x[off[k]:off[k + 1]], M, imgs[k], qs[k], ts[k])
        assert len(a) == sizes[k] or sizes[k] == 0, (k, len(a), sizes[k])
        n_cpu += 1
    cpu_s = time.perf_counter() - t1
    same = bool((mbs[0].depth() == O.associate_to_map(oc, x[off[(F - 1) % n_distinct]:off[(F - 1) % n_distinct + 1]], M, imgs[(F - 1) % n_distinct],
                                                      qs[(F - 1) % n_distinct], ts[(F - 1) % n_distinct])[0]).all())
    out = {"metric": "colour-projection frames/sec (associateToMap + depthFill + rgb_map accumulation, independent streams in lock-step)",
           "value": round(frames / el, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8 image ops / f64 projection", "data": "synthetic",
           "config": {"workload": "S3 VLP-32-shaped scans (32 x 1800) into 1241 x 376 RGB noise images, colour projection (SURVEY 8f-3, config 5)",
                      "streams": B, "frames_per_step": F, "mean_points_in": round(float(np.mean(npts)), 1), "mean_points_out": round(float(np.mean(sizes[sizes > 0])), 1)},
           "roofline": {"bound": "hbm", "kernel": "frame chain: k_colour_splat, k_depth_fill<5>, k_colour_count, k_colour_lift (wall clock incl. the host round trip per batch)",
                        "achieved": round(alg * frames / el / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg * frames / el / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic": None, "algorithmic_bytes_per_frame": round(alg, 1)},
           "cpu_baseline": {"value": round(n_cpu / cpu_s, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                            "sample": "%d of the same frames, oracle/lo_colour.c (-O3), 1 thread" % n_cpu},
           "depth_map_equals_cpu": same}
    print(json.dumps(out), flush=True)


def bench_posegraph(args):
    """Workload of SURVEY config 4 / row 8f-2 (a new feature: the reference has no pose graph): one S4 loop-closure graph of
    `--keyframes` keyframes (KITTI-00 has 4541 scans) optimised with 5 LM iterations per step.  With --gpus N every rank holds the
    graph, linearises the edges of its keyframe range, and ONE all-reduce per round sums the normal equations (RCCL); the
    factorisation is replicated, so N > 1 shards only the linearisation ("strong" scaling of one fixed graph)."""
    import torch
    import lmono_amd
    from lmono_amd import sharding
    from workloads import s4
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available() and world == args.gpus
    torch.cuda.set_device(local)
    multi = world > 1 or os.environ.get("LMONO_BENCH_FORCE_COLLECTIVES") == "1"      # one rank through the RCCL all-reduce (tests/test_rccl_gpu.py)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29573")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    g = s4.make_graph(n=args.keyframes, laps=2.2)
    ctx = lmono_amd.Context(local)
    pg = lmono_amd.PoseGraph(ctx, g["odom"], g["loops"], g["loop_info"])
    buf = torch.zeros(pg.reduce_count, dtype=torch.float64, device="cuda:%d" % local)
    pg.use_reduce_tensor(buf)

    def run():
        pg.reset()
        return sharding.pose_graph_rounds(pg, rank, world, max_iter=5, force_collective=multi)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rounds = run()
    fence()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda:%d" % local)
    if multi:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    out, st = pg.result()
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return
    # ---- cpu_baseline leg: the only place the oracle is touched
    from oracle import oracle as O
    t1 = time.perf_counter()
    ref, rs = O.pose_graph_optimize(g["odom"], g["loops"], g["loop_info"], max_iter=5)
    cpu_s = time.perf_counter() - t1
    w = pg.bandwidth
    # the factorisation sweeps the (w+1) x (w+1) block window once per keyframe: read + write of ~ (w+1)^2 / 2 blocks of 128 B
    alg = args.keyframes * (w + 1) * (w + 1) * 128.0 * st["iterations"]
    res = {"metric": "loop-closure pose-graph optimisations/sec (4-DoF keyframe graph, 5 LM iterations)", "value": round(args.steps / el, 3), "unit": "graphs/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "S4 loop-closure graph (new feature, SURVEY 8f-2 / config 4 shape)", "keyframes": args.keyframes, "loops": int(len(g["loops"])),
                      "edges": pg.n_edges, "half_bandwidth_blocks": w, "lm_iterations": st["iterations"], "rounds": rounds,
                      "all_reduce_bytes_per_round": 8 * pg.reduce_count,
                      "collective_backend": dist.get_backend() if multi else None, "collective_ranks": world if multi else 0,
                      "collective_lib": _rccl_mapped() if multi else None},
           "roofline": {"bound": "hbm", "kernel": "k_pg_step (one workgroup: block-banded Cholesky, 8-column panels in LDS, trailing window on f64 MFMA; bound by the dependent chain of 4541 pivots)",
                        "achieved": round(alg * args.steps / el / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg * args.steps / el / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic": None},
           "cpu_baseline": {"value": round(1.0 / cpu_s, 3), "unit": "graphs/s", "cores": 1, "kind": "port", "sample": "the same graph, oracle/lo_posegraph.c (-O3), 1 thread"},
           "max_keyframe_diff_vs_cpu_m": float(np.abs(out[:, :3] - ref[:, :3]).max()),
           "ate_vs_truth_m": {"odometry": round(s4.ate(g["odom"], g["truth"]), 4), "optimised": round(s4.ate(out, g["truth"]), 4)}}
    if args.dump_poses:
        np.savez(args.dump_poses, poses=out)
    print(json.dumps(res), flush=True)
    if multi:
        dist.destroy_process_group()


GOLDEN = os.path.join(ROOT, "tests", "golden", "s1_seq%02d_oracle.npz")


def _rccl_mapped():
    """The RCCL shared object mapped into this process (from /proc/self/maps), or None: evidence that a collective went through RCCL."""
    try:
        with open("/proc/self/maps") as fh:
            for ln in fh:
                if "librccl" in ln:
                    return ln.split()[-1]
    except OSError:
        pass
    return None


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as fresh child processes (the
    driver's own command line: torch.distributed.run, one rank per GPU) BEFORE anything in this process touches the GPU,
    and exit with the children's code.  The parent imports neither torch nor the HIP library."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def probe_ranks(args):
    """--probe-ranks: the launcher check used by the CPU tests -- every rank joins a gloo group, the ranks are counted with
    one all-reduce and rank 0 prints the count.  No GPU call."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"probe": "ranks", "n_gpus": world, "ranks_seen": int(t.item()), "requested": args.gpus}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def ba_secondary(ctx, windows=1024, steps=3):
    """BASELINE configs[2] shape beside the headline: batched Estimator::optimization() solves and the
    batch-1 latency the reference's own operating point (one sequence, one window per frame) sees."""
    import lmono_amd
    from workloads import s2 as K
    base = [K.make_window(s) for s in range(16)]
    b = lmono_amd.BaBatch(ctx, [base[k % 16] for k in range(windows)])
    b.reset(); b.solve(30); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.reset(); b.solve(30)
    ctx.synchronize()
    el = (time.perf_counter() - t0) / steps
    _, _, _, sm = b.read()
    n_obs = float(np.mean([len(w["obs_feat"]) for w in base])); iters = float(sm[:, 2].mean())
    flops = (2.0e3 * n_obs + 72.0 ** 3 / 3.0) * iters * windows
    one = lmono_amd.BaBatch(ctx, [base[0]])
    one.reset(); one.solve(30); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        one.reset(); one.solve(30)
    ctx.synchronize()
    lat = (time.perf_counter() - t0) / 10
    return {"workload": "S2 11-frame windows (configs[2] shape), Estimator::optimization() solve, <= 30 dogleg iterations",
            "windows_per_s": round(windows / el, 1), "batch": windows, "single_window_latency_ms": round(lat * 1e3, 3),
            "mean_iterations": iters, "fp64_tflops": round(flops / el / 1e12, 3), "fp64_frac": round(flops / el / 1e12 / FP64_PEAK_TFLOPS, 5)}


def ba_streams_secondary(streams=(1, 64, 256), frames=160):
    """The Estimator frame loop over N independent sequences in lock-step (EstimatorBatch) beside the headline: frames/s at a few stream counts on
    short S2 streams, and whether every stream's output is the bytes of its single-stream run.  The C++ host mirror runs in a child process (its own GPU context);
    the full-length figures (2761 frames per stream) are `bench.py --workload ba-seq --seq-streams N` (profiles/r6/ba_seq_*streams.json)."""
    import subprocess
    import tempfile
    d = tempfile.mkdtemp()
    n_files = 4
    # (generated in this process, one after the other: a process that holds a GPU context does not fork workers)
    files = [_make_seq_stream((frames, 2 + k, os.path.join(d, "stream%d.bin" % k))) for k in range(n_files)]
    exe = os.path.join(ROOT, "lmono_amd", "host", "estimator_seq")
    single = {}
    out = {"workload": "S2 frame streams (configs[2] shape), %d frames each, %d different files; EstimatorBatch: one batched C-ABI call per numeric step, marginalisation overlapped" % (frames, n_files),
           "frames_per_s": {}, "every_stream_equals_its_single_stream_run": True}
    # (the largest count once more as TWO lock-step batches driven by one thread: a batch's host passes run under the other's solve)
    for N, G in [(N, 1) for N in streams] + [(max(streams), 2)] * (max(streams) >= 128):
        if N == 1:
            r = subprocess.run([exe, files[0], "-", "async"], capture_output=True, text=True, timeout=300)
        else:
            r = subprocess.run([exe, files[0], "-", "async", "streams=%d" % N, "groups=%d" % G, "digest"] + files[1:], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-500:])
        lines = r.stdout.splitlines()
        tim = [ln for ln in lines if ln.startswith("TIM")][0].split()
        out["frames_per_s"][str(N) if G == 1 else "%d as %d batches" % (N, G)] = round(max(N, 1) * 1e3 / float(tim[2]), 1)
        dig = {int(ln.split()[1]): ln.split()[2] for ln in lines if ln.startswith("DIG")}
        if N > 1:
            for k in range(min(n_files, N)):
                if k not in single:
                    q = subprocess.run([exe, files[k], "-", "async"], capture_output=True, text=True, timeout=300)
                    single[k] = [ln for ln in q.stdout.splitlines() if ln.startswith("DIG")][0].split()[2]
                for s_ in range(k, N, n_files):
                    out["every_stream_equals_its_single_stream_run"] = out["every_stream_equals_its_single_stream_run"] and dig[s_] == single[k]
    return out


def streamed_extra(ctx, off, xyzi_d, args, gold_poses):
    """PCIe-inclusive operation (SURVEY 8d "PCIe H2D floor"): the sequence sits in PINNED host memory and streams through
    two device working sets in n_chunks scan ranges -- lmono_batch_stage_h copies range j + 1 on the library's copy stream while range j is
    registered and its chains run (scan-range shard with a lead-in, the range boundary validated like a rank boundary).  Reports the
    streamed rate, the copy-only and compute-only times of the same ranges, and how much of the shorter one was hidden."""
    import torch
    import lmono_amd
    from lmono_amd import trajectory
    n = len(off) - 1
    lead = args.lead
    n_chunks = max(1, min(args.stream_chunks, n // 8))
    total_bytes = int(off[-1]) * 16
    pinned = ctx.host_alloc(total_bytes)
    torch.from_numpy(pinned.view(np.float32).reshape(-1, 4)).copy_(xyzi_d)          # the sequence back in (pinned) host memory
    base = pinned.ctypes.data
    ranges = []
    for j in range(n_chunks):
        s, e = j * n // n_chunks, (j + 1) * n // n_chunks
        lb = max(s - lead, 0)
        ranges.append((lb, s, e))
    max_scans = max(e - lb for lb, s, e in ranges)
    max_pts = max(int(off[e] - off[lb]) for lb, s, e in ranges)
    batches = [lmono_amd.ScanBatch(ctx, max_scans, max_pts) for _ in range(2)]
    chains = max(1, args.stream_chains if args.stream_chains > 0 else args.chains // n_chunks)
    incr_all = torch.zeros((n, 7), dtype=torch.float64, device=xyzi_d.device)
    incr_all[:, 3] = 1.0
    incr_loc = torch.zeros((max_scans, 7), dtype=torch.float64, device=xyzi_d.device)
    poses = torch.zeros((n, 7), dtype=torch.float64, device=xyzi_d.device)

    def stage(j):
        lb, s, e = ranges[j]
        batches[j % 2].stage_host(base + int(off[lb]) * 16, int(off[e] - off[lb]), keepalive=pinned)

    def compute(j, staged):
        lb, s, e = ranges[j]
        b = batches[j % 2]
        o = (off[lb:e + 1] - off[lb]).astype(np.int64)
        if staged:
            b.scanreg_staged(o, 64, 5.0)
        else:
            b.scanreg(xyzi_d.data_ptr() + int(off[lb]) * 16, o, 64, 5.0, keepalive=xyzi_d)
        b.odometry_shard_d(chains, lead, s - lb, incr_loc.data_ptr())
        if j > 0:
            b.shard_validate(incr_all[s - 1].cpu().numpy(), incr_loc.data_ptr())
        else:
            ctx.synchronize()
        incr_all[s:e] = incr_loc[s - lb:e - lb]

    def run_streamed():
        stage(0)
        for j in range(n_chunks):
            if j + 1 < n_chunks:
                stage(j + 1)
            compute(j, True)
        torch.cuda.synchronize()

    def run_compute_only():
        for j in range(n_chunks):
            compute(j, False)
        torch.cuda.synchronize()

    def timed(fn, reps=2):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    t_compute = timed(run_compute_only)
    t_stream = timed(run_streamed)
    ctx.pose_prefix_d(incr_all.data_ptr(), 0, n, poses.data_ptr())
    torch.cuda.synchronize()
    p = poses.cpu().numpy()
    # copy-only: the same stage calls, then wait for the device (hipDeviceSynchronize covers the copy stream)
    def run_copy():
        for j in range(n_chunks):
            stage(j)
        torch.cuda.synchronize()
    t_copy = timed(run_copy)
    out = {"scans_per_s": round(n / t_stream, 1), "ms_per_pass": round(t_stream * 1e3, 2), "chunks": n_chunks, "chains_per_chunk": chains,
           "h2d_GBps": round(total_bytes / t_copy / 1e9, 1), "copy_only_ms": round(t_copy * 1e3, 2), "compute_only_ms": round(t_compute * 1e3, 2),
           "overlap_frac": round((t_copy + t_compute - t_stream) / min(t_copy, t_compute), 3),
           "pcie_floor_scans_per_s": round(n / t_copy, 1),
           "note": "pinned host memory -> two device working sets; H2D of range j + 1 under scanRegistration + laserOdometry of range j; "
                   "overlap_frac 1 = the shorter of (copy, compute) fully hidden"}
    if gold_poses is not None and len(gold_poses) >= n:
        out["ate_vs_cpu_m"] = round(trajectory.ate(p, gold_poses[:n]), 6)
    for b in batches:
        b.close()
    ctx.host_free(pinned)
    return out


def stress_extra(ctx, args, chains, dev):
    """Untimed extra: the SAME pass over the cluttered stress world (seq 2: 200 small boxes, 20 % stray returns, moving cylinders,
    dropped ring sectors) -- the headline's world-dependence in the driver line.  Same layout (chains, lead), three passes timed."""
    import torch
    import lmono_amd
    from lmono_amd import trajectory
    from workloads import s1 as S1
    n = args.scans
    w = S1.S1World(seed=4242, n_az=args.az, clutter=True)
    xyzi, off = w.scans(w.trajectory(n))
    xd = torch.from_numpy(xyzi).to(dev)
    del xyzi
    b = lmono_amd.ScanBatch(ctx, n, int(off[-1]))
    incr = torch.zeros((n, 7), dtype=torch.float64, device=dev)
    poses = torch.zeros((n, 7), dtype=torch.float64, device=dev)

    def one():
        b.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
        b.odometry_d(chains, args.lead, incr.data_ptr(), None)
        rep = b.boundary_report()
        ctx.pose_prefix_d(incr.data_ptr(), 0, n, poses.data_ptr())
        return rep
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = [one() for _ in range(3)]
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 3
    rep = reps[-1]
    out = {"world": "stress sequence (seq 2: cluttered world), %d scans, the headline's layout (%d chains, lead %d)" % (n, chains, args.lead),
           "scans_per_s": round(n / el, 1), "ms_per_step": round(el * 1e3, 3), "boundaries": rep["n_chains"] - 1, "flagged": rep["flagged"],
           "pairs_rerun": rep["pairs_rerun"], "rounds": rep["rounds"], "unresolved": rep["unresolved"], "repair_ms": round(float(np.mean([r["repair_ms"] for r in reps])), 3)}
    if args.az == 2000 and os.path.exists(GOLDEN % 2):
        gp = np.load(GOLDEN % 2)["poses"]
        hi = min(n, len(gp))
        out["ate_vs_cpu_m"] = round(trajectory.ate(poses.cpu().numpy()[:hi], gp[:hi]), 6)
    return out


def latency_extra(ctx, xyzi_d, off, n_steps=64):
    """One scan per call (lmono_odom_step): wall time of a callback for a scan already in HBM -- scanRegistration of ONE scan plus ONE scan
    pair of laserOdometry -- and of the scanRegistration part alone (a 1-scan batch)."""
    import torch
    import lmono_amd
    cap = int(np.diff(off[:n_steps + 2]).max())
    st = lmono_amd.OdomStream(ctx, cap, 64, 5.0, history=8)
    ts = []
    for k in range(n_steps + 1):
        a, e = int(off[k]), int(off[k + 1])
        t0 = time.perf_counter()
        st.step(dev_ptr=xyzi_d.data_ptr() + a * 16, n_points=e - a)
        ts.append(time.perf_counter() - t0)
    st.close()
    ts = np.array(ts[1:]) * 1e3
    b1 = lmono_amd.ScanBatch(ctx, 1, cap)
    tr = []
    for k in range(1, 17):
        a, e = int(off[k]), int(off[k + 1])
        o = np.array([0, e - a], np.int64)
        t0 = time.perf_counter()
        b1.scanreg(xyzi_d.data_ptr() + a * 16, o, 64, 5.0, keepalive=xyzi_d)
        ctx.synchronize()
        tr.append(time.perf_counter() - t0)
    b1.close()
    reg = float(np.median(tr[2:])) * 1e3
    return {"online_step_ms": round(float(np.median(ts)), 3), "online_step_p99_ms": round(float(np.quantile(ts, 0.99)), 3),
            "scanreg_ms": round(reg, 3), "odom_pair_ms": round(float(np.median(ts)) - reg, 3), "steps": n_steps,
            "note": "lmono_odom_step on a scan resident in HBM: scanRegistration of one scan + one scan pair (2 x [search + <= 4 LM iterations]), "
                    "host-synchronous; scanreg_ms = the front end of one scan alone"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scans", type=int, default=4541, help="scans per GPU (weak) or in total (strong); KITTI seq 00 = 4541")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank gets --scans scans of one long trajectory; strong: --scans scans in total, sharded "
                         "over the ranks (BASELINE configs[3]: seq 00 sharded across 8)")
    ap.add_argument("--chains", type=int, default=256, help="concurrent odometry chains per GPU (strong scaling: in total)")
    ap.add_argument("--lead", type=int, default=6, help="lead-in scans of a chain that does not start at scan 0")
    ap.add_argument("--lead-full", type=int, default=3,
                    help="lead-in scan pairs of a chain (the last ones) that use all feature points; the earlier ones a quarter (-1: all use all)")
    ap.add_argument("--lead-seed", type=int, default=-1, help="LMONO_OPT_LEAD_SEED (-1: the library's default): 1 = lead-in states re-seeded from the neighbouring chains' first results")
    ap.add_argument("--kitti-dir", default="", help="read the scans of a KITTI-layout sequence directory (velodyne/%%06d.bin + times.txt, e.g. "
                    "dataset/sequences/00: /root/reference/README.md:48-60) instead of generating S1 scans; --scans caps the count")
    ap.add_argument("--seq", type=int, default=0, choices=[0, 1, 2], help="0: the S1 figure-8 sequence (headline); 1: the held-out sequence (other world, clover trajectory); "
                    "2: the stress sequence (cluttered world: small boxes, 20 %% stray returns, moving cylinders, dropped ring sectors)")
    ap.add_argument("--az", type=int, default=2000, help="azimuth steps per ring (2000 = HDL-64 at 10 Hz)")
    ap.add_argument("--cpu-sample", type=int, default=128, help="scans of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="ba-seq: frames of the CPU oracle's replay (-1 = the whole stream)")
    ap.add_argument("--stream-chunks", type=int, default=8, help="streamed extra: scan ranges the sequence streams through two working sets in")
    ap.add_argument("--stream-chains", type=int, default=64, help="streamed extra: chains per range (0 = --chains / --stream-chunks)")
    ap.add_argument("--only-streamed", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (sequential run, BA secondary)")
    ap.add_argument("--probe-ranks", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dump-poses", default="", help=argparse.SUPPRESS)      # rank 0's poses / increments of the timed run as .npz (tests)
    ap.add_argument("--workload", default="lidar", choices=["lidar", "ba", "ba-seq", "map", "colour", "posegraph"],
                    help="lidar = headline (BASELINE configs[1]); ba = configs[2]-shaped sliding-window BA solves (secondary); "
                         "map = laserMapping over a synthetic sequence, one stream (SURVEY 8f-1)")
    ap.add_argument("--windows", type=int, default=1024, help="ba: independent windows per GPU")
    ap.add_argument("--frames-seq", type=int, default=2761, help="ba-seq: frames of the stream (KITTI seq 05 = 2761)")
    ap.add_argument("--streams", type=int, default=64, help="map / colour: independent streams advanced in lock-step")
    ap.add_argument("--seq-groups", type=int, default=1, help="ba-seq: the streams as this many independent lock-step groups (own context, HIP stream and host thread each)")
    ap.add_argument("--seq-streams", type=int, default=1, help="ba-seq: independent sequences stepped in lock-step by EstimatorBatch (1 = the reference's one sequence)")
    ap.add_argument("--frames", type=int, default=20, help="colour: frames per stream and step")
    ap.add_argument("--keyframes", type=int, default=4541, help="posegraph: keyframes of the graph")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.probe_ranks:
        return probe_ranks(args)
    if args.workload == "ba":
        return bench_ba(args)
    if args.workload == "ba-seq":
        return bench_ba_seq(args)
    if args.workload == "map":
        return bench_map(args)
    if args.workload == "colour":
        return bench_colour(args)
    if args.workload == "posegraph":
        return bench_posegraph(args)

    import torch
    import torch.distributed as dist
    import lmono_amd
    from lmono_amd import sharding, trajectory
    from workloads import s1 as S1       # synthetic S1 scans (input plumbing)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus == 1:
            args.gpus = world              # started by a launcher without --gpus: adopt its world size
        else:
            sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    # LMONO_BENCH_REHEARSE=1: every rank on cuda:0 and the collectives over gloo on CPU copies -- the multi-rank step of this file run end to
    # end on a one-GPU box (tests/test_lidar_gpu.py); never a measurement
    rehearse = world > 1 and os.environ.get("LMONO_BENCH_REHEARSE") == "1"
    # LMONO_BENCH_SAME_DEVICE=1: every rank on cuda:0 but the collectives through "nccl" (RCCL) itself -- several ranks on the ONE card of a test box
    # (tests/test_rccl_gpu.py: works only if the library admits two ranks on one device); never a measurement either
    same_device = world > 1 and os.environ.get("LMONO_BENCH_SAME_DEVICE") == "1"
    if rehearse or same_device:
        local_rank = 0
    # LMONO_BENCH_FORCE_COLLECTIVES=1: a ONE-rank run takes the multi-rank step -- a world-1 "nccl" (RCCL) group, the all-gathers and
    # all-reduces on device tensors -- so that the collective path runs on the one GPU a test box has (tests/test_rccl_gpu.py)
    multi = world > 1 or os.environ.get("LMONO_BENCH_FORCE_COLLECTIVES") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    coll = (lambda t: t.cpu()) if rehearse else (lambda t: t)        # the tensor a collective runs on
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kitti_stamps = None
    if args.kitti_dir:
        from lmono_amd import kitti_io
        kitti_stamps = kitti_io.read_times(os.path.join(args.kitti_dir, "times.txt"))
        args.scans = min(args.scans, len(kitti_stamps)) if world == 1 or args.scaling == "strong" else min(args.scans, len(kitti_stamps) // world)
        if args.scaling != "strong" and world > 1:
            pass                                   # weak scaling: every rank takes the next --scans scans of the sequence
    if multi and args.lead < 1:
        sys.exit("bench.py: --gpus > 1 needs --lead >= 1 (a rank's first owned scan pair needs the scan before it, which the lead-in loads)")
    strong = args.scaling == "strong"
    n_total = args.scans if strong else args.scans * world
    load_begin, own_begin, own_end = sharding.shard_range(n_total, world, rank, args.lead)
    lead_r = own_begin - load_begin
    n_local = own_end - load_begin
    n_own = own_end - own_begin

    t0 = time.time()
    traj = None
    if args.kitti_dir:
        xyzi, off, _ = kitti_io.load_scans(args.kitti_dir, load_begin, own_end - load_begin)
    else:
        if args.seq == 0:
            w = S1.S1World(n_az=args.az)
            traj = w.trajectory(n_total)
        elif args.seq == 1:           # the held-out sequence: another world, another trajectory (tests/golden/s1_seq01_oracle.npz)
            w = S1.S1World(seed=777, n_az=args.az)
            traj = w.trajectory_clover(n_total)
        else:                         # the stress sequence: the cluttered world (tests/golden/s1_seq02_oracle.npz)
            w = S1.S1World(seed=4242, n_az=args.az, clutter=True)
            traj = w.trajectory(n_total)
        xyzi, off = w.scans(traj[load_begin:own_end], scan_id0=load_begin)
    gen_s = time.time() - t0
    total_pts = int(off[-1])

    ctx = lmono_amd.Context(local_rank)
    if os.environ.get("LMONO_LEAD_FULL") is None:
        ctx.set_option(ctx.OPT_LEAD_FULL, args.lead_full)
    if args.lead_seed >= 0:
        ctx.set_option(ctx.OPT_LEAD_SEED, args.lead_seed)
    xyzi_t = torch.from_numpy(xyzi)
    t0 = time.time()
    xyzi_d = xyzi_t.to(dev)
    torch.cuda.synchronize()
    h2d_s = time.time() - t0
    sample = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        m = min(args.cpu_sample, n_local)
        sample = (np.ascontiguousarray(xyzi[:off[m]]), off[:m + 1].copy())
    del xyzi, xyzi_t

    batch = lmono_amd.ScanBatch(ctx, n_local, total_pts)
    incr_d = torch.zeros((n_local, 7), dtype=torch.float64, device=dev)
    poses_d = torch.zeros((n_own, 7), dtype=torch.float64, device=dev)
    # chains of this rank.  Weak scaling: --chains per rank.  Strong scaling: a rank's share of the scans is cut by RULE, not by --chains // world:
    # as many chains as keep a chain at least 3 lead-ins long -- the length the single-GPU layout was tuned to (4541 scans / 256 chains = 17.7 scans at
    # lead 6) --, at most --chains; with 4541 scans and lead 6 that gives 256 / 126 / 63 / 31 chains per rank at 1 / 2 / 4 / 8 ranks
    chains = args.chains
    if strong and world > 1:
        chains = min(args.chains, max(1, n_own // (3 * max(args.lead, 1))))
    chains = max(1, min(chains, n_local))

    boundary, shard_rounds = [], [0]

    def step():
        batch.scanreg(xyzi_d.data_ptr(), off, 64, 5.0, keepalive=xyzi_d)
        if not multi:
            batch.odometry_d(chains, args.lead, incr_d.data_ptr(), None)
        else:
            # the rank's chains run over its owned scans (chain 0's lead-in = the previous rank's last scans); then the rank boundaries
            # are validated like the chain boundaries inside a rank: one all-gather of every rank's last increment per round
            # main pass, then ONE all-gather of every rank's last increment, then one validation of all the rank's boundaries -- the chain
            # boundaries inside it and the one to the previous rank -- in the same repair rounds
            batch.odometry_shard_main_d(chains, args.lead, lead_r, incr_d.data_ptr())
            shard_rounds[0] = sharding.validate_rank_boundaries(lambda: coll(incr_d[-1]), lambda prev: batch.shard_validate(prev, incr_d.data_ptr()), rank, world, deferred=True)
        boundary.append(batch.boundary_report())
        ctx.pose_prefix_d(incr_d.data_ptr(), lead_r, n_local, poses_d.data_ptr())
        if multi:
            bases = sharding.gather_bases(coll(poses_d[-1].clone())).to(dev)
            # the first owned increment composes onto the previous rank's last pose
            ctx.pose_rebase_d(bases.data_ptr(), rank, poses_d.data_ptr(), n_own)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_reset()
    del boundary[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        te = coll(torch.tensor([elapsed], dtype=torch.float64, device=dev))
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    groups, n_reg, n_odo = ctx.timing()

    cnt = batch.counts()
    status_or = int(np.bitwise_or.reduce(cnt[:, 5]))
    n_cloud_mean = float(cnt[:, 0].mean())
    gpu_incr = incr_d.cpu().numpy()
    gpu_poses = poses_d.cpu().numpy()

    # ---- tolerance of the timed run: its poses against the committed trajectory of the strictly sequential CPU path over
    # the WHOLE sequence (tests/golden/s1_seq00_oracle.npz: data, not the oracle), every owned scan the fixture covers
    parity = None
    if not args.kitti_dir and args.az == 2000 and os.path.exists(GOLDEN % args.seq):
        gold = np.load(GOLDEN % args.seq)
        gp, gi, gc = gold["poses"], gold["incr"], gold["feat_counts"]
        hi = min(own_end, len(gp))
        sums = np.zeros(6)
        if hi > own_begin:
            m = hi - own_begin
            s2, cnt_a = trajectory.ate_sums(gpu_poses[:m], gp[own_begin:hi])
            st, sr, cnt_r = trajectory.rpe_from_relative(gpu_incr[lead_r:lead_r + m], gi[own_begin:hi])
            feat_equal = float((cnt[lead_r:lead_r + m, 1:5] == gc[own_begin:hi]).all())
            sums = np.array([s2, cnt_a, st, sr, cnt_r, 1.0 - feat_equal])
        if multi:
            ts = coll(torch.from_numpy(sums).to(dev))
            dist.all_reduce(ts)
            sums = ts.cpu().numpy()
        if sums[1] > 0:
            parity = {"reference": "tests/golden/s1_seq%02d_oracle.npz (sequential CPU oracle, n_chains 1, lead 0)" % args.seq,
                      "scans_compared": int(sums[1]),
                      "ate_vs_cpu_m": round(float(np.sqrt(sums[0] / sums[1])), 6),
                      "rpe_vs_cpu": {"delta_scans": 1, "trans_rmse_m": round(float(np.sqrt(sums[2] / sums[4])), 7),
                                     "rot_rmse_deg": round(float(np.degrees(np.sqrt(sums[3] / sums[4]))), 7)},
                      "feature_counts_equal": bool(sums[5] == 0)}
            if world == 1:
                r100 = trajectory.rpe(gpu_poses[:hi], gp[:hi], delta=100)
                parity["rpe_vs_cpu_100"] = {"delta_scans": 100, "trans_rmse_m": round(r100["trans_rmse_m"], 6), "rot_rmse_deg": round(r100["rot_rmse_deg"], 6)}

    if rank == 0:
        scans_per_s = n_total * args.steps / elapsed
        # ---- roofline of the dominant kernel group (device time from hipEvents on the launch stream)
        # algorithmic bytes per scan (DESIGN.md "Kernels"): N = raw points per scan, Nc = points kept in the ring-sorted cloud
        N = total_pts / n_local
        Nc = n_cloud_mean
        nlf = float(cnt[:, 4].mean()); nls = float(cnt[:, 2].mean())
        alg = {   # algorithmic bytes per scan of each kernel
            "k_ring_sort": 16 * N + 16 * Nc,
            "k_curvature": 16 * Nc + 5 * Nc,
            "k_select": 5 * Nc + 1 * Nc,
            "k_voxel": 17 * Nc + 16 * nlf,
            "k_compact": 32 * (nlf + nls),
            "k_grid_build": 48 * (nlf + nls),
            "k_line_index": 32 * (nlf + nls),
            # SURVEY 8d odometry figure: 0.77 MB per scan over its 2 correspondence + 2 solve launches; the record
            # traffic of the solve is part of it, so the whole figure is booked on k_correspond
            "k_correspond": 0.77e6,
        }
        launches_per_step = {k: 1.0 for k in alg}
        pairs = max(groups["odometry_launch_pairs"], 1.0)
        launches_per_step["k_correspond"] = pairs / max(n_odo, 1)
        ms_step = {k: groups[k] / max(args.steps, 1) for k in alg}          # device ms per bench step
        dom = max(ms_step, key=lambda k: ms_step[k])
        ms_launch = ms_step[dom] / launches_per_step[dom]
        # scans handled by one launch: every front-end kernel sees all local scans; a k_correspond launch advances
        # every chain of ITS chain group by half a scan (2 launches per scan-to-scan step).  The timed launches are
        # group 0's (they carry the events); the other G-1 groups run the same kernels beside them on their own streams
        chain_groups = ctx.odom_chain_groups(chains)
        scans_per_launch = n_local if dom != "k_correspond" else chains / chain_groups * 0.5
        ach = alg[dom] * scans_per_launch / (ms_launch * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE and
        # WRITE_SIZE in separate passes, gfx950 correction applied; scripts/profile_round.sh); null if not collected for it
        traffic = None
        sq = None           # SQ instruction counters of a launch of the dominant kernel (the committed summary's, like traffic)
        for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
            pmc_path = os.path.join(ROOT, "profiles", rnd, "pmc_%s.json" % dom)
            if os.path.exists(pmc_path) and chains == 256:
                try:
                    with open(pmc_path) as fh:
                        pmc = json.load(fh)
                    if dom != "k_correspond" or pmc.get("chain_groups", 1) == chain_groups:     # counters of a launch of this size only
                        traffic = pmc["hbm_bytes_per_launch"]
                        if "valu_wave_insts_per_launch" in pmc:
                            sq = (float(pmc["valu_wave_insts_per_launch"]), float(pmc.get("salu_wave_insts_per_launch", 0.0)), "profiles/%s/pmc_%s.json" % (rnd, dom))
                        break
                except (OSError, ValueError, KeyError, AttributeError):      # an unreadable summary: the line carries null
                    pass
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "ms_per_launch": round(ms_launch, 4), "launches_per_step": round(launches_per_step[dom], 1),
                    # the G groups' launches run side by side, each on its share of the CUs: one launch's bytes over its own
                    # duration (achieved, as rocprof sees it) understates the chip by about G; G launches' bytes over that duration:
                    "chain_groups": chain_groups, "achieved_all_groups": round(ach * (chain_groups if dom == "k_correspond" else 1), 2),
                    "frontend_fused_frac": round(37 * N * n_local / (groups["frontend_total"] / max(n_reg, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "whole_step_frac": round((37 * N + 0.77e6) * n_local / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                    "group_ms_per_step": {k: round(v / max(args.steps, 1), 3) for k, v in groups.items() if k != "odometry_launch_pairs"}}
        if sq is not None:
            # INSTRUCTION roofline (VERDICT r5 #4a): the search is bound by instruction issue, so its distance from THAT peak is the figure that says how
            # good the kernel is.  A SIMD issues one wave64 VALU instruction per 4 cycles (MI355X_MICROARCH.md: 16 lanes per cycle); 4 SIMDs per CU.
            # The chain groups' launches share the chip, so the launch's own count is multiplied by the groups that run beside it.
            simds = 4 * (torch.cuda.get_device_properties(dev).multi_processor_count or 256)
            clock_hz = 2.4e9
            issue_peak = simds / 4.0 * clock_hz * (ms_launch * 1e-3)           # wave-level VALU instructions the chip can issue during one launch
            conc = chain_groups if dom == "k_correspond" else 1
            roofline["issue_frac"] = round(sq[0] * conc / issue_peak, 4)
            roofline["issue_frac_valu_plus_salu"] = round((sq[0] + sq[1]) * conc / issue_peak, 4)
            roofline["issue"] = {"valu_wave_insts_per_launch": sq[0], "salu_wave_insts_per_launch": sq[1], "concurrent_launches": conc, "simds": simds, "clock_GHz": 2.4,
                                 "peak_wave_insts_per_launch_time": round(issue_peak), "source": sq[2],
                                 "note": "VALU wave instructions of one launch x the launches that run side by side / (SIMDs / 4 cycles x 2.4 GHz x the launch's duration); "
                                         "SALU instructions issue on their own port: the second figure is an upper bound on how busy the issue stage is"}
        if dom == "k_correspond":
            # "k_correspond" is the library's timing group of the search; the kernel rocprofv3 lists under it is lmono::k_corr_flat
            roofline["kernel_symbol"] = "lmono::k_corr_flat (+ k_correspond_list for deferred features: 0 here)"
            roofline["note"] = ("exact nearest-neighbour + scan-line walk over a (line, azimuth-bin) index, candidates swept as 64-byte chunks; the kernel is bound by "
                                "instruction issue and by the dependent rounds of its slowest workgroup, not by HBM or the L1 (14.4 M VALU + 6.3 M SALU wave instructions per "
                                "64-chain main-pass launch -- issue_frac; round 4's '46 M + 19 M per 256-chain launch' averaged ALL launches of a validated run at lead 4 / 2 full pairs, "
                                "thinned lead-in and short repair launches included: the same kernel, another averaging set; VALU ~80 % busy while the CUs are full, UTCL1 misses 0.04 %; "
                                "profiles/r4/NOTES.md, profiles/r6/NOTES.md) -- the HBM fraction is reported because SURVEY 8d prices the path in bytes")
        out = {
            "metric": "KITTI HDL-64 scans/sec (scanRegistration + laserOdometry)", "value": round(scans_per_s, 1),
            "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32 features / f64 solve", "data": "synthetic",
            "config": {"workload": "KITTI-seq-00-shaped synthetic S1 HDL-64, laserOdometry-only (configs[1])",
                       "scans_total": n_total, "scans_per_gpu": n_own, "points_per_scan": round(N), "azimuth_steps": args.az,
                       "odometry_chains_per_gpu": chains, "chain_lead_in": args.lead, "lead_in_full_pairs": ctx.get_option(ctx.OPT_LEAD_FULL), "lead_in_seed": ctx.get_option(ctx.OPT_LEAD_SEED), "odometry_chain_groups": ctx.odom_chain_groups(chains),
                       "parallelism": ("scan-range shard x%d, one %s all-gather of 7 doubles per rank per exchange" % (world, "gloo (rehearsal)" if rehearse else "RCCL"))
                                      if multi else "no collective (1 rank)",
                       "collective_ranks": world if multi else 0, "collective_backend": (dist.get_backend() if multi else None),
                       "collective_lib": _rccl_mapped() if multi and not rehearse else None, "status_or": status_or, "gen_s": round(gen_s, 1), "h2d_s": round(h2d_s, 2),
                       "h2d_GBps": round(total_pts * 16 / h2d_s / 1e9, 1)},
            "roofline": roofline,
        }
        # the chained schedule validated itself in every timed step (LMONO_OPT_BOUNDARY_TOL; rank 0's chains): per step
        rep = boundary[-1]
        out["boundaries_rerun"] = rep["chains_rerun"]
        out["boundary_validation"] = {"tol": rep["tol"], "boundaries": rep["n_chains"] - 1, "flagged": rep["flagged"], "chains_rerun": rep["chains_rerun"],
                                      "pairs_rerun": rep["pairs_rerun"], "rounds": rep["rounds"], "unresolved": rep["unresolved"],
                                      "max_residual": float(rep["max_resid"]), "residual_q50_q90_q99": [float(v) for v in np.quantile(rep["resid"][1:], [0.5, 0.9, 0.99])] if rep["n_chains"] > 1 else None,
                                      "repair_ms_per_step": round(float(np.mean([b["repair_ms"] for b in boundary])), 3),
                                      "rank_boundary_rounds": shard_rounds[0] if multi else None,
                                      "residual": "max(|dq_i|, 0.1 |dt_i| / m) between a chain's own lead-in estimate of the pair before its first owned one and its predecessor's increment for that pair"}
        if parity is not None:
            out["ate_vs_cpu_m"] = parity["ate_vs_cpu_m"]
            out["rpe_vs_cpu"] = parity["rpe_vs_cpu"]
            out["parity"] = parity
        if traj is not None:
            m_gt = min(n_own, len(traj))
            out["ate_vs_truth_m"] = round(trajectory.ate(gpu_poses[:m_gt], _gt_relative(traj[:m_gt])), 4) if rank == 0 and own_begin == 0 else None
        else:
            out["data"] = "KITTI-layout sequence " + args.kitti_dir
            out["config"]["workload"] = "KITTI-layout sequence on disk, laserOdometry-only (configs[1])"
        if world == 1 and not args.no_extras:
            # ---- untimed extras.  (1) A-LOAM's own schedule: ONE chain, no lead-in (every scan pair warm-started from the
            # previous increment) -- the run whose poses must EQUAL the CPU path's, and the throughput of that schedule
            seq_incr = torch.zeros((n_local, 7), dtype=torch.float64, device=dev)
            seq_poses = torch.zeros((n_local, 7), dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch.odometry_d(1, 0, seq_incr.data_ptr(), None)
            ctx.pose_prefix_d(seq_incr.data_ptr(), 0, n_local, seq_poses.data_ptr())
            torch.cuda.synchronize()
            seq_s = time.perf_counter() - t0
            out["sequential"] = {"schedule": "n_chains 1, lead 0 (the reference's: strictly sequential warm starts), odometry stage only",
                                 "odometry_scans_per_s": round(n_local / seq_s, 1)}
            if parity is not None:
                sp = seq_poses.cpu().numpy()
                hi = min(n_local, len(gp))
                out["sequential"]["max_abs_pose_diff_vs_cpu"] = float(np.abs(sp[:hi] - gp[:hi]).max())
                out["sequential"]["ate_vs_cpu_m"] = round(trajectory.ate(sp[:hi], gp[:hi]), 9)
            # (2) one scan per call: the latency of a callback
            try:
                out["latency"] = latency_extra(ctx, xyzi_d, off)
            except Exception as e:
                out["latency"] = {"error": repr(e)}
            # (3) PCIe-inclusive streaming of the whole sequence from pinned host memory
            try:
                out["streamed"] = streamed_extra(ctx, off, xyzi_d, args, gp if parity is not None else None)
            except Exception as e:
                out["streamed"] = {"error": repr(e)}
            # (4) the Estimator half beside the headline
            try:
                out["secondary"] = {"ba": ba_secondary(ctx)}
            except Exception as e:       # the secondary line must never take the headline down
                out["secondary"] = {"ba": {"error": repr(e)}}
            try:
                out["secondary"]["ba_streams"] = ba_streams_secondary()
            except Exception as e:
                out["secondary"]["ba_streams"] = {"error": repr(e)}
            # (5) the same pass over the cluttered world: the chained schedule's price is world-dependent, its result is not
            if args.seq == 0 and not args.kitti_dir:
                try:
                    del seq_incr, seq_poses
                    out["secondary"]["stress"] = stress_extra(ctx, args, chains, dev)
                except Exception as e:
                    out["secondary"]["stress"] = {"error": repr(e)}
        if sample is not None:
            # ---- cpu_baseline leg: the only place the oracle is touched
            from oracle import oracle as O
            sx, so = sample
            m = len(so) - 1
            t0 = time.time()
            ref = O.run_sequence(sx, so, threads=1)
            cpu_s = time.time() - t0
            out["cpu_baseline"] = {"value": round(m / cpu_s, 2), "unit": "scans/s", "cores": 1, "kind": "port",
                                   "sample": "first %d scans of the same sequence, oracle/ C restatement (-O3, kd-tree), 1 thread: scanreg %.0f ms + odometry %.0f ms"
                                             % (m, ref["stage_ms"][0], ref["stage_ms"][1])}
            # the same CPU path on all host cores the process may use: scans in parallel (front end), one odometry chain per
            # thread with the same lead-in as the GPU run; a reported baseline, not the target
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = min(cores, 16)     # the host share that goes with one GPU of the box
            if cores > 1:
                t0 = time.time()
                ref_mt = O.run_sequence(sx, so, n_chains=cores, lead=args.lead, threads=cores)
                cpu_mt = time.time() - t0
                out["cpu_baseline_all_cores"] = {"value": round(m / cpu_mt, 2), "unit": "scans/s", "cores": cores, "kind": "port",
                                                 "sample": "the same %d scans, OpenMP over scans / %d odometry chains with lead-in %d: scanreg %.0f ms + odometry %.0f ms"
                                                           % (m, cores, args.lead, ref_mt["stage_ms"][0], ref_mt["stage_ms"][1])}
        if args.dump_poses:
            np.savez(args.dump_poses, poses=gpu_poses, incr=gpu_incr)
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def _gt_relative(poses):
    """Ground-truth sensor poses relative to scan 0 as [n,7] (q xyzw, t): the planar S1 trajectory (x, y, z, yaw)."""
    out = np.zeros((len(poses), 7))
    x0, y0, z0, yaw0 = poses[0]
    c, s = np.cos(-yaw0), np.sin(-yaw0)
    dx, dy = poses[:, 0] - x0, poses[:, 1] - y0
    out[:, 4] = c * dx - s * dy; out[:, 5] = s * dx + c * dy; out[:, 6] = poses[:, 2] - z0
    half = 0.5 * (poses[:, 3] - yaw0)
    out[:, 2] = np.sin(half); out[:, 3] = np.cos(half)
    return out


if __name__ == "__main__":
    main()
