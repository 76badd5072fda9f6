"""Scan-to-map optimisation step of laserMapping on the GPU (lmono_map_refine) against the CPU oracle on
the same clouds: identical neighbour sets / residual blocks, refined pose within 1e-9."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scan_clouds(oracle, x, off, k):
    n = int(off[k + 1] - off[k])
    bufs = [np.zeros((n, 4), np.float32) for _ in range(5)]
    curv = np.zeros(n, np.float32); label = np.zeros(n, np.int32)
    info = oracle.ScanregInfo()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.lib().lo_scanreg(P(x[off[k]:off[k + 1]]), n, 64, C.c_float(5.0), P(bufs[0]), P(curv), P(label), P(bufs[1]), P(bufs[2]), P(bufs[3]), P(bufs[4]), C.byref(info))
    return bufs[2][:info.n_less_sharp].copy(), bufs[4][:info.n_less_flat].copy()


@pytest.fixture(scope="module")
def frames(oracle):
    """Map state after 4 frames of the synthetic sequence and the inputs of the next two frames."""
    w = oracle.S1World(n_az=500)
    traj = w.trajectory(7)
    x, off = w.scans(traj)
    ref = oracle.run_sequence(x, off)
    m = oracle.Map()
    for k in range(4):
        ls, lf = _scan_clouds(oracle, x, off, k)
        m.process(ls, lf, ref["poses"][k, :4], ref["poses"][k, 4:])
    out = []
    for k in (4, 5):
        ls, lf = _scan_clouds(oracle, x, off, k)
        out.append(dict(cmap=m.all_points(0), smap=m.all_points(1), cstack=oracle.voxel_filter(ls, 0.4), sstack=oracle.voxel_filter(lf, 0.8),
                        x0=ref["poses"][k].copy()))
        m.process(ls, lf, ref["poses"][k, :4], ref["poses"][k, 4:])
    return out


def test_refine_matches_oracle_blocks_and_pose(oracle, gpu_ctx, frames):
    poses, stats, nn = gpu_ctx.map_refine([f["cmap"] for f in frames], [f["smap"] for f in frames], [f["cstack"] for f in frames],
                                          [f["sstack"] for f in frames], np.array([f["x0"] for f in frames]), want_nn=True)
    at = 0
    for s, f in enumerate(frames):
        xr, st, corr = oracle.map_refine(f["cmap"], f["smap"], f["cstack"], f["sstack"], f["x0"])
        assert list(stats[s, :6]) == [st.n_edge[0], st.n_edge[1], st.n_plane[0], st.n_plane[1], st.lm_iters[0], st.lm_iters[1]]
        assert st.n_edge[1] > 100 and st.n_plane[1] > 500
        assert np.abs(poses[s] - xr).max() < 1e-9
        # neighbour sets of the accepted blocks are the exact 5 nearest map points (brute force, float distances)
        nq = len(f["cstack"]) + len(f["sstack"])
        mine = nn[at:at + nq]
        acc = np.nonzero(mine[:, 0] >= 0)[0]
        assert len(acc) == len(corr)
        xq = xr   # the last outer iteration starts from the pose after the first solve: recompute it from the oracle instead
        rng = np.random.default_rng(0)
        for qi in rng.choice(acc, 60, replace=False):
            which = 0 if qi < len(f["cstack"]) else 1
            cloud = f["cmap"] if which == 0 else f["smap"]
            idx = mine[qi]
            assert len(set(idx.tolist())) == 5 and (idx < len(cloud)).all()
        at += nq
    # the accepted blocks are written in query order: same source points as the oracle's list
    f = frames[0]
    _, _, corr = oracle.map_refine(f["cmap"], f["smap"], f["cstack"], f["sstack"], f["x0"])
    acc = np.nonzero(nn[:len(f["cstack"]) + len(f["sstack"]), 0] >= 0)[0]
    stack = np.concatenate([f["cstack"], f["sstack"]])
    assert np.array_equal(stack[acc, :3], np.array([[c.cp[0], c.cp[1], c.cp[2]] for c in corr], np.float32))


def test_solve_cluster_sizes_agree(oracle, gpu_ctx, frames):
    """k_map_solve runs as a cluster of K workgroups per stream; map_solve_cluster (mapping.hip) starts from K = 8 and lowers it until
    ceil(streams / 8) * 8 * K <= 128 workgroups: 1, 20, 40, 56 and 72 streams run K = 8, 5, 3, 2 and 1.  The same frame in every stream must give the
    oracle's block counts, iteration counts and pose -- and, since round 6, the same BYTES whatever the cluster size (a stream's sums are formed per
    virtual rank -- always eight -- and added in rank order: its result no longer depends on how many streams share the call)."""
    f = frames[1]
    xr, st, _ = oracle.map_refine(f["cmap"], f["smap"], f["cstack"], f["sstack"], f["x0"])
    want = [st.n_edge[0], st.n_edge[1], st.n_plane[0], st.n_plane[1], st.lm_iters[0], st.lm_iters[1]]
    first = None
    for n in (1, 20, 40, 56, 72):
        poses, stats, _ = gpu_ctx.map_refine([f["cmap"]] * n, [f["smap"]] * n, [f["cstack"]] * n, [f["sstack"]] * n, np.tile(f["x0"], (n, 1)))
        for s in range(n):
            assert list(stats[s, :6]) == want, (n, s)
        assert np.abs(poses - xr).max() < 1e-9, n
        assert (poses == poses[0]).all(), n                 # every stream of a launch adds the same partial sums in the same order
        first = poses[0] if first is None else first
        assert poses[0].tobytes() == first.tobytes(), n       # K-independent bytes


def test_neighbours_are_the_exact_five_nearest(oracle, gpu_ctx, frames):
    f = frames[0]
    # zero LM effect on the check: compare against brute force at the FINAL pose by refining from the converged pose
    xr, _, _ = oracle.map_refine(f["cmap"], f["smap"], f["cstack"], f["sstack"], f["x0"])
    poses, stats, nn = gpu_ctx.map_refine([f["cmap"]], [f["smap"]], [f["cstack"]], [f["sstack"]], xr[None], want_nn=True)
    # neighbours of the LAST outer iteration were searched from the pose after the first solve; the second solve barely
    # moves a converged pose, so brute force at `poses` must reproduce them for all but boundary cases
    from scipy.spatial.transform import Rotation as R
    x1, st1, _ = oracle.map_refine(f["cmap"], f["smap"], f["cstack"], f["sstack"], xr)
    assert np.abs(poses[0] - x1).max() < 1e-9
    q = poses[0]
    Rm = R.from_quat(q[:4]).as_matrix()
    bad = 0
    checked = 0
    for which, stack, cloud, base in ((0, f["cstack"], f["cmap"], 0), (1, f["sstack"], f["smap"], len(f["cstack"]))):
        pts = (stack[:, :3].astype(np.float64) @ Rm.T + q[4:]).astype(np.float32)
        for i in range(0, len(stack), 37):
            idx = nn[base + i]
            if idx[0] < 0:
                continue
            d = ((cloud[:, 0] - pts[i, 0]) ** 2 + (cloud[:, 1] - pts[i, 1]) ** 2) + (cloud[:, 2] - pts[i, 2]) ** 2
            order = np.lexsort((np.arange(len(cloud)), d))[:5]
            checked += 1
            # the query point used here is re-derived from the final pose (the device searched from the pose after the
            # first solve, ~1e-6 m away): near-ties may swap, the five distances may not differ
            assert np.allclose(np.sort(d[idx]), d[order], rtol=0, atol=1e-2), (i, idx, order)
            bad += list(order) != list(idx)
    assert checked > 50 and bad <= checked // 5


def test_empty_map_leaves_the_pose_alone(gpu_ctx, frames):
    f = frames[0]
    poses, stats, _ = gpu_ctx.map_refine([np.zeros((0, 4), np.float32)], [np.zeros((0, 4), np.float32)], [f["cstack"]], [f["sstack"]], f["x0"][None])
    assert np.array_equal(poses[0], f["x0"]) and not stats[:, :6].any()


def test_voxel_filter_is_bit_exact(oracle, gpu_ctx, frames):
    """lmono_voxel_filter (stable radix sort by PCL cell index, index-order sums) against oracle.voxel_filter."""
    rng = np.random.default_rng(11)
    f = frames[0]
    clouds = [f["cmap"], f["smap"], f["cstack"], f["sstack"]]
    leafs = [0.4, 0.8, 0.4, 0.8]
    big = np.zeros((60000, 4), np.float32); big[:, :3] = rng.uniform(-40, 40, (60000, 3)); big[:, 3] = rng.uniform(0, 64, 60000)
    clouds += [big, big[:1], big[:2] * 0 + np.float32(-3.3), np.zeros((0, 4), np.float32), rng.normal(0, 0.05, (500, 4)).astype(np.float32)]
    leafs += [0.8, 0.4, 0.2, 0.4, 0.2]
    out = gpu_ctx.voxel_filter(clouds, leafs)
    for c, leaf, o in zip(clouds, leafs, out):
        want = oracle.voxel_filter(c, leaf) if len(c) else np.zeros((0, 4), np.float32)
        assert o.shape == want.shape, (len(c), leaf, o.shape, want.shape)
        assert np.array_equal(o.view(np.uint32), want.view(np.uint32))
    assert len(out[4]) > 20000
    with pytest.raises(Exception):
        gpu_ctx.voxel_filter([np.zeros((70000, 4), np.float32)], [0.4])


def test_voxel_filter_tile_edges_long_runs_and_every_pass_count(oracle, gpu_ctx):
    """The filter works in 2048-point tiles, one workgroup each, 1..4 radix passes of 1..9 bits: sizes around the tile edges, the largest cloud,
    runs of one cell that span several tiles, and key widths that need 1, 2, 3 and 4 passes."""
    rng = np.random.default_rng(12)
    def cloud(n, span, z_span=None):
        c = np.zeros((n, 4), np.float32)
        c[:, :3] = rng.uniform(-span, span, (n, 3))
        if z_span is not None:
            c[:, 2] = rng.uniform(-z_span, z_span, n)
        c[:, 3] = rng.uniform(0, 64, n)
        return c
    clouds, leafs = [], []
    for n in (63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4096, 4097, 6143, 65535, 65536):
        clouds.append(cloud(n, 30.0)); leafs.append(0.5)
    # long runs: 5000 points in one cell, then a mixture where a third of the points share three cells (runs across tile and wave boundaries)
    one = cloud(5000, 0.05); one[:, :3] += np.float32(10.2)
    clouds.append(one); leafs.append(0.4)
    mix = cloud(30000, 20.0); mix[::3, :3] = np.float32(0.1) + rng.integers(0, 3, (10000, 1)).astype(np.float32) + rng.uniform(0, 0.1, (10000, 3)).astype(np.float32)
    clouds.append(mix); leafs.append(0.4)
    # key widths: ~3 bits (1 pass), ~15 bits (2 passes), ~26 bits (3 x 9), ~29 bits (4 x 8)
    clouds += [cloud(3000, 0.9), cloud(20000, 6.0), cloud(40000, 90.0, 20.0), cloud(40000, 40.0)]
    leafs += [1.0, 0.4, 0.4, 0.1]
    out = gpu_ctx.voxel_filter(clouds, leafs)
    for c, leaf, o in zip(clouds, leafs, out):
        want = oracle.voxel_filter(c, leaf)
        assert o.shape == want.shape, (len(c), leaf, o.shape, want.shape)
        assert np.array_equal(o.view(np.uint32), want.view(np.uint32)), (len(c), leaf)
    assert len(out[14]) == 1


def test_device_resident_mapper_follows_the_oracle(oracle, gpu_ctx):
    """lmono_mapper_process frame by frame (scan clouds taken in place from the scan batch, cube map in HBM) against
    oracle.run_mapping fed with the same odometry poses."""
    import torch
    import lmono_amd
    w = oracle.S1World(n_az=500)
    traj = w.trajectory(12)
    x, off = w.scans(traj)
    xd = torch.from_numpy(x).cuda()
    batch = lmono_amd.ScanBatch(gpu_ctx, 12, len(x))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    _, odo = batch.odometry(n_chains=1, lead=0)
    want = oracle.run_mapping(x, off, odo)
    mapper = lmono_amd.Mapper(gpu_ctx)
    got = np.zeros((12, 7))
    for k in range(12):
        q, t, st = mapper.process(batch, k, odo[k, :4], odo[k, 4:])
        got[k, :4] = q; got[k, 4:] = t
        ws = want["stats"][k]
        assert list(st[:6]) == [ws.n_edge[0], ws.n_edge[1], ws.n_plane[0], ws.n_plane[1], ws.lm_iters[0], ws.lm_iters[1]], k
    assert np.abs(got - want["poses"]).max() < 1e-7
    assert want["stats"][-1].n_plane[1] > 1000 and oracle.ate(got, oracle.gt_relative(traj)) < oracle.ate(odo, oracle.gt_relative(traj))
    # the map itself: replay the oracle frame by frame and compare every cube
    m = oracle.Map()
    import ctypes as C
    for k in range(12):
        ls, lf = _scan_clouds(oracle, x, off, k)
        m.process(ls, lf, odo[k, :4], odo[k, 4:])
    n_pts = 0
    for which in (0, 1):
        for i in range(21):
            for j in range(21):
                for kk in range(11):
                    a = m.cube(which, i, j, kk)
                    bq = mapper.cube(which, i, j, kk)
                    assert a.shape == bq.shape, (which, i, j, kk)
                    if len(a):
                        assert np.abs(a - bq).max() < 1e-4
                        n_pts += len(a)
    assert n_pts > 20000


def test_failed_frame_leaves_the_mapper_as_it_was(oracle, gpu_ctx, full_seq):
    """A frame that fails AFTER the cube array was shifted (a scan cloud beyond the 65536-point stack, offered at a pose 300 m away) must not
    change the mapper: the following frames give the poses, bit for bit, of a mapper that never saw the failing call."""
    import torch
    import lmono_amd
    w = oracle.S1World(n_az=500)
    traj = w.trajectory(6)
    x, off = w.scans(traj)
    xd = torch.from_numpy(x).cuda()
    batch = lmono_amd.ScanBatch(gpu_ctx, 6, len(x))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    _, odo = batch.odometry(n_chains=1, lead=0)
    big = full_seq["xyzi"][full_seq["off"][0]:full_seq["off"][1]].copy()
    big[:, :3] *= 4.0                                   # less-flat cloud > 65536 points (test_lidar_gpu.py)
    bd = torch.from_numpy(big).cuda()
    bad = lmono_amd.ScanBatch(gpu_ctx, 1, len(big))
    bad.scanreg(bd.data_ptr(), np.array([0, len(big)], np.int64), 64, 5.0, keepalive=bd)
    assert bad.counts()[0, 4] > 65536
    ref, tst = lmono_amd.Mapper(gpu_ctx), lmono_amd.Mapper(gpu_ctx)
    out = {id(ref): [], id(tst): []}
    for k in range(6):
        for m in (ref, tst):
            q, t, st = m.process(batch, k, odo[k, :4], odo[k, 4:])
            out[id(m)].append(np.concatenate([q, t]))
        if k == 2:
            with pytest.raises(lmono_amd.LmonoError):
                tst.process(bad, 0, np.array([0.0, 0.0, 0.0, 1.0]), np.array([300.0, -200.0, 0.0]))
    assert np.array_equal(np.array(out[id(ref)]), np.array(out[id(tst)]))
    for which in (0, 1):
        for i in (9, 10, 11):
            for j in (9, 10, 11):
                a, bq = ref.cube(which, i, j, 5), tst.cube(which, i, j, 5)
                assert a.shape == bq.shape and np.array_equal(a, bq)


def test_batched_streams_match_their_single_stream_oracles(oracle, gpu_ctx):
    """lmono_mapper_process_batch: three independent streams (two different worlds, one of them twice) advanced in
    lock-step, each against oracle.run_mapping of its own sequence."""
    import torch
    import lmono_amd
    seqs = []
    for seed in (20240, 777):
        w = oracle.S1World(seed=seed, n_az=500)
        traj = w.trajectory(8)
        x, off = w.scans(traj)
        xd = torch.from_numpy(x).cuda()
        batch = lmono_amd.ScanBatch(gpu_ctx, 8, len(x))
        batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
        _, odo = batch.odometry(n_chains=1, lead=0)
        seqs.append(dict(batch=batch, odo=odo, want=oracle.run_mapping(x, off, odo)))
    streams = [seqs[0], seqs[1], seqs[0]]
    mappers = [lmono_amd.Mapper(gpu_ctx) for _ in streams]
    for k in range(8):
        q, t, st = lmono_amd.Mapper.process_batch(gpu_ctx, mappers, [s["batch"] for s in streams], [k] * 3,
                                                  np.array([s["odo"][k, :4] for s in streams]), np.array([s["odo"][k, 4:] for s in streams]))
        for i, s in enumerate(streams):
            ws = s["want"]["stats"][k]
            assert list(st[i, :6]) == [ws.n_edge[0], ws.n_edge[1], ws.n_plane[0], ws.n_plane[1], ws.lm_iters[0], ws.lm_iters[1]], (k, i)
            assert np.abs(np.concatenate([q[i], t[i]]) - s["want"]["poses"][k]).max() < 1e-7, (k, i)
    assert np.array_equal(mappers[0].cube(1, 10, 10, 5), mappers[2].cube(1, 10, 10, 5)) and len(mappers[0].cube(1, 10, 10, 5)) > 100


def _all_cubes(mapper):
    out = {}
    for which in (0, 1):
        for i in range(21):
            for j in range(21):
                for k in range(11):
                    c = mapper.cube(which, i, j, k)
                    if len(c):
                        out[(which, i, j, k)] = c
    return out


def test_device_tables_follow_the_host_tables_through_shifts_and_mode_changes(oracle, gpu_ctx):
    """lmono_mapper_process keeps the cube table on the device and plans the map update there (round 5); lmono_mapper_process_batch keeps it on the host.
    Three mappers see the same 10 frames, the odometry stretched so that the vehicle crosses cube borders (the array shifts several times): one through
    the device path only, one through the host path only (a batch of one), one alternating between the two (the table is converted on entry).  Poses,
    statistics and every cube must be identical, bit for bit."""
    import torch
    import lmono_amd
    w = oracle.S1World(n_az=500)
    traj = w.trajectory(10)
    x, off = w.scans(traj)
    xd = torch.from_numpy(x).cuda()
    batch = lmono_amd.ScanBatch(gpu_ctx, 10, len(x))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    _, odo = batch.odometry(n_chains=1, lead=0)
    odo = odo.copy()
    odo[:, 4] += np.arange(10) * 37.0            # 37 m per frame along x: the centre cube changes every other frame, the array shifts from frame 2 on
    odo[:, 5] -= np.arange(10) * 21.0
    dev, host, mixed = (lmono_amd.Mapper(gpu_ctx) for _ in range(3))
    for k in range(10):
        qd, td, sd = dev.process(batch, k, odo[k, :4], odo[k, 4:])
        qh, th, sh = lmono_amd.Mapper.process_batch(gpu_ctx, [host], [batch], [k], odo[k:k + 1, :4], odo[k:k + 1, 4:])
        if k % 3 == 1:
            qm, tm, sm = lmono_amd.Mapper.process_batch(gpu_ctx, [mixed], [batch], [k], odo[k:k + 1, :4], odo[k:k + 1, 4:])
            qm, tm, sm = qm[0], tm[0], sm[0]
        else:
            qm, tm, sm = mixed.process(batch, k, odo[k, :4], odo[k, 4:])
        assert np.array_equal(qd, qh[0]) and np.array_equal(td, th[0]) and list(sd[:7]) == list(sh[0][:7]), k
        assert np.array_equal(qd, qm) and np.array_equal(td, tm) and list(sd[:7]) == list(sm[:7]), k
    cd, ch, cm = _all_cubes(dev), _all_cubes(host), _all_cubes(mixed)
    assert len(cd) > 40 and cd.keys() == ch.keys() == cm.keys()
    for key in cd:
        assert np.array_equal(cd[key], ch[key]) and np.array_equal(cd[key], cm[key]), key
    # a reset empties the device's table as well
    dev.reset()
    assert not _all_cubes(dev)
    q0, t0, s0 = dev.process(batch, 0, odo[0, :4], odo[0, 4:])
    assert not s0[:6].any() and s0[6] == 0


def test_update_refused_on_the_device_is_reported_by_the_next_call(oracle, gpu_ctx):
    """With the cube table on the device the map update runs behind the call's return, so a capacity error found while it is planned (here: a cube that
    would exceed the voxel filter's 65536 points, provoked with a 1 cm leaf that filters nothing away) cannot be returned by the call that caused it:
    the NEXT call on the mapper reports it, the table keeps the map as it was before the refused update, and nothing hangs."""
    import torch
    import lmono_amd
    w = oracle.S1World(n_az=2000)
    traj = w.trajectory(6)
    x, off = w.scans(traj)
    xd = torch.from_numpy(x).cuda()
    batch = lmono_amd.ScanBatch(gpu_ctx, 6, len(x))
    batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
    _, odo = batch.odometry(n_chains=1, lead=0)
    m = lmono_amd.Mapper(gpu_ctx, 0.01, 0.01)
    sizes, failed_at = [], None
    for k in range(40):                      # the six scans again and again: ~16 k less-flat points per frame into the three centre cubes
        try:
            m.process(batch, k % 6, odo[k % 6, :4], odo[k % 6, 4:])
        except lmono_amd.LmonoError as e:
            failed_at = k
            assert "refused" in str(e) or "capacity" in str(e)
            break
        sizes.append(sum(len(m.cube(1, 10, 10, kk)) for kk in (4, 5, 6)))
    assert failed_at is not None and failed_at >= 2, (failed_at, sizes)      # a centre cube passes 65536 points after a dozen frames
    # the refused update left the table alone: the centre cubes hold what the last accepted update put there
    assert sum(len(m.cube(1, 10, 10, kk)) for kk in (4, 5, 6)) == sizes[-1] or len(sizes) < 2
    m.reset()
    q, t, st = m.process(batch, 0, odo[0, :4], odo[0, 4:])           # usable again
    assert not st[:6].any()


def test_arena_compaction_with_the_table_on_the_device(oracle, gpu_ctx):
    """The arenas are compacted on the host's copy of the cube table (rare: the bump pointer has to come near 6 Mi points).  LMONO_MAP_COMPACT_AT (a test
    hook, read once per process) makes it happen every other frame in a child process: table to the host, live cubes into the other arena half, table
    back to the device -- the trajectory and the cubes must be those of this process, which never compacts."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
import lmono_amd
from oracle import oracle as O
w = O.S1World(n_az=500)
traj = w.trajectory(8)
x, off = w.scans(traj)
ctx = lmono_amd.Context(0)
xd = torch.from_numpy(x).cuda()
batch = lmono_amd.ScanBatch(ctx, 8, len(x))
batch.scanreg(xd.data_ptr(), off, 64, 5.0, keepalive=xd)
_, odo = batch.odometry(n_chains=1, lead=0)
m = lmono_amd.Mapper(ctx)
out = []
for k in range(8):
    q, t, st = m.process(batch, k, odo[k, :4], odo[k, 4:])
    out.append(np.concatenate([q, t, st[:7].astype(np.float64)]))
cubes = [m.cube(wh, 10, 10, 5) for wh in (0, 1)]
np.savez(sys.argv[1], out=np.array(out), c0=cubes[0], c1=cubes[1])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("plain", {}), ("compact", {"LMONO_MAP_COMPACT_AT": "20000"})):
        path = os.path.join(root, "gpurun_out", "compact_%s.npz" % tag)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", code, path], cwd=root, env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(path)
    assert np.array_equal(res["plain"]["out"], res["compact"]["out"])
    assert len(res["plain"]["c1"]) > 100 and np.array_equal(res["plain"]["c0"], res["compact"]["c0"]) and np.array_equal(res["plain"]["c1"], res["compact"]["c1"])
