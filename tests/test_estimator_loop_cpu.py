"""The Estimator frame loop restated in the oracle (oracle/estimator_ref.py): state machine, keyframe decision, static frames,
marginalisation branches, loop re-anchoring, trajectory of record.  CPU only."""
import numpy as np

from tests import estimator_stream as S


def test_state_machine_and_trajectory(oracle):
    from workloads import s2
    from oracle import estimator_ref as E
    st = s2.make_stream(48, seed=1, stops=(30, 31))
    est, log = S.replay_oracle(st)
    stages = [r[1] for r in log]
    assert stages[:10] == [E.NOT_INITED] * 10 and all(s == E.INITED for s in stages[10:])          # INITED with the 11th frame (:442-459)
    tr = np.array(est.trajectory)
    assert tr.shape == (38, 8)                                                                    # one new_odometry row per INITED frame
    assert np.allclose(tr[:, 0], st["headers"][10:])                                              # Header[WINDOW_SIZE] = the newest frame
    # not exactly unit: Rs = rlc^T L0_R with the 8-digit laser_to_camera0 of the YAML, whose 3x3 block is a rotation to ~1e-8 only
    assert np.abs(np.linalg.norm(tr[:, 4:], axis=1) - 1).max() < 1e-7
    # static_status exactly on the frames where the LiDAR moved < 0.1 m (:259-265), and those solves still terminate
    static = [r[2] for r in log]
    assert [k for k, v in enumerate(static) if v] == [0, 30, 31]                                   # frame 0: |t - 0| < 0.1
    # both marginalisation branches ran; a MARGIN_SECOND_NEW elimination leaves [ex, pose0..pose8]
    kinds = [m for m in est.marg_log]
    assert any(f == E.MARGIN_OLD for f, _ in kinds) and any(f == E.MARGIN_SECOND_NEW and nb == 10 for f, nb in kinds)
    keyframes = [r[0] for r in log]
    assert 0 < sum(keyframes[10:]) < len(keyframes[10:])                                          # keyframes and non-keyframes both occur
    # the fused trajectory follows the ground truth (LiDAR increments with 1 cm / 0.05 deg noise chained over 38 frames)
    err = np.linalg.norm(tr[:, 1:4] - st["gt_P"][10:], axis=1)
    assert err.max() < 1.0 and err[:5].max() < 0.1
    # window bookkeeping invariants: every track starts inside the window and ends at most at the newest frame
    for f in est.feature:
        assert 0 <= f.start_frame <= E.WINDOW_SIZE and f.end_frame() <= E.WINDOW_SIZE and len(f.obs) >= 1


def test_second_new_drops_the_newest_lidar_frame_as_written(oracle):
    """all_image_frame.erase(end() - 1) (Estimator.cc:735) removes the newest frame's LiDAR pose on a non-keyframe: the restatement keeps
    that behaviour, so after such a slide the window holds 10 image frames whose last one is the SECOND-newest."""
    from workloads import s2
    from oracle import estimator_ref as E
    st = s2.make_stream(48, seed=1, stops=(30, 31))
    seen = []

    def on_frame(k, est):
        if est.stage_flag == E.INITED and est.marginalization_flag == E.MARGIN_SECOND_NEW:
            seen.append((k, len(est.frames), est.frames[-1][0], st["headers"][k - 1]))
    S.replay_oracle(st, on_frame=on_frame)
    assert seen, "the stream must contain a non-keyframe after initialisation"
    for k, n, last_header, prev_header in seen:
        assert n == 10 and last_header != st["headers"][k]


def test_loop_correction_reanchors_the_window_rigidly(oracle):
    """loopCorrection (:309-365): every window pose keeps its pose relative to the matched frame, which takes the corrected pose."""
    from workloads import s2
    from oracle import estimator_ref as E
    st = s2.make_stream(24, seed=2)
    e = S.loop_event(st, 18)
    snap = {}

    def on_frame(k, est):
        if k == 17:
            snap["Rs"] = [R.copy() for R in est.Rs]; snap["Ps"] = [P.copy() for P in est.Ps]; snap["H"] = list(est.Header)
    est = E.EstimatorRef(st["tlc"])
    est0, _ = S.replay_oracle(st, on_frame=on_frame)          # no loop: reference run
    est1, _ = S.replay_oracle(st, loops=[e])
    assert e["stamp"] in snap["H"][:E.WINDOW_SIZE]
    # the runs agree before the loop frame and differ after it
    t0 = np.array(est0.trajectory); t1 = np.array(est1.trajectory)
    assert np.array_equal(t0[:7], t1[:7]) and np.abs(t0[8:, 1:4] - t1[8:, 1:4]).max() > 1e-3
    # rigidity of the correction itself, on a copy of the window before frame 18
    est2 = E.EstimatorRef(st["tlc"])
    est2.Rs = [R.copy() for R in snap["Rs"]]; est2.Ps = [P.copy() for P in snap["Ps"]]; est2.Header = list(snap["H"])
    est2.setLoopFrame(e["stamp"], e["old_T"], e["old_Q"], e["correct_T"], e["correct_Q"])
    est2.loopCorrection()
    i = snap["H"].index(e["stamp"])
    assert np.allclose(est2.Ps[i], e["correct_T"])
    for j in range(E.WINDOW_SIZE + 1):
        rel_before = snap["Rs"][i].T @ (snap["Ps"][j] - snap["Ps"][i]); rel_after = est2.Rs[i].T @ (est2.Ps[j] - est2.Ps[i])
        # to the orthonormality of the window rotations (~1e-8: they inherit the YAML extrinsic's 8 digits)
        assert np.abs(rel_before - rel_after).max() < 1e-6
        assert np.abs(snap["Rs"][i].T @ snap["Rs"][j] - est2.Rs[i].T @ est2.Rs[j]).max() < 1e-6


def test_cxx_mirror_over_the_c_oracle_prints_the_python_replays_trajectory(oracle, tmp_path):
    """The C++ host mirror (lmono_amd/host/lmono_host.cpp, the code the product's frame loop runs) linked against the C oracle through
    oracle/cpu_shim.cpp instead of the HIP library -- oracle/estimator_seq_cpu, `bench.py --workload ba-seq`'s CPU baseline -- against the
    Python replay of the same oracle: the same decisions on every frame and the same trajectory (two host-side orchestrations of the same C
    numerics; they differ in nothing but the order of a few double additions)."""
    import os
    import subprocess
    from workloads import s2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "estimator_seq_cpu"])
    st = s2.make_stream(120, seed=0, stops=(60, 61))
    est, log = S.replay_oracle(st)
    fx = tmp_path / "stream.bin"
    s2.write_stream(fx, st)
    out = subprocess.run([os.path.join(root, "oracle", "estimator_seq_cpu"), str(fx), "-"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    odo = np.array([[float(v) for v in ln.split()[1:]] for ln in out.stdout.splitlines() if ln.startswith("ODO")])
    ref = np.array(est.trajectory)
    assert odo.shape == ref.shape
    assert np.abs(odo[:, 1:4] - ref[:, 1:4]).max() < 1e-6 and np.abs(np.abs(odo[:, 4:8]) - np.abs(ref[:, 4:8])).max() < 1e-7
    frm = [ln.split()[1:] for ln in out.stdout.splitlines() if ln.startswith("FRM")]
    assert len(frm) == len(log)
    for k, (row, r) in enumerate(zip(frm, log)):
        assert (int(row[1]), int(row[2]), int(row[3])) == (r[0], r[1], r[2]) and (int(row[7]), int(row[8])) == (r[6], r[7]), "frame %d" % k


def _split_streams(stdout):
    """estimator_seq streams=N output -> {stream: [its lines]}, {stream: digest}"""
    cur, by, dig = None, {}, {}
    for ln in stdout.splitlines():
        if ln.startswith("STR "):
            cur = int(ln.split()[1]); by[cur] = []
        elif ln.startswith("DIG "):
            dig[int(ln.split()[1])] = ln.split()[2]
        elif ln.startswith(("TIM", "FLP")):
            continue
        elif cur is not None:
            by[cur].append(ln)
    return by, dig


def test_lockstep_batch_of_estimators_prints_every_streams_own_lines(oracle, tmp_path):
    """EstimatorBatch: N independent Estimators stepped in lock-step, every numeric step ONE C-ABI call over the N windows, the host
    halves on a thread pool.  Host logic only here (the C ABI is the CPU shim over the oracle): five streams replaying three different stream files --
    static stretches, loop events, keyframe / non-keyframe frames that differ between the streams, so the MARGIN_OLD and MARGIN_SECOND_NEW groups of
    a frame are both non-empty and change from frame to frame -- print, stream by stream, exactly the lines of the single-stream runs of their
    files, with marginalisation inline and overlapped."""
    import os
    import subprocess
    from workloads import s2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "estimator_seq_cpu"])
    exe = os.path.join(root, "oracle", "estimator_seq_cpu")
    files, single = [], []
    for k, (seed, stops) in enumerate(((2, (20, 21, 30)), (0, ()), (3, (25,)))):
        st = s2.make_stream(48, seed=seed, stops=stops)
        loops = [S.loop_event(st, 30 + 3 * k)] if k != 1 else []
        for e in loops:                         # (a fixed corrected pose: the event needs no live window here)
            e["correct_T"] = np.array([0.1 * k, 0.2, 0.3]); e["correct_Q"] = np.array([1.0, 0.0, 0.001, 0.0])
        fx = tmp_path / ("s%d.bin" % k)
        S.write_stream(fx, st, loops)
        files.append(str(fx))
        out = subprocess.run([exe, str(fx), "-"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        single.append([ln for ln in out.stdout.splitlines() if not ln.startswith(("TIM", "FLP", "DIG"))])
        assert [ln for ln in out.stdout.splitlines() if ln.startswith("DIG")]
    kinds = set()
    for lines in single:
        kinds |= {ln.split()[2] for ln in lines if ln.startswith("FRM")}
    assert kinds == {"0", "1"}                  # keyframes and non-keyframes both occur
    for mode in ("sync", "async"):
        out = subprocess.run([exe, files[0], "-", mode, "streams=5", files[1], files[2]], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, LMONO_HOST_THREADS="3"))
        assert out.returncode == 0, out.stderr[-2000:]
        by, dig = _split_streams(out.stdout)
        assert sorted(by) == [0, 1, 2, 3, 4] and len(dig) == 5
        for s in range(5):
            assert by[s] == single[s % 3], "%s: stream %d differs from the single-stream run of its file" % (mode, s)
        assert dig[0] == dig[3] and dig[1] == dig[4] and len({dig[0], dig[1], dig[2]}) == 3
    # digest-only output (what the bench reads for many streams) carries the same digests
    out = subprocess.run([exe, files[0], "-", "streams=5", "digest", files[1], files[2]], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith(("FRM", "ODO"))]
    assert _split_streams(out.stdout)[1] == dig
    # the same streams as two independent lock-step groups on two host threads (own contexts): same digests
    out = subprocess.run([exe, files[0], "-", "async", "streams=5", "groups=2", "digest", files[1], files[2]], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert _split_streams(out.stdout)[1] == dig


def test_lockstep_loop_under_thread_sanitizer(oracle, tmp_path):
    """The lock-step frame loop is host threads around the numeric calls: the pool that walks the streams (every thread owns a share of them and takes
    from the others' when done), the marginalisation worker beside the next frame, two batches interleaved by one driving thread.  The host mirror over the
    oracle shim, built with -fsanitize=thread: 24 frames x 8 streams x 2 groups on 4 threads must finish without a report (a report makes the run fail:
    halt_on_error) and print one digest for the eight streams of one file.  (GPU AddressSanitizer / XNACK are not available: sanitizers run on the CPU build.)"""
    import os
    import shutil
    import subprocess
    from workloads import s2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None:
        import pytest
        pytest.skip("no g++")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "estimator_seq_cpu"])          # (cpu_shim_stubs.o, liblmono_oracle.so)
    host = os.path.join(root, "lmono_amd", "host")
    exe = str(tmp_path / "eseq_tsan")
    cmd = ["g++", "-O1", "-g", "-fsanitize=thread", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17", "-pthread", "-I" + host,
           os.path.join(root, "oracle", "cpu_shim.cpp"), os.path.join(host, "lmono_host.cpp"), os.path.join(host, "estimator_seq.cpp"), os.path.join(host, "kitti_io.cpp"),
           os.path.join(root, "oracle", "cpu_shim_stubs.o"), "-o", exe, "-L" + os.path.join(root, "oracle"), "-llmono_oracle", "-Wl,-rpath," + os.path.join(root, "oracle"), "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("tsan" in b.stderr or "sanitize" in b.stderr):
        import pytest
        pytest.skip("no ThreadSanitizer runtime: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    fx = tmp_path / "s.bin"
    s2.write_stream(str(fx), s2.make_stream(24, seed=2, stops=()))
    env = dict(os.environ, LMONO_HOST_THREADS="4", TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    out = subprocess.run([exe, str(fx), "-", "async", "streams=8", "groups=2", "digest"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
    digs = {ln.split()[2] for ln in out.stdout.splitlines() if ln.startswith("DIG")}
    assert len(digs) == 1 and sum(ln.startswith("DIG") for ln in out.stdout.splitlines()) == 8
