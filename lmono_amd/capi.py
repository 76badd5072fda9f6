"""ctypes binding of include/lmono_hip.h (the drop-in C ABI).  Fails loudly when the library is absent."""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_QUERIES = 64 * 6 * 2 + 64 * 6 * 4

# every symbol include/lmono_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "lmono_create", "lmono_destroy", "lmono_last_error", "lmono_set_stream", "lmono_use_own_stream", "lmono_set_option", "lmono_get_option", "lmono_synchronize", "lmono_version",
    "lmono_batch_create", "lmono_batch_destroy", "lmono_scanreg_batch", "lmono_scanreg_batch_h", "lmono_host_alloc", "lmono_host_free", "lmono_batch_stage_h", "lmono_scanreg_batch_staged", "lmono_batch_counts", "lmono_batch_get_cloud",
    "lmono_batch_get_curvature", "lmono_odom_batch", "lmono_odom_batch_d", "lmono_odom_shard_d", "lmono_odom_shard_main_d", "lmono_odom_shard_validate", "lmono_odom_boundary_report", "lmono_odom_stream_create", "lmono_odom_stream_destroy", "lmono_odom_step", "lmono_odom_stream_scan", "lmono_odom_correspond", "lmono_timing_reset", "lmono_timing_read",
    "lmono_pose_prefix_d", "lmono_pose_rebase_d", "lmono_map_refine", "lmono_voxel_filter", "lmono_mapper_create", "lmono_mapper_destroy", "lmono_mapper_reset", "lmono_mapper_process", "lmono_mapper_process_batch", "lmono_mapper_cube",
    "lmono_map_builder_create", "lmono_map_builder_destroy", "lmono_associate_to_map", "lmono_associate_to_map_batch", "lmono_map_builder_depth",
    "lmono_map_builder_cloud", "lmono_map_builder_map", "lmono_map_builder_clear",
    "lmono_pose_graph_create", "lmono_pose_graph_destroy", "lmono_pose_graph_reset", "lmono_pose_graph_info", "lmono_pose_graph_reduce_buffer", "lmono_pose_graph_set_reduce_buffer", "lmono_pose_graph_linearise",
    "lmono_pose_graph_step", "lmono_pose_graph_optimize", "lmono_pose_graph_result", "lmono_factor_eval", "lmono_factor_eval_d", "lmono_factor_eval_blocks", "lmono_factor_eval_blocks_d",
    "lmono_triangulate", "lmono_outlier_scores", "lmono_shift_depth", "lmono_shift_depth_batch", "lmono_marginalize", "lmono_marg_evaluate", "lmono_marg_second_new", "lmono_ba_batch_create", "lmono_ba_batch_update", "lmono_ba_batch_destroy", "lmono_ba_solve", "lmono_ba_batch_reset", "lmono_ba_batch_read", "lmono_debug_bounds",
]


class LmonoError(RuntimeError):
    pass


def lib_path():
    # LMONO_HIP_LIB: a diagnostic build of the same sources (scripts/prof_tile.py); the product path is the in-tree library
    return os.environ.get("LMONO_HIP_LIB") or os.path.join(_HERE, "lib", "liblmono_hip.so")


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise LmonoError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
    L = C.CDLL(p)
    L.lmono_create.restype = C.c_void_p
    L.lmono_create.argtypes = [C.c_int]
    L.lmono_destroy.argtypes = [C.c_void_p]
    L.lmono_last_error.restype = C.c_char_p
    L.lmono_last_error.argtypes = [C.c_void_p]
    L.lmono_version.restype = C.c_char_p
    L.lmono_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.lmono_use_own_stream.argtypes = [C.c_void_p]
    L.lmono_synchronize.argtypes = [C.c_void_p]
    L.lmono_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.lmono_get_option.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.lmono_batch_create.restype = C.c_void_p
    L.lmono_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    L.lmono_batch_destroy.argtypes = [C.c_void_p]
    L.lmono_scanreg_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.lmono_scanreg_batch_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.lmono_host_alloc.restype = C.c_void_p
    L.lmono_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.lmono_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.lmono_batch_stage_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.lmono_scanreg_batch_staged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.lmono_batch_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.lmono_batch_get_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.lmono_batch_get_curvature.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.lmono_odom_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.lmono_odom_batch_d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.lmono_odom_shard_d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lmono_odom_shard_main_d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.lmono_odom_shard_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.lmono_odom_boundary_report.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.lmono_odom_stream_create.restype = C.c_void_p
    L.lmono_odom_stream_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]
    L.lmono_odom_stream_destroy.argtypes = [C.c_void_p]
    L.lmono_odom_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.lmono_odom_stream_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.lmono_odom_correspond.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.lmono_timing_reset.argtypes = [C.c_void_p]
    L.lmono_triangulate.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_double, C.c_int]
    L.lmono_outlier_scores.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_double, C.c_void_p]
    L.lmono_shift_depth.argtypes = [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 3
    L.lmono_shift_depth_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    L.lmono_marginalize.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 14
    L.lmono_marg_evaluate.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    L.lmono_marg_second_new.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
    L.lmono_ba_batch_create.restype = C.c_void_p
    L.lmono_ba_batch_create.argtypes = [C.c_void_p, C.c_void_p]
    L.lmono_ba_batch_destroy.argtypes = [C.c_void_p]
    L.lmono_ba_batch_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.lmono_ba_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.lmono_ba_batch_reset.argtypes = [C.c_void_p, C.c_void_p]
    L.lmono_ba_batch_read.argtypes = [C.c_void_p] * 6
    L.lmono_debug_bounds.argtypes = [C.c_void_p, C.c_void_p]
    L.lmono_factor_eval.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.lmono_factor_eval_d.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.lmono_factor_eval_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
    L.lmono_factor_eval_blocks_d.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
    L.lmono_pose_prefix_d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.lmono_pose_rebase_d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.lmono_timing_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _lib = L
    return L


class Context:
    """lmono_ctx wrapper.  One per process / GPU."""

    def __init__(self, device=0):
        self.L = load_library()
        self._own_stream = False          # the context starts on the null stream
        # objects created on this context (batches, streams, mappers ...): closed before the context whatever order the
        # interpreter drops them in -- their destroy calls touch the context's stream
        self._children = weakref.WeakSet()
        self.h = self.L.lmono_create(int(device))
        if not self.h:
            raise LmonoError("lmono_create(%d) failed: no usable HIP device" % device)
        self.device = device
        if os.environ.get("LMONO_CORR_TILE") is not None:       # A/B switches for measurements
            self.L.lmono_set_option(self.h, 0, int(os.environ["LMONO_CORR_TILE"]))
        if os.environ.get("LMONO_LEAD_FULL") is not None:
            self.L.lmono_set_option(self.h, 3, int(os.environ["LMONO_LEAD_FULL"]))
        if os.environ.get("LMONO_ODOM_STREAMS") is not None:
            self.L.lmono_set_option(self.h, 2, int(os.environ["LMONO_ODOM_STREAMS"]))
        if os.environ.get("LMONO_BOUNDARY_TOL") is not None:
            self.L.lmono_set_option(self.h, 4, int(os.environ["LMONO_BOUNDARY_TOL"]))

    def check(self, rc):
        if rc < 0:
            raise LmonoError("lmono error %d: %s" % (rc, self.L.lmono_last_error(self.h).decode()))
        return rc

    def last_error(self):
        return self.L.lmono_last_error(self.h).decode()

    def set_stream(self, raw_stream):
        self.check(self.L.lmono_set_stream(self.h, C.c_void_p(raw_stream)))
        self._own_stream = bool(raw_stream)

    def use_own_stream(self):
        """Run on a non-blocking stream of the library's (two contexts on two host threads then overlap)."""
        self.check(self.L.lmono_use_own_stream(self.h))
        self._own_stream = True

    def synchronize(self):
        self.check(self.L.lmono_synchronize(self.h))

    OPT_CORR_TILE = 0
    OPT_DEFER_EVERY = 1
    OPT_ODOM_STREAMS = 2
    OPT_LEAD_FULL = 3
    OPT_BOUNDARY_TOL = 4
    OPT_BA_CLUSTER = 5
    OPT_LEAD_SEED = 6

    def set_option(self, key, value):
        self.check(self.L.lmono_set_option(self.h, int(key), int(value)))

    def get_option(self, key):
        v = C.c_int(0)
        self.check(self.L.lmono_get_option(self.h, int(key), C.byref(v)))
        return int(v.value)

    def odom_chain_groups(self, n_chains):
        """Chain groups (HIP streams) an odometry call with n_chains chains runs on: the rule of odom_run in lmono_hip.hip."""
        g = max(1, min(8, self.get_option(self.OPT_ODOM_STREAMS)))
        if self.get_option(self.OPT_CORR_TILE) != 3:
            return 1
        if self._own_stream:
            g = min(g, 3)
        while g > 1 and n_chains // g < 32:
            g -= 1
        return g

    def host_alloc(self, nbytes):
        """Pinned host memory from the library (lmono_host_alloc) as a numpy uint8 array; free with host_free(array)."""
        p = self.L.lmono_host_alloc(self.h, int(nbytes))
        if not p:
            raise LmonoError("lmono_host_alloc(%d) failed: %s" % (nbytes, self.last_error()))
        arr = np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(p))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self.L.lmono_host_free(self.h, C.c_void_p(p))

    def timing_reset(self):
        self.check(self.L.lmono_timing_reset(self.h))

    def timing(self):
        """Summed device ms per kernel group since timing_reset(): dict + call counts."""
        ms = np.zeros(52)
        nr, no = C.c_int(0), C.c_int(0)
        self.check(self.L.lmono_timing_read(self.h, ms.ctypes.data, 52, C.byref(nr), C.byref(no)))
        self.diag = ms[13:].copy()
        names = ["frontend_total", "odometry_total", "k_ring_sort", "k_curvature", "k_select", "k_voxel", "k_compact",
                 "k_grid_build", "k_line_index", "k_correspond", "k_lm_solve", "odometry_launch_pairs", "deferred_features"]
        return dict(zip(names, ms.tolist())), nr.value, no.value

    FACTOR_DIMS = {0: (14, 24, 36, 6, 84), 1: (22, 4, 4, 2, 44), 2: (7, 16, 2, 6, 42), 3: (1, 44, 1, 2, 2)}

    def factor_eval(self, kind, params, consts, info, want_jac=True):
        """Batched Evaluate of one BA factor kind on the GPU (host arrays in / out)."""
        npar, ncon, ninf, nres, njac = self.FACTOR_DIMS[kind]
        params = np.ascontiguousarray(params, np.float64).reshape(-1, npar)
        consts = np.ascontiguousarray(consts, np.float64).reshape(-1, ncon)
        info = np.ascontiguousarray(info, np.float64).reshape(ninf)
        n = len(params)
        r = np.zeros((n, nres)); J = np.zeros((n, njac)) if want_jac else None
        self.check(self.L.lmono_factor_eval(self.h, kind, n, params.ctypes.data, consts.ctypes.data, info.ctypes.data,
                                            r.ctypes.data, J.ctypes.data if want_jac else None))
        return r, J

    def factor_eval_blocks(self, kind, params, consts, info, block_mask, J_init=None):
        """Evaluate with ceres' per-block contract: block_mask [n] uint8, bit k = jacobians[k] != NULL.  J starts as J_init (or NaN) so
        that the caller can see which blocks were written."""
        npar, ncon, ninf, nres, njac = self.FACTOR_DIMS[kind]
        params = np.ascontiguousarray(params, np.float64).reshape(-1, npar)
        consts = np.ascontiguousarray(consts, np.float64).reshape(-1, ncon)
        info = np.ascontiguousarray(info, np.float64).reshape(ninf)
        n = len(params)
        mask = np.ascontiguousarray(block_mask, np.uint8).reshape(n)
        r = np.zeros((n, nres))
        J = np.full((n, njac), np.nan) if J_init is None else np.ascontiguousarray(J_init, np.float64).copy()
        self.check(self.L.lmono_factor_eval_blocks(self.h, kind, n, params.ctypes.data, consts.ctypes.data, info.ctypes.data,
                                                   r.ctypes.data, J.ctypes.data, mask.ctypes.data))
        return r, J

    def factor_eval_d(self, kind, count, params_ptr, consts_ptr, info_ptr, r_ptr, J_ptr=None):
        self.check(self.L.lmono_factor_eval_d(self.h, kind, count, C.c_void_p(params_ptr), C.c_void_p(consts_ptr),
                                              C.c_void_p(info_ptr), C.c_void_p(r_ptr), C.c_void_p(J_ptr or 0)))

    @staticmethod
    def _pack_windows(windows):
        """windows: list of dicts(Rs [n,3,3], Ps [n,3], tlc 4x4, trk_start, trk_off, trk_pts)."""
        W = len(windows)
        Rs = np.zeros((W, 11, 9)); Ps = np.zeros((W, 11, 3))
        for k, w in enumerate(windows):
            n = len(w["Rs"]); Rs[k, :n] = np.asarray(w["Rs"]).reshape(n, 9); Ps[k, :n] = w["Ps"]
        tlc = np.ascontiguousarray([np.asarray(w["tlc"]).ravel() for w in windows], np.float64)
        feat_off = np.concatenate([[0], np.cumsum([len(w["trk_start"]) for w in windows])]).astype(np.int32)
        start = np.ascontiguousarray(np.concatenate([w["trk_start"] for w in windows]), np.int32)
        offs, base = [0], 0
        for w in windows:
            offs.extend((np.asarray(w["trk_off"][1:]) + base).tolist()); base += int(w["trk_off"][-1])
        obs_off = np.array(offs, np.int32)
        pts = np.ascontiguousarray(np.concatenate([np.asarray(w["trk_pts"]).reshape(-1, 2) for w in windows]), np.float64)
        return W, feat_off, Rs, Ps, tlc, start, obs_off, pts

    def triangulate(self, windows, depth, track_cnt=3, window_size=10, weight=1500.0, refine_iters=50):
        W, feat_off, Rs, Ps, tlc, start, obs_off, pts = self._pack_windows(windows)
        d = np.ascontiguousarray(depth, np.float64).copy(); flag = np.zeros(len(d), np.int32)
        self.check(self.L.lmono_triangulate(self.h, W, feat_off.ctypes.data, Rs.ctypes.data, Ps.ctypes.data, tlc.ctypes.data, start.ctypes.data,
                                            obs_off.ctypes.data, pts.ctypes.data, d.ctypes.data, flag.ctypes.data, track_cnt, window_size, weight, refine_iters))
        return d, flag

    def outlier_scores(self, windows, depth, track_cnt=3, weight=1500.0):
        W, feat_off, Rs, Ps, tlc, start, obs_off, pts = self._pack_windows(windows)
        d = np.ascontiguousarray(depth, np.float64); sc = np.zeros(len(d))
        self.check(self.L.lmono_outlier_scores(self.h, W, feat_off.ctypes.data, Rs.ctypes.data, Ps.ctypes.data, tlc.ctypes.data, start.ctypes.data,
                                               obs_off.ctypes.data, pts.ctypes.data, d.ctypes.data, track_cnt, weight, sc.ctypes.data))
        return sc

    def shift_depth(self, back_R0, back_P0, R1, P1, tlc, pt_i, depth):
        a = [np.ascontiguousarray(v, np.float64).ravel() for v in (back_R0, back_P0, R1, P1, tlc)]
        pt = np.ascontiguousarray(pt_i, np.float64).reshape(-1, 2); d = np.ascontiguousarray(depth, np.float64); out = np.zeros(len(d))
        self.check(self.L.lmono_shift_depth(self.h, *[v.ctypes.data for v in a], len(d), pt.ctypes.data, d.ctypes.data, out.ctypes.data))
        return out

    def shift_depth_batch(self, frames, pt_i_list, depth_list):
        """frames: [n][40] (back_R0, back_P0, R1, P1, TLC per window); pt_i_list / depth_list: per window arrays.  Returns the list of shifted depths."""
        fr = np.ascontiguousarray(frames, np.float64).reshape(-1, 40)
        off = np.concatenate([[0], np.cumsum([len(d) for d in depth_list])]).astype(np.int32)
        pt = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float64).reshape(-1, 2) for p in pt_i_list]) if off[-1] else np.zeros((0, 2)), np.float64)
        d = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float64).ravel() for v in depth_list]) if off[-1] else np.zeros(0), np.float64)
        out = np.zeros(len(d))
        self.check(self.L.lmono_shift_depth_batch(self.h, len(fr), fr.ctypes.data, off.ctypes.data, pt.ctypes.data, d.ctypes.data, out.ctypes.data))
        return [out[off[k]:off[k + 1]] for k in range(len(fr))]

    def debug_bounds(self):
        """(hits, line of the first, byte offset of the first, block of the first) of a -DLMONO_BOUNDS build; raises LmonoError on the product build."""
        out = (C.c_ulonglong * 4)()
        self.check(self.L.lmono_debug_bounds(self.h, out))
        return tuple(int(v) for v in out)

    def marginalize(self, windows):
        """windows: list of dicts(poses [11,7], ex [7], invd [F0], obs_feat, obs_j, pts [O,4], laser01 [24], laser_info, mono_info)."""
        W = len(windows)
        feat_off = np.concatenate([[0], np.cumsum([len(w["invd"]) for w in windows])]).astype(np.int32)
        obs_off = np.concatenate([[0], np.cumsum([len(w["obs_j"]) for w in windows])]).astype(np.int32)
        cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(w[k], dt).reshape(len(w[k]), -1) for w in windows]).ravel(), dt)
        poses = np.ascontiguousarray([w["poses"] for w in windows], np.float64); ex = np.ascontiguousarray([w["ex"] for w in windows], np.float64)
        invd = cat("invd", np.float64); of = cat("obs_feat", np.int32); oj = cat("obs_j", np.int32); pts = cat("pts", np.float64)
        l01 = np.ascontiguousarray([w["laser01"] for w in windows], np.float64)
        li = np.ascontiguousarray(windows[0]["laser_info"], np.float64); mi = np.ascontiguousarray(windows[0]["mono_info"], np.float64)
        J = np.zeros((W, 66, 66)); r = np.zeros((W, 66)); st = np.zeros(W, np.int32)
        self.check(self.L.lmono_marginalize(self.h, W, feat_off.ctypes.data, obs_off.ctypes.data, poses.ctypes.data, ex.ctypes.data, invd.ctypes.data,
                                            of.ctypes.data, oj.ctypes.data, pts.ctypes.data, l01.ctypes.data, li.ctypes.data, mi.ctypes.data,
                                            J.ctypes.data, r.ctypes.data, st.ctypes.data))
        return J, r, st

    def marg_evaluate(self, lin_J, lin_r, x0, x):
        lin_J = np.ascontiguousarray(lin_J, np.float64); lin_r = np.ascontiguousarray(lin_r, np.float64)
        x0 = np.ascontiguousarray(x0, np.float64); x = np.ascontiguousarray(x, np.float64)
        W = len(lin_r); res = np.zeros((W, 66))
        self.check(self.L.lmono_marg_evaluate(self.h, W, lin_J.ctypes.data, lin_r.ctypes.data, x0.ctypes.data, x.ctypes.data, res.ctypes.data))
        return res

    def marg_second_new(self, lin_J, lin_r, x0, x, drop_block):
        """MARGIN_SECOND_NEW: lin_J [W, n0, n0], lin_r [W, n0], x0 / x [W, nb, 7] -> (J [W, n, n], r [W, n], status [W]), n = n0 - 6."""
        lin_J = np.ascontiguousarray(lin_J, np.float64); lin_r = np.ascontiguousarray(lin_r, np.float64)
        x0 = np.ascontiguousarray(x0, np.float64); x = np.ascontiguousarray(x, np.float64)
        W, nb = x.shape[0], x.shape[1]
        n = 6 * nb - 6
        J = np.zeros((W, n, n)); r = np.zeros((W, n)); st = np.zeros(W, np.int32)
        self.check(self.L.lmono_marg_second_new(self.h, W, nb, int(drop_block), lin_J.ctypes.data, lin_r.ctypes.data, x0.ctypes.data, x.ctypes.data,
                                                J.ctypes.data, r.ctypes.data, st.ctypes.data))
        return J, r, st

    def pose_prefix_d(self, incr_ptr, first, n, poses_ptr):
        self.check(self.L.lmono_pose_prefix_d(self.h, C.c_void_p(incr_ptr), first, n, C.c_void_p(poses_ptr)))

    def pose_rebase_d(self, bases_ptr, n_bases, poses_ptr, n):
        self.check(self.L.lmono_pose_rebase_d(self.h, C.c_void_p(bases_ptr or 0), n_bases, C.c_void_p(poses_ptr), n))

    def close(self):
        if getattr(self, "h", None):
            for child in list(self._children):
                child.close()
            self.L.lmono_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


    def voxel_filter(self, clouds, leafs):
        """pcl::VoxelGrid on a list of [n,4] float32 clouds (one leaf size each): list of filtered clouds."""
        arrs = [np.ascontiguousarray(a, np.float32).reshape(-1, 4) for a in clouds]
        off = np.zeros(len(arrs) + 1, np.int64)
        off[1:] = np.cumsum([len(a) for a in arrs])
        cat = np.concatenate(arrs) if off[-1] > 0 else np.zeros((0, 4), np.float32)
        leaf = np.ascontiguousarray(leafs, np.float32)
        out = np.zeros_like(cat) if len(cat) else np.zeros((1, 4), np.float32)
        oo = np.zeros(len(arrs) + 1, np.int64)
        self.L.lmono_voxel_filter.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        self.check(self.L.lmono_voxel_filter(self.h, len(arrs), cat.ctypes.data, off.ctypes.data, leaf.ctypes.data, out.ctypes.data, oo.ctypes.data))
        return [out[oo[k]:oo[k + 1]].copy() for k in range(len(arrs))]

    def map_refine(self, corner_maps, surf_maps, corner_stacks, surf_stacks, poses_qt, want_nn=False):
        """Scan-to-map optimisation step of laserMapping for a batch of independent streams: lists of [n,4] float32 clouds
        per stream, poses_qt [n_streams,7] (q xyzw, t) initial guesses.  Returns (poses [n_streams,7], stats
        [n_streams,8], nn [total stack points,5] or None)."""
        ns = len(corner_maps)

        def cat(lst):
            arrs = [np.ascontiguousarray(a, np.float32).reshape(-1, 4) for a in lst]
            off = np.zeros(ns + 1, np.int64)
            off[1:] = np.cumsum([len(a) for a in arrs])
            return (np.concatenate(arrs) if off[-1] > 0 else np.zeros((0, 4), np.float32)), off
        cm, cmo = cat(corner_maps); sm, smo = cat(surf_maps); cs, cso = cat(corner_stacks); ss, sso = cat(surf_stacks)
        poses = np.ascontiguousarray(poses_qt, np.float64).reshape(ns, 7).copy()
        stats = np.zeros((ns, 8), np.int32)
        nn = np.zeros((int(cso[-1] + sso[-1]), 5), np.int32) if want_nn else None
        self.L.lmono_map_refine.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 11
        self.check(self.L.lmono_map_refine(self.h, ns, cm.ctypes.data, cmo.ctypes.data, sm.ctypes.data, smo.ctypes.data,
                                           cs.ctypes.data, cso.ctypes.data, ss.ctypes.data, sso.ctypes.data,
                                           poses.ctypes.data, stats.ctypes.data, nn.ctypes.data if want_nn else None))
        return poses, stats, nn

class OdomStream:
    """Online laserOdometry (lmono_odom_stream): one scan per step(), the previous scan's features stay on the device."""

    def __init__(self, ctx, max_points, n_lines=64, min_range=5.0, history=8):
        self.ctx = ctx
        ctx._children.add(self)
        self.h = ctx.L.lmono_odom_stream_create(ctx.h, int(max_points), int(n_lines), float(min_range), int(history))
        if not self.h:
            raise LmonoError("lmono_odom_stream_create failed: %s" % ctx.L.lmono_last_error(ctx.h).decode())

    def step(self, xyzi=None, dev_ptr=None, n_points=None, warm_start=None):
        """xyzi: [n,4] float32 host array, or dev_ptr + n_points for a scan resident in HBM.  Returns (incr [7] = q_last_curr xyzw +
        t_last_curr, pose [7] = q_w_curr + t_w_curr, info [8])."""
        incr = np.zeros(7); pose = np.zeros(7); info = np.zeros(8, np.int32)
        use = 0
        if warm_start is not None:
            incr[:] = np.asarray(warm_start, np.float64); use = 1
        if dev_ptr is None:
            xyzi = np.ascontiguousarray(xyzi, np.float32)
            ptr, n, on_dev = xyzi.ctypes.data, len(xyzi), 0
        else:
            ptr, n, on_dev = dev_ptr, int(n_points), 1
        self.ctx.check(self.ctx.L.lmono_odom_step(self.ctx.h, self.h, C.c_void_p(ptr), n, on_dev, use, incr.ctypes.data, incr.ctypes.data + 32,
                                                  pose.ctypes.data, pose.ctypes.data + 32, info.ctypes.data))
        return incr, pose, info

    def scan(self):
        """(raw lmono_scan_batch handle, scan index) of the newest scan, for Mapper.process_raw / cloud()."""
        bh = C.c_void_p(0); sc = C.c_int(0)
        self.ctx.check(self.ctx.L.lmono_odom_stream_scan(self.h, C.byref(bh), C.byref(sc)))
        return bh.value, sc.value

    def cloud(self, which, cap):
        bh, sc = self.scan()
        out = np.zeros((max(cap, 1), 4), np.float32)
        n = self.ctx.check(self.ctx.L.lmono_batch_get_cloud(self.ctx.h, C.c_void_p(bh), sc, which, out.ctypes.data, cap))
        return out[:n]

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.lmono_odom_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BoundaryReport(C.Structure):
    """lmono_boundary_report (include/lmono_hip.h)"""
    _fields_ = [("n_chains", C.c_int), ("flagged", C.c_int), ("chains_rerun", C.c_int), ("pairs_rerun", C.c_int), ("rounds", C.c_int),
                ("unresolved", C.c_int), ("tol", C.c_double), ("max_resid", C.c_double), ("repair_ms", C.c_double)]


class ScanBatch:
    """Device-resident working set of a batch of scans (lmono_scan_batch)."""

    def __init__(self, ctx, n_scans_cap, total_points_cap):
        self.ctx = ctx
        ctx._children.add(self)
        self.h = ctx.L.lmono_batch_create(ctx.h, int(n_scans_cap), int(total_points_cap))
        if not self.h:
            raise LmonoError("lmono_batch_create failed: %s" % ctx.L.lmono_last_error(ctx.h).decode())
        self.n_scans = 0
        self._keep = None

    def scanreg(self, xyzi_dev_ptr, offsets, n_lines=64, min_range=5.0, keepalive=None):
        """xyzi_dev_ptr: raw device pointer of a [total,4] float32 array resident in HBM."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        self.n_scans = len(offsets) - 1
        self._keep = keepalive
        self.ctx.check(self.ctx.L.lmono_scanreg_batch(self.ctx.h, self.h, C.c_void_p(xyzi_dev_ptr), offsets.ctypes.data,
                                                       self.n_scans, int(n_lines), float(min_range)))

    def scanreg_host(self, xyzi, offsets, n_lines=64, min_range=5.0):
        """xyzi: [total,4] float32 numpy array in host memory (KITTI .bin layout): staged to HBM by the library."""
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        offsets = np.ascontiguousarray(offsets, np.int64)
        self.n_scans = len(offsets) - 1
        self._keep = xyzi
        self.ctx.check(self.ctx.L.lmono_scanreg_batch_h(self.ctx.h, self.h, xyzi.ctypes.data, offsets.ctypes.data,
                                                         self.n_scans, int(n_lines), float(min_range)))

    def stage_host(self, host_ptr, total_points, keepalive=None):
        """Asynchronous H2D of a working set (pinned host memory at raw address host_ptr) into this batch's staging buffer."""
        self._keep = keepalive
        self.ctx.check(self.ctx.L.lmono_batch_stage_h(self.ctx.h, self.h, C.c_void_p(host_ptr), int(total_points)))

    def scanreg_staged(self, offsets, n_lines=64, min_range=5.0):
        offsets = np.ascontiguousarray(offsets, np.int64)
        self.n_scans = len(offsets) - 1
        self.ctx.check(self.ctx.L.lmono_scanreg_batch_staged(self.ctx.h, self.h, offsets.ctypes.data, self.n_scans, int(n_lines), float(min_range)))

    def counts(self):
        out = np.zeros((self.n_scans, 6), np.int32)
        self.ctx.check(self.ctx.L.lmono_batch_counts(self.ctx.h, self.h, out.ctypes.data))
        return out

    def cloud(self, scan, which, cap):
        out = np.zeros((max(cap, 1), 4), np.float32)
        n = self.ctx.check(self.ctx.L.lmono_batch_get_cloud(self.ctx.h, self.h, scan, which, out.ctypes.data, cap))
        return out[:n]

    def curvature(self, scan, cap):
        cv = np.zeros(max(cap, 1), np.float32)
        lb = np.zeros(max(cap, 1), np.int32)
        n = self.ctx.check(self.ctx.L.lmono_batch_get_curvature(self.ctx.h, self.h, scan, cv.ctypes.data, lb.ctypes.data, cap))
        return cv[:n], lb[:n]

    def odometry(self, n_chains=1, lead=0):
        incr = np.zeros((self.n_scans, 7))
        poses = np.zeros((self.n_scans, 7))
        self.ctx.check(self.ctx.L.lmono_odom_batch(self.ctx.h, self.h, n_chains, lead, incr.ctypes.data, poses.ctypes.data))
        return incr, poses

    def odometry_d(self, n_chains, lead, incr_ptr=None, poses_ptr=None):
        self.ctx.check(self.ctx.L.lmono_odom_batch_d(self.ctx.h, self.h, n_chains, lead,
                                                      C.c_void_p(incr_ptr or 0), C.c_void_p(poses_ptr or 0)))

    def odometry_shard_d(self, n_chains, lead, first_owned, incr_ptr=None):
        """Rank-local odometry of a scan-range shard: the first `first_owned` scans of the batch are the previous rank's (lead-in)."""
        self.ctx.check(self.ctx.L.lmono_odom_shard_d(self.ctx.h, self.h, n_chains, lead, int(first_owned), C.c_void_p(incr_ptr or 0)))

    def odometry_shard_main_d(self, n_chains, lead, first_owned, incr_ptr=None):
        """odometry_shard_d without the validation of the rank's inner boundaries: shard_validate (after the exchange of the last
        increments) then validates every boundary of the rank, the external one too, in one set of repair rounds."""
        self.ctx.check(self.ctx.L.lmono_odom_shard_main_d(self.ctx.h, self.h, n_chains, lead, int(first_owned), C.c_void_p(incr_ptr or 0)))

    def shard_validate(self, prev_incr, incr_ptr=None):
        """Checks / repairs chain 0's warm start against the previous rank's last increment (None: the rank owns the sequence's first scan;
        only valid as the deferred validation after odometry_shard_main_d); True when this rank's last increment changed."""
        prev = None if prev_incr is None else np.ascontiguousarray(prev_incr, np.float64).reshape(7)
        ch = C.c_int(0)
        self.ctx.check(self.ctx.L.lmono_odom_shard_validate(self.ctx.h, self.h, None if prev is None else prev.ctypes.data, C.c_void_p(incr_ptr or 0), C.byref(ch)))
        return bool(ch.value)

    def boundary_report(self):
        """What the boundary validation of the last odometry call did: dict + per-chain residuals and re-run pair counts."""
        rep = BoundaryReport()
        self.ctx.check(self.ctx.L.lmono_odom_boundary_report(self.ctx.h, self.h, C.byref(rep), None, None, 0))
        n = rep.n_chains
        resid = np.zeros(max(n, 1)); rerun = np.zeros(max(n, 1), np.int32)
        self.ctx.check(self.ctx.L.lmono_odom_boundary_report(self.ctx.h, self.h, C.byref(rep), resid.ctypes.data, rerun.ctypes.data, n))
        d = {k: getattr(rep, k) for k, _ in BoundaryReport._fields_}
        d["resid"] = resid[:n]; d["rerun"] = rerun[:n]
        return d

    def correspond(self, scan, q, t):
        q = np.ascontiguousarray(q, np.float64); t = np.ascontiguousarray(t, np.float64)
        out = np.zeros((MAX_QUERIES, 4), np.int32)
        n = self.ctx.check(self.ctx.L.lmono_odom_correspond(self.ctx.h, self.h, scan, q.ctypes.data, t.ctypes.data, out.ctypes.data, MAX_QUERIES))
        return out[:n]

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.lmono_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BaDesc(C.Structure):
    _fields_ = [("n_windows", C.c_int), ("feat_off", C.c_void_p), ("obs_off", C.c_void_p), ("flags", C.c_void_p),
                ("poses", C.c_void_p), ("ex", C.c_void_p), ("inv_depth", C.c_void_p), ("obs_feat", C.c_void_p),
                ("obs_i", C.c_void_p), ("obs_j", C.c_void_p), ("obs_pts", C.c_void_p), ("laser_consts", C.c_void_p),
                ("prior_T", C.c_void_p), ("laser_info", C.c_void_p), ("mono_info", C.c_void_p), ("prior_w", C.c_void_p)]


class BaBatch:
    """Batch of independent BA windows resident in HBM (lmono_ba_batch).  `windows`: list of dicts with the keys of
    tests/ba_cases.make_window (poses [n,7], ex, inv_depth, obs_feat/obs_i/obs_j, obs_pts, laser_consts, prior_T, flags)."""

    def __init__(self, ctx, windows):
        self.ctx = ctx
        ctx._children.add(self)
        d = self._desc(windows)
        self.h = ctx.L.lmono_ba_batch_create(ctx.h, C.byref(d))
        if not self.h:
            raise LmonoError("lmono_ba_batch_create failed: %s" % ctx.L.lmono_last_error(ctx.h).decode())

    def update(self, windows):
        """Load another set of windows into the same device arrays (lmono_ba_batch_update)."""
        d = self._desc(windows)
        self.ctx.check(self.ctx.L.lmono_ba_batch_update(self.ctx.h, self.h, C.byref(d)))

    def _desc(self, windows):
        W = len(windows)
        self.W = W
        self.n_poses = [len(w["poses"]) for w in windows]
        feat_off = np.concatenate([[0], np.cumsum([len(w["inv_depth"]) for w in windows])]).astype(np.int32)
        obs_off = np.concatenate([[0], np.cumsum([len(w["obs_feat"]) for w in windows])]).astype(np.int32)
        flags = np.array([[len(w["poses"]), int(w["use_prior"]), int(w["ex_constant"]), int(w["use_mono"])] for w in windows], np.int32)
        poses = np.zeros((W, 11, 7)); poses[:, :, 6] = 1.0
        laser = np.zeros((W, 10, 24))
        for k, w in enumerate(windows):
            poses[k, :len(w["poses"])] = w["poses"]
            laser[k, :len(w["laser_consts"])] = w["laser_consts"]
        cat = lambda key, dt: np.ascontiguousarray(np.concatenate([np.asarray(w[key], dt).reshape(len(w[key]), -1) for w in windows]).ravel(), dt)
        self._keep = dict(feat_off=feat_off, obs_off=obs_off, flags=flags, poses=poses,
                          ex=np.ascontiguousarray([w["ex"] for w in windows], np.float64), inv_depth=cat("inv_depth", np.float64),
                          obs_feat=cat("obs_feat", np.int32), obs_i=cat("obs_i", np.int32), obs_j=cat("obs_j", np.int32),
                          obs_pts=cat("obs_pts", np.float64), laser=laser,
                          prior=np.ascontiguousarray([np.asarray(w["prior_T"]).ravel() for w in windows], np.float64),
                          li=np.ascontiguousarray(windows[0]["laser_info"], np.float64), mi=np.ascontiguousarray(windows[0]["mono_info"], np.float64),
                          pw=np.ascontiguousarray(windows[0]["prior_w"], np.float64))
        k = self._keep
        d = BaDesc(W, *[k[n].ctypes.data for n in ("feat_off", "obs_off", "flags", "poses", "ex", "inv_depth", "obs_feat", "obs_i",
                                                    "obs_j", "obs_pts", "laser", "prior", "li", "mi", "pw")])
        self.feat_off = feat_off
        return d

    def solve(self, max_iter=30):
        self.ctx.check(self.ctx.L.lmono_ba_solve(self.ctx.h, self.h, max_iter))

    def reset(self):
        self.ctx.check(self.ctx.L.lmono_ba_batch_reset(self.ctx.h, self.h))

    def read(self):
        poses = np.zeros((self.W, 11, 7)); ex = np.zeros((self.W, 7)); invd = np.zeros(int(self.feat_off[-1])); sm = np.zeros((self.W, 6))
        self.ctx.check(self.ctx.L.lmono_ba_batch_read(self.ctx.h, self.h, poses.ctypes.data, ex.ctypes.data, invd.ctypes.data, sm.ctypes.data))
        return poses, ex, invd, sm

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.lmono_ba_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mapper:
    """laserMapping with a device-resident cube map (lmono_mapper_*): process(batch, scan, q_wodom, t_wodom) per frame."""

    def __init__(self, ctx, line_res=0.4, plane_res=0.8):
        self.ctx = ctx
        ctx._children.add(self)
        L = ctx.L
        L.lmono_mapper_create.restype = C.c_void_p
        L.lmono_mapper_create.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.lmono_mapper_destroy.argtypes = [C.c_void_p]
        L.lmono_mapper_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.lmono_mapper_cube.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        self.h = L.lmono_mapper_create(ctx.h, line_res, plane_res)
        if not self.h:
            raise LmonoError("lmono_mapper_create failed: " + ctx.last_error())

    def process(self, batch, scan, q_wodom, t_wodom):
        q = np.ascontiguousarray(q_wodom, np.float64); t = np.ascontiguousarray(t_wodom, np.float64)
        qo = np.zeros(4); to = np.zeros(3); st = np.zeros(8, np.int32)
        self.ctx.check(self.ctx.L.lmono_mapper_process(self.ctx.h, self.h, batch.h, int(scan), q.ctypes.data, t.ctypes.data,
                                                       qo.ctypes.data, to.ctypes.data, st.ctypes.data))
        return qo, to, st

    def reset(self):
        self.ctx.L.lmono_mapper_reset.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx.check(self.ctx.L.lmono_mapper_reset(self.ctx.h, self.h))

    @staticmethod
    def process_batch(ctx, mappers, batches, scans, q_wodom, t_wodom):
        """One frame of several independent streams in one call: lists of Mapper / ScanBatch, scans [n], q_wodom [n,4],
        t_wodom [n,3].  Returns (q [n,4], t [n,3], stats [n,8])."""
        n = len(mappers)
        mh = (C.c_void_p * n)(*[m.h for m in mappers]); bh = (C.c_void_p * n)(*[b.h for b in batches])
        sc = np.ascontiguousarray(scans, np.int32)
        q = np.ascontiguousarray(q_wodom, np.float64).reshape(n, 4); t = np.ascontiguousarray(t_wodom, np.float64).reshape(n, 3)
        qo = np.zeros((n, 4)); to = np.zeros((n, 3)); st = np.zeros((n, 8), np.int32)
        ctx.L.lmono_mapper_process_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        ctx.check(ctx.L.lmono_mapper_process_batch(ctx.h, n, mh, bh, sc.ctypes.data, q.ctypes.data, t.ctypes.data, qo.ctypes.data, to.ctypes.data, st.ctypes.data))
        return qo, to, st

    def cube(self, which, i, j, k):
        n = self.ctx.L.lmono_mapper_cube(self.ctx.h, self.h, which, i, j, k, None, 0)
        self.ctx.check(min(n, 0))
        out = np.zeros((max(n, 1), 4), np.float32)
        if n > 0:
            self.ctx.check(min(self.ctx.L.lmono_mapper_cube(self.ctx.h, self.h, which, i, j, k, out.ctypes.data, n), 0))
        return out[:n]

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.lmono_mapper_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass



class Camera(C.Structure):
    """lmono_camera: PINHOLE intrinsics of the cam yaml + kernel_size / kernel_type / blur_type of the map config."""
    _fields_ = [("width", C.c_int), ("height", C.c_int),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double),
                ("kernel_size", C.c_int), ("kernel_type", C.c_int), ("blur_type", C.c_int)]


POINT_RGB = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("bgra", np.uint32)])


def lidar_to_camera(rlc, tlc):
    """The 4 x 4 of map_build_node.cc:216-220: [rlc^T | -rlc^T tlc]."""
    rlc = np.asarray(rlc, np.float64).reshape(3, 3); tlc = np.asarray(tlc, np.float64).reshape(3)
    M = np.eye(4)
    M[:3, :3] = rlc.T
    M[:3, 3] = (-1.0 * rlc.T) @ tlc
    return M


class MapBuilder:
    """MapBuilder::associateToMap / depthFill / rgb_map accumulation on the device (lmono_map_builder_*)."""

    def __init__(self, ctx, camera, max_cloud_points=1 << 18, map_capacity_points=None):
        self.ctx = ctx
        ctx._children.add(self)
        self.cam = camera
        L = ctx.L
        L.lmono_map_builder_create.restype = C.c_void_p
        L.lmono_map_builder_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64]
        L.lmono_map_builder_destroy.argtypes = [C.c_void_p]
        L.lmono_associate_to_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.lmono_associate_to_map_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.lmono_map_builder_depth.argtypes = [C.c_void_p] * 3
        L.lmono_map_builder_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.lmono_map_builder_map.restype = C.c_int64
        L.lmono_map_builder_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.lmono_map_builder_clear.argtypes = [C.c_void_p, C.c_void_p]
        if map_capacity_points is None:
            map_capacity_points = 10 * camera.width * camera.height      # processMapping flushes every 10 frames
        self.h = L.lmono_map_builder_create(ctx.h, C.byref(camera), int(max_cloud_points), int(map_capacity_points))
        if not self.h:
            raise LmonoError("lmono_map_builder_create failed: " + ctx.last_error())

    def associate(self, xyzi, transform, bgr, q_wc, t_wc):
        """One associateToMap on host buffers; returns the size of the coloured cloud."""
        xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4); M = np.ascontiguousarray(transform, np.float64).reshape(16)
        bgr = np.ascontiguousarray(bgr, np.uint8)
        if bgr.shape != (self.cam.height, self.cam.width, 3):
            raise LmonoError("image must be [height][width][3] uint8")
        q = np.ascontiguousarray(q_wc, np.float64).reshape(4); t = np.ascontiguousarray(t_wc, np.float64).reshape(3)
        n = C.c_int(0)
        self.ctx.check(self.ctx.L.lmono_associate_to_map(self.ctx.h, self.h, xyzi.ctypes.data, len(xyzi), M.ctypes.data, bgr.ctypes.data,
                                                         q.ctypes.data, t.ctypes.data, C.addressof(n)))
        return n.value

    @staticmethod
    def associate_batch(ctx, builders, xyzi_ptrs, n_points, transforms, bgr_ptrs, q_wc, t_wc):
        """One frame of several independent builders; xyzi_ptrs / bgr_ptrs are device pointers (ints).  Returns sizes [n]."""
        n = len(builders)
        mh = (C.c_void_p * n)(*[b.h for b in builders])
        xp = (C.c_void_p * n)(*[int(p) for p in xyzi_ptrs]); bp = (C.c_void_p * n)(*[int(p) for p in bgr_ptrs])
        npt = np.ascontiguousarray(n_points, np.int32)
        M = np.ascontiguousarray(transforms, np.float64).reshape(n, 16)
        q = np.ascontiguousarray(q_wc, np.float64).reshape(n, 4); t = np.ascontiguousarray(t_wc, np.float64).reshape(n, 3)
        out = np.zeros(n, np.int32)
        ctx.check(ctx.L.lmono_associate_to_map_batch(ctx.h, n, mh, xp, npt.ctypes.data, M.ctypes.data, bp, q.ctypes.data, t.ctypes.data, out.ctypes.data))
        return out

    def depth(self):
        d = np.zeros((self.cam.height, self.cam.width), np.uint8)
        self.ctx.check(self.ctx.L.lmono_map_builder_depth(self.ctx.h, self.h, d.ctypes.data))
        return d

    def cloud(self, which=0):
        n = self.ctx.L.lmono_map_builder_cloud(self.ctx.h, self.h, which, None, 0)
        self.ctx.check(min(n, 0))
        out = np.zeros(max(n, 1), POINT_RGB)
        self.ctx.check(min(self.ctx.L.lmono_map_builder_cloud(self.ctx.h, self.h, which, out.ctypes.data, n), 0))
        return out[:n]

    def map(self):
        n = self.ctx.L.lmono_map_builder_map(self.ctx.h, self.h, None, 0)
        self.ctx.check(min(n, 0))
        out = np.zeros(max(n, 1), POINT_RGB)
        self.ctx.check(min(self.ctx.L.lmono_map_builder_map(self.ctx.h, self.h, out.ctypes.data, n), 0))
        return out[:n]

    def clear(self):
        self.ctx.check(self.ctx.L.lmono_map_builder_clear(self.ctx.h, self.h))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.lmono_map_builder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PoseGraph:
    """Loop-closure pose graph (lmono_pose_graph_*; a new feature): 4-DoF keyframe graph over odometry poses and
    loop_info records.  optimize() on one GPU; linearise() / reduce_buffer / step() for the multi-GPU round (sharding.py)."""

    def __init__(self, ctx, poses_tq, loops, loop_info):
        self.ctx = ctx
        ctx._children.add(self)
        L = ctx.L
        L.lmono_pose_graph_create.restype = C.c_void_p
        L.lmono_pose_graph_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.lmono_pose_graph_destroy.argtypes = [C.c_void_p]
        L.lmono_pose_graph_info.argtypes = [C.c_void_p] * 4
        L.lmono_pose_graph_reduce_buffer.restype = C.c_void_p
        L.lmono_pose_graph_reduce_buffer.argtypes = [C.c_void_p]
        L.lmono_pose_graph_linearise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.lmono_pose_graph_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.lmono_pose_graph_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lmono_pose_graph_result.argtypes = [C.c_void_p] * 4
        P = np.ascontiguousarray(poses_tq, np.float64).reshape(-1, 7)
        lp = np.ascontiguousarray(loops, np.int32).reshape(-1, 2); li = np.ascontiguousarray(loop_info, np.float64).reshape(-1, 8)
        if len(lp) != len(li):
            raise LmonoError("loops and loop_info differ in length")
        self.n = len(P)
        self.h = L.lmono_pose_graph_create(ctx.h, self.n, P.ctypes.data, len(lp), lp.ctypes.data, li.ctypes.data)
        if not self.h:
            raise LmonoError("lmono_pose_graph_create failed: " + ctx.last_error())
        rc = C.c_int64(0); bw = C.c_int(0); ne = C.c_int(0)
        L.lmono_pose_graph_info(self.h, C.addressof(rc), C.addressof(bw), C.addressof(ne))
        self.reduce_count, self.bandwidth, self.n_edges = rc.value, bw.value, ne.value
        self.reduce_ptr = L.lmono_pose_graph_reduce_buffer(self.h)

    def reset(self):
        self.ctx.L.lmono_pose_graph_reset.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx.check(self.ctx.L.lmono_pose_graph_reset(self.ctx.h, self.h))

    def use_reduce_tensor(self, tensor):
        """Make a caller-owned contiguous fp64 device tensor of reduce_count elements the buffer linearise() fills and step()
        reads (the tensor handed to torch.distributed.all_reduce)."""
        if tensor.numel() != self.reduce_count or tensor.element_size() != 8 or not tensor.is_contiguous():
            raise LmonoError("reduce tensor must be contiguous fp64 with %d elements" % self.reduce_count)
        self.ctx.L.lmono_pose_graph_set_reduce_buffer.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx.check(self.ctx.L.lmono_pose_graph_set_reduce_buffer(self.h, tensor.data_ptr()))
        self.reduce_tensor = tensor
        self.reduce_ptr = tensor.data_ptr()

    def linearise(self, rank=0, world=1):
        self.ctx.check(self.ctx.L.lmono_pose_graph_linearise(self.ctx.h, self.h, rank, world))

    def step(self, max_iter=5):
        done = C.c_int(0)
        self.ctx.check(self.ctx.L.lmono_pose_graph_step(self.ctx.h, self.h, max_iter, C.addressof(done)))
        return bool(done.value)

    def optimize(self, max_iter=5):
        self.ctx.check(self.ctx.L.lmono_pose_graph_optimize(self.ctx.h, self.h, max_iter))
        return self.result()

    def result(self):
        out = np.zeros((self.n, 7)); st = np.zeros(6)
        self.ctx.check(self.ctx.L.lmono_pose_graph_result(self.ctx.h, self.h, out.ctypes.data, st.ctypes.data))
        return out, dict(iterations=int(st[0]), initial_cost=st[1], final_cost=st[2], bandwidth=int(st[3]), accepted=int(st[4]), rejected=int(st[5]))

    def close(self):
        if getattr(self, "h", None):
            self.ctx.L.lmono_pose_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
