"""lmono_amd -- MI355X-native (gfx950) implementation of lmono's per-scan numeric hot path.

The product is the C-ABI shared library lmono_amd/lib/liblmono_hip.so (include/lmono_hip.h) built from the
hand-written HIP kernels in lmono_amd/csrc/.  This Python package is plumbing only: it loads the library with
ctypes and uses torch for device buffers / streams / torch.distributed.  There is no CPU fallback: every entry
point raises when the HIP library or a GPU is missing.
"""
from .capi import LmonoError, Context, ScanBatch, OdomStream, BaBatch, Mapper, MapBuilder, PoseGraph, Camera, lidar_to_camera, lib_path, load_library  # noqa: F401
