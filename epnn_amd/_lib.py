"""ctypes binding of libepnn_hip.so (include/epnn.h).  There is no CPU fallback: a missing library or a
missing MI355X raises, it never silently computes somewhere else."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EPNN_LIB") or os.path.join(_HERE, "libepnn_hip.so")      # (EPNN_LIB: a development build, tools/)

W_MSG, W_UPD, W_PAS = 0, 1, 2


class EpnnConfig(C.Structure):
    _fields_ = [("nx", C.c_int32), ("h_dim", C.c_int32), ("e_dim", C.c_int32), ("T", C.c_int32),
                ("hidden", C.c_int32), ("cutoff", C.c_float), ("eta", C.c_float), ("near_tol", C.c_float)]


class EpnnError(RuntimeError):
    pass


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/epnn.h declares
# int exchange(void *ctx, float *d_rows, int row_len, int n_rows, int row_lo, int row_hi)   (epnn_set_partition)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)

SIGNATURES = {
    "epnn_last_error": (C.c_char_p, []),
    "epnn_version": (C.c_int, []),
    "epnn_device_count": (C.c_int, []),
    "epnn_create": (C.c_int, [C.POINTER(EpnnConfig), C.c_int, C.POINTER(_vp)]),
    "epnn_destroy": (C.c_int, [_vp]),
    "epnn_skip_hw_queues": (C.c_int, [C.c_int, C.c_int]),
    "epnn_set_weights": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "epnn_get_weights": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "epnn_weight_shape": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _ip, _ip]),
    "epnn_set_update_layers": (C.c_int, [_vp, C.c_int, _ip]),
    "epnn_edges": (C.c_int, [_vp, C.c_int, _fp, _fp]),
    "epnn_edges_ex": (C.c_int, [_vp, C.c_int, _fp, C.c_int, C.c_double, C.c_double, _fp, C.POINTER(C.c_double)]),
    "epnn_forward_xyz": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _fp, _fp, _fp, _fp]),
    "epnn_forward_xyz_begin": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _fp, _fp, _fp]),
    "epnn_forward_xyz_end": (C.c_int, [_vp, _fp]),
    "epnn_set_partition": (C.c_int, [_vp, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "epnn_forward_xyz_dev": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _vp, _vp, _vp, _vp]),
    "epnn_model_forward_dense": (C.c_int, [_vp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    "epnn_model_forward_dense_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epnn_gnn_forward": (C.c_int, [_vp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    "epnn_epn_forward": (C.c_int, [_vp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    "epnn_mlp_forward": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "epnn_mlp_forward_layers": (C.c_int, [_vp, C.c_int, C.c_int, _ip, C.POINTER(_fp), C.POINTER(_fp), _fp, _fp, C.c_int]),
    "epnn_train_init": (C.c_int, [_vp, C.c_float, C.c_float, C.c_float, C.c_float]),
    "epnn_param_count": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "epnn_train_step_dense": (C.c_int, [_vp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int]),
    "epnn_train_step_xyz": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int]),
    "epnn_get_gradients": (C.c_int, [_vp, _fp, C.c_int64]),
    "epnn_set_gradients": (C.c_int, [_vp, _fp, C.c_int64]),
    "epnn_train_apply": (C.c_int, [_vp]),
    "epnn_comm_unique_id": (C.c_int, [C.c_char_p]),
    "epnn_comm_init": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    "epnn_comm_count": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "epnn_debug_pairs": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]),
    "epnn_comm_allreduce": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_int32, C.c_int32]),
    "epnn_dev_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "epnn_dev_free": (C.c_int, [_vp, _vp]),
    "epnn_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "epnn_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "epnn_sync": (C.c_int, [_vp]),
    "epnn_timer_begin": (C.c_int, [_vp]),
    "epnn_timer_end": (C.c_int, [_vp, _fp]),
    "epnn_last_timing": (C.c_int, [_vp, _fp]),
    "epnn_timing_at": (C.c_int, [_vp, C.c_int, _fp]),
    "epnn_edge_basis_residual": (C.c_double, [_vp]),
    "epnn_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "epnn_last_stats": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
}

_lib = None


def visible_gpu_count():
    """GPUs this process will see, WITHOUT initialising HIP in it (the runtime reads GPU_MAX_HW_QUEUES when it starts, so a
    driver that wants to size its queues by ranks per device must count devices first): EPNN_NDEV if a launcher exported it,
    the *_VISIBLE_DEVICES lists if set, else the library's own answer from a short-lived child process (the KFD topology in
    /sys also lists devices a container cannot open).  0 when nothing can be said."""
    import subprocess
    import sys
    v = os.environ.get("EPNN_NDEV")
    if v and v.isdigit():
        return int(v)
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([t for t in v.split(",") if t.strip() != ""])
    try:
        out = subprocess.run([sys.executable, "-c",
                              "import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); lib.epnn_device_count.restype = ctypes.c_int; print(lib.epnn_device_count())",
                              LIB_PATH], capture_output=True, text=True, timeout=300)
        n = int(out.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.SubprocessError):
        return 0
    os.environ["EPNN_NDEV"] = str(n)                          # children of this process (launch_ranks) need not ask again
    return n


def queues_for_shared_device(ranks_on_device, default=16):
    """Hardware queues a process should ask for when `ranks_on_device` processes share one GPU.  A device serves a limited
    number of hardware queues at once; beyond it the firmware time-slices whole queues (milliseconds at a time): two bench
    ranks with 16 queues each beside a third process fell from 200 M to 14 M atoms/s (gpurun_out/r2_fulltests2.log).  One
    rank per device keeps the default."""
    if ranks_on_device <= 1:
        return default
    return max(2, 8 // int(ranks_on_device))


def load():
    """Load libepnn_hip.so (built by __graft_entry__.build()); raises EpnnError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EpnnError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback for the EPNN hot path.")
    # Several handles (= HIP streams) keep batches in flight concurrently (engine.Pipeline).  The HIP runtime maps a
    # process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them the null stream's): kernels of
    # streams that share a queue serialise.  Ask for 16 unless the caller decided otherwise (six batches in flight, and an
    # RCCL communicator in the same process takes queues of its own: with 8 the pipeline loses 15 %); read when HIP initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise EpnnError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        raise EpnnError(lib.epnn_last_error().decode(errors="replace"))


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_fp)


def iptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)
