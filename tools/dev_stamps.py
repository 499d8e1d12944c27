#!/usr/bin/env python3
"""Diagnostic build (-DEPNN_STAMPS): where does one workgroup of the fused kernel spend its cycles?
Never quote this build's run time (guide section 7, In-kernel stamps); read the shares."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "epnn_amd", "libepnn_hip_stamps.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffinite-math-only", "-fno-signed-zeros", "-shared", "-fPIC", "-DEPNN_STAMPS=" + os.environ.get("STAMPS", "1"), "-DEPNN_ABL=" + os.environ.get("ABL", "0"),
                "-o", so, os.path.join(ROOT, "epnn_amd/csrc/epnn_api.hip"), "-L/opt/rocm/lib", "-lrccl"], check=True)
from epnn_amd import _lib
_lib.LIB_PATH = so
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine
eng = Engine(nx=9, T=5)
eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights")))
eng.set_option("size_classes", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
offsets, xyz, x, Q, N = synth.qm9_like_batch(B=B, seed=0)
for _ in range(3):
    q = eng.forward_xyz(offsets, xyz, x, Q, N)
buf = np.zeros(B * 4 * 64, dtype=np.uint64)
eng.lib.epnn_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
eng.lib.epnn_debug_stamps(eng.h, buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf.reshape(B, 4, 64)
labels = ["init"] + sum([[f"g{t}.A", f"g{t}.barA", f"g{t}.B", f"g{t}.barB", f"g{t}.C", f"g{t}.barC"] for t in range(5)], [])
for blk in (0, B // 2, B - 1):
    n = int(st[blk, 0, 63] >> np.uint64(32)); npairs = int(st[blk, 0, 63] & np.uint64(0xFFFFFFFF)); ns = int(st[blk, 0, 62])
    t0 = st[blk, :, 0].min()
    print(f"block {blk}: n={n} pairs={npairs} stamps={ns}; total cycles (wave0) {int(st[blk,0,ns-1]-st[blk,0,0])}")
    for w in range(4):
        d = np.diff(st[blk, w, :ns].astype(np.int64))
        print(f"  wave {w}: first 20 segment cycles {d[:20].tolist()}")
        # GNN segments: after init(1 seg) come 6 per step; EPN steps have no stamps inside -> last big segment
    d0 = np.diff(st[blk, 0, :ns].astype(np.int64))
    print("  wave0 GNN total", int(d0[1:31].sum()), "EPN+rest", int(d0[31:].sum()))
eng.close()
