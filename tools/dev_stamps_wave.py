#!/usr/bin/env python3
"""Diagnostic build (-DEPNN_STAMPS) of the wave-autonomous kernel: cycles per phase of one wavefront.
Never quote this build's run time; read the shares."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "epnn_amd", "libepnn_hip_stamps.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffinite-math-only", "-fno-signed-zeros", "-shared", "-fPIC", "-DEPNN_STAMPS=1",
                "-o", so, os.path.join(ROOT, "epnn_amd/csrc/epnn_api.hip"), "-L/opt/rocm/lib", "-lrccl"], check=True, stderr=subprocess.DEVNULL)
from epnn_amd import _lib
_lib.LIB_PATH = so
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine
eng = Engine(nx=9, T=5)
eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
offsets, xyz, x, Q, N = synth.qm9_like_batch(B=B, seed=0)
for _ in range(3):
    q = eng.forward_xyz(offsets, xyz, x, Q, N)
buf = np.zeros(B * 64, dtype=np.uint64)
eng.lib.epnn_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
eng.lib.epnn_debug_stamps(eng.h, buf.ctypes.data_as(C.c_void_p), buf.size)
st = buf.reshape(B, 64)
labels = ["init", "g0:G+proj", "g0:pairs", "g0:U1U2", "g0:Gtiles", "g0:proj", "g1:pairs", "g1:U1U2", "g1:Gtiles", "g1:proj",
          "g2..4+h", "e0:PR", "e0:pairs", "e0:q", "e1:PR", "e1:pairs", "e1:q", "e2..4+out"]
for blk in (0, B // 8, B // 2, B - 1):
    n = int(st[blk, 63] >> np.uint64(32)); npairs = int(st[blk, 63] & np.uint64(0xFFFFFFFF)); ns = int(st[blk, 62])
    d = np.diff(st[blk, :ns].astype(np.int64))
    print(f"block {blk}: n={n} pairs={npairs} total {int(st[blk, ns-1] - st[blk, 0])}")
    print("   " + "  ".join(f"{l}={v}" for l, v in zip(labels, d.tolist())))
t0 = st[:, 0].astype(np.int64); t1 = np.array([st[b, int(st[b, 62]) - 1] for b in range(B)]).astype(np.int64)
print("kernel span (first start -> last end)", int(t1.max() - t0.min()), "mean wave life", int((t1 - t0).mean()), "max", int((t1 - t0).max()),
      "start spread", int(t0.max() - t0.min()))
eng.close()
