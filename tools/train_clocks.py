#!/usr/bin/env python3
"""Where a row-fused training launch spends its time: a development build of the library (-DEPNN_TF_CLOCKS, built HERE into
tools/_dev/, never the shipped .so) stamps the 100 MHz wall clock of workgroup 0 at the phase boundaries of every pair-sweep launch of one
training step.  Usage: python tools/train_clocks.py [B]"""
import os, sys, subprocess, ctypes as C, tarfile, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = os.path.join(ROOT, "tools", "_dev", "libepnn_clocks.so")
if "--build" in sys.argv or not os.path.exists(DEV):
    os.makedirs(os.path.dirname(DEV), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffinite-math-only", "-fno-signed-zeros", "-mllvm",
                    "-amdgpu-mfma-vgpr-form", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DEPNN_TF_CLOCKS", "-shared", "-fPIC", "-o", DEV, os.path.join(ROOT, "epnn_amd/csrc/epnn_api.hip"),
                    "-L/opt/rocm/lib", "-lrccl"], check=True)
    if "--build" in sys.argv: sys.exit(0)
from epnn_amd import _lib
_lib.LIB_PATH = DEV
from epnn_amd import checkpoint, charge_gn
from epnn_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
d = tempfile.mkdtemp(); tarfile.open(os.path.join(ROOT, "tests/golden/mixed_val.tar.gz")).extractall(d)
names = [str(n) for n in np.load(os.path.join(ROOT, "tests/golden/val_names.npy"), allow_pickle=True)][:64]
mols = [charge_gn.read_xyz(os.path.join(d, "mixed_val", nm + ".xyz"), 9) for nm in names]
eng = Engine(nx=9, T=5); eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights")))
eng.train_init()
eng.set_option("train_graph", 0)          # the development build allocates its clock buffer inside the step
for a in sys.argv[1:]:
    if a.startswith("--opt="): k, v = a[6:].split(":"); eng.set_option(k, int(v))
def batch(k):
    ms = mols[k * B:(k + 1) * B]
    off = np.zeros(len(ms) + 1, np.int32); off[1:] = np.cumsum([len(m[1]) for m in ms])
    return off, np.concatenate([m[0] for m in ms]), np.concatenate([m[1] for m in ms]), np.array([m[2] for m in ms], np.float32), np.zeros(off[-1], np.float32)
for k in range(3): eng.train_step_xyz(*batch(k), 41)
lib = _lib.load()
buf = (C.c_ulonglong * (64 * 16))()
lib.epnn_debug_train_clocks.restype = C.c_int
lib.epnn_debug_train_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
assert lib.epnn_debug_train_clocks(eng.h, buf, 64 * 16) == 0
c = np.array(buf[:], dtype=np.uint64).reshape(64, 16).astype(np.int64)
print("""phase boundaries of workgroup 0 in units of 10 ns (wall_clock64) from the workgroup's start; '-' = not stamped in this launch
forward  (k_tf_pair_fwd<MODE, true>):  1 rows staged (barrier)   2 both Dense layers done (barrier)   3 column sums + third Dense (message) /
         per-row third Dense (pass)   5 end: update MLP (message, on the last wavefront) / pair sum (pass)
backward (k_tb_pair_bwd_mm<MODE>):     1 chain and staging met (barrier)   2 dz2, dz1 done (barrier)   3 end of the weight-gradient jobs
         4 chain: prologue's sums   5 chain: gradient at a_i   6 chain: update MLP backward (message)   7 chain: end   8 staging wavefronts: end
         9 dz1 MFMAs   10 dz stores   11 update partials stored   12 first weight-gradient job starts   13 its K loop ends""")
for l in range(64):
    if c[l, 0] == 0: continue
    row = c[l]; last = max(k for k in range(16) if row[k])
    kind = "msg fwd" if l < 5 else "pass fwd" if l < 10 else "pass bwd" if l < 15 else "msg bwd"
    cyc = f"  core clock {(row[15] - row[14]) / max(1, row[3] - row[0]) / 10:.2f} GHz" if row[14] and row[15] else ""
    last = min(last, 13)
    print(f"{l:3d} {kind:9s}", " ".join(f"{(row[k] - row[0]):6d}" if row[k] else "     -" for k in range(1, last + 1)) + cyc)
