#!/usr/bin/env python3
"""Where the fused kernel's wavefronts spend their cycles: a development build of the library (-DEPNN_STAMPS, built HERE into
tools/_dev/, never the shipped .so) lets lane 0 of every wavefront of k_wave_forward stamp s_memtime (shader clocks) at its
phase boundaries.  The bench batch is launched `copies` times over in ONE launch so that every SIMD holds its two wavefronts.
Usage: python tools/wave_clocks.py [--lib path/to/stamps/build.so] [--copies 4] [--init-detail]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
INIT = "--init-detail" in args          # four more stamps inside the front-end (-DEPNN_STAMPS_INIT)
DEV = os.path.join(ROOT, "tools", "_dev", "libepnn_stamps_init.so" if INIT else "libepnn_stamps.so")
if "--lib" in args:
    DEV = os.path.abspath(args[args.index("--lib") + 1])
elif "--build" in args or not os.path.exists(DEV):
    os.makedirs(os.path.dirname(DEV), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffinite-math-only", "-fno-signed-zeros", "-mllvm",
                    "-amdgpu-mfma-vgpr-form", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DEPNN_STAMPS"] + (["-DEPNN_STAMPS_INIT"] if INIT else []) + ["-shared", "-fPIC", "-o", DEV, os.path.join(ROOT, "epnn_amd/csrc/epnn_api.hip"),
                    "-L/opt/rocm/lib", "-lrccl"], check=True)
    if "--build" in args:
        sys.exit(0)
copies = int(args[args.index("--copies") + 1]) if "--copies" in args else 4
from epnn_amd import _lib
_lib.LIB_PATH = DEV
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine

off, xyz, x, Q, N = synth.qm9_like_batch(1024, 0, 29)
ns = np.diff(off)
if "--same" in args:          # every wavefront the same molecule (the first one of that size): the SIMD partners run in step
    m = int(np.flatnonzero(ns == int(args[args.index("--same") + 1]))[0])
    xyz, x, Q, ns = xyz[off[m]:off[m + 1]], x[off[m]:off[m + 1]], Q[m:m + 1], ns[m:m + 1]
    xyz, x, Q, ns = np.tile(xyz, (1024, 1)), np.tile(x, (1024, 1)), np.tile(Q, 1024), np.tile(ns, 1024)
offs = np.concatenate([[0], np.cumsum(np.tile(ns, copies))]).astype(np.int32)
eng = Engine()
eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models", "decay_model_weights")))
eng.set_option("wave2", 0)
for a_ in args:                     # developer switches: --opt=name:value
    if a_.startswith("--opt="):
        k_, v_ = a_[6:].split(":")
        eng.set_option(k_, int(v_))
for _ in range(3):
    q = eng.forward_xyz(offs, np.tile(xyz, (copies, 1)), np.tile(x, (copies, 1)), np.tile(Q, copies), N)
lib = _lib.load()
W = 1024 * copies
buf = (C.c_ulonglong * (64 * W))()
lib.epnn_debug_stamps.restype = C.c_int
lib.epnn_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_size_t]
assert lib.epnn_debug_stamps(eng.h, buf, 64 * W) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(W, 64).astype(np.int64)
names = ["init (front-end)", "G tiles + projections 0", "sweep 0", "update 0", "G tiles 1", "projections 1", "sweep 1", "update 1",
         "G tiles 2", "projections 2", "GNN steps 2.. + h", "EPN P,R 0", "EPN blocks 0", "charge update 0", "EPN P,R 1", "EPN blocks 1",
         "charge update 1", "EPN steps 2.."]
if INIT:
    names = ["init: record, coordinates -> LDS", "init: inputs requested, pair map cleared", "init: pair slots (f64 cutoff compare)",
             "init: edge coordinates (table)", "init: zero row, first e rows"] + names[1:]
nst = st[:, 62]
wn = (st[:, 63] >> 32).astype(int)
npair = (st[:, 63] & 0xFFFFFFFF).astype(int)
k = int(np.median(nst))
d = np.diff(st[:, :k], axis=1).astype(np.float64)
tot = (st[:, k - 1] - st[:, 0]).astype(np.float64)
print(f"{W} wavefronts in one launch ({copies} x the bench batch), {k} stamps each; shader clocks per wavefront (mean over all / over n = 17..19)")
sel = (wn >= 17) & (wn <= 19)
for i in range(d.shape[1]):
    nm = names[i] if i < len(names) else f"phase {i}"
    print(f"  {nm:28s} {d[:, i].mean():10.0f} {100 * d[:, i].sum() / tot.sum():6.1f} %   {d[sel, i].mean():10.0f}")
print(f"  {'total':28s} {tot.mean():10.0f}            {tot[sel].mean():10.0f}")
mf = 0
for n_, p_ in zip(wn, npair):
    pass
span = st[:, k - 1].max() - st[:, 0].min()
print(f"launch span {span} clocks; sum of wavefront lifetimes / (2048 slots) = {tot.sum() / 2048:.0f}")
# placement: HW_ID bits 0-3 wave slot, 4-5 SIMD, 8-11 CU, 12 shader array, 13-15 shader engine; XCC_ID in the high word
hw = st[:, 61]
print('raw HW_ID / XCC_ID of the first wavefronts:', [hex(int(v)) for v in hw[:6]], [hex(int(v)) for v in st[5, 56:64]])
simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0s, t1s = st[:, 0], st[:, k - 1]
occ = {}
for kk in np.unique(key):
    m = key == kk
    ev = sorted([(t, 1, s_) for t, s_ in zip(t0s[m], simd[m])] + [(t, -1, s_) for t, s_ in zip(t1s[m], simd[m])])
    cnt = [0, 0, 0, 0]
    last = ev[0][0]
    hist = {}
    for t, d_, s_ in ev:
        state = (sum(cnt), tuple(sorted(cnt)))
        hist[state] = hist.get(state, 0) + (t - last)
        last = t
        cnt[s_] += d_
    for st_, v in hist.items():
        occ[st_] = occ.get(st_, 0) + v
tot_t = float(sum(occ.values()))
print("time share of a CU's state (waves resident, sorted waves per SIMD), over %d CUs:" % len(np.unique(key)))
for st_, v in sorted(occ.items(), key=lambda kv: -kv[1])[:12]:
    print(f"   {st_[0]} waves {st_[1]}: {100 * v / tot_t:5.1f} %")
rt = (st[:, 60] - st[:, 59]).astype(np.float64)          # 100 MHz ticks of every wavefront's lifetime
ghz = tot / np.maximum(rt, 1.0) * 0.1
print(f"shader clock seen by the wavefronts (s_memtime / s_memrealtime): mean {ghz.mean():.3f} GHz, 5 % .. 95 %: {np.percentile(ghz, 5):.3f} .. {np.percentile(ghz, 95):.3f}; "
      f"launch lasted {(st[:, 60].max() - st[:, 59].min()) / 100.0:.1f} us on the 100 MHz clock")
by = {}
for n_ in sorted(set(wn)):
    by[n_] = tot[wn == n_].mean()
print("lifetime by atoms:", {int(a): int(b) for a, b in by.items()})
if "--by-size" in args:
    # the phases of the wavefronts of some sizes side by side: where the step from 16 to 17 atoms (a second column block) goes
    sizes = [n_ for n_ in (12, 14, 15, 16, 17, 18, 20, 22, 24) if (wn == n_).sum() >= 8]
    print("phase (mean shader clocks) by atoms:   " + " ".join(f"{n_:>7d}" for n_ in sizes))
    for i in range(d.shape[1]):
        nm = names[i] if i < len(names) else f"phase {i}"
        print(f"  {nm:36s}" + " ".join(f"{d[wn == n_, i].mean():7.0f}" for n_ in sizes))
    print(f"  {'total':36s}" + " ".join(f"{tot[wn == n_].mean():7.0f}" for n_ in sizes))
    print(f"  {'near pairs (unordered)':36s}" + " ".join(f"{npair[wn == n_].mean():7.0f}" for n_ in sizes))
