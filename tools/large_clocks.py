#!/usr/bin/env python3
"""Where the tiled path's small launches spend their time at protein size: a development build of the library (-DEPNN_LG_CLOCKS,
built into tools/_dev/, never the shipped .so) stamps the 100 MHz wall clock of workgroup 0 at phase boundaries of the GNN
tail launches (k_lg_gnn_tail) and the EPN-step launches (k_lg_epn_step) of one forward of the 2220-atom protein.
    python tools/large_clocks.py [--build] [--lib path] [--forwards n] [--detail]"""
import os, sys, subprocess, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = os.path.join(ROOT, "tools", "_dev", "libepnn_lgclocks.so")
if "--lib" in sys.argv: DEV = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])      # another development build (-DEPNN_LG_CLOCKS)
if "--build" in sys.argv or not os.path.exists(DEV):
    os.makedirs(os.path.dirname(DEV), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffinite-math-only", "-fno-signed-zeros", "-mllvm",
                    "-amdgpu-mfma-vgpr-form", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DEPNN_LG_CLOCKS", "-shared", "-fPIC", "-o", DEV, os.path.join(ROOT, "epnn_amd/csrc/epnn_api.hip"),
                    "-L/opt/rocm/lib", "-lrccl"], check=True)
    if "--build" in sys.argv: sys.exit(0)
from epnn_amd import _lib
_lib.LIB_PATH = DEV
from epnn_amd import checkpoint, charge_gn
from epnn_amd.engine import Engine
eng = Engine(nx=9, T=5); eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights")))
xyz, x, Q, _ = charge_gn.read_xyz(os.path.join(ROOT, "tests/golden/protein/6qlp_capped.xyz"), 9)
offsets = np.array([0, len(x)], dtype=np.int32); Q = np.array([Q], dtype=np.float32); N = len(x)
d = [eng.to_device(a) for a in (xyz, x, Q)]; dq = eng.alloc(N * 4)
nfw = int(sys.argv[sys.argv.index("--forwards") + 1]) if "--forwards" in sys.argv else 5
for _ in range(nfw): eng.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
eng.sync()
lib = _lib.load()
NB = 128 + 4 * 1024
buf = (C.c_ulonglong * NB)()
lib.epnn_debug_large_clocks.restype = C.c_int
lib.epnn_debug_large_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
assert lib.epnn_debug_large_clocks(eng.h, buf, NB) == 0
c = np.array(buf[:], dtype=np.uint64).astype(np.int64)
print("units of 10 ns from the start of workgroup 0 (the stamps of the LAST launch of each kind in the forward)")
print("k_lg_gnn_tail: 1 weights / image requested   2 S reduced (partials, slot rows)   3 barrier   4 update MLP done   5 barrier (then the projections)")
for run, name in ((0, "last GNN step (no projections)"), (1, "middle steps (next step's P, R)"), (2, "hand-over to the EPN stack (static projections)")):
    r = c[16 * run:16 * run + 8]
    if r[0]: print(f"  {name:48s}", " ".join(f"{(r[k] - r[0]):6d}" if r[k] else "     -" for k in range(1, 6)))
print("k_lg_epn_step: 1 q of both atoms (slot sums of the previous step)   2 G tile, P / R rows, weights in registers   3 MFMAs done   4 end")
for t in range(5):
    r = c[64 + 8 * t:64 + 8 * t + 8]
    if r[0]: print(f"  step {t}", " ".join(f"{(r[k] - r[0]):6d}" if r[k] else "     -" for k in range(1, 5)))

# the merged launches of the compact entry's front (100 MHz stamps of one workgroup of every kind of work; absolute differences)
def span(a, b):
    return f"{(c[b] - c[a]) * 0.01:6.2f}" if c[a] and c[b] else "     -"
print("front-end launches, us (start -> end of ONE workgroup of each kind; kernel start = its first stamp):")
print(f"  k_lg_first   tile workgroup 0: rows built {span(104, 105)}  a_eo stored / hashes {span(105, 106)}  projections {span(106, 107)}   total {span(104, 107)} | first count workgroup {span(109, 110)} (starts {span(104, 109)} after the tile workgroup)")
print(f"  k_lg_scan_types   scan {span(114, 115)} | type table {span(122, 123)}      (first -> scan start {span(104, 114)})")
print(f"  k_lg_fill_assign  fill workgroup 0 {span(112, 113)}                        (scan start -> fill start {span(114, 112)})")
print(f"  k_lg_second  link workgroup 0 {span(116, 117)} | type sums {span(118, 119)} | correction tiles {span(120, 121)}   (fill start -> second start {span(112, 116)})")

# k_lg_sweep2 (the LAST sweep launch of the forward): per workgroup s_memtime / s_memrealtime at the start and the end of its task
sw = c[128:].reshape(1024, 4)
sw = sw[sw[:, 0] > 0]
if len(sw):
    dt, dr = (sw[:, 2] - sw[:, 0]).astype(float), (sw[:, 3] - sw[:, 1]).astype(float)
    ghz = dt / np.maximum(dr, 1) * 0.1
    t0 = sw[:, 1].min()
    st, en = (sw[:, 1] - t0) * 0.01, (sw[:, 3] - t0) * 0.01
    print(f"k_lg_sweep2, last launch of the forward ({nfw} forwards in a row): {len(sw)} workgroups; shader clock seen by them "
          f"(s_memtime / s_memrealtime) mean {ghz.mean():.3f} GHz, 5 % .. 95 %: {np.percentile(ghz, 5):.3f} .. {np.percentile(ghz, 95):.3f}")
    print(f"  task length {dr.mean() * 0.01:.1f} us (min {dr.min() * 0.01:.1f}, max {dr.max() * 0.01:.1f}); starts after the first workgroup's: "
          f"median {np.median(st):.1f} us, 95 % {np.percentile(st, 95):.1f}, last {st.max():.1f}; ends: first {en.min():.1f} us, median {np.median(en):.1f}, last {en.max():.1f}")
    # which tasks are the slow ones: 18 tile groups x 28 j-chunks on the protein (block = group * nchunk + chunk)
    if "--detail" in sys.argv:
        dur = dr * 0.01
        nchunk = 7 if len(sw) in (490, 512) else 1
        g = dur[:len(dur) // nchunk * nchunk].reshape(-1, nchunk)
        print("  mean task length by tile group:", " ".join(f"{v:.0f}" for v in g.mean(axis=1)))
        print("  mean task length by j-chunk:   ", " ".join(f"{v:.0f}" for v in g.mean(axis=0)))
        print("  histogram of task lengths (us):", np.histogram(dur, bins=[0, 30, 60, 75, 80, 85, 90, 95, 100, 110, 200])[0].tolist(), "bins 0,30,60,75,80,85,90,95,100,110+")
