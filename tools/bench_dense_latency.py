#!/usr/bin/env python3
"""Latency of ONE literal model([h,e,x,q,mask]) call per molecule (the reference's calling convention, infer.py:62-76), dense
inputs resident in HBM, by molecule size at N = 41: the default (a molecule that fills more than 55 % of N takes the row-fused
forward kernels, a workgroup per atom slot), the fused inference kernels only ("dense_rowfused" = 0), and with "wave2" = 0,
"wave3" = 0 as well (k_wave_forward / the tiled kernels).
    python tools/bench_dense_latency.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epnn_amd import checkpoint, synth, charge_gn
from epnn_amd.engine import Engine

N, nx = 41, 9
eng = Engine(nx=nx, T=5)
eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights")))
rng = np.random.default_rng(2)
fn = eng.lib.epnn_model_forward_dense_dev
print("n    default ms   fused kernels only (dense_rowfused = 0) ms   one wavefront per molecule / tiled (and wave2 = wave3 = 0) ms")
for n in (9, 16, 18, 23, 29, 32, 35, 38, 41):
    span = 1.6 * n ** (1 / 3.0) * 1.3
    while True:
        pts = rng.uniform(0, span, size=(n, 3))
        d = np.linalg.norm(pts[:, None] - pts[None], axis=-1) + np.eye(n) * 10
        if d.min() > 0.8:
            break
    h = np.zeros((1, N, N, 48), np.float32); e = np.zeros((1, N, N, 48), np.float32)
    x = np.zeros((1, N, N, nx), np.float32); q = np.zeros((1, N, N, 1), np.float32); m = np.zeros((1, N, N, 1), np.float32)
    e[0, :n, :n] = charge_gn.get_init_edges(pts.astype(np.float32), np.array([]), num=48)[0]
    x[0, :n, :n] = synth.features(rng.choice(["H", "C", "N", "O"], size=n))[None]
    m[0, :n, :n, 0] = 1
    d = [eng.to_device(a) for a in (h, e, x, q, m)]
    out = eng.alloc(N * 4)
    row = []
    for opts in ((-1, 1, 1), (-1, 1, 0), (0, 0, 0)):
        eng.set_option("wave2", opts[0]); eng.set_option("wave3", opts[1]); eng.set_option("dense_rowfused", opts[2])
        def run():
            assert fn(eng.h, 1, N, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, out.ptr) == 0, eng.lib.epnn_last_error()
        for _ in range(20): run()                      # (the first calls of a process load kernels and bring the clocks up)
        eng.sync(); eng.timer_begin()
        for _ in range(50): run()
        row.append(eng.timer_end() / 50)
    print(f"{n:2d}   {row[0]:.3f}        {row[1]:.3f}        {row[2]:.3f}", flush=True)
    for b in d + [out]: b.free()
