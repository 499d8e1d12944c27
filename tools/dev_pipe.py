#!/usr/bin/env python3
"""Throughput with 1, 2, 3 batches in flight (one Engine = one stream + workspace each)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine
w = checkpoint.load_epnn_weights("models/decay_model_weights")
offsets, xyz, x, Q, N = synth.qm9_like_batch(B=1024, seed=0)
A = int(offsets[-1])
for depth in (1, 2, 3):
    engs = []
    for k in range(depth):
        e = Engine(nx=9, T=5); e.set_weights(w)
        d = [e.to_device(a) for a in (xyz, x, Q)]; dq = e.alloc(A * 4)
        engs.append((e, d, dq))
    for k in range(10):
        e, d, dq = engs[k % depth]; e.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
    for e, _, _ in engs: e.sync()
    steps = 100
    t0 = time.perf_counter()
    for k in range(steps):
        e, d, dq = engs[k % depth]; e.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
    for e, _, _ in engs: e.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"depth {depth}: {dt*1e6:.1f} us/step -> {A/dt/1e6:.1f} M atoms/s", flush=True)
    for e, d, dq in engs: e.close()
