#!/usr/bin/env python3
"""Wall-clock per forward without any profiling events (what bench.py's `value` sees)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine
eng = Engine(nx=9, T=5); eng.set_weights(checkpoint.load_epnn_weights("models/decay_model_weights"))
for k, v in [a.split("=") for a in sys.argv[1:]]:
    eng.set_option(k, int(v))
offsets, xyz, x, Q, N = synth.qm9_like_batch(B=1024, seed=0)
d = [eng.to_device(a) for a in (xyz, x, Q)]; dq = eng.alloc(int(offsets[-1]) * 4)
for _ in range(10): eng.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
eng.sync()
for steps in (50, 200):
    t0 = time.perf_counter()
    for _ in range(steps): eng.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
    t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    print(f"steps={steps}: enqueue {1e6*(t1-t0)/steps:.1f} us/step, total {1e6*(t2-t0)/steps:.1f} us/step", flush=True)
