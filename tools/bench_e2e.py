#!/usr/bin/env python3
"""Host-to-host rate of the compact entry (PCIe inclusive): xyz/x/Q in host memory -> charges in host memory.
Not the bench's `value` (that one starts with the inputs resident in HBM); quoted in DESIGN.md section 5."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine
eng = Engine(nx=9, T=5)
eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
offsets, xyz, x, Q, N = synth.qm9_like_batch(B=B, seed=0)
for _ in range(5):
    q = eng.forward_xyz(offsets, xyz, x, Q, N)
t0 = time.perf_counter()
K = 50
for _ in range(K):
    q = eng.forward_xyz(offsets, xyz, x, Q, N)
dt = (time.perf_counter() - t0) / K
print(f"host->host, one call at a time: {dt * 1e3:.3f} ms per batch of {B} molecules, {offsets[-1] / dt / 1e6:.1f} M atoms/s")
eng.close()

# the same through Pipeline.map: DIFFERENT batches (a new plan every call), several in flight
from epnn_amd.engine import Pipeline
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pipe = Pipeline(depth=depth, nx=9, T=5)
pipe.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights")))
batches = [synth.qm9_like_batch(B=B, seed=s) for s in range(12)]
N = max(b[4] for b in batches)
stream = [batches[k % 12][:4] for k in range(240)]
for q in pipe.map(stream[:24], N):
    pass
t0 = time.perf_counter()
atoms = 0
for q in pipe.map(stream, N):
    atoms += q.shape[0]
dt = (time.perf_counter() - t0) / len(stream)
print(f"host->host, Pipeline.map depth {depth}, a different batch every call: {dt * 1e3:.3f} ms per batch of {B} molecules, "
      f"{atoms / len(stream) / dt / 1e6:.1f} M atoms/s")
pipe.close()
