#!/usr/bin/env python3
"""Developer check: Pipeline.map on rotating batches, one configuration, for a rocprofv3 --kernel-trace."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Pipeline
depth, nb = int(sys.argv[1]), int(sys.argv[2])
w = checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights"))
pipe = Pipeline(depth=depth, nx=9, T=5, device=0)
pipe.set_weights(w)
batches = [synth.qm9_like_batch(B=1024, seed=1000 + s, N=29)[:4] for s in range(nb)]
stream = [batches[k % nb] for k in range(600)]
for rep in range(2):
    t0 = time.perf_counter()
    atoms = sum(q.shape[0] for q in pipe.map(stream, 29))
    dt = time.perf_counter() - t0
print(f"depth {depth} batches {nb}: {dt / len(stream) * 1e6:.1f} us per call, {atoms / dt / 1e6:.1f} M atoms/s", flush=True)
pipe.close()
