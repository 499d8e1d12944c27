#!/usr/bin/env python3
"""Developer timing loop: kernel-stage times of the QM9-like batch under a few engine options."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine

def run(opts, steps=30, B=int(os.environ.get("B", "1024")), seed=0):
    w = checkpoint.load_epnn_weights("models/decay_model_weights")
    eng = Engine(nx=9, T=5)
    eng.set_weights(w)
    for k, v in opts.items():
        eng.set_option(k, v)
    offsets, xyz, x, Q, N = synth.qm9_like_batch(B=B, seed=seed)
    d = [eng.to_device(a) for a in (xyz, x, Q)]
    dq = eng.alloc(int(offsets[-1]) * 4)
    for _ in range(5):
        eng.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
    eng.sync()
    eng.set_option("profile", steps)
    for _ in range(steps):
        eng.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
    eng.sync()
    st = np.array([eng.timing_at(k) for k in range(steps)])
    stats = eng.last_stats()
    eng.close()
    return st.mean(0), st.min(0), stats

if __name__ == "__main__":
    for spec in sys.argv[1:] or ["{}"]:
        opts = json.loads(spec)
        mean, mn, stats = run(opts)
        print(spec, "mean ms front/fused/tiled/total", np.round(mean, 4), "min", np.round(mn, 4), "stats", stats, flush=True)
