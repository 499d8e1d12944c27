import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine
w = checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights"))
eng = Engine(nx=9, T=5, device=0); eng.set_weights(w)
def timed(fn, reps):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
for B in (128, 256, 512, 768, 1024, 1536, 2048, 4096):
    off, bx, bz, bQ, _ = synth.qm9_like_batch(B=B, seed=3, N=29)
    d = [eng.to_device(v) for v in (bx, bz, bQ)]
    dq = eng.alloc(int(off[-1]) * 4)
    row = []
    for thr in (0, 17, 20, 22, 25):
        eng.set_option("wave2", thr)
        def dev():
            eng.forward_xyz_dev(off, d[0], d[1], d[2], dq, 29); eng.sync()
        row.append(timed(dev, 50))
    print(f"B={B:5d} blocking device-resident ms by wave2 = 0 / 17 / 20 / 22 / 25: " + "  ".join(f"{t:.3f}" for t in row), flush=True)
