#!/usr/bin/env python3
"""Developer check: fused vs layer-by-layer train step on one molecule, varying N."""
import os, sys, tarfile, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_molecules, GOLDEN
from epnn_amd import checkpoint
from epnn_amd.engine import Engine
d = tempfile.mkdtemp(); tarfile.open(GOLDEN + "/mixed_val.tar.gz").extractall(d); vd = d + "/mixed_val"
names = [str(n) for n in np.load(GOLDEN + "/val_names.npy", allow_pickle=True)]
labs = np.load(GOLDEN + "/test_lab_charges.npy")
nx, T = 9, 5
w = checkpoint.load_epnn_weights(ROOT + "/models/decay_model_weights")
tn = ["upd"] + [f"msg{t}" for t in range(T)] + [f"pas{t}" for t in range(T)]
for mi in (0, 3, 7):
  for N in (18, 20, 41):
    mols, offsets, xyz, x, Q = load_molecules(vd, [names[mi]], nx)
    n = mols[0][1].shape[0]
    if n > N: continue
    y = labs[mi, :n].astype(np.float32)
    gs = []
    for fused in (1, 0):
        eng = Engine(nx=nx, T=T); eng.set_option("train_fused", fused); eng.set_option("train_graph", 0); eng.set_weights(w); eng.train_init()
        qq, loss = eng.train_step_xyz(offsets, xyz, x, Q, y, N, apply=False)
        g1 = eng.get_gradients().astype(np.float64)
        qq2, loss2 = eng.train_step_xyz(offsets, xyz, x, Q, y, N, apply=False)
        g2 = eng.get_gradients().astype(np.float64)
        gs.append((g1, qq, loss, np.abs(g1 - g2).max()))
        eng.close()
    (ga, qa, la, ra), (gb, qb, lb, rb) = gs
    pos = 0; rows = []
    for nm, m in zip(tn, [w["upd"]] + w["msg"] + w["pas"]):
        for l, (W, b) in enumerate(m):
            for kind, arr in (("W", W), ("b", b)):
                sl = slice(pos, pos + arr.size); sc = np.abs(gb[sl]).max()
                if sc > 0 and np.abs(ga[sl] - gb[sl]).max() / sc > 2e-4:
                    rows.append(f"{nm}.{l}.{kind} {np.abs(ga[sl] - gb[sl]).max() / sc:.1e}")
                pos += arr.size
    print(f"{names[mi]} n={n} Q={float(Q[0])} N={N}: |dq| fused vs layered {np.abs(qa - qb).max():.1e}; rerun diffs {ra:.1e} {rb:.1e}; tensors off: {rows}", flush=True)
