#!/usr/bin/env python3
import os, sys, tarfile, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_molecules, GOLDEN
from oracle import epnn_oracle as orc, epnn_oracle_train as ot
from epnn_amd import checkpoint
from epnn_amd.engine import Engine
d = tempfile.mkdtemp(); tarfile.open(GOLDEN + "/mixed_val.tar.gz").extractall(d); vd = d + "/mixed_val"
names = [str(n) for n in np.load(GOLDEN + "/val_names.npy", allow_pickle=True)]
labs = np.load(GOLDEN + "/test_lab_charges.npy")
nx, T, N = 9, 5, 18
w = checkpoint.load_epnn_weights(ROOT + "/models/decay_model_weights")
mols, offsets, xyz, x, Q = load_molecules(vd, [names[0]], nx)
n = 18
y = labs[0, :n].astype(np.float32)
dense = [orc.dense_inputs(m[0], m[1], m[2], N) for m in mols]
h, e, xd, q, mask = (np.stack([dd[k] for dd in dense]) for k in range(5))
yd = np.zeros((1, N, 1)); yd[0, :n, 0] = labs[0, :n]
refs = {s: ot.flatten(ot.loss_and_grads(h, e, xd, q, mask, yd, w, kink_shift=s)[2]) for s in (0.0, 1e-6, 2e-5, -2e-5, 1e-4, 1e-3)}
tn = ["upd"] + [f"msg{t}" for t in range(T)] + [f"pas{t}" for t in range(T)]
for fused in (1, 0):
    eng = Engine(nx=nx, T=T); eng.set_option("train_fused", fused); eng.set_weights(w); eng.train_init()
    qq, loss = eng.train_step_xyz(offsets, xyz, x, Q, y, N, apply=False)
    g = eng.get_gradients().astype(np.float64); eng.close()
    for s, gr in refs.items():
        pos = 0; rows = []
        for nm, m in zip(tn, [w["upd"]] + w["msg"] + w["pas"]):
            for l, (W, b) in enumerate(m):
                for kind, arr in (("W", W), ("b", b)):
                    sl = slice(pos, pos + arr.size); sc = np.abs(refs[0.0][sl]).max()
                    if sc > 0: rows.append((np.abs(g[sl] - gr[sl]).max() / sc, f"{nm}.{l}.{kind}"))
                    pos += arr.size
        rows.sort(reverse=True)
        print(f"fused={fused} vs oracle with relu'(z)=[z>{s}]: " + ", ".join(f"{n} {e:.1e}" for e, n in rows[:4]), flush=True)
