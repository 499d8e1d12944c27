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
nx, T = 9, 5
w = checkpoint.load_epnn_weights(ROOT + "/models/decay_model_weights")
mi = 3
mols, offsets, xyz, x, Q = load_molecules(vd, [names[mi]], nx)
n = mols[0][1].shape[0]; N = n
y = labs[mi, :n].astype(np.float32)
dense = [orc.dense_inputs(m[0], m[1], m[2], N) for m in mols]
h, e, xd, q, mask = (np.stack([dd[k] for dd in dense]) for k in range(5))
yd = np.zeros((1, N, 1)); yd[0, :n, 0] = labs[mi, :n]
loss, pred, gref = ot.loss_and_grads(h, e, xd, q, mask, yd, w)
gr = ot.flatten(gref)
g32 = ot.flatten(ot.loss_and_grads(h, e, xd, q, mask, yd, w, dtype=np.float32)[2]).astype(np.float64)
res = {}
for fused in (1, 0):
    eng = Engine(nx=nx, T=T); eng.set_option("train_fused", fused); eng.set_weights(w); eng.train_init()
    qq, l = eng.train_step_xyz(offsets, xyz, x, Q, y, N, apply=False)
    res[fused] = (eng.get_gradients().astype(np.float64), qq); eng.close()
print("pred err fused/layered", np.abs(res[1][1] - pred[0, :n, 0]).max(), np.abs(res[0][1] - pred[0, :n, 0]).max())
tn = ["upd"] + [f"msg{t}" for t in range(T)] + [f"pas{t}" for t in range(T)]
pos = 0
for nm, m in zip(tn, [w["upd"]] + w["msg"] + w["pas"]):
    for l, (W, b) in enumerate(m):
        for kind, arr in (("W", W), ("b", b)):
            sl = slice(pos, pos + arr.size); sc = np.abs(gr[sl]).max()
            if sc > 0:
                ef, el, e32 = (np.abs(v[sl] - gr[sl]).max() / sc for v in (res[1][0], res[0][0], g32))
                if ef > 1e-4:
                    k = int(np.abs(res[1][0][sl] - gr[sl]).argmax())
                    print(f"{nm}.{l}.{kind} shape {arr.shape} scale {sc:.2e}: fused {ef:.1e} layered {el:.1e} oracle32 {e32:.1e}; worst entry {np.unravel_index(k, arr.shape)} "
                          f"fused {res[1][0][sl][k]:.6e} layered {res[0][0][sl][k]:.6e} f64 {gr[sl][k]:.6e}")
            pos += arr.size
