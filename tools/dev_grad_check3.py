#!/usr/bin/env python3
"""Developer check: isolate the fused train step's error on SSI dimers."""
import os, sys, tarfile, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_molecules, GOLDEN, random_weights
from epnn_amd import checkpoint
from epnn_amd.engine import Engine
d = tempfile.mkdtemp(); tarfile.open(GOLDEN + "/mixed_val.tar.gz").extractall(d); vd = d + "/mixed_val"
names = [str(n) for n in np.load(GOLDEN + "/val_names.npy", allow_pickle=True)]
labs = np.load(GOLDEN + "/test_lab_charges.npy")
nx = 9
wd = checkpoint.load_epnn_weights(ROOT + "/models/decay_model_weights")

def run(tag, w, T, mi, N, y=None, mod=None):
    mols, offsets, xyz, x, Q = load_molecules(vd, [names[mi]], nx)
    if mod: xyz, x = mod(xyz.copy(), x.copy())
    n = x.shape[0]
    yy = labs[mi, :n].astype(np.float32) if y is None else y(n)
    gs = []
    for fused in (1, 0):
        eng = Engine(nx=nx, T=T); eng.set_option("train_fused", fused); eng.set_weights(w); eng.train_init()
        qq, loss = eng.train_step_xyz(offsets, xyz, x, Q, yy, N, apply=False)
        gs.append(eng.get_gradients().astype(np.float64)); eng.close()
    ga, gb = gs
    tn = ["upd"] + [f"msg{t}" for t in range(T)] + [f"pas{t}" for t in range(T)]
    pos = 0; rows = []
    for nm, m in zip(tn, [w["upd"]] + w["msg"] + w["pas"]):
        for l, (W, b) in enumerate(m):
            for kind, arr in (("W", W), ("b", b)):
                sl = slice(pos, pos + arr.size); sc = np.abs(gb[sl]).max()
                if sc > 0: rows.append((np.abs(ga[sl] - gb[sl]).max() / sc, f"{nm}.{l}.{kind}"))
                pos += arr.size
    rows.sort(reverse=True)
    print(f"{tag}: " + ", ".join(f"{n} {e:.1e}" for e, n in rows[:5]), flush=True)

w1 = {"msg": [wd["msg"][4]], "upd": wd["upd"], "pas": [wd["pas"][4]]}
run("dimer decay T=5", wd, 5, 0, 18)
run("dimer decay T=1 (step-4 weights)", w1, 1, 0, 18)
run("dimer random T=5", random_weights(9, 5, seed=9, scale=0.4), 5, 0, 18)
run("dimer random T=1", random_weights(9, 1, seed=9, scale=0.4), 1, 0, 18)
run("qm9 decay T=5", wd, 5, 1, 18)
run("dimer decay T=5, y random", wd, 5, 0, 18, y=lambda n: (np.random.default_rng(0).normal(size=n) * 0.2).astype(np.float32))
run("dimer decay T=5, y = 0", wd, 5, 0, 18, y=lambda n: np.zeros(n, np.float32))
def nosulfur(xyz, x):
    s = x[:, 0] == 16
    x[s] = 0; x[s, 0] = 8; x[s, 4] = 1
    return xyz, x
run("dimer decay T=5, S -> O", wd, 5, 0, 18, mod=nosulfur)
def closer(xyz, x):
    return xyz * 0.8, x
run("dimer decay T=5, coordinates x0.8", wd, 5, 0, 18, mod=closer)
