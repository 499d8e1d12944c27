"""Randomised check of the forward against the float64 oracle (not collected by pytest; run by hand on a GPU box):
    python tests/fuzz_forward.py [seed] [seconds]
Random weights, nx 9 / 10, T 1..5, padded size 8..69, batches mixing molecules of 1..60 atoms (fused and tiled kernels),
both front-ends.  Round 1: 518 batches, worst error 7 % of the tolerance max(1e-5, 4 x float32 noise of the oracle); round 2
(seed 21, 90 s, after the tiled path's fused tails and the wave priority): 157 batches, worst 3.9 %; with the three-block
kernel for 33..48 atoms in the mix (seeds 41, 42): 344 batches, worst 4.7 %; with the block-per-wavefront kernel
(molecules of 17..32 atoms split over two wavefronts, smaller ones in pairs; seeds 51, 52): 309 batches, worst 7.0 %; with 33..48 atoms on three wavefronts (seeds 61, 62): 325 batches, worst 5.9 %.
Round 5: two batches in five with other `layers` of the update MLP (1..3 hidden layers of 1..72 units: the [32, 32] kernels on a padded
copy, the 64-unit fused kernel, the generic update stage; seeds 81, 82: 412 batches, worst 5.4 %), one in four with h_dim = e_dim in 1..47 (seed 92: 193 batches, worst 4.6 %)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_weights
from epnn_amd.engine import Engine
from oracle import epnn_oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); worst = 0.0; ncase = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    nx = int(rng.choice([9, 10])); T = int(rng.integers(1, 6)); N = int(rng.integers(8, 70))
    hd = 48 if rng.random() < 0.75 else int(rng.integers(1, 48))      # h_dim = e_dim of the model (gen_padded_init_state(path, h_dim, e_dim))
    w = random_weights(nx, T, seed=int(rng.integers(1 << 30)), scale=float(rng.uniform(0.2, 0.5)), h_dim=hd)
    layers = None
    if rng.random() < 0.4:       # make_model(layers): one to three hidden layers of 1..72 units in the update MLP (charge_gn.py:369-371)
        layers = [int(v) for v in rng.integers(1, 73, size=int(rng.integers(1, 4)))]
        dims = [hd + 32] + layers + [hd]
        sc = float(rng.uniform(0.2, 0.5))
        w["upd"] = [(rng.uniform(-1, 1, (i, o)).astype(np.float32) * np.float32(sc * np.sqrt(6.0 / (i + o))), rng.uniform(-0.1, 0.1, (o,)).astype(np.float32))
                    for i, o in zip(dims[:-1], dims[1:])]
    B = int(rng.integers(1, 12))
    mols = []
    for b in range(B):
        n = int(rng.integers(1, min(N, 60) + 1))
        dens = rng.uniform(0.7, 2.0)
        xyz = (rng.normal(size=(n, 3)) * dens * max(1.0, n ** (1 / 3)) * 0.8).astype(np.float32)
        x = np.zeros((n, nx), np.float32); el = rng.integers(1, nx, size=n); x[np.arange(n), el] = 1; x[:, 0] = rng.integers(1, 10, size=n)
        mols.append((xyz, x, np.float32(rng.integers(-2, 3))))
    off = np.zeros(B + 1, np.int32); off[1:] = np.cumsum([m[1].shape[0] for m in mols])
    eng = Engine(nx=nx, T=T, h_dim=hd, e_dim=hd); eng.set_weights(w)
    for front in (1, 0):
        eng.set_option("wave_front", front)
        q = eng.forward_xyz(off, np.concatenate([m[0] for m in mols]), np.concatenate([m[1] for m in mols]), np.array([m[2] for m in mols], np.float32), N)
        for k, m in enumerate(mols):
            ref = orc.forward_xyz(m[0], m[1], m[2], w, N=N, dtype=np.float64, h_dim=hd)
            ref32 = orc.forward_xyz(m[0], m[1], m[2], w, N=N, dtype=np.float32, h_dim=hd)
            n = m[1].shape[0]
            err = np.abs(q[off[k]:off[k + 1]] - ref[:n]).max(); noise = np.abs(ref32 - ref).max()
            worst = max(worst, err / max(1e-5, 4 * noise))
            if err > max(1e-5, 4 * noise):
                print("FAIL", dict(nx=nx, T=T, N=N, n=n, front=front, layers=layers, h_dim=hd, err=float(err), noise=float(noise))); sys.exit(1)
            assert abs(float(q[off[k]:off[k + 1]].sum(dtype=np.float64)) - float(m[2])) < 5e-5
    eng.close(); ncase += 1
print(f"fuzz ok: {ncase} batches, worst err / tolerance {worst:.3f}")
