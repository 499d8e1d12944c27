"""Randomised check of the layer-level entries (GNN_layer.call, EPN_layer.call) on ARBITRARY dense inputs -- asymmetric e,
non-zero diagonal, fractional and asymmetric masks, padded atoms, systems of 1..47 atoms -- against the float64 oracle
(not collected by pytest; run by hand on a GPU box):   python tests/fuzz_dense.py [seed] [seconds]
Round 1 found with it: a fractional node mask (only possible when an atom's mask column sums to less than 1, i.e. in
tiny systems with fractional masks) was applied twice to the h block of the first update step.
Round 2 (seed 22, 90 s): 551 cases, worst error 4.0 % of the tolerance."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_weights          # noqa: E402
from epnn_amd import charge_gn               # noqa: E402
from oracle import epnn_oracle as orc        # noqa: E402


def random_case(rng, nx=9):
    T, B, N = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(2, 68))
    e = np.zeros((B, N, N, 48), np.float32)
    mask = np.zeros((B, N, N, 1), np.float32)
    x = np.zeros((B, N, nx), np.float32)
    h = np.zeros((B, N, 48), np.float32)
    q = np.zeros((B, N, 1), np.float32)
    for b in range(B):
        nr = int(rng.integers(1, N + 1))
        dense = rng.random((nr, nr)) < rng.uniform(0.1, 0.9)
        ev = (rng.random((nr, nr, 48)) * 0.3 * dense[..., None]).astype(np.float32)
        sym = rng.random((nr, nr)) < 0.5
        e[b, :nr, :nr] = np.where(sym[..., None] & sym.T[..., None], np.maximum(ev, ev.transpose(1, 0, 2)), ev)
        mask[b, :nr, :nr, 0] = rng.choice([0.0, 0.5, 1.0], size=(nr, nr), p=[0.1, 0.2, 0.7])
        x[b, :nr, 0] = rng.choice([1, 6, 7, 8], size=nr)
        x[b, np.arange(nr), 1 + rng.integers(0, 4, size=nr)] = 1
        h[b, :nr] = rng.normal(size=(nr, 48)) * 0.2
        q[b, :nr, 0] = rng.normal(size=nr) * 0.1
    return T, h, e, x, q, mask


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < budget:
        T, h, e, x, q, mask = random_case(rng)
        w = random_weights(9, T, seed=int(rng.integers(1 << 30)), scale=0.35)
        gnn = charge_gn.GNN_layer(charge_gn.MLP_layer, charge_gn.MLP_layer([32, 32], out_dim=48), T)
        epn = charge_gn.EPN_layer(charge_gn.MLP_layer, T=T)
        for t in range(T):
            gnn.message_fns[t].set_weights(w["msg"][t])
            epn.pass_fns[t].set_weights(w["pas"][t])
        gnn.update_fn.set_weights(w["upd"])
        h_gpu = gnn.call(h, e, x, q, mask)
        h_ref = orc.gnn_layer(h, e, x, q, mask, w["msg"], w["upd"], dtype=np.float64)
        h_r32 = orc.gnn_layer(h, e, x, q, mask, w["msg"], w["upd"], dtype=np.float32)
        eh = np.abs(h_gpu - h_ref).max() / max(1e-5, 3 * np.abs(h_r32 - h_ref).max())
        q_gpu = epn.call(h, e, x, q, mask)
        q_ref = orc.epn_layer(h, e, x, q, mask, w["pas"], dtype=np.float64)
        q_r32 = orc.epn_layer(h, e, x, q, mask, w["pas"], dtype=np.float32)
        eq = np.abs(q_gpu - q_ref).max() / max(1e-5, 3 * np.abs(q_r32 - q_ref).max())
        worst = max(worst, eh, eq)
        if max(eh, eq) > 1:
            print("FAIL", dict(T=T, B=e.shape[0], N=e.shape[1], gnn=float(eh), epn=float(eq)))
            sys.exit(1)
        n += 1
    print(f"dense fuzz ok: {n} cases, worst err / tolerance {worst:.3f}")


if __name__ == "__main__":
    main()
