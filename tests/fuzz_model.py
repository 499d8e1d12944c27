"""Randomised check of the literal make_model call (epnn_model_forward_dense: the (B,N,N,.) tensors and their reductions,
charge_gn.py:382-384) on arbitrary inputs -- tiled like the featuriser's or not, rank-3 or rank-4 mask -- against the float64
oracle (not collected by pytest; run by hand on a GPU box: python tests/fuzz_model.py [seed] [seconds]).  Round 1: 1867 cases, worst error 6 % of the tolerance.
Round 2 (seed 23, 60 s): 1673 cases, worst error 5.5 % of the tolerance.  Round 5: one case in three with h_dim in 1..47 and
other `layers` of the update MLP (seed 91, 90 s: 849 cases, worst 5.9 %)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_weights
from epnn_amd import charge_gn
from oracle import epnn_oracle as orc
from fuzz_dense import random_case
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 21)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
t0 = time.time(); n = 0; worst = 0
while time.time() - t0 < budget:
    T, h, e, x, q, mask = random_case(rng)
    B, N = e.shape[:2]
    # one case in three: a model with h_dim (= the channels of e, charge_gn.py:376-377) below the scripts' 48 and other `layers`
    hd = 48 if rng.random() < 0.67 else int(rng.integers(1, 48))
    layers = [32, 32] if hd == 48 else [int(v) for v in rng.integers(1, 70, size=int(rng.integers(1, 4)))]
    w = random_weights(9, T, seed=int(rng.integers(1 << 30)), scale=0.35, h_dim=hd)
    if layers != [32, 32]:
        dims = [hd + 32] + layers + [hd]
        w["upd"] = [(rng.uniform(-1, 1, (i, o)).astype(np.float32) * np.float32(0.35 * np.sqrt(6.0 / (i + o))), rng.uniform(-0.1, 0.1, (o,)).astype(np.float32))
                    for i, o in zip(dims[:-1], dims[1:])]
    h, e = np.ascontiguousarray(h[..., :hd]), np.ascontiguousarray(e[..., :hd])      # (any e is a valid input of the dense call)
    # literal make_model inputs: (B,N,N,.) tensors, NOT necessarily tiled: random per-pair values times the mask
    mode = int(rng.integers(0, 3))
    if mode == 0:      # tiled like gen_padded_init_state (row j*n+k of the tiled per-atom array is atom k)
        h_inp = np.broadcast_to(h[:, None, :, :], (B, N, N, hd)) * (mask > 0)
        x_inp = np.broadcast_to(x[:, None, :, :], (B, N, N, 9)) * (mask > 0)
        q_inp = np.broadcast_to(q[:, None, :, :], (B, N, N, 1)) * (mask > 0)
    else:              # arbitrary
        h_inp = (rng.normal(size=(B, N, N, hd)) * 0.2).astype(np.float32) * (rng.random((B, N, N, 1)) < 0.7)
        x_inp = np.broadcast_to(x[:, None, :, :], (B, N, N, 9)) * (rng.random((B, N, N, 1)) < 0.8)
        q_inp = (rng.normal(size=(B, N, N, 1)) * 0.1).astype(np.float32)
    h_inp, x_inp, q_inp = (np.ascontiguousarray(t, dtype=np.float32) for t in (h_inp, x_inp, q_inp))
    m_inp = mask if mode < 2 else mask[..., 0]          # rank-3 mask is accepted like Keras does
    model = charge_gn.make_model(layers, hd, T, 9, N)
    model.set_weights_dict(w)
    p = model([h_inp, e, x_inp, q_inp, m_inp])
    ref = orc.model_forward(h_inp, e, x_inp, q_inp, m_inp, w, np.float64)
    r32 = orc.model_forward(h_inp, e, x_inp, q_inp, m_inp, w, np.float32)
    err = np.abs(p - ref).max() / max(1e-5, 3 * np.abs(r32 - ref).max())
    worst = max(worst, err)
    if err > 1:
        print("FAIL", dict(T=T, B=B, N=N, mode=mode, h_dim=hd, layers=layers, err=float(err))); sys.exit(1)
    n += 1
print(f"model fuzz ok: {n} cases, worst err / tolerance {worst:.3f}")
