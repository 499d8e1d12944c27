"""Randomised check of the train step's gradients against the float64 training oracle (not collected by pytest; run by
hand on a GPU box):   python tests/fuzz_train.py [seed] [seconds]
Random weights, nx 9 / 10, T 1..4, N 3..25, 1..3 molecules per batch; the row-fused kernels, and the layer-by-layer ones
whenever the fused gradient is off by more than 2e-4 of the largest entry.  Round 1 (seed 3, 2141 cases): every case
agrees to ~1e-6 except (a) batches whose loss gradient is numerically zero (nothing to compare) and (b) ONE case in
which the fused gradient is 2.4e-2 off while the layer-by-layer one is 1.4e-6 off: a pre-activation within one float32
ulp of 0 that the two summation orders put on different sides of the ReLU kink (the forward is unaffected, 1.7e-7);
scaling the weights by 1 +- 1e-4 makes both agree with the oracle to 7e-7 again.
Round 2 (seed 24, 1398 cases, 5 flagged): every flagged case is now checked against the oracle's kink bracket
(relu'(z) = [z > +-tau] separately for the GNN rows and the pass network's two row sets, tau = 4e-6): a case whose
gradient lies within the bracket is a ReLU-kink decision, anything else makes the script fail.
Round 3 (seed 31, 5016 cases after the matrix-pipe rewrite of the row-fused kernels, 6 flagged): five kink decisions and one batch
of a one-atom and a three-atom molecule where BOTH implementations are 3.5e-4 off and the float32 oracle itself 3.1e-4 -- the
noise floor of that problem; the tolerance is now max(2e-4, 4 x the float32 oracle's distance from the float64 one), as in
the config-3 test."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_weights                      # noqa: E402
from test_train_oracle import _tiny_batch                # noqa: E402
from epnn_amd.engine import Engine                       # noqa: E402
from oracle import epnn_oracle_train as ot               # noqa: E402


def run(nx, T, w, batch, fused):
    eng = Engine(nx=nx, T=T)
    eng.set_option("train_fused", fused)
    eng.set_weights(w)
    eng.train_init()
    pred, _ = eng.train_step_dense(*batch, apply=False)
    g = eng.get_gradients().astype(np.float64)
    eng.close()
    return pred, g


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    t0, n, flagged, unexplained = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        nx, T, N, B = int(rng.choice([9, 10])), int(rng.integers(1, 5)), int(rng.integers(3, 26)), int(rng.integers(1, 4))
        ns = [int(rng.integers(1, N + 1)) for _ in range(B)]
        w = random_weights(nx, T, seed=int(rng.integers(1 << 30)), scale=0.5)
        batch = _tiny_batch(nx, N, ns, seed=int(rng.integers(1 << 30)))
        _, pr, gr = ot.loss_and_grads(*batch, w)
        gr = ot.flatten(gr)
        glob = np.abs(gr).max()
        pred, g = run(nx, T, w, batch, 1)
        assert np.abs(pred - pr).max() < 3e-5, (nx, T, N, ns)
        n += 1
        if glob < 1e-12:
            continue
        err = np.abs(g - gr).max() / glob
        if err > 2e-4:
            _, g0 = run(nx, T, w, batch, 0)
            flagged += 1
            band = np.zeros_like(gr)
            for where in ("gnn", "listed", "swapped"):
                lo = ot.flatten(ot.loss_and_grads(*batch, w, kink_shift=+4e-6, kink_where=where)[2])
                hi = ot.flatten(ot.loss_and_grads(*batch, w, kink_shift=-4e-6, kink_where=where)[2])
                band += np.abs(hi - lo)
            # the float32 noise floor of the problem itself: the reference algorithm evaluated in float32 against float64 (a batch
            # of one- and three-atom molecules has residuals and gradients so small that float32 rounding of the forward alone
            # moves them by more than 2e-4 of the largest entry -- both implementations are then off by the same amount)
            g32 = ot.flatten(ot.loss_and_grads(*batch, w, dtype=np.float32)[2]).astype(np.float64)
            noise = float(np.abs(g32 - gr).max() / glob)
            out = [float(np.maximum(np.abs(v - gr) - 2 * band, 0).max() / glob) for v in (g, g0)]
            kink = max(out) <= max(2e-4, 4 * noise)
            unexplained += not kink
            print(f"case nx={nx} T={T} N={N} ns={ns}: fused vs oracle {err:.2e}, layer-by-layer vs oracle "
                  f"{np.abs(g0 - gr).max() / glob:.2e}, fused vs layer-by-layer {np.abs(g - g0).max() / glob:.2e}; kink bracket up to "
                  f"{band.max() / glob:.1e} -> " + ("a ReLU-kink decision" if kink else f"NOT explained (outside by {max(out):.1e})") + f"; float32 oracle noise {noise:.1e}")
    print(f"train fuzz: {n} cases, {flagged} flagged, {unexplained} not explained by a ReLU kink")
    if unexplained:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
