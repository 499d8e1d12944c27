"""The training oracle's hand-written backward vs central finite differences of its forward (float64). CPU only."""
import os

import numpy as np

from conftest import random_weights


def _tiny_batch(nx, N, ns, seed=0):
    from oracle import epnn_oracle as orc
    rng = np.random.default_rng(seed)
    B = len(ns)
    h = np.zeros((B, N, N, 48)); e = np.zeros((B, N, N, 48)); x = np.zeros((B, N, N, nx))
    q = np.zeros((B, N, N, 1)); mask = np.zeros((B, N, N)); y = np.zeros((B, N, 1))
    for b, n in enumerate(ns):
        xyz = (rng.normal(size=(n, 3)) * 1.3).astype(np.float32)
        feat = np.zeros((n, nx), np.float32)
        feat[:, 0] = rng.choice([1, 6, 7, 8], size=n)
        feat[np.arange(n), 1 + rng.integers(0, 4, size=n)] = 1
        hp, ep, xp, qp, mk = orc.dense_inputs(xyz, feat, np.float32(rng.integers(-1, 2)), N)
        h[b], e[b], x[b], q[b], mask[b] = hp, ep, xp, qp, mk
        y[b, :n, 0] = rng.normal(size=n) * 0.3
    return h, e, x, q, mask, y


def test_backward_matches_finite_differences():
    from oracle import epnn_oracle_train as ot
    nx, T = 9, 2
    w = random_weights(nx, T, seed=4, scale=0.5)
    h, e, x, q, mask, y = _tiny_batch(nx, 6, [5, 3])
    loss, pred, g = ot.loss_and_grads(h, e, x, q, mask, y, w)
    theta = ot.flatten(w)
    gflat = ot.flatten(g)
    assert gflat.shape == theta.shape
    rng = np.random.default_rng(0)
    idx = np.concatenate([rng.choice(theta.size, size=120, replace=False), np.flatnonzero(np.abs(gflat) > 1e-3)[:40]])
    worst = 0.0
    for k in idx:
        d = 1e-5 * max(1.0, abs(theta[k]))
        tp, tm = theta.copy(), theta.copy()
        tp[k] += d
        tm[k] -= d
        lp = ot.loss_and_grads(h, e, x, q, mask, y, ot.unflatten(tp, w))[0]
        lm = ot.loss_and_grads(h, e, x, q, mask, y, ot.unflatten(tm, w))[0]
        fd = (lp - lm) / (2 * d)
        worst = max(worst, abs(fd - gflat[k]) / max(1e-6, abs(fd), abs(gflat[k])) if max(abs(fd), abs(gflat[k])) > 1e-7 else 0.0)
    assert worst < 2e-5, worst
    # structure the reference's gradients have: the last pass bias gets exactly zero gradient (cancels in f_ij - f_ji)
    assert all(np.all(m[2][1] == 0) for m in g["pas"])


def test_adam_matches_keras_formula():
    from oracle import epnn_oracle_train as ot
    opt = ot.Adam(3)
    theta = np.array([1.0, -2.0, 0.5])
    grad = np.array([0.1, -0.3, 0.0])
    t1 = opt.step(theta, grad)
    # first step of Adam moves every parameter with non-zero gradient by ~lr * sign(grad)
    np.testing.assert_allclose(t1 - theta, [-1e-3, 1e-3, 0.0], rtol=1e-4, atol=1e-12)   # eps=1e-7 shifts it by ~3e-5
    t2 = opt.step(t1, grad)
    assert np.all(np.abs(t2 - t1)[:2] < 1.01e-3)
