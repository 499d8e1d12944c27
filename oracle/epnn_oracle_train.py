"""CPU oracle for the training step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/epnn_oracle.py).

Restates reference charge_gn.py:393-402 (train_step) on top of the literal dense forward:
    loss = MSE(y, predictions) over the size-1 last axis -> (1,N) tensor of (y-p)^2        (:397)
    gradients = tape.gradient(loss, vars): TensorFlow sums a non-scalar target -> d/dtheta sum_atoms (y-p)^2   (:398)
    Adam (Keras-2 defaults, :419): lr 1e-3, beta1 0.9, beta2 0.999, eps 1e-7,
        theta -= lr * sqrt(1-beta2^t)/(1-beta1^t) * m / (sqrt(v) + eps)
The backward is written by hand (no autograd is available here) and is itself checked against central finite
differences of the forward in float64 (tests/test_train_oracle.py).

PARITY UNPINNED: the reference ships no stored gradients or optimizer trajectories; TensorFlow is not installed.
"""
from __future__ import annotations

import numpy as np

from . import epnn_oracle as orc


_KINK_SHIFT = 0.0     # see loss_and_grads(kink_shift=...)
_KINK_WHERE = "all"


def _mlp_fwd(rows, layers):
    """Returns output and the layer inputs / pre-activations needed by the backward."""
    acts, pres = [rows], [None]
    x = rows
    for W, b in layers[:-1]:
        z = x @ W + b
        x = np.maximum(z, 0)
        acts.append(x)
        pres.append(z)
    W, b = layers[-1]
    return x @ W + b, (acts, pres)


def _mlp_bwd(dout, tape, layers, where="gnn"):
    """dout: gradient wrt the MLP output.  Returns (d rows, [(dW, db)] per layer)."""
    acts, pres = tape
    shift = _KINK_SHIFT if _KINK_WHERE in ("all", where) else 0.0
    grads = [None] * len(layers)
    W, b = layers[-1]
    grads[-1] = (acts[-1].T @ dout, dout.sum(0))
    d = dout @ W.T
    for l in range(len(layers) - 2, -1, -1):
        d = d * (pres[l + 1] > shift)                  # relu'(z) = [z > 0]; shifted only to bracket float32 kink decisions
        W, b = layers[l]
        grads[l] = (acts[l].T @ d, d.sum(0))
        d = d @ W.T
    return d, grads


def _cast(w, dtype):
    c = lambda m: [(np.asarray(W, dtype), np.asarray(b, dtype)) for W, b in m]
    return {"msg": [c(m) for m in w["msg"]], "upd": c(w["upd"]), "pas": [c(m) for m in w["pas"]]}


def zero_grads(w):
    z = lambda m: [(np.zeros_like(W, dtype=np.float64), np.zeros_like(b, dtype=np.float64)) for W, b in m]
    return {"msg": [z(m) for m in w["msg"]], "upd": z(w["upd"]), "pas": [z(m) for m in w["pas"]]}


def _acc(dst, src):
    for k, (dW, db) in enumerate(src):
        dst[k] = (dst[k][0] + dW, dst[k][1] + db)


def loss_and_grads(h_inp, e_inp, x_inp, q_inp, mask_inp, y, weights, dtype=np.float64, kink_shift=0.0, kink_where="all"):
    """Batch of B molecules (dense make_model inputs).  Returns (loss_sum, predictions (B,N,1), grads dict);
    loss_sum = sum over molecules and atoms of (y - p)^2, i.e. what tape.gradient differentiates when the B
    molecules' gradients are summed (data-parallel training: one molecule per rank, all-reduce sum).

    kink_shift: the backward takes relu'(z) = [z > kink_shift] (forward unchanged).  The gradient of a ReLU network is
    discontinuous where a pre-activation crosses 0; a float32 implementation whose z differs from the float64 one by its
    rounding (~1e-6 here) may land on the other side.  Evaluating with kink_shift = +tau and -tau brackets every such
    decision: where no |z| < tau exists both equal the kink_shift = 0 gradient exactly.  kink_where restricts the shift to
    the GNN's networks ("gnn"), to the pass network's rows [a_i|a_j|e] ("listed") or to its rows [a_j|a_i|e] ("swapped"):
    every pair is evaluated once in each of the two row sets, an implementation may flip one and not the other."""
    global _KINK_SHIFT, _KINK_WHERE
    _KINK_SHIFT, _KINK_WHERE = float(kink_shift), kink_where
    try:
        return _loss_and_grads(h_inp, e_inp, x_inp, q_inp, mask_inp, y, weights, dtype)
    finally:
        _KINK_SHIFT, _KINK_WHERE = 0.0, "all"


def _loss_and_grads(h_inp, e_inp, x_inp, q_inp, mask_inp, y, weights, dtype):
    w = _cast(weights, dtype)
    f32 = [np.asarray(t, dtype=np.float32) for t in (h_inp, e_inp, x_inp, q_inp, mask_inp)]
    h0, x, q0, mask = orc.model_reduce(f32[0], f32[2], f32[3], f32[4], dtype)
    e = f32[1].astype(dtype)
    y = np.asarray(y, dtype=dtype).reshape(q0.shape)
    B, N = e.shape[0], e.shape[1]
    E = e.shape[-1]
    nm = np.clip(mask.sum(axis=1), 0, 1)                                   # (B,N,1)
    T = len(w["msg"])
    tape_g = []
    h = h0
    for t in range(T):
        a = np.concatenate([x, h, q0], -1)
        F = a.shape[-1]
        X = np.concatenate([np.broadcast_to(a[:, :, None, :], (B, N, N, F)),
                            np.broadcast_to(a[:, None, :, :], (B, N, N, F)), e], -1).reshape(B * N * N, -1)
        m, acts = _mlp_fwd(X, w["msg"][t])
        M = m.reshape(B, N, N, -1).sum(2)
        U0 = np.concatenate([h, M], 2) * nm
        hn, uacts = _mlp_fwd(U0.reshape(B * N, -1), w["upd"])
        tape_g.append((acts, uacts, F))
        h = hn.reshape(B, N, -1) * nm
    feats = h
    tol = np.float32(1e-5)
    near = (np.clip(f32[1], tol, np.float32(1e5)).max(-1) != tol).astype(dtype)
    wgt = mask.max(-1) * near                                               # (B,N,N)
    q = q0
    tape_e = []
    for t in range(T):
        a = np.concatenate([x, feats, q], -1)
        F = a.shape[-1]
        ai = np.broadcast_to(a[:, :, None, :], (B, N, N, F))
        aj = np.broadcast_to(a[:, None, :, :], (B, N, N, F))
        fN, actsN = _mlp_fwd(np.concatenate([ai, aj, e], -1).reshape(B * N * N, -1), w["pas"][t])
        fT, actsT = _mlp_fwd(np.concatenate([aj, ai, e], -1).reshape(B * N * N, -1), w["pas"][t])
        anti = 0.5 * (fN.reshape(B, N, N) - fT.reshape(B, N, N)) * wgt
        q = q + anti.sum(2)[..., None]
        tape_e.append((actsN, actsT, F))
    pred = q
    loss = float(((y - pred) ** 2).sum())

    # ------------------------------------------------------------------ backward
    g = zero_grads(w)
    gq = -2.0 * (y - pred)                                                  # (B,N,1)
    gfeat = np.zeros_like(feats)
    nh = feats.shape[-1]
    nx = x.shape[-1]
    for t in range(T - 1, -1, -1):
        actsN, actsT, F = tape_e[t]
        ganti = np.broadcast_to(gq, (B, N, N)) * wgt                        # d q_i / d anti_ij = 1
        dN = (0.5 * ganti).reshape(-1, 1)
        dT = (-0.5 * ganti).reshape(-1, 1)
        dXN, gN = _mlp_bwd(dN, actsN, w["pas"][t], "listed")
        dXT, gT = _mlp_bwd(dT, actsT, w["pas"][t], "swapped")
        _acc(g["pas"][t], gN)
        _acc(g["pas"][t], gT)
        dXN = dXN.reshape(B, N, N, -1)
        dXT = dXT.reshape(B, N, N, -1)
        # rows of X_N are [a_i | a_j | e], rows of X_T are [a_j | a_i | e]
        ga = dXN[..., :F].sum(2) + dXN[..., F:2 * F].sum(1) + dXT[..., :F].sum(1) + dXT[..., F:2 * F].sum(2)
        gfeat = gfeat + ga[..., nx:nx + nh]
        gq = gq + ga[..., nx + nh:nx + nh + 1]
    gh = gfeat
    for t in range(T - 1, -1, -1):
        acts, uacts, F = tape_g[t]
        dhn = (gh * nm).reshape(B * N, -1)
        dU0, gu = _mlp_bwd(dhn, uacts, w["upd"])
        _acc(g["upd"], gu)
        dU0 = dU0.reshape(B, N, -1) * nm
        gh_prev = dU0[..., :nh]
        gM = dU0[..., nh:]
        dm = np.broadcast_to(gM[:, :, None, :], (B, N, N, gM.shape[-1])).reshape(B * N * N, -1)
        dX, gm = _mlp_bwd(dm, acts, w["msg"][t])
        _acc(g["msg"][t], gm)
        dX = dX.reshape(B, N, N, -1)
        ga = dX[..., :F].sum(2) + dX[..., F:2 * F].sum(1)
        gh = gh_prev + ga[..., nx:nx + nh]                                  # q is a constant inside the GNN
    return loss, pred, g


def flatten(wd):
    """Keras trainable_variables order (charge_gn.py:371-374): update MLP, message MLPs, pass MLPs; kernel, bias."""
    out = []
    for m in [wd["upd"]] + list(wd["msg"]) + list(wd["pas"]):
        for W, b in m:
            out += [np.asarray(W, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()]
    return np.concatenate(out)


def unflatten(vec, like):
    out = {"msg": [], "upd": None, "pas": []}
    pos = 0

    def take(m):
        nonlocal pos
        res = []
        for W, b in m:
            Wn = vec[pos:pos + W.size].reshape(W.shape)
            pos += W.size
            bn = vec[pos:pos + b.size].reshape(b.shape)
            pos += b.size
            res.append((Wn, bn))
        return res

    out["upd"] = take(like["upd"])
    out["msg"] = [take(m) for m in like["msg"]]
    out["pas"] = [take(m) for m in like["pas"]]
    return out


class Adam:
    """tf.keras.optimizers.Adam() with Keras-2 defaults (charge_gn.py:419)."""

    def __init__(self, n, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, dtype=np.float64):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = np.zeros(n, dtype)
        self.v = np.zeros(n, dtype)
        self.t = 0

    def step(self, theta, grad):
        self.t += 1
        self.m = self.b1 * self.m + (1 - self.b1) * grad
        self.v = self.b2 * self.v + (1 - self.b2) * grad * grad
        alpha = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        return theta - alpha * self.m / (np.sqrt(self.v) + self.eps)
