"""Training step on the GPU (epnn_train_step_*) vs the float64 training oracle. GPU only."""
import os

import numpy as np
import pytest

from conftest import random_weights
from test_train_oracle import _tiny_batch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-12, np.abs(b).max())


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("nx,T,N,ns", [(9, 2, 8, [6, 8]), (10, 3, 12, [12, 5, 9])])
def test_gradients_match_oracle(gpu_engine_factory, nx, T, N, ns, fused):
    """Gradient of sum (y-p)^2 over a small batch: float32 HIP kernels vs the finite-difference-checked float64
    oracle, per parameter tensor, plus the structural zero (last pass bias)."""
    from oracle import epnn_oracle_train as ot
    w = random_weights(nx, T, seed=3, scale=0.5)
    h, e, x, q, mask, y = _tiny_batch(nx, N, ns, seed=2)
    loss_ref, pred_ref, g_ref = ot.loss_and_grads(h, e, x, q, mask, y, w)
    eng = gpu_engine_factory(nx=nx, T=T)
    eng.set_option("train_fused", fused)       # 1: one workgroup per atom and pair MLP; 0: one launch per Dense layer
    eng.set_weights(w)
    eng.train_init()
    pred, loss = eng.train_step_dense(h, e, x, q, mask, y, apply=False)
    assert np.abs(pred - pred_ref).max() < 2e-5
    assert abs(loss - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref))
    g = eng.get_gradients().astype(np.float64)
    gr = ot.flatten(g_ref)
    assert g.shape == gr.shape == (eng.param_count(),)
    # per tensor: relative to that tensor's largest gradient entry
    pos = 0
    worst = 0.0
    for m in [w["upd"]] + w["msg"] + w["pas"]:
        for W, b in m:
            for arr in (W, b):
                sl = slice(pos, pos + arr.size)
                scale = np.abs(gr[sl]).max()
                if scale > 0:
                    worst = max(worst, np.abs(g[sl] - gr[sl]).max() / scale)
                else:
                    assert np.all(g[sl] == 0)
                pos += arr.size
    print(f"fused={fused} nx={nx} T={T} N={N}: worst per-tensor relative gradient error {worst:.2e}; loss {loss:.6f} vs {loss_ref:.6f}")
    assert worst < 2e-4
    # weights untouched with apply=False
    w2 = eng.get_weights()
    assert np.array_equal(w2["msg"][0][0][0], w["msg"][0][0][0])


@pytest.fixture(scope="module")
def config3_case(val_dir, val_names, golden_dir, weights_decay):
    """Inputs and float64 / float32 oracle gradients of the config-3 gradient test (shared by the two implementations)."""
    from conftest import load_molecules
    from oracle import epnn_oracle as orc
    from oracle import epnn_oracle_train as ot
    nx, T, N = 9, 5, 41
    labs = np.load(os.path.join(golden_dir, "test_lab_charges.npy"))
    sizes = [orc.parse_xyz(os.path.join(val_dir, nm + ".xyz"), nx)[1].shape[0] for nm in val_names]
    pick = [val_names.index("dsgdb9nsd_081300"), val_names.index("SSI-081ILE-085ARG-1-dimer"), int(np.argmax(sizes))]
    assert sizes[pick[2]] == 38                              # the split's largest; the directory maximum 41 is a training file
    mols, offsets, xyz, x, Q = load_molecules(val_dir, [val_names[i] for i in pick], nx)
    y = np.concatenate([labs[i, :sizes[i]] for i in pick]).astype(np.float32)
    dense = [orc.dense_inputs(m[0], m[1], m[2], N) for m in mols]
    h, e, xd, q, mask = (np.stack([d[k] for d in dense]) for k in range(5))
    yd = np.zeros((len(mols), N, 1))
    for b, i in enumerate(pick):
        yd[b, :sizes[i], 0] = labs[i, :sizes[i]]
    out = {"batch": (offsets, xyz, x, Q, y), "sizes": [sizes[i] for i in pick], "N": N}
    # ReLU kinks.  The gradient jumps where a pre-activation crosses 0, and a float32 evaluation of a tiny |z| may land on
    # either side -- every row of the pass network is evaluated twice (as [a_i|a_j|e] of atom i and as the swapped row of
    # atom j), each in its own summation order.  With the shipped weights SSI-001ASN-030MET-1-dimer has z = 2.9e-7 in the
    # last pass network: the two train-step implementations then differ by 1e-2 in that layer's bias gradient, the float64
    # value lies between.  The oracle brackets such decisions (relu'(z) = [z > +tau] / [z > -tau], separately for the GNN's
    # rows and for each of the pass network's two row sets).  The molecules are chosen so that the shipped weights have no
    # such decision within tau = 2e-5 (|z| reaches ~90 there; asserted).  With random weights (|z| = O(1), tau = 2e-6) the
    # 2.4 M pre-activations of a forward always include a few: their bracket widens the tolerance of the tensors they touch.
    for which, w, tau in (("decay_model_weights", weights_decay, 2e-5), ("random", random_weights(nx, T, seed=13, scale=0.4), 2e-6)):
        loss_ref, pred_ref, g_ref = ot.loss_and_grads(h, e, xd, q, mask, yd, w)
        gr = ot.flatten(g_ref)
        g32 = ot.flatten(ot.loss_and_grads(h, e, xd, q, mask, yd, w, dtype=np.float32)[2]).astype(np.float64)
        band = np.zeros_like(gr)
        for where in ("gnn", "listed", "swapped"):
            g_lo = ot.flatten(ot.loss_and_grads(h, e, xd, q, mask, yd, w, kink_shift=+tau, kink_where=where)[2])
            g_hi = ot.flatten(ot.loss_and_grads(h, e, xd, q, mask, yd, w, kink_shift=-tau, kink_where=where)[2])
            band += np.abs(g_hi - g_lo)
        if which == "decay_model_weights":
            assert band.max() == 0.0
        out[which] = (w, loss_ref, pred_ref, gr, g32, band)
    return out


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("which", ["decay_model_weights", "random"])
def test_gradients_match_oracle_at_config3_shape(gpu_engine_factory, config3_case, which, fused):
    """BASELINE.json configs[2] shape: N = 41, T = 5, real molecules of the reference's `mixed` set (a QM9 molecule, a
    charged SSI dimer, the largest system of the validation split) with their stored MBIS labels, through the entry
    train.py uses (train_step_xyz).  Gradient of sum (y-p)^2 (charge_gn.py:397-398) per parameter tensor vs the float64
    oracle, with the shipped checkpoint (collapsed GNN: its message / update tensors get exactly zero gradient in both)
    and with random non-degenerate weights, for both train-step implementations.  Tolerance per tensor: 2e-4 of its
    largest entry, or 4x the float32 noise of the reference algorithm itself (the float32 oracle against the float64
    one) where that is larger: with a trained model the residuals y - p are ~1e-3, so the float32 rounding of p alone
    (3e-7) moves every gradient by ~3e-4 of its size."""
    nx, T, N = 9, 5, config3_case["N"]
    offsets, xyz, x, Q, y = config3_case["batch"]
    w, loss_ref, pred_ref, gr, g32, band = config3_case[which]
    eng = gpu_engine_factory(nx=nx, T=T)
    eng.set_option("train_fused", fused)
    eng.set_weights(w)
    eng.train_init()
    qq, loss = eng.train_step_xyz(offsets, xyz, x, Q, y, N, apply=False)
    for b, n in enumerate(config3_case["sizes"]):
        assert np.abs(qq[offsets[b]:offsets[b + 1]] - pred_ref[b, :n, 0]).max() < 2e-5
    assert abs(loss - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref))
    g = eng.get_gradients().astype(np.float64)
    pos, worst, worst_noise, zero_tensors, kinked = 0, 0.0, 0.0, 0, 0
    table = []
    for mi, m in enumerate([w["upd"]] + w["msg"] + w["pas"]):
        for li, (W, b) in enumerate(m):
            for arr in (W, b):
                sl = slice(pos, pos + arr.size)
                scale = np.abs(gr[sl]).max()
                if scale > 0:
                    noise = np.abs(g32[sl] - gr[sl]).max() / scale
                    kink = band[sl].max() / scale                          # 0 for every tensor with the shipped weights
                    err = np.abs(g[sl] - gr[sl]).max() / scale
                    kinked += kink > 0
                    table.append((mi, li, arr.shape, float(err), float(noise)))
                    if os.environ.get("EPNN_TEST_VERBOSE"):
                        print(table[-1])
                    if kink == 0:
                        worst, worst_noise = max(worst, err), max(worst_noise, noise)
                    assert err <= max(2e-4, 4 * noise) + 2 * kink, (err, noise, kink)
                else:
                    zero_tensors += 1
                    assert np.all(g[sl] == 0)
                pos += arr.size
    print(f"{which} fused={fused} N=41 T=5: worst per-tensor relative gradient error {worst:.2e} (float32 oracle: {worst_noise:.2e}) over the "
          f"tensors no ReLU kink touches; {kinked} touched, {zero_tensors} with zero gradient; loss {loss:.6f} vs {loss_ref:.6f}")


def test_adam_trajectory_matches_oracle(gpu_engine_factory):
    """Ten optimizer steps (one small batch per step, like charge_gn.py:443-451) vs the oracle's Adam in float64."""
    from oracle import epnn_oracle_train as ot
    nx, T, N = 9, 2, 8
    w = random_weights(nx, T, seed=8, scale=0.5)
    batches = [_tiny_batch(nx, N, [7, 5], seed=s) for s in range(3)]
    eng = gpu_engine_factory(nx=nx, T=T)
    eng.set_weights(w)
    eng.train_init()
    theta = ot.flatten(w)
    opt = ot.Adam(theta.size)
    losses, losses_ref = [], []
    for step in range(10):
        h, e, x, q, mask, y = batches[step % 3]
        lr, _, g = ot.loss_and_grads(h, e, x, q, mask, y, ot.unflatten(theta, w))
        theta = opt.step(theta, ot.flatten(g))
        losses_ref.append(lr)
        _, l = eng.train_step_dense(h, e, x, q, mask, y, apply=True)
        losses.append(l)
    wt = eng.get_weights()
    got = ot.flatten(wt)
    # Adam normalises the step, so float32 noise in tiny gradients can flip early steps by ~lr; compare loosely on
    # the parameters and tightly on the loss curve
    print("loss curve", np.round(losses, 5), "ref", np.round(losses_ref, 5))
    assert np.abs(np.array(losses) - np.array(losses_ref)).max() < 1e-3 * max(losses_ref)
    assert np.abs(got - theta).max() < 2.5e-3
    assert np.mean(np.abs(got - theta) < 2e-4) > 0.97
    # the model now predicts with the trained weights (inference path picks them up)
    h, e, x, q, mask, y = batches[0]
    pred = eng.model_forward_dense(h, e, x, q, mask)
    ref = __import__("oracle.epnn_oracle", fromlist=["x"]).model_forward(h, e, x, q, mask, wt)
    assert np.abs(pred - ref).max() < 2e-5


def test_graph_replay_equals_kernel_by_kernel(gpu_engine_factory):
    """The train step replays its launch sequence as a hipGraph ("train_graph", an option): losses, predictions and the
    weights after five Adam steps must be bit-identical to launching kernel by kernel."""
    nx, T, N = 9, 2, 8
    w = random_weights(nx, T, seed=3, scale=0.5)
    batches = [_tiny_batch(nx, N, [7, 5], seed=s) for s in range(2)]
    out = []
    for graph in (1, 0):
        eng = gpu_engine_factory(nx=nx, T=T)
        eng.set_option("train_graph", graph)
        eng.set_weights(w)
        eng.train_init()
        tr = []
        for step in range(5):
            h, e, x, q, mask, y = batches[step % 2]
            p, l = eng.train_step_dense(h, e, x, q, mask, y, apply=True)
            tr.append((p.copy(), l))
        out.append((tr, eng.get_weights()))
    (ta, wa), (tb, wb) = out
    for (pa, la), (pb, lb) in zip(ta, tb):
        assert la == lb and np.array_equal(pa, pb)
    from oracle import epnn_oracle_train as ot
    assert np.array_equal(ot.flatten(wa), ot.flatten(wb))


def test_graph_replay_keeps_the_captures_of_alternating_batch_sizes(gpu_engine_factory):
    """A training loop alternates between its batch size and the epoch's last, smaller batch, with and without the optimizer
    step (validation): the step's hipGraph is captured once per (B, N, buffers, apply) and the last few captures are kept, the
    device-side step counter follows the host's across them.  Losses, predictions and the weights after twelve steps of mixed
    kinds are bit-identical to launching kernel by kernel."""
    nx, T, N = 9, 2, 8
    w = random_weights(nx, T, seed=6, scale=0.5)
    kinds = [([7, 5], True), ([6], True), ([7, 5], False), ([3, 8, 4], True)]
    batches = [(_tiny_batch(nx, N, ns, seed=10 + k), ap) for k, (ns, ap) in enumerate(kinds)]
    order = [0, 1, 0, 2, 3, 1, 0, 3, 2, 0, 1, 3]
    out = []
    for graph in (1, 0):
        eng = gpu_engine_factory(nx=nx, T=T)
        eng.set_option("train_graph", graph)
        eng.set_weights(w)
        eng.train_init()
        tr = []
        for k in order:
            (h, e, x, q, mask, y), ap = batches[k]
            p, l = eng.train_step_dense(h, e, x, q, mask, y, apply=ap)
            tr.append((p.copy(), l))
        out.append((tr, eng.get_weights()))
    (ta, wa), (tb, wb) = out
    for (pa, la), (pb, lb) in zip(ta, tb):
        assert la == lb and np.array_equal(pa, pb)
    from oracle import epnn_oracle_train as ot
    assert np.array_equal(ot.flatten(wa), ot.flatten(wb))


def test_fused_step_equals_layer_by_layer_at_full_size(gpu_engine_factory, val_dir, val_names, weights_decay):
    """configs[2] shape (N = 41, T = 5, shipped checkpoint, real molecules): the row-fused kernels and the layer-by-layer
    kernels are two implementations of the same literal algorithm; predictions, loss and every gradient tensor agree
    to float32 rounding."""
    from conftest import load_molecules
    names = [nm for nm in val_names if nm.startswith("dsgdb9nsd")][:4]
    mols, offsets, xyz, x, Q = load_molecules(val_dir, names, 9)
    rng = np.random.default_rng(1)
    y = (rng.normal(size=int(offsets[-1])) * 0.2).astype(np.float32)
    res = []
    for fused in (1, 0):
        eng = gpu_engine_factory(nx=9, T=5)
        eng.set_option("train_fused", fused)
        eng.set_weights(weights_decay)
        eng.train_init()
        q, loss = eng.train_step_xyz(offsets, xyz, x, Q, y, 41, apply=False)
        res.append((q, loss, eng.get_gradients().astype(np.float64)))
    (qa, la, ga), (qb, lb, gb) = res
    assert np.abs(qa - qb).max() < 2e-6
    assert abs(la - lb) < 1e-5 * max(1.0, abs(lb))
    pos = 0
    worst = 0.0
    for m in [weights_decay["upd"]] + weights_decay["msg"] + weights_decay["pas"]:
        for W, b in m:
            for arr in (W, b):
                sl = slice(pos, pos + arr.size)
                scale = np.abs(gb[sl]).max()
                if scale > 0:
                    worst = max(worst, np.abs(ga[sl] - gb[sl]).max() / scale)
                else:
                    assert np.all(ga[sl] == 0)
                pos += arr.size
    print(f"fused vs layer-by-layer at N=41, T=5: worst per-tensor relative gradient difference {worst:.2e}")
    assert worst < 1e-4


@pytest.mark.parametrize("B", [1, 3, 8])
def test_workgroups_per_atom_do_not_change_a_bit(gpu_engine_factory, val_dir, val_names, B):
    """The matrix-pipe backward hands an atom's weight-gradient jobs to 1..6 workgroups ("train_split"; automatic: as many as
    fit the 32 CUs of the XCD the atom's workgroups share -- 5 for one molecule at N = 41, 2 for three, 1 for eight).  A job is the same arithmetic whichever workgroup
    runs it and everything else is computed by all of them from the previous launch's copies: predictions, loss, gradients
    and the weights after two optimizer steps are bit-identical for every split, random weights (every tensor gets a gradient)."""
    from conftest import load_molecules
    names = [nm for nm in val_names if nm.startswith("dsgdb9nsd")][:B]
    mols, offsets, xyz, x, Q = load_molecules(val_dir, names, 9)
    rng = np.random.default_rng(B)
    y = (rng.normal(size=int(offsets[-1])) * 0.2).astype(np.float32)
    w = random_weights(9, 5, seed=21, scale=0.4)
    ref = None
    for split in (0, 1, 2, 5):
        eng = gpu_engine_factory(nx=9, T=5)
        eng.set_option("train_split", split)
        eng.set_weights(w)
        eng.train_init()
        q, loss = eng.train_step_xyz(offsets, xyz, x, Q, y, 41, apply=False)
        g = eng.get_gradients()
        assert np.abs(g).max() > 0
        q1, l1 = eng.train_step_xyz(offsets, xyz, x, Q, y, 41)
        q2, l2 = eng.train_step_xyz(offsets, xyz, x, Q, y, 41)
        out = (q, np.float32(loss), g, q2, np.float32(l2))
        if ref is None:
            ref = out
        else:
            for a, b_ in zip(ref, out):
                assert np.array_equal(a, b_), split


@pytest.mark.parametrize("B", [1, 3, 8])
def test_skipping_padded_slots_does_not_change_a_bit(gpu_engine_factory, val_dir, val_names, B):
    """train_step_xyz pads every molecule to N slots with exact zeros; the matrix-pipe kernels leave the workgroups of those
    slots at once ("train_skip_padded", default), read the padded partners' h and q as the zeros they are and skip their
    weight-gradient partials in the reduction.  A padded atom's partials are exact zeros when they ARE computed, so
    predictions, loss, gradients and the weights after two optimizer steps are bit-identical either way -- molecules of 9..29
    atoms at N = 41, random weights."""
    from conftest import load_molecules
    names = [nm for nm in val_names if nm.startswith("dsgdb9nsd")][3:3 + B]
    mols, offsets, xyz, x, Q = load_molecules(val_dir, names, 9)
    assert max(np.diff(offsets)) < 41
    rng = np.random.default_rng(40 + B)
    y = (rng.normal(size=int(offsets[-1])) * 0.2).astype(np.float32)
    w = random_weights(9, 5, seed=23, scale=0.4)
    ref = None
    for skip in (1, 0):
        eng = gpu_engine_factory(nx=9, T=5)
        eng.set_option("train_skip_padded", skip)
        eng.set_weights(w)
        eng.train_init()
        q, loss = eng.train_step_xyz(offsets, xyz, x, Q, y, 41, apply=False)
        g = eng.get_gradients()
        assert np.abs(g).max() > 0
        eng.train_step_xyz(offsets, xyz, x, Q, y, 41)
        q2, l2 = eng.train_step_xyz(offsets, xyz, x, Q, y, 41)
        out = (q, np.float32(loss), g, q2, np.float32(l2), eng.get_weights_flat() if hasattr(eng, "get_weights_flat") else q2)
        if ref is None:
            ref = out
        else:
            for a, b_ in zip(ref, out):
                assert np.array_equal(a, b_)


def test_a_step_that_returns_behind_its_forward_pass_is_the_same_step(gpu_engine_factory, val_dir, val_names):
    """train_step_xyz returns when its forward pass is done ("train_async", default): loss and predictions are on the host
    then, the backward pass and the optimizer step run on and everything that looks at gradients or weights, changes the
    batch shape or runs inference on the handle waits for them first.  A sequence that mixes all of that -- steps on two
    batch shapes, a gradient read-out without an optimizer step, a forward between two steps, the weights at the end --
    returns the same bits as with every step waiting for its own end."""
    from conftest import load_molecules
    qm9 = [nm for nm in val_names if nm.startswith("dsgdb9nsd")]
    b1 = load_molecules(val_dir, qm9[:3], 9)
    b2 = load_molecules(val_dir, qm9[3:4], 9)
    rng = np.random.default_rng(4)
    ys = [(rng.normal(size=int(b[1][-1])) * 0.2).astype(np.float32) for b in (b1, b2)]
    w = random_weights(9, 5, seed=29, scale=0.4)
    outs = []
    for asyn in (1, 0):
        eng = gpu_engine_factory(nx=9, T=5)
        eng.set_option("train_async", asyn)
        eng.set_weights(w)
        eng.train_init()
        log = []

        def step(b, y, N, apply=True):
            mols, offsets, xyz, x, Q = b
            q, loss = eng.train_step_xyz(offsets, xyz, x, Q, y, N, apply=apply)
            log.append((q.copy(), np.float32(loss)))
        for _ in range(3):
            step(b1, ys[0], 41)
        step(b2, ys[1], 30)                               # another batch shape: buffers are re-made
        step(b1, ys[0], 41, apply=False)
        log.append((eng.get_gradients(),))
        step(b1, ys[0], 41)
        mols, offsets, xyz, x, Q = b2
        log.append((eng.forward_xyz(offsets, xyz, x, Q, 30),))      # inference with the weights trained so far
        for _ in range(2):
            step(b2, ys[1], 30)
        log.append((np.concatenate([np.ravel(a) for m in ([eng.get_weights()["upd"]] + eng.get_weights()["msg"] + eng.get_weights()["pas"]) for W, b in m for a in (W, b)]),))
        outs.append(log)
    for a, b_ in zip(*outs):
        for u, v in zip(a, b_):
            assert np.array_equal(u, v)


def test_train_step_xyz_equals_dense(gpu_engine_factory, val_dir, val_names):
    from conftest import load_molecules
    from oracle import epnn_oracle as orc
    nx, T, N = 9, 2, 24
    w = random_weights(nx, T, seed=1, scale=0.5)
    names = [nm for nm in val_names if nm.startswith("dsgdb9nsd")][:3]
    mols, offsets, xyz, x, Q = load_molecules(val_dir, names, nx)
    rng = np.random.default_rng(0)
    y = (rng.normal(size=int(offsets[-1])) * 0.2).astype(np.float32)
    eng = gpu_engine_factory(nx=nx, T=T)
    eng.set_weights(w)
    eng.train_init()
    qa, la = eng.train_step_xyz(offsets, xyz, x, Q, y, N, apply=False)
    ga = eng.get_gradients()
    dense = [orc.dense_inputs(m[0], m[1], m[2], N) for m in mols]
    h, e, xd, q, mask = (np.stack([d[k] for d in dense]) for k in range(5))
    yd = np.zeros((len(mols), N, 1), np.float32)
    for b in range(len(mols)):
        yd[b, :offsets[b + 1] - offsets[b], 0] = y[offsets[b]:offsets[b + 1]]
    pb, lb = eng.train_step_dense(h, e, xd, q, mask, yd, apply=False)
    gb = eng.get_gradients()
    assert abs(la - lb) < 1e-5 * max(1, abs(lb))
    assert np.abs(ga - gb).max() <= 1e-5 * max(1e-6, np.abs(gb).max())
    for b in range(len(mols)):
        n = offsets[b + 1] - offsets[b]
        assert np.abs(qa[offsets[b]:offsets[b + 1]] - pb[b, :n, 0]).max() < 1e-6


def test_rccl_single_rank_allreduce(gpu_engine_factory):
    """World-size-1 communicator: checks linkage and that the all-reduce leaves the gradient unchanged."""
    from epnn_amd.engine import Engine
    nx, T, N = 9, 1, 6
    w = random_weights(nx, T, seed=5, scale=0.5)
    h, e, x, q, mask, y = _tiny_batch(nx, N, [4], seed=1)
    eng = gpu_engine_factory(nx=nx, T=T)
    eng.set_weights(w)
    eng.train_init()
    eng.comm_init(Engine.comm_unique_id(), 0, 1)
    eng.train_step_dense(h, e, x, q, mask, y, apply=False)
    g0 = eng.get_gradients()
    eng.train_apply()
    assert np.array_equal(eng.get_gradients(), g0)
    assert not np.array_equal(eng.get_weights()["pas"][0][0][0], w["pas"][0][0][0])


def test_collectives_fail_closed_world1(gpu_engine_factory):
    """The status guard in front of the payload collectives (comm_guard, csrc/epnn_host.h): with a world-size-1 communicator and the
    developer switch that runs the collectives anyway, (a) a train step with apply=1 goes guard -> gradient all-reduce -> Adam and
    gives the bits of the plain step; (b) an injected failure on "a rank" makes the call return non-zero with the guard's message
    BEFORE the payload collective is enqueued, leaves the weights untouched, and the handle works afterwards; (c) a step that fails
    on its way to the collective (molecule larger than N) still runs its status collective from the exit path; (d) the same three for
    the row exchange of a partitioned system."""
    from epnn_amd import synth
    from epnn_amd._lib import EpnnError
    from epnn_amd.engine import Engine
    nx, T, N = 9, 2, 12
    w = random_weights(nx, T, seed=8, scale=0.5)
    mols_n = [7, 11]
    rng = np.random.default_rng(3)
    off = np.array([0, 7, 18], np.int32)
    xyz = rng.uniform(0, 4.0, (18, 3)).astype(np.float32)
    x = synth.features(rng.choice(["H", "C", "N", "O"], size=18)).astype(np.float32)
    Q = np.zeros(2, np.float32)
    y = rng.normal(0, 0.2, 18).astype(np.float32)
    plain = gpu_engine_factory(nx=nx, T=T)
    plain.set_weights(w)
    plain.train_init()
    q0, l0 = plain.train_step_xyz(off, xyz, x, Q, y, N, apply=True)
    w_plain = plain.get_weights()
    eng = gpu_engine_factory(nx=nx, T=T)
    eng.set_weights(w)
    eng.train_init()
    eng.comm_init(Engine.comm_unique_id(), 0, 1)
    eng.set_option("part_collective", 1)
    # (b) first: the failing step must leave everything as it was
    eng.set_option("comm_inject_fail", 1)
    with pytest.raises(EpnnError, match="aborted on every rank"):
        eng.train_step_xyz(off, xyz, x, Q, y, N, apply=True)
    assert np.array_equal(eng.get_weights()["pas"][0][0][0], w["pas"][0][0][0])
    # (c) a failure in front of the collective: the exit path reports it through the status collective
    with pytest.raises(EpnnError, match="aborted on every rank; this rank failed: .*does not fit N"):
        eng.train_step_xyz(off, xyz, x, Q, y, 8, apply=True)
    # (a)
    q1, l1 = eng.train_step_xyz(off, xyz, x, Q, y, N, apply=True)
    assert np.array_equal(q1, q0) and l1 == l0
    w_coll = eng.get_weights()
    for a, b in zip(w_coll["pas"][0] + w_coll["msg"][1] + w_coll["upd"], w_plain["pas"][0] + w_plain["msg"][1] + w_plain["upd"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # (d) partitioned forward, rows over the communicator
    offs, bxyz, bx, bQ, bN = synth.box_system(n_atoms=300, seed=4)
    fwd = gpu_engine_factory(nx=nx, T=T)
    fwd.set_weights(w)
    whole = fwd.forward_xyz(offs, bxyz, bx, bQ, bN)
    fwd.comm_init(Engine.comm_unique_id(), 0, 1)
    fwd.set_option("part_collective", 1)
    fwd.set_option("comm_inject_fail", 1)
    with pytest.raises(EpnnError, match="row exchange.*aborted on every rank"):
        fwd.forward_xyz(offs, bxyz, bx, bQ, bN)
    assert np.array_equal(fwd.forward_xyz(offs, bxyz, bx, bQ, bN), whole)
    fwd.set_option("comm_guard", 0)
    assert np.array_equal(fwd.forward_xyz(offs, bxyz, bx, bQ, bN), whole)


def test_train_script_and_mirror(tmp_path, golden_dir):
    """train.py end to end on the four-file fixture directory (two epochs), and the charge_gn.train_step mirror."""
    import subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), os.path.join(golden_dir, "qm9_small"), "--epochs", "2",
                          "--n-elems", "9", "--init", os.path.join(ROOT, "models", "decay_model_weights"), "--out", str(tmp_path / "w")],
                         cwd=tmp_path, capture_output=True, text=True, timeout=600)          # arrays go to the cwd like the reference's
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("Epoch")]
    assert len(lines) == 2 and "Test Acc" in lines[0]
    assert os.path.exists(tmp_path / "w.index") and os.path.exists(tmp_path / "val_names.npy")
    from epnn_amd import charge_gn, checkpoint
    w = checkpoint.load_epnn_weights(str(tmp_path / "w"))            # the file train.py wrote is a readable bundle
    assert len(w["msg"]) == 5
    x, h, q, e, Q, y, mask, names = charge_gn.gen_padded_init_state(os.path.join(golden_dir, "qm9_small") + "/", 48, 48, n_elems=9)
    model = charge_gn.make_model([32, 32], 48, 5, 9, x.shape[1])
    model.load_weights(os.path.join(ROOT, "models", "decay_model_weights"))
    opt = charge_gn.Adam()
    before = model([h[:1], e[:1], x[:1], q[:1], mask[:1]])
    loss, acc = charge_gn.Mean(), charge_gn.MeanAbsoluteError()
    for _ in range(3):
        pred = charge_gn.train_step(model, opt, h[:1], e[:1], x[:1], q[:1], y[:1], mask[:1], loss, acc)
    assert pred.shape == before.shape and np.isfinite(loss.result())
    after = model([h[:1], e[:1], x[:1], q[:1], mask[:1]])
    assert np.abs(after - before).max() > 0                          # weights moved, inference sees them
    tv = model.trainable_variables
    assert len(tv) == 66 and sum(v.size for v in tv) == 74037


@pytest.mark.parametrize("N,ns", [(50, [50, 33]), (96, [96]), (97, [60])])
def test_fused_step_at_larger_padded_sizes(gpu_engine_factory, N, ns):
    """Padded sizes beyond one pass of the fused kernels' row loops (48 rows per pass), their LDS limit N = 96, and the
    fallback to the layer-by-layer kernels above it: both settings of "train_fused" give the same loss and gradients."""
    nx, T = 9, 2
    w = random_weights(nx, T, seed=11, scale=0.4)
    h, e, x, q, mask, y = _tiny_batch(nx, N, ns, seed=4)
    res = []
    for fused in (1, 0):
        eng = gpu_engine_factory(nx=nx, T=T)
        eng.set_option("train_fused", fused)
        eng.set_weights(w)
        eng.train_init()
        pred, loss = eng.train_step_dense(h, e, x, q, mask, y, apply=False)
        res.append((pred, loss, eng.get_gradients().astype(np.float64)))
    (pa, la, ga), (pb, lb, gb) = res
    assert np.abs(pa - pb).max() < 5e-6
    assert abs(la - lb) < 1e-5 * max(1.0, abs(lb))
    pos, worst = 0, 0.0
    for m in [w["upd"]] + w["msg"] + w["pas"]:
        for W, b in m:
            for arr in (W, b):
                sl = slice(pos, pos + arr.size)
                scale = np.abs(gb[sl]).max()
                if scale > 0:
                    worst = max(worst, np.abs(ga[sl] - gb[sl]).max() / scale)
                pos += arr.size
    print(f"N={N}: fused vs layer-by-layer worst per-tensor relative gradient difference {worst:.2e}")
    assert worst < 1e-4


def _merged_split_dir(tmp_path, train_dir, val_dir):
    """One directory with the training and the validation files, like the reference's data/mixed/ (symlinks)."""
    d = tmp_path / "mixed"
    d.mkdir()
    for src in (train_dir, val_dir):
        for f in os.listdir(src):
            os.symlink(os.path.join(src, f), d / f)
    return str(d)


@pytest.mark.parametrize("world,per", [(1, 1), (2, 1), (1, 3), (2, 2)])
def test_train_script_on_the_recorded_split(tmp_path, golden_dir, train_dir, val_dir, train_names, world, per):
    """train.py --names on the reference's recorded split (BASELINE.json configs[2]; models/model_systems/train_names.npy,
    val_names.npy), N = 41: six optimizer steps from the shipped checkpoint, at world size 1 and with two rank processes
    started by `--gpus 2` (they share this box's one GPU, so the gradient sum goes through the host; one GPU per rank
    uses the RCCL all-reduce).  Row k of train_pred_charges / train_lab_charges must belong to training molecule k of
    the recorded order -- whichever rank computed it, and whichever of the `per` molecules of that rank's step it was
    (`--molecules-per-rank`, the form of data parallelism that scales: DESIGN.md section 6): the labels are that molecule's, the
    prediction has exactly its atoms and sums to its total charge.  The validation arrays cover all 871 systems in the recorded
    order."""
    import subprocess, sys
    from conftest import ROOT
    from epnn_amd import charge_gn
    data = _merged_split_dir(tmp_path, train_dir, val_dir)
    steps = 6
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), data, "--epochs", "1", "--n-elems", "9", "--init",
           os.path.join(ROOT, "models", "decay_model_weights"), "--out", str(tmp_path / "ck" / "w"), "--outdir", str(tmp_path / "arr"),
           "--names", os.path.join(golden_dir, "train_names.npy"), os.path.join(golden_dir, "val_names.npy"), "--max-steps", str(steps)]
    if world > 1:
        cmd += ["--gpus", str(world)]
    if per > 1:
        cmd += ["--molecules-per-rank", str(per)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert len([l for l in res.stdout.splitlines() if l.startswith("Epoch 0, Loss:")]) == 1
    assert len([l for l in res.stdout.splitlines() if "molecules/s" in l and f"{world} rank(s) x {per} molecule(s)" in l]) == 1
    arr = tmp_path / "arr"
    assert [str(n) for n in np.load(arr / "train_names.npy", allow_pickle=True)] == train_names
    pred, lab = np.load(arr / "train_pred_charges.npy"), np.load(arr / "train_lab_charges.npy")
    assert pred.shape == lab.shape == (steps * world * per, 41)
    for k in range(steps * world * per):
        xyz, x, Q, _ = charge_gn.read_xyz(os.path.join(train_dir, train_names[k] + ".xyz"), 9)
        y = np.load(os.path.join(train_dir, train_names[k] + ".npy")).ravel()
        n = len(y)
        assert np.array_equal(lab[k, :n], y.astype(np.float32)) and np.all(lab[k, n:] == 0)
        assert np.all(pred[k, :n] != 0) and np.all(pred[k, n:] == 0)
        assert abs(float(pred[k].sum(dtype=np.float64)) - float(Q)) < 1e-4
        assert np.abs(pred[k, :n] - y).max() < 0.5                   # a trained model: predictions are near the labels
    vp, vl = np.load(arr / "test_pred_charges.npy"), np.load(arr / "test_lab_charges.npy")
    gold_lab = np.load(os.path.join(golden_dir, "test_lab_charges.npy"))
    assert vp.shape == vl.shape == (871, 41)
    assert np.abs(vl - gold_lab).max() < 1e-6                        # the reference's own stored validation labels, same order
    gold_pred = np.load(os.path.join(golden_dir, "test_pred_charges.npy"))
    # six Adam steps (lr 1e-3 on every parameter) away from the checkpoint that wrote the stored predictions
    assert np.abs(vp - gold_pred).max() < 0.3 and np.abs(vp - gold_pred).mean() < 0.01
    from epnn_amd import checkpoint
    assert len(checkpoint.load_epnn_weights(str(tmp_path / "ck" / "w"))["msg"]) == 5


def test_loss_curve_on_the_recorded_split_matches_oracle(gpu_engine_factory, train_dir, train_names, weights_decay):
    """configs[2] numerics over a trajectory: 50 optimizer steps (one molecule per step, charge_gn.py:443-451) from
    decay_model_weights on the first 50 molecules of the recorded training split, N = 41 -- the per-step losses follow
    the float64 oracle's (its own Adam on its own gradients)."""
    from conftest import load_molecules
    from oracle import epnn_oracle as orc
    from oracle import epnn_oracle_train as ot
    nx, N, steps = 9, 41, 50
    mols, offsets, xyz, x, Q = load_molecules(train_dir, train_names[:steps], nx)
    ys = [np.load(os.path.join(train_dir, nm + ".npy")).ravel().astype(np.float32) for nm in train_names[:steps]]
    eng = gpu_engine_factory(nx=nx, T=5)
    eng.set_weights(weights_decay)
    eng.train_init()
    theta = ot.flatten(weights_decay)
    opt = ot.Adam(theta.size)
    losses, ref = [], []
    for s in range(steps):
        xyz_s, x_s, Q_s = mols[s]
        n = x_s.shape[0]
        _, l = eng.train_step_xyz(np.array([0, n], np.int32), xyz_s, x_s, np.array([Q_s], np.float32), ys[s], N, apply=True)
        losses.append(l)
        h, e, xd, q, mask = (a[None] for a in orc.dense_inputs(xyz_s, x_s, Q_s, N))
        yd = np.zeros((1, N, 1))
        yd[0, :n, 0] = ys[s]
        lr, _, g = ot.loss_and_grads(h, e, xd, q, mask, yd, ot.unflatten(theta, weights_decay))
        theta = opt.step(theta, ot.flatten(g))
        ref.append(lr)
    losses, ref = np.array(losses), np.array(ref)
    rel = np.abs(losses - ref) / np.maximum(ref, 1e-6)
    print(f"50 steps on the recorded split: loss {ref[0]:.5f} .. {ref[-1]:.5f}; worst relative loss difference {rel.max():.2e} "
          f"(step {int(rel.argmax())}); sum of losses {losses.sum():.5f} vs {ref.sum():.5f}")
    assert rel[0] < 1e-4                                             # step 0: same weights, forward parity
    assert rel.max() < 2e-2 and abs(losses.sum() - ref.sum()) < 2e-3 * ref.sum()


_RCCL_DP_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
from epnn_amd.engine import Engine
from epnn_amd.rendezvous import Rendezvous
from conftest import random_weights
from test_train_oracle import _tiny_batch
from oracle import epnn_oracle_train as ot
r = Rendezvous()
nx, T, N = 9, 2, 10
w = random_weights(nx, T, seed=4, scale=0.5)
h, e, x, q, mask, y = _tiny_batch(nx, N, [9, 7], seed=3)          # molecule `rank` is this rank's share of the step
eng = Engine(nx=nx, T=T, device=r.rank)
eng.set_weights(w); eng.train_init()
eng.comm_init(r.broadcast(Engine.comm_unique_id() if r.rank == 0 else None, name="id"), r.rank, r.world)
k = r.rank
eng.train_step_dense(h[k:k+1], e[k:k+1], x[k:k+1], q[k:k+1], mask[k:k+1], y[k:k+1], apply=True)   # all-reduce + Adam on the device
g = eng.get_gradients().astype(np.float64)                        # the summed gradient (the all-reduce is in place)
ref = ot.flatten(ot.loss_and_grads(h, e, x, q, mask, y, w)[2])    # oracle gradient of BOTH molecules = sum of the ranks' gradients
assert np.abs(g - ref).max() <= 2e-4 * np.abs(ref).max(), float(np.abs(g - ref).max() / np.abs(ref).max())
wts = r.all_gather(ot.flatten(eng.get_weights()).tobytes(), name="w")
assert all(b == wts[0] for b in wts)                              # every rank took the same Adam step
r.barrier(); r.close(); eng.close()
if r.rank == 0:
    print("RCCL_DP_OK", r.world)
'''


def test_rccl_gradient_allreduce_two_gpus(tmp_path):
    """Data-parallel train step on two GPUs (BASELINE.json configs[2] at world size 2): every rank runs forward + backward
    of its own molecule, ONE ncclAllReduce of the flat gradient on the device, the same Adam step everywhere.  The
    all-reduced gradient equals the float64 oracle's gradient of the two-molecule batch.  Needs two devices; skipped on a
    one-GPU box (RCCL cannot join two ranks of one device; train.py's host-staged sum covers the rest of the path there)."""
    import subprocess, sys
    from conftest import ROOT
    from epnn_amd import _lib
    if _lib.load().epnn_device_count() < 2:
        pytest.skip("one GPU visible: a multi-rank RCCL communicator needs one device per rank")
    script = tmp_path / "w.py"
    script.write_text(_RCCL_DP_WORKER)
    drv = ("import sys; sys.path.insert(0, sys.argv[1]); from epnn_amd.rendezvous import launch_ranks; "
           "sys.exit(launch_ranks(sys.argv[2], sys.argv[1:2], 2))")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", drv, ROOT, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_DP_OK 2" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
