"""Training entry point, same flow as the reference's ``charge_gn.py`` ``__main__`` block (charge_gn.py:412-471):
read a directory of .xyz (+ .npy labels), 80/20 split with random_state=42 (or the reference's RECORDED split, ``--names``),
EPOCHS passes of one optimizer step per training molecule, validation pass, save_weights on the best validation MAE,
dump names / predictions / labels, print the reference's epoch line.  Arithmetic on the MI355X through ``epnn_amd``
(no TensorFlow, no torch).

Data parallel (BASELINE.json configs[2]): ``python train.py DIR --gpus 8`` starts one rank process per GPU itself (fresh
children, before this process makes any GPU call; ``python -m torch.distributed.run --nproc-per-node 8 train.py DIR`` works
as well).  Each optimizer step then takes WORLD_SIZE consecutive training molecules, one per rank; the gradients are
summed with ONE RCCL all-reduce of the flat 296 KB gradient on the GPU (the 128-byte RCCL id travels over
``epnn_amd.rendezvous``), then every rank takes the same Adam step.  The reference is single-device batch 1, so
trajectories are comparable only at world size 1.  With fewer GPUs than ranks (rehearsal on a one-GPU box) the ranks
share devices, RCCL cannot join two ranks of one device, and the gradient sum goes through the host instead.

One molecule per GPU and step is LATENCY-bound by construction (a step is ~21 dependent launches of ~8 us, DESIGN.md
section 6): eight GPUs at one molecule each take an optimizer step in about the time one GPU does, i.e. at best ~1.15x the
molecules/s of one GPU that puts eight molecules into its step.  ``--molecules-per-rank B`` is the form that scales: every rank
sums the gradients of B molecules in its step (the arithmetic of tests/test_shard_synth.py::
test_data_parallel_gradient_sum_world2_gloo), the all-reduce stays one per step.  Rank 0 prints the molecules/s of every epoch's
training pass, to be compared with the expectation in DESIGN.md.

    python train.py xyz_dir/ [--epochs E] [--n-elems 10] [--init PREFIX] [--out models/model_weights] [--limit M]
                             [--names train_names.npy val_names.npy] [--gpus N] [--max-steps S] [--outdir DIR]
                             [--molecules-per-rank B]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--epochs", type=int, default=500)            # charge_gn.py:424
    ap.add_argument("--n-elems", type=int, default=10)            # charge_gn.py:418
    ap.add_argument("--init", default=None, help="checkpoint prefix to start from (default: Glorot init)")
    ap.add_argument("--out", default="models/model_weights")      # charge_gn.py:462
    ap.add_argument("--limit", type=int, default=0, help="use only the first M molecules (smoke runs)")
    ap.add_argument("--names", nargs=2, metavar=("TRAIN_NPY", "VAL_NPY"), default=None,
                    help="the recorded split (models/model_systems/train_names.npy val_names.npy, charge_gn.py:433-434) "
                         "instead of a fresh train_test_split; only the named molecules are read")
    ap.add_argument("--gpus", type=int, default=0, help="start this many rank processes (one per GPU)")
    ap.add_argument("--max-steps", type=int, default=0, help="optimizer steps per epoch (0: the whole training split)")
    ap.add_argument("--outdir", default=".", help="where the names / predictions / labels arrays go (the reference: cwd)")
    ap.add_argument("--molecules-per-rank", type=int, default=1,
                    help="molecules whose gradients a rank sums in one optimizer step (1: the reference's batch of one per device)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from epnn_amd.rendezvous import launch_ranks
        sys.exit(launch_ranks(__file__, sys.argv[1:] if argv is None else argv, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    from epnn_amd import _lib, charge_gn, shard
    from epnn_amd.engine import Engine

    rdzv = None
    if world > 1:
        from epnn_amd.rendezvous import Rendezvous
        rdzv = Rendezvous(rank, world)
        ndev = _lib.load().epnn_device_count()
        if ndev < 1:
            raise SystemExit("train.py: no HIP device visible")
        # one GPU per rank; ranks beyond the device count share (charge_gn reads LOCAL_RANK when the module is imported,
        # so the device is fixed here before the first engine exists)
        charge_gn._DEVICE = int(os.environ.get("LOCAL_RANK", str(rank))) % ndev
        rccl = ndev >= world

    h_dim, e_dim, layers, T = 48, 48, [32, 32], 5                  # charge_gn.py:413-417
    path = args.path if args.path.endswith("/") else args.path + "/"
    if args.names:
        tr = [str(n) for n in np.load(args.names[0], allow_pickle=True)]
        va = [str(n) for n in np.load(args.names[1], allow_pickle=True)]
        if args.limit:
            tr, va = tr[:args.limit], va[:max(1, args.limit // 4)]
        files = [n + ".xyz" for n in tr + va]
    else:
        files = sorted(f for f in os.listdir(path) if f.endswith(".xyz"))
        if args.limit:
            files = files[:args.limit]
    mols, labels, names = [], [], []
    for f in files:
        xyz, x, Q, nlines = charge_gn.read_xyz(path + f, args.n_elems)
        lab = path + f[:-4] + ".npy"
        y = np.array(np.load(lab), dtype=np.float32).ravel() if os.path.exists(lab) else np.zeros(nlines - 2, np.float32)
        mols.append((xyz, x, Q))
        labels.append(y)
        names.append(f[:-4])
    N = max(len(y) for y in labels)                                 # the directory maximum (charge_gn.py:340)
    if args.names:
        it, ie = np.arange(len(tr)), np.arange(len(tr), len(tr) + len(va))
    else:
        from sklearn.model_selection import train_test_split
        it, ie = train_test_split(np.arange(len(mols)), test_size=0.2, random_state=42)     # charge_gn.py:431
    os.makedirs(args.outdir, exist_ok=True)
    out = lambda f: os.path.join(args.outdir, f)
    if rank == 0:
        np.save(out("train_names.npy"), np.array([names[i] for i in it]), allow_pickle=True)    # charge_gn.py:433-434
        np.save(out("val_names.npy"), np.array([names[i] for i in ie]), allow_pickle=True)

    model = charge_gn.make_model(layers, h_dim, T, args.n_elems, N)
    if args.init:
        model.load_weights(args.init)
    elif world > 1:
        model.set_weights_dict(rdzv.broadcast(model.weights_dict() if rank == 0 else None, name="init"))   # same start everywhere
    eng = model.engine()
    opt = charge_gn.Adam()
    opt.bind(model)
    if world > 1 and rccl:
        eng.comm_init(rdzv.broadcast(Engine.comm_unique_id() if rank == 0 else None, name="rccl_id"), rank, world)

    def padded(v):
        p = np.zeros(N, np.float32)
        p[:len(v)] = v
        return p

    def batch_of(idx):
        """flat batch (offsets, xyz, x, Q) of the molecules `idx`"""
        off = np.zeros(len(idx) + 1, np.int32)
        off[1:] = np.cumsum([len(mols[i][1]) for i in idx])
        return (off, np.concatenate([mols[i][0] for i in idx]), np.concatenate([mols[i][1] for i in idx]),
                np.array([mols[i][2] for i in idx], np.float32))

    train_loss, train_acc = charge_gn.Mean("train_loss"), charge_gn.MeanAbsoluteError("train_acc")
    test_loss, test_acc = charge_gn.Mean("test_loss"), charge_gn.MeanAbsoluteError("test_acc")
    best = np.inf
    per = max(1, args.molecules_per_rank)
    nsteps = len(it) // (world * per)
    if args.max_steps:
        nsteps = min(nsteps, args.max_steps)
    my_val = [int(i) for i in ie[rank::world]]                      # validation molecules are independent: sharded
    for epoch in range(args.epochs):
        for m in (train_loss, train_acc, test_loss, test_acc):
            m.reset_states()
        train_rows = []                                             # (position in `it`, prediction padded to N) of this rank
        t_epoch = time.perf_counter()
        for s in range(nsteps):
            mine = shard.dp_step_molecules(it, world, s, per)[rank]
            mine = [mine] if per == 1 else mine
            off, xyz, x, Q = batch_of(mine)
            y = np.concatenate([labels[i] for i in mine])
            if world > 1 and not rccl:
                q, _ = eng.train_step_xyz(off, xyz, x, Q, y, N, apply=False)
                g = rdzv.all_gather(eng.get_gradients(), name="grad")
                eng.set_gradients(np.sum(np.stack(g).astype(np.float64), axis=0).astype(np.float32))      # rank order: same bits everywhere
                eng.train_apply()
            else:
                q, _ = eng.train_step_xyz(off, xyz, x, Q, y, N, apply=True)               # all-reduce (RCCL) + Adam on the device
            for k, i in enumerate(mine):
                qk = padded(q[off[k]:off[k + 1]])
                train_rows.append(((s * world + rank) * per + k, qk))
                train_loss((padded(labels[i]) - qk) ** 2)                                  # charge_gn.py:400-401
                train_acc(qk, padded(labels[i]))
        eng.sync()
        t_epoch = time.perf_counter() - t_epoch
        val_rows = []
        if my_val:
            off, xyz, x, Q = batch_of(my_val)
            qv = eng.forward_xyz(off, xyz, x, Q, N)                 # the validation pass (charge_gn.py:453-459) as ONE batch
            for k, i in enumerate(my_val):
                p = padded(qv[off[k]:off[k + 1]])
                val_rows.append((rank + k * world, p))
                test_loss((padded(labels[i]) - p) ** 2)
                test_acc(p, padded(labels[i]))
        if world > 1:
            parts = rdzv.all_gather((train_rows, val_rows, [(m._sum, m._n) for m in (train_loss, train_acc, test_loss, test_acc)]), name="epoch")
            train_rows = sorted((r for p in parts for r in p[0]), key=lambda r: r[0])
            val_rows = sorted((r for p in parts for r in p[1]), key=lambda r: r[0])
            for k, m in enumerate((train_loss, train_acc, test_loss, test_acc)):
                m._sum, m._n = sum(p[2][k][0] for p in parts), sum(p[2][k][1] for p in parts)
        if test_acc.result() < best:                                # every rank holds the same metrics: same decision
            best = test_acc.result()
            if rank == 0:
                model.save_weights(args.out)                                                            # charge_gn.py:462
                np.save(out("train_pred_charges.npy"), np.array([p for _, p in train_rows]))             # charge_gn.py:465-468
                np.save(out("train_lab_charges.npy"), np.array([padded(labels[it[pos]]) for pos, _ in train_rows]))
                np.save(out("test_pred_charges.npy"), np.array([p for _, p in val_rows]))
                np.save(out("test_lab_charges.npy"), np.array([padded(labels[ie[pos]]) for pos, _ in val_rows]))
        if rank == 0:
            template = 'Epoch {}, Loss: {}, Acc: {}, Test Loss: {}, Test Acc: {}'
            print(template.format(epoch, train_loss.result(), train_acc.result(), test_loss.result(), test_acc.result()), flush=True)
            print(f"  training pass: {nsteps} optimizer steps of {world} rank(s) x {per} molecule(s) in {t_epoch:.3f} s = "
                  f"{nsteps * world * per / t_epoch:.0f} molecules/s ({t_epoch / max(1, nsteps) * 1e3:.3f} ms per step)", flush=True)
    if rdzv is not None:
        rdzv.barrier("end")
        rdzv.close()


if __name__ == "__main__":
    main()
