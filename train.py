"""Training entry point, same flow as the reference's ``charge_gn.py`` ``__main__`` block (charge_gn.py:412-471):
read a directory of .xyz (+ .npy labels), 80/20 split with random_state=42, EPOCHS passes of one optimizer step per
training molecule, validation pass, save_weights on the best validation MAE, dump names / predictions / labels, print
the reference's epoch line.  Arithmetic on the MI355X through ``epnn_amd`` (no TensorFlow).

Data parallel (BASELINE.json configs[2]): launch with
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py data/mixed/ ...
Each optimizer step then takes WORLD_SIZE consecutive training molecules, one per GPU; the gradients are summed with
one RCCL all-reduce of the flat 296 KB gradient (the reference is single-device batch 1, so trajectories are only
comparable at world size 1).

    python train.py xyz_dir/ [--epochs E] [--n-elems 10] [--init PREFIX] [--out models/model_weights] [--limit M]
"""
import argparse
import os
import sys

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
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)     # only carries the 128-byte RCCL id

    from sklearn.model_selection import train_test_split
    from epnn_amd import charge_gn, shard
    from epnn_amd.engine import Engine

    h_dim, e_dim, layers, T = 48, 48, [32, 32], 5                  # charge_gn.py:413-417
    path = args.path if args.path.endswith("/") else args.path + "/"
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
    idx = np.arange(len(mols))
    it, ie = train_test_split(idx, test_size=0.2, random_state=42)  # charge_gn.py:431
    if rank == 0:
        np.save("train_names.npy", np.array([names[i] for i in it]), allow_pickle=True)
        np.save("val_names.npy", np.array([names[i] for i in ie]), allow_pickle=True)

    model = charge_gn.make_model(layers, h_dim, T, args.n_elems, N)
    if args.init:
        model.load_weights(args.init)
    eng = model.engine()
    opt = charge_gn.Adam()
    opt.bind(model)
    if world > 1:
        ids = [Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        eng.comm_init(ids[0], rank, world)

    def one(i):
        xyz, x, Q = mols[i]
        return (np.array([0, len(x)], np.int32), xyz, x, np.array([Q], np.float32), labels[i])

    train_loss, train_acc = charge_gn.Mean("train_loss"), charge_gn.MeanAbsoluteError("train_acc")
    test_loss, test_acc = charge_gn.Mean("test_loss"), charge_gn.MeanAbsoluteError("test_acc")
    best = np.inf
    for epoch in range(args.epochs):
        for m in (train_loss, train_acc, test_loss, test_acc):
            m.reset_states()
        train_preds, test_preds = [], []
        nsteps = len(it) // world
        for s in range(nsteps):
            i = shard.dp_step_molecules(it, world, s)[rank]
            off, xyz, x, Q, y = one(i)
            q, _ = eng.train_step_xyz(off, xyz, x, Q, y, N, apply=True)
            pad = np.zeros(N, np.float32)
            pad[:len(q)] = q
            ypad = np.zeros(N, np.float32)
            ypad[:len(y)] = y
            train_loss((ypad - pad) ** 2)
            train_acc(pad, ypad)
            train_preds.append(pad)
        for i in ie:
            off, xyz, x, Q, y = one(int(i))
            q = eng.forward_xyz(off, xyz, x, Q, N)
            pad = np.zeros(N, np.float32)
            pad[:len(q)] = q
            ypad = np.zeros(N, np.float32)
            ypad[:len(y)] = y
            test_loss((ypad - pad) ** 2)
            test_acc(pad, ypad)
            test_preds.append(pad)
        if test_acc.result() < best and rank == 0:
            best = test_acc.result()
            model.save_weights(args.out)
            np.save("train_pred_charges.npy", np.array(train_preds))
            np.save("train_lab_charges.npy", np.array([np.pad(labels[i], (0, N - len(labels[i]))) for i in it[:len(train_preds)]]))
            np.save("test_pred_charges.npy", np.array(test_preds))
            np.save("test_lab_charges.npy", np.array([np.pad(labels[i], (0, N - len(labels[i]))) for i in ie]))
        if rank == 0:
            template = 'Epoch {}, Loss: {}, Acc: {}, Test Loss: {}, Test Acc: {}'
            print(template.format(epoch, train_loss.result(), train_acc.result(), test_loss.result(), test_acc.result()), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
