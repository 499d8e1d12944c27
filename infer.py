"""Inference entry point, same flow and printed lines as the reference's ``infer.py`` (derekmetcalf/epnn):
featurise a directory of .xyz files -> make_model -> load_weights('./models/decay_model_weights') -> save
test_names.npy -> one model call per molecule with its wall-clock time -> "avg inference time" / "avg feature time".

The reference script is not runnable as committed (``path = ''`` must be edited by hand, ``repeats`` is undefined,
``n_elems = 9`` contradicts the 10-column featuriser it calls).  Here the directory is the first
command-line argument (default: tests/golden/qm9_small/), ``repeats`` defaults to 1, and ``n_elems = 9`` selects the
8-element table the script itself defines (infer.py:13-30), which is what the shipped checkpoint was trained with.
The arithmetic runs on the MI355X through ``epnn_amd.charge_gn`` (no TensorFlow).

    python infer.py [xyz_dir/] [--repeats R] [--weights PREFIX] [--batched]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from epnn_amd import charge_gn  # noqa: E402


def test_step(model, h, e, x, q, y, mask):
    predictions = model([h, e, x, q, mask])
    return predictions


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("path", nargs="?", default=os.path.join(ROOT, "tests", "golden", "qm9_small") + "/")
    ap.add_argument("--repeats", type=int, default=1)
    ap.add_argument("--weights", default=os.path.join(ROOT, "models", "decay_model_weights"))
    ap.add_argument("--batched", action="store_true",
                    help="also run every molecule in ONE call through the compact entry (coordinates, no dense tensors)")
    ap.add_argument("--out", default="test_preds.npy")
    args = ap.parse_args(argv)

    h_dim = 48
    e_dim = 48
    layers = [32, 32]
    T = 5
    path = args.path if args.path.endswith("/") else args.path + "/"
    n_elems = 9
    repeats = args.repeats

    timeA = time.time()
    x, h, q, e, Q, y, mask, names = charge_gn.gen_padded_init_state(path, h_dim, e_dim, n_elems=n_elems)
    timeB = time.time()

    model = charge_gn.make_model(layers, h_dim, T, n_elems, x.shape[1])
    model.load_weights(args.weights)

    np.save("test_names.npy", names, allow_pickle=True)

    test_preds = []
    timeC = timeD = time.time()
    for i in range(len(x)):
        # one system per call, batch axis of length 1 (infer.py:62-69)
        hb, eb, xb, qb, yb, maskb = (np.asarray(t[i])[None] for t in (h, e, x, q, y, mask))
        timeC = time.time()
        for _ in range(repeats):
            t_call = time.time()
            test_preds.append(test_step(model, hb, eb, xb, qb, yb, maskb))
            print(time.time() - t_call)                       # the reference prints every call's seconds (infer.py:74)
        timeD = time.time()
    np.save(args.out, np.array(test_preds))

    print(f"avg inference time: {(timeD-timeC)/repeats}")
    print(f"avg feature time:{(timeB-timeA)}")

    if args.batched:
        mols = [charge_gn.read_xyz(path + str(nm) + ".xyz", n_elems) for nm in names]
        offsets = np.zeros(len(mols) + 1, dtype=np.int32)
        offsets[1:] = np.cumsum([m[1].shape[0] for m in mols])
        t0 = time.time()
        qf = model.predict_xyz(offsets, np.concatenate([m[0] for m in mols]), np.concatenate([m[1] for m in mols]),
                               np.array([m[2] for m in mols], dtype=np.float32))
        t1 = time.time()
        worst = 0.0
        for i in range(len(mols)):
            n = mols[i][1].shape[0]
            worst = max(worst, float(np.abs(qf[offsets[i]:offsets[i + 1]] - test_preds[i * repeats][0, :n, 0]).max()))
        print(f"batched compact entry: {len(mols)} molecules in {t1-t0:.6f} s, max |dq| vs per-molecule calls {worst:.2e}")
        # the same through the pipeline that keeps several batches in flight (here: one molecule per batch)
        singles = [(np.array([0, m[1].shape[0]], dtype=np.int32), m[0], m[1], np.array([m[2]], dtype=np.float32)) for m in mols]
        t0 = time.time()
        streamed = list(model.predict_xyz_stream(singles))
        t1 = time.time()
        worst = max(float(np.abs(streamed[i] - qf[offsets[i]:offsets[i + 1]]).max()) for i in range(len(mols)))
        print(f"pipelined compact entry: {len(mols)} calls in {t1-t0:.6f} s (incl. creating the pipeline), max |dq| vs the batch {worst:.2e}")
    return np.array(test_preds), names


if __name__ == "__main__":
    main()
