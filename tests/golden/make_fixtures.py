#!/usr/bin/env python3
"""Regenerates everything under tests/golden/ and models/ from the read-only reference checkout.

Run ONLY in the build container (needs /root/reference); the outputs are committed, the GPU box never runs this.

What it copies (data files, not source):
  * models/<ckpt>.{index,data-*}            the reference's three TF tensor-bundle checkpoints
  * tests/golden/test_pred_charges.npy ...   stored TF predictions for the 871-system validation split
  * tests/golden/mixed_val.tar.gz            the 871 validation xyz files (+ label .npy) out of data/mixed.tar.gz
  * tests/golden/mixed_train.tar.gz          the 3480 training xyz files (+ label .npy) of the recorded split (train_names.npy)
  * tests/golden/protein/                    6qlp_capped.xyz + preds.npy out of data/protein.tar.gz
  * tests/golden/qm9_small/                  four small xyz files (+ labels) for the infer.py plumbing test
What it computes with the reference's own NumPy/SciPy featuriser (charge_gn.get_init_edges /
gen_padded_init_state, imported with a stub `tensorflow` module because TensorFlow is not installed here):
  * tests/golden/featurise_qm9_small.npz     x, h, q, e, Q, y, mask, names for tests/golden/qm9_small/
  * tests/golden/edges_081300.npz            xyz -> (e, C) for one molecule
"""
import io
import os
import shutil
import sys
import tarfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SMALL = ["dsgdb9nsd_081300", "dsgdb9nsd_000228", "CCxdOXyOmY.psi4-opt-b3lyp-adz", "SSI-001ASN-030MET-1-dimer"]


def stub_tensorflow():
    tf = types.ModuleType("tensorflow")
    keras = types.ModuleType("tensorflow.keras")
    layers = types.ModuleType("tensorflow.keras.layers")

    class Layer:
        def __init__(self, *a, **k):
            pass

    layers.Layer = Layer
    layers.Dense = type("Dense", (), {"__init__": lambda self, *a, **k: None})
    keras.layers = layers
    keras.Model = type("Model", (), {})
    tf.keras = keras
    tf.function = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.keras"] = keras
    sys.modules["tensorflow.keras.layers"] = layers


def main():
    os.makedirs(os.path.join(ROOT, "models"), exist_ok=True)
    for f in os.listdir(os.path.join(REF, "models")):
        src = os.path.join(REF, "models", f)
        if os.path.isfile(src):
            shutil.copyfile(src, os.path.join(ROOT, "models", f))
    ms = os.path.join(REF, "models", "model_systems")
    for f in ["test_pred_charges.npy", "test_lab_charges.npy", "val_names.npy", "train_names.npy"]:
        shutil.copyfile(os.path.join(ms, f), os.path.join(HERE, f))

    names = [str(n) for n in np.load(os.path.join(ms, "val_names.npy"), allow_pickle=True)]
    want = set(names) | set(SMALL)
    small_dir = os.path.join(HERE, "qm9_small")
    os.makedirs(small_dir, exist_ok=True)
    with tarfile.open(os.path.join(REF, "data", "mixed.tar.gz")) as src, \
            tarfile.open(os.path.join(HERE, "mixed_val.tar.gz"), "w:gz") as dst:
        members = {m.name: m for m in src.getmembers() if m.isfile()}
        for name in sorted(members):
            base = os.path.basename(name)
            stem = base[:-4]
            if base.endswith(".xyz") and stem in want:
                blob = src.extractfile(members[name]).read()
                if stem in set(names):
                    ti = tarfile.TarInfo("mixed_val/" + base)
                    ti.size = len(blob)
                    dst.addfile(ti, io.BytesIO(blob))
                if stem in SMALL:
                    with open(os.path.join(small_dir, base), "wb") as f:
                        f.write(blob)
                # labels: QM9 files share the stem; SSI/ion files use "<...>_mbis-mtp.npy"-style names
                lab = os.path.dirname(name) + "/" + stem + ".npy"
                if lab in members:
                    lblob = src.extractfile(members[lab]).read()
                    if stem in set(names):
                        ti = tarfile.TarInfo("mixed_val/" + stem + ".npy")
                        ti.size = len(lblob)
                        dst.addfile(ti, io.BytesIO(lblob))
                    if stem in SMALL:
                        with open(os.path.join(small_dir, stem + ".npy"), "wb") as f:
                            f.write(lblob)

    # the recorded training split (models/model_systems/train_names.npy; charge_gn.py:431-434): BASELINE.json configs[2]'s data
    tnames = set(str(n) for n in np.load(os.path.join(ms, "train_names.npy"), allow_pickle=True))
    with tarfile.open(os.path.join(REF, "data", "mixed.tar.gz")) as src, \
            tarfile.open(os.path.join(HERE, "mixed_train.tar.gz"), "w:gz", compresslevel=9) as dst:
        members = {m.name: m for m in src.getmembers() if m.isfile()}
        for name in sorted(members):
            base = os.path.basename(name)
            if base[:-4] in tnames and (base.endswith(".xyz") or (base.endswith(".npy") and not base.endswith("splits.npy"))):
                blob = src.extractfile(members[name]).read()
                ti = tarfile.TarInfo("mixed_train/" + base)
                ti.size = len(blob)
                dst.addfile(ti, io.BytesIO(blob))

    pdir = os.path.join(HERE, "protein")
    os.makedirs(pdir, exist_ok=True)
    with tarfile.open(os.path.join(REF, "data", "protein.tar.gz")) as src:
        for want_f in ["protein/6qlp_capped.xyz", "protein/preds.npy"]:
            with open(os.path.join(pdir, os.path.basename(want_f)), "wb") as f:
                f.write(src.extractfile(want_f).read())

    # featurisation goldens through the reference's own NumPy code
    stub_tensorflow()
    sys.path.insert(0, REF)
    import charge_gn as ref  # noqa: E402

    # infer.py's element table (8 elements, nx = 9) is what the shipped decay_model_weights expects
    for table, tag in [(None, "nx10"), ("infer", "nx9")]:
        if table == "infer":
            ref.atom_num_dict = {'H': 1, 'C': 6, 'N': 7, 'O': 8, 'F': 9, 'S': 16, 'Cl': 17, 'Br': 35}
            ref.elem_dict = {'H': 0, 'C': 1, 'N': 2, 'O': 3, 'F': 4, 'S': 5, 'Cl': 6, 'Br': 7}
        x, h, q, e, Q, y, mask, nm = ref.gen_padded_init_state(small_dir + "/", 48, 48)
        order = np.argsort(nm)  # os.listdir order is filesystem dependent
        np.savez_compressed(os.path.join(HERE, f"featurise_qm9_small_{tag}.npz"),
                            x=x[order], h=h[order], q=q[order], e=e[order].astype(np.float32),
                            Q=np.array(Q)[order], y=y[order], mask=mask[order], names=nm[order])
    xyz = []
    for line in open(os.path.join(small_dir, "dsgdb9nsd_081300.xyz")).readlines()[2:]:
        xyz.append(line.split()[1:4])
    xyz = np.array(xyz, dtype=np.float32)
    e, C = ref.get_init_edges(xyz, np.array([]), num=48)
    np.savez_compressed(os.path.join(HERE, "edges_081300.npz"), xyz=xyz, e=e, C=C[:, :, 0])
    print("fixtures written")


if __name__ == "__main__":
    main()
