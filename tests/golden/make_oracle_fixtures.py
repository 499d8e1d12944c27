#!/usr/bin/env python3
"""Stores OUR float64 oracle's output for the two large synthetic systems of the GPU parity tests, so that the GPU suite does
not spend three minutes of host time per run recomputing them.

NOT reference data: these arrays are produced by oracle/epnn_oracle.py (this repo's CPU restatement of charge_gn.py) on synthetic
inputs with random weights -- they pin nothing about the reference, they only cache a deterministic computation.  The fixture
records a SHA-256 of the inputs (coordinates, features, every weight tensor); the tests recompute the inputs, compare the hash and
fall back to running the oracle when it differs, and tests/test_oracle_golden.py::test_cached_oracle_fixtures_are_current
recomputes the cheaper one on the CPU to show the cache is the oracle's current output.

    python tests/golden/make_oracle_fixtures.py          (about 4 minutes on 8 cores)
writes tests/golden/oracle_box1500.npz and tests/golden/oracle_subbox4096.npz.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def inputs_hash(xyz, x, Q, N, w):
    h = hashlib.sha256()
    for a in (xyz, x, np.asarray(Q, np.float32), np.asarray([N], np.int64)):
        h.update(np.ascontiguousarray(a).tobytes())
    for t in range(len(w["msg"])):
        for W, b in w["msg"][t]:
            h.update(np.ascontiguousarray(W).tobytes()); h.update(np.ascontiguousarray(b).tobytes())
    for W, b in w["upd"]:
        h.update(np.ascontiguousarray(W).tobytes()); h.update(np.ascontiguousarray(b).tobytes())
    for t in range(len(w["pas"])):
        for W, b in w["pas"][t]:
            h.update(np.ascontiguousarray(W).tobytes()); h.update(np.ascontiguousarray(b).tobytes())
    return h.hexdigest()


def box1500_case():
    """inputs of tests/test_gpu_parity.py::test_box_system_vs_oracle"""
    from conftest import random_weights
    from epnn_amd import synth
    w = random_weights(9, 2, seed=21, scale=0.35)
    offsets, xyz, x, Q, N = synth.box_system(n_atoms=1500, seed=0)
    return xyz, x, Q, N, w


def subbox4096_case(box100k=None):
    """inputs of tests/test_gpu_parity.py::test_box_subbox_4096_vs_oracle"""
    from conftest import random_weights
    from epnn_amd import synth
    _, xyz_all, x_all, _, _ = box100k if box100k is not None else synth.box_system(n_atoms=100_000, seed=0)
    order = np.argsort(xyz_all.max(axis=1), kind="stable")[:4096]
    order.sort()
    xyz, x = xyz_all[order], x_all[order]
    w = random_weights(9, 2, seed=21, scale=0.35)
    for t in range(2):
        w["msg"][t][2] = (w["msg"][t][2][0] / 64.0, w["msg"][t][2][1] / 64.0)
    return xyz, x, np.array([1.0], np.float32), 4096, w


def load(name, xyz, x, Q, N, w):
    """the cached oracle arrays of fixture `name` if they were made from exactly these inputs, else None"""
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        return None
    z = np.load(path)
    if str(z["inputs_sha256"]) != inputs_hash(xyz, x, Q, N, w):
        return None
    return z


def main():
    from oracle import epnn_oracle as orc
    xyz, x, Q, N, w = box1500_case()
    ref = orc.forward_xyz(xyz, x, Q[0], w, N=N, dtype=np.float64, row_block=128)
    ref32 = orc.forward_xyz(xyz, x, Q[0], w, N=N, dtype=np.float32, row_block=128)
    np.savez_compressed(os.path.join(HERE, "oracle_box1500.npz"), q_float64=ref, q_float32=ref32,
                        inputs_sha256=inputs_hash(xyz, x, Q, N, w),
                        made_by="tests/golden/make_oracle_fixtures.py: oracle.epnn_oracle.forward_xyz (this repo's oracle, not reference data)")
    print("box1500 done", flush=True)
    xyz, x, Q, N, w = subbox4096_case()
    ref = orc.forward_xyz_large(xyz, x, np.float32(1.0), w, dtype=np.float64, row_block=64)
    np.savez_compressed(os.path.join(HERE, "oracle_subbox4096.npz"), q_float64=ref, inputs_sha256=inputs_hash(xyz, x, Q, N, w),
                        made_by="tests/golden/make_oracle_fixtures.py: oracle.epnn_oracle.forward_xyz_large (this repo's oracle, not reference data)")
    print("subbox4096 done", flush=True)


if __name__ == "__main__":
    main()
