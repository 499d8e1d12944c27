import os
import sys
import tarfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def val_dir(tmp_path_factory):
    """The 871 validation xyz files of the reference's recorded split (tests/golden/mixed_val.tar.gz)."""
    d = tmp_path_factory.mktemp("mixed_val")
    with tarfile.open(os.path.join(GOLDEN, "mixed_val.tar.gz")) as tf:
        tf.extractall(d)
    return os.path.join(str(d), "mixed_val")


@pytest.fixture(scope="session")
def train_dir(tmp_path_factory):
    """The 3480 training xyz + label files of the reference's recorded split (tests/golden/mixed_train.tar.gz)."""
    d = tmp_path_factory.mktemp("mixed_train")
    with tarfile.open(os.path.join(GOLDEN, "mixed_train.tar.gz")) as tf:
        tf.extractall(d)
    return os.path.join(str(d), "mixed_train")


@pytest.fixture(scope="session")
def train_names():
    return [str(n) for n in np.load(os.path.join(GOLDEN, "train_names.npy"), allow_pickle=True)]


@pytest.fixture(scope="session")
def val_names():
    return [str(n) for n in np.load(os.path.join(GOLDEN, "val_names.npy"), allow_pickle=True)]


@pytest.fixture(scope="session")
def val_gold():
    return np.load(os.path.join(GOLDEN, "test_pred_charges.npy"))


@pytest.fixture(scope="session")
def weights_decay():
    from epnn_amd import checkpoint
    return checkpoint.load_epnn_weights(os.path.join(ROOT, "models", "decay_model_weights"))


@pytest.fixture(scope="session")
def weights_full():
    from epnn_amd import checkpoint
    return checkpoint.load_epnn_weights(os.path.join(ROOT, "models", "model_weights"))


def random_weights(nx, T, seed=0, scale=1.0, h_dim=48):
    """Glorot-uniform kernels like Keras Dense defaults (charge_gn.py:38-39) but with non-zero biases, so that
    every term of the path is exercised (the shipped decay_model_weights has a collapsed GNN)."""
    rng = np.random.default_rng(seed)
    F = nx + h_dim + 1

    def dense(i, o, bias=0.1):
        lim = scale * np.sqrt(6.0 / (i + o))
        return (rng.uniform(-lim, lim, (i, o)).astype(np.float32), rng.uniform(-bias, bias, (o,)).astype(np.float32))

    return {"msg": [[dense(2 * F + h_dim, 32), dense(32, 32), dense(32, 32)] for _ in range(T)],
            "upd": [dense(h_dim + 32, 32), dense(32, 32), dense(32, h_dim)],
            "pas": [[dense(2 * F + h_dim, 32), dense(32, 32), dense(32, 1, 0.0)] for _ in range(T)]}


def load_molecules(val_dir, names, nx=9):
    from oracle import epnn_oracle as orc
    mols = [orc.parse_xyz(os.path.join(val_dir, nm + ".xyz"), nx) for nm in names]
    offsets = np.zeros(len(mols) + 1, dtype=np.int32)
    offsets[1:] = np.cumsum([m[1].shape[0] for m in mols])
    xyz = np.concatenate([m[0] for m in mols]).astype(np.float32)
    x = np.concatenate([m[1] for m in mols]).astype(np.float32)
    Q = np.array([m[2] for m in mols], dtype=np.float32)
    return mols, offsets, xyz, x, Q


@pytest.fixture(scope="session")
def gpu_engine_factory():
    engines = []

    def make(**kw):
        from epnn_amd.engine import Engine
        e = Engine(**kw)
        engines.append(e)
        return e

    yield make
    for e in engines:
        e.close()
