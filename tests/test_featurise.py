"""Featuriser: the oracle's restatement and the host-side logic of epnn_amd.charge_gn (file parsing, padding) vs arrays
produced by the reference's own NumPy/SciPy code (tests/golden/make_fixtures.py imported charge_gn.get_init_edges /
gen_padded_init_state).  The product's edge features are computed on the device (epnn_edges_ex): those tests are
marked gpu."""
import os

import numpy as np
import pytest


@pytest.mark.parametrize("nx", [9, 10])
def test_oracle_gen_padded_init_state_matches_reference(golden_dir, nx):
    from oracle import epnn_oracle as orc
    fx = np.load(os.path.join(golden_dir, f"featurise_qm9_small_nx{nx}.npz"))
    path = os.path.join(golden_dir, "qm9_small") + "/"
    x2, h2, q2, e2, Q2, y2, mask2, names2 = orc.gen_padded_init_state(path, 48, 48, nx)
    assert list(names2) == list(fx["names"])
    assert np.array_equal(x2, fx["x"]) and np.array_equal(e2.astype(np.float32), fx["e"])
    assert np.array_equal(q2, fx["q"]) and np.array_equal(mask2, fx["mask"]) and np.array_equal(y2, fx["y"])


def test_oracle_get_init_edges_matches_reference(golden_dir):
    from oracle import epnn_oracle as orc
    fx = np.load(os.path.join(golden_dir, "edges_081300.npz"))
    e2, C2 = orc.get_init_edges(fx["xyz"], num=48)
    assert np.array_equal(e2, fx["e"]) and np.array_equal(C2, fx["C"])
    # properties the kernels rely on: symmetric, zero diagonal, zero beyond the cutoff
    assert np.array_equal(e2, e2.transpose(1, 0, 2)) and np.all(e2[np.arange(18), np.arange(18)] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("nx", [9, 10])
def test_gen_padded_init_state_matches_reference(golden_dir, nx):
    """The product's featuriser: parsing / padding exactly as the reference's arrays; the edge features come from the
    device kernel (float64 cos / exp of the device's math library, then the float32 cast: equal to the reference's
    NumPy values except at float32 rounding boundaries)."""
    from epnn_amd import charge_gn
    fx = np.load(os.path.join(golden_dir, f"featurise_qm9_small_nx{nx}.npz"))
    path = os.path.join(golden_dir, "qm9_small") + "/"
    x, h, q, e, Q, y, mask, names = charge_gn.gen_padded_init_state(path, 48, 48, n_elems=nx)
    assert list(names) == list(fx["names"])
    assert x.dtype == np.float64 and e.dtype == np.float64 and mask.shape == fx["mask"].shape
    assert np.array_equal(x, fx["x"]) and np.array_equal(h, fx["h"]) and np.array_equal(mask, fx["mask"])
    assert np.array_equal(q, fx["q"]) and np.array_equal(y, fx["y"])
    assert np.array_equal(np.array(Q, dtype=np.float32), fx["Q"])
    e32 = e.astype(np.float32)
    assert np.abs(e32 - fx["e"]).max() <= 1e-7 and np.mean(e32 != fx["e"]) < 1e-3
    assert np.array_equal(e32 > 1e-5, fx["e"] > 1e-5)                  # the same near pairs (charge_gn.py:90-94)


@pytest.mark.gpu
def test_get_init_edges_matches_reference(golden_dir):
    from epnn_amd import charge_gn
    from oracle import epnn_oracle as orc
    fx = np.load(os.path.join(golden_dir, "edges_081300.npz"))
    e, C = charge_gn.get_init_edges(fx["xyz"], np.array([]), num=48)
    assert e.dtype == np.float32 and e.shape == fx["e"].shape and C.dtype == np.float64 and C.shape == fx["e"].shape
    assert np.abs(e - fx["e"]).max() <= 1e-7 and np.mean(e != fx["e"]) < 1e-3
    assert np.abs(C[:, :, 0] - fx["C"]).max() <= 1e-15 and np.array_equal(C[:, :, 0], C[:, :, 47])
    assert np.array_equal(e, e.transpose(1, 0, 2)) and np.all(e[np.arange(18), np.arange(18)] == 0)
    # the reference's default num = 32 and other constants (charge_gn.py:122)
    e32, C32 = charge_gn.get_init_edges(fx["xyz"], np.array([]))
    ref32, cref = orc.get_init_edges(fx["xyz"], num=32)
    assert e32.shape == (18, 18, 32) and np.abs(e32 - ref32).max() <= 1e-7
    e5, _ = charge_gn.get_init_edges(fx["xyz"], np.array([]), num=20, cutoff=2.5, eta=3.0)
    ref5, _ = orc.get_init_edges(fx["xyz"], num=20, cutoff=2.5, eta=3.0)
    assert np.abs(e5 - ref5).max() <= 1e-7
    with pytest.raises(ValueError):
        charge_gn.get_init_edges(fx["xyz"], np.array([3, 7]))     # the reference exit()s here (charge_gn.py:134-145)


def test_xyz_parsing_rules(tmp_path):
    """Atom count comes from the line count, not the header; total charge is the first token of line 2; extra
    columns are ignored (charge_gn.py:317-323)."""
    from epnn_amd import charge_gn
    (tmp_path / "m.xyz").write_text("99\n-1 2\nO 0.0 0.0 0.0 0.5\nH 0.96 0.0 0.0 junk\nCl 0.0 1.1 0.0\n")
    xyz, x, Q, nlines = charge_gn.read_xyz(str(tmp_path / "m.xyz"), 9)
    assert xyz.shape == (3, 3) and xyz.dtype == np.float32 and float(Q) == -1.0 and nlines == 5
    assert x.shape == (3, 9) and x[0, 0] == 8 and x[0, 4] == 1 and x[2, 0] == 17 and x[2, 7] == 1
    _, x10, _, _ = charge_gn.read_xyz(str(tmp_path / "m.xyz"), 10)
    assert x10.shape == (3, 10) and x10[2, 8] == 1
    with pytest.raises(KeyError):
        (tmp_path / "p.xyz").write_text("1\n0 1\nP 0 0 0\n")
        charge_gn.read_xyz(str(tmp_path / "p.xyz"), 9)           # P is not in infer.py's table


def test_horton_txt2npy(tmp_path):
    """Label tooling (reference data/horton_txt2npy.py): 4 header lines, then token 4 of every line is the charge."""
    from epnn_amd import labels
    d = tmp_path / "out"
    (d / "sub").mkdir(parents=True)
    body = "header 1\nheader 2\nheader 3\nheader 4\n" + "".join(
        f"{k} X 0.0 0.0 {q:.6f} 1.0 2.0\n" for k, q in enumerate([-0.25, 0.5, 0.125]))
    (d / "a-mtp.txt").write_text(body)
    (d / "sub" / "b-mtp.txt").write_text(body)
    (d / "ignored.txt").write_text("nothing")
    written = labels.horton_txt2npy(str(d))
    assert sorted(os.path.relpath(w, d) for w in written) == ["a-mtp.npy", os.path.join("sub", "b-mtp.npy")]
    for w in written:
        q = np.load(w, allow_pickle=True)
        assert q.dtype == np.float64 and np.array_equal(q, np.array([-0.25, 0.5, 0.125]))
