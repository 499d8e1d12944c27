"""TF tensor-bundle reader/writer (epnn_amd/checkpoint.py) on the reference's real checkpoint files. CPU only."""
import os
import shutil

import numpy as np
import pytest

from conftest import ROOT

MODELS = os.path.join(ROOT, "models")


@pytest.mark.parametrize("name,nx,T,params", [("decay_model_weights", 9, 5, 74037), ("model_weights", 10, 5, 74677),
                                               ("model2_weights", 9, 3, 46515)])
def test_reads_reference_checkpoints(name, nx, T, params):
    from epnn_amd import checkpoint as ck
    w = ck.load_epnn_weights(os.path.join(MODELS, name))      # verifies every block and tensor CRC32C
    F = nx + 49
    assert len(w["msg"]) == T and len(w["pas"]) == T
    total = 0
    for grp, last in (("msg", 32), ("pas", 1)):
        for mlp in w[grp]:
            assert [k.shape for k, _ in mlp] == [(2 * F + 48, 32), (32, 32), (32, last)]
            total += sum(k.size + b.size for k, b in mlp)
    assert [k.shape for k, _ in w["upd"]] == [(80, 32), (32, 32), (32, 48)]
    total += sum(k.size + b.size for k, b in w["upd"])
    assert total == params
    # the last pass bias never gets gradient (cancels in f_ij - f_ji) and is exactly 0 in every shipped checkpoint
    assert all(float(m[2][1][0]) == 0.0 for m in w["pas"])


def test_message_fn_alias_is_last_step():
    """'message_fn' / 'pass_fn' keys hold step T-1 (charge_gn.py:61,99 alias the last list element)."""
    from epnn_amd import checkpoint as ck
    num_shards, entries = ck.read_index(os.path.join(MODELS, "decay_model_weights"))
    assert num_shards == 2
    keys = list(entries)
    assert any(k.startswith("layer_with_weights-0/message_fn/") for k in keys)
    assert not any(k.startswith("layer_with_weights-0/message_fns/4/") for k in keys)
    assert any(k.startswith("layer_with_weights-0/message_fns/3/") for k in keys)


def test_write_read_round_trip(tmp_path):
    from epnn_amd import checkpoint as ck
    w = ck.load_epnn_weights(os.path.join(MODELS, "decay_model_weights"))
    graph = ck.read_object_graph(os.path.join(MODELS, "decay_model_weights"))
    assert graph is not None and b"layer_with_weights-0" in graph
    ck.save_epnn_weights(str(tmp_path / "ck" / "w"), w, graph)
    w2 = ck.load_epnn_weights(str(tmp_path / "ck" / "w"))
    for grp in ("msg", "pas"):
        for a, b in zip(w[grp], w2[grp]):
            for (k1, b1), (k2, b2) in zip(a, b):
                assert np.array_equal(k1, k2) and np.array_equal(b1, b2)
    assert ck.read_object_graph(str(tmp_path / "ck" / "w")) == graph


def test_corruption_is_detected(tmp_path):
    from epnn_amd import checkpoint as ck
    for f in os.listdir(MODELS):
        if f.startswith("model2_weights"):
            shutil.copy(os.path.join(MODELS, f), tmp_path / f)
    data = tmp_path / "model2_weights.data-00001-of-00002"
    raw = bytearray(data.read_bytes())
    raw[1000] ^= 0x40
    data.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        ck.load_epnn_weights(str(tmp_path / "model2_weights"))
    with pytest.raises(ValueError):
        (tmp_path / "bad.index").write_bytes(b"\x00" * 64)
        ck.read_index(str(tmp_path / "bad"))
