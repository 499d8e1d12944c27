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


def test_rewrite_of_reference_checkpoint_is_byte_identical(tmp_path):
    """The only pin available without TensorFlow: re-writing the reference's single-shard checkpoint
    (models/model_weights, written by charge_gn.py:462) from its decoded tensors reproduces BOTH files byte for byte --
    tensor order in the data file, string-tensor layout and its two checksums, index block, footer."""
    from epnn_amd import checkpoint as ck
    src = os.path.join(MODELS, "model_weights")
    w = ck.load_epnn_weights(src)
    for graph in (ck.read_object_graph(src), None):          # the stored graph string, and the generated one
        dst = str(tmp_path / ("a" if graph else "b") / "model_weights")
        ck.save_epnn_weights(dst, w, graph)
        for suf in (".index", ".data-00000-of-00001"):
            with open(src + suf, "rb") as f, open(dst + suf, "rb") as g:
                assert f.read() == g.read(), suf


@pytest.mark.parametrize("name,T", [("decay_model_weights", 5), ("model_weights", 5), ("model2_weights", 3)])
def test_generated_object_graph_equals_the_stored_one(name, T):
    """keras_object_graph(T) is the graph string Keras wrote into every shipped checkpoint (read_object_graph verifies
    the string tensor's length checksum and entry checksum on the way)."""
    from epnn_amd import checkpoint as ck
    stored = ck.read_object_graph(os.path.join(MODELS, name))
    assert ck.keras_object_graph(T) == stored
    keys = ck.object_graph_keys(stored)
    assert len(keys) == 6 * (2 * T + 1)
    assert keys[0] == "layer_with_weights-0/update_fn/layer_set/0/kernel/.ATTRIBUTES/VARIABLE_VALUE"


def test_string_tensor_checksums_follow_tensorflow(tmp_path):
    """Known answers from the reference's own file: length checksum 904473428 and entry crc 4144287813 of
    models/model_weights' _CHECKPOINTABLE_OBJECT_GRAPH; a flipped payload bit or length checksum is refused."""
    import struct
    from epnn_amd import checkpoint as ck
    src = os.path.join(MODELS, "model_weights")
    graph = ck.read_object_graph(src)
    raw, crc = ck._string_tensor(graph)
    assert struct.unpack("<I", raw[2:6])[0] == 904473428 and crc == 4144287813
    _, entries = ck.read_index(src)
    assert entries["_CHECKPOINTABLE_OBJECT_GRAPH"]["crc32c"] == crc
    for f in os.listdir(MODELS):
        if f.startswith("model_weights"):
            shutil.copy(os.path.join(MODELS, f), tmp_path / f)
    data = tmp_path / "model_weights.data-00000-of-00001"
    good = data.read_bytes()
    for pos in (298708 + 3, 298708 + 100):                   # inside the length checksum / inside the payload
        bad = bytearray(good)
        bad[pos] ^= 0x10
        data.write_bytes(bytes(bad))
        with pytest.raises(ValueError, match="checksum"):
            ck.read_object_graph(str(tmp_path / "model_weights"))


def test_tensor_without_graph_node_is_refused(tmp_path):
    from epnn_amd import checkpoint as ck
    with pytest.raises(ValueError, match="object graph"):
        ck.write_bundle(str(tmp_path / "x"), {"not/in/graph": np.zeros(3, np.float32)}, ck.keras_object_graph(2))


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


@pytest.mark.parametrize("widths", [[16], [64, 32], [8, 24, 40]])
def test_round_trip_with_other_update_layers(tmp_path, widths):
    """make_model(layers, ...) sizes the update MLP (charge_gn.py:371): the writer's object graph and keys follow the number of
    Dense layers under update_fn/layer_set, the reader finds them all again; the reference's own [32, 32] graph is unchanged."""
    from conftest import random_weights
    from epnn_amd import checkpoint
    w = random_weights(9, 2, seed=4)
    rng = np.random.default_rng(0)
    dims = [80] + widths + [48]
    w["upd"] = [(rng.normal(size=(i, o)).astype(np.float32), rng.normal(size=(o,)).astype(np.float32)) for i, o in zip(dims[:-1], dims[1:])]
    checkpoint.save_epnn_weights(str(tmp_path / "g"), w)
    back = checkpoint.load_epnn_weights(str(tmp_path / "g"))
    assert len(back["upd"]) == len(widths) + 1
    for (k0, b0), (k1, b1) in zip(w["upd"], back["upd"]):
        assert np.array_equal(k0, k1) and np.array_equal(b0, b1)
    for t in range(2):
        for l in range(3):
            assert np.array_equal(w["msg"][t][l][0], back["msg"][t][l][0]) and np.array_equal(w["pas"][t][l][1], back["pas"][t][l][1])
    assert checkpoint.keras_object_graph(2) == checkpoint.keras_object_graph(2, n_upd=3)
