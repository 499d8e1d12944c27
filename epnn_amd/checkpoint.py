"""TensorFlow-free reader/writer for the TF2 "tensor bundle" checkpoints the reference ships.

The reference restores its weights with ``model.load_weights('./models/decay_model_weights')``
(reference ``infer.py:57``) and stores them with ``model.save_weights('models/model_weights')``
(reference ``charge_gn.py:462``).  Both go through TensorFlow's tensor-bundle format:

``<prefix>.index``
    a LevelDB-style SSTable: ``[data block]* [metaindex block] [index block] [48-byte footer]``.
    Every block is followed by a 1-byte compression tag (0 = none) and a 4-byte masked CRC32C.
    Block entries are ``varint shared | varint non_shared | varint value_len | key suffix | value``
    with prefix-compressed keys; a block ends with its restart offsets (u32 each) and their count (u32).
    The footer holds two block handles (varint offset, varint size) for the metaindex and the index
    block, zero padding, and the magic ``0xdb4775248b80fb57`` (little endian).
    Key ``""`` maps to a ``BundleHeaderProto`` (field 1 = num_shards), every other key to a
    ``BundleEntryProto`` (1 dtype, 2 shape{2 dim{1 size}}, 3 shard_id, 4 offset, 5 size, 6 fixed32 crc32c).
``<prefix>.data-%05d-of-%05d``
    raw little-endian row-major tensor bytes at ``offset`` of shard ``shard_id``.

Only what the EPNN checkpoints need is implemented: float32 tensors and the scalar string tensor that
holds the Keras object graph (``_CHECKPOINTABLE_OBJECT_GRAPH``).
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DT_FLOAT = 1
_DT_STRING = 7
_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"

# --------------------------------------------------------------------------- crc32c (Castagnoli)
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78
        tab = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if (c & 1) else (c >> 1)
            tab[i] = c
        _CRC_TABLE = tab
    return _CRC_TABLE


def crc32c(data: bytes, crc: int = 0) -> int:
    tab = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask(crc: int) -> int:
    """LevelDB/TF masked crc: rotate right by 15 and add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# --------------------------------------------------------------------------- varint / protobuf
def _get_varint(buf: bytes, pos: int):
    shift = 0
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not (b & 0x80):
            return val, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf: bytes):
    """Yield (field_number, wire_type, value) of one protobuf message (wire types 0, 1, 2, 5)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = struct.unpack("<I", buf[pos:pos + 4])[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, val


def _parse_entry(buf: bytes):
    ent = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None}
    for fno, _, val in _parse_proto(buf):
        if fno == 1:
            ent["dtype"] = val
        elif fno == 2:
            dims = []
            for f2, _, v2 in _parse_proto(val):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = v3
                    dims.append(size)
            ent["shape"] = dims
        elif fno == 3:
            ent["shard_id"] = val
        elif fno == 4:
            ent["offset"] = val
        elif fno == 5:
            ent["size"] = val
        elif fno == 6:
            ent["crc32c"] = val
    return ent


# --------------------------------------------------------------------------- sstable
def _read_block(data: bytes, offset: int, size: int, verify: bool):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed tensor-bundle index blocks are not supported")
    if verify:
        stored = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
        if _mask(crc32c(raw + bytes([ctype]))) != stored:
            raise ValueError("tensor-bundle index block checksum mismatch")
    nrestart = struct.unpack("<I", raw[-4:])[0]
    end = len(raw) - 4 - 4 * nrestart
    pos = 0
    key = b""
    out = []
    while pos < end:
        shared, pos = _get_varint(raw, pos)
        non_shared, pos = _get_varint(raw, pos)
        vlen, pos = _get_varint(raw, pos)
        key = key[:shared] + raw[pos:pos + non_shared]
        pos += non_shared
        out.append((key, raw[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(prefix: str, verify: bool = True):
    """Return (num_shards, OrderedDict key -> entry dict) of ``<prefix>.index``."""
    with open(prefix + ".index", "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != _MAGIC:
        raise ValueError(f"{prefix}.index is not a tensor-bundle index (bad magic)")
    footer = data[-48:]
    pos = 0
    _, pos = _get_varint(footer, pos)  # metaindex offset
    _, pos = _get_varint(footer, pos)  # metaindex size
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    entries = OrderedDict()
    num_shards = 1
    for _, handle in _read_block(data, ioff, isize, verify):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for key, val in _read_block(data, boff, bsize, verify):
            if key == b"":
                for fno, _, v in _parse_proto(val):
                    if fno == 1:
                        num_shards = v
            else:
                entries[key.decode()] = _parse_entry(val)
    return num_shards, entries


def read_bundle(prefix: str, verify: bool = True):
    """Read every float32 tensor of a bundle: OrderedDict key -> np.ndarray (float32)."""
    num_shards, entries = read_index(prefix, verify)
    shards = {}
    out = OrderedDict()
    for key, ent in entries.items():
        if ent["dtype"] != _DT_FLOAT:
            continue
        sid = ent["shard_id"]
        if sid not in shards:
            with open(f"{prefix}.data-{sid:05d}-of-{num_shards:05d}", "rb") as f:
                shards[sid] = f.read()
        raw = shards[sid][ent["offset"]:ent["offset"] + ent["size"]]
        if len(raw) != ent["size"]:
            raise ValueError(f"tensor {key}: data shard truncated")
        if verify and ent["crc32c"] is not None and _mask(crc32c(raw)) != ent["crc32c"]:
            raise ValueError(f"tensor {key}: checksum mismatch")
        out[key] = np.frombuffer(raw, dtype="<f4").reshape(ent["shape"]).copy()
    return out


# --------------------------------------------------------------------------- EPNN weight layout
def _mlp_from(bundle, base):
    layers = []
    k = 0
    while f"{base}/layer_set/{k}/kernel{_SUFFIX}" in bundle:
        layers.append((bundle[f"{base}/layer_set/{k}/kernel{_SUFFIX}"],
                       bundle[f"{base}/layer_set/{k}/bias{_SUFFIX}"]))
        k += 1
    if not layers:
        raise KeyError(f"no MLP found under {base}")
    return layers


def load_epnn_weights(prefix: str, verify: bool = True):
    """Decode an EPNN checkpoint into ``{"msg": [T][3](W,b), "upd": [3](W,b), "pas": [T][3](W,b)}``.

    Key layout (reference ``charge_gn.py:48-54, 80-85, 371-374``): ``layer_with_weights-0`` is the
    ``GNN_layer`` (``update_fn`` + ``message_fns/0..T-2`` + ``message_fn`` = step T-1, because the attribute
    assigned at ``charge_gn.py:61`` aliases the last list element) and ``layer_with_weights-1`` the
    ``EPN_layer`` (``pass_fns/0..T-2`` + ``pass_fn`` = step T-1, ``charge_gn.py:99``).
    """
    bundle = read_bundle(prefix, verify)
    g, p = "layer_with_weights-0", "layer_with_weights-1"

    def steps(root, listname, lastname):
        out = []
        t = 0
        while f"{root}/{listname}/{t}/layer_set/0/kernel{_SUFFIX}" in bundle:
            out.append(_mlp_from(bundle, f"{root}/{listname}/{t}"))
            t += 1
        out.append(_mlp_from(bundle, f"{root}/{lastname}"))
        return out

    return {"msg": steps(g, "message_fns", "message_fn"),
            "upd": _mlp_from(bundle, f"{g}/update_fn"),
            "pas": steps(p, "pass_fns", "pass_fn")}


# --------------------------------------------------------------------------- writer (save_weights)
def _proto_field(fno: int, wt: int, payload) -> bytes:
    tag = _put_varint((fno << 3) | wt)
    if wt == 0:
        return tag + _put_varint(payload)
    if wt == 2:
        return tag + _put_varint(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack("<I", payload)
    raise ValueError(wt)


def _entry_proto(dtype, shape, shard_id, offset, size, crc) -> bytes:
    out = _proto_field(1, 0, dtype)
    dims = b"".join(_proto_field(2, 2, _proto_field(1, 0, d)) for d in shape)
    out += _proto_field(2, 2, dims)
    if shard_id:
        out += _proto_field(3, 0, shard_id)
    if offset:
        out += _proto_field(4, 0, offset)
    out += _proto_field(5, 0, size)
    out += _proto_field(6, 5, crc)
    return out


def _build_block(items, restart_interval=16) -> bytes:
    buf = bytearray()
    restarts = []
    prev = b""
    for n, (key, val) in enumerate(items):
        if n % restart_interval == 0:
            restarts.append(len(buf))
            shared = 0
        else:
            shared = 0
            m = min(len(prev), len(key))
            while shared < m and prev[shared] == key[shared]:
                shared += 1
        buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val))
        buf += key[shared:] + val
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def keras_object_graph(T: int, n_plain_layers: int = 14, n_upd: int = 3) -> bytes:
    """The ``TrackableObjectGraph`` proto Keras writes for the reference's ``make_model`` (``charge_gn.py:369-391``).

    TensorFlow restores by walking this graph, so a file without the real layout cannot be loaded by the reference's
    ``model.load_weights`` (``infer.py:57``).  The layout is fully determined by the model's Python object tree:
    node ids are handed out in breadth-first order of discovery from the Model; a node lists its children as
    (node_id, local_name) and a variable lists one attribute (VARIABLE_VALUE, Keras variable name, checkpoint key).

      root: ``layer-0 .. layer-13`` (the five Inputs and the TF-op layers of ``:382-384``; no weights, no children),
            ``layer_with_weights-0`` = ``layer-14`` = the GNN_layer, ``layer_with_weights-1`` = ``layer-15`` = the EPN_layer
      GNN_layer (``:48-54``): ``message_fns`` (list 0..T-1), ``update_fn``, ``message_fn`` (the attribute set at ``:61``: the
            SAME object as list element T-1)              EPN_layer (``:80-85``): ``pass_fns``, ``pass_fn`` (``:99``)
      MLP_layer (``:31-39``): ``nodes`` (plain list, no children), ``layer_set`` (list 0..2 of Dense: ``kernel``, ``bias``)

    Keras variable names follow creation order in ``make_model`` (``:371-374``): ``dense .. dense_2`` update MLP,
    then the T message MLPs, then the T pass MLPs.  Checked byte for byte against the graph strings of all three shipped
    checkpoints (``tests/test_checkpoint.py``).  ``n_upd``: Dense layers of the update MLP (``len(layers) + 1`` of
    ``make_model(layers, ...)``; 3 for the reference's ``[32, 32]``) -- the same tree with a longer or shorter ``layer_set``.
    """
    class Obj:
        def __init__(self):
            self.children = []        # (local_name, Obj)
            self.attr = None          # (full_name, checkpoint_key)

    dense_counter = [0]

    def mlp(n_dense=3):
        m = Obj()
        m.children.append(("nodes", Obj()))
        ls = Obj()
        for l in range(n_dense):
            d = Obj()
            stem = "dense" if dense_counter[0] == 0 else f"dense_{dense_counter[0]}"
            dense_counter[0] += 1
            for kind in ("kernel", "bias"):
                v = Obj()
                v.attr = f"{stem}/{kind}"
                d.children.append((kind, v))
            ls.children.append((str(l), d))
        m.children.append(("layer_set", ls))
        return m

    upd = mlp(n_upd)
    msgs = [mlp() for _ in range(T)]
    pas = [mlp() for _ in range(T)]
    gnn, epn = Obj(), Obj()
    lst = Obj()
    lst.children = [(str(t), msgs[t]) for t in range(T)]
    gnn.children = [("message_fns", lst), ("update_fn", upd), ("message_fn", msgs[T - 1])]
    lst = Obj()
    lst.children = [(str(t), pas[t]) for t in range(T)]
    epn.children = [("pass_fns", lst), ("pass_fn", pas[T - 1])]
    root = Obj()
    root.children = [(f"layer-{k}", Obj()) for k in range(n_plain_layers)]
    root.children += [("layer_with_weights-0", gnn), (f"layer-{n_plain_layers}", gnn),
                      ("layer_with_weights-1", epn), (f"layer-{n_plain_layers + 1}", epn)]

    # breadth-first numbering; the checkpoint key of a variable is the path by which it was FIRST reached
    ids, order, path = {id(root): 0}, [root], {id(root): ""}
    head = 0
    while head < len(order):
        o = order[head]
        head += 1
        for name, c in o.children:
            if id(c) not in ids:
                ids[id(c)] = len(order)
                order.append(c)
                path[id(c)] = (path[id(o)] + "/" + name) if path[id(o)] else name
    out = b""
    for o in order:
        nd = b""
        for name, c in o.children:
            nd += _proto_field(1, 2, _proto_field(1, 0, ids[id(c)]) + _proto_field(2, 2, name.encode()))
        if o.attr is not None:
            nd += _proto_field(2, 2, _proto_field(1, 2, b"VARIABLE_VALUE") + _proto_field(2, 2, o.attr.encode())
                               + _proto_field(3, 2, (path[id(o)] + _SUFFIX).encode()))
        out += _proto_field(1, 2, nd)
    return out


def object_graph_keys(graph_bytes: bytes):
    """Checkpoint keys of the variables in a ``TrackableObjectGraph``, in node order (= TensorFlow's save order)."""
    keys = []
    for fno, _, node in _parse_proto(graph_bytes):
        if fno != 1:
            continue
        for f2, _, val in _parse_proto(node):
            if f2 == 2:
                for f3, _, v3 in _parse_proto(val):
                    if f3 == 3:
                        keys.append(v3.decode())
    return keys


def _string_tensor(payload: bytes):
    """Bytes and entry checksum of a scalar string tensor the way TensorFlow's ``WriteStringTensor`` lays it out:
    ``varint length | u32 masked crc32c of the length as uint32 LE | bytes``; the entry's crc32c runs over the uint32
    length, then the 4 checksum bytes, then the payload."""
    ln = struct.pack("<I", len(payload))
    crc = crc32c(ln)
    len_ck = struct.pack("<I", _mask(crc))
    crc = crc32c(payload, crc32c(len_ck, crc))
    return _put_varint(len(payload)) + len_ck + payload, _mask(crc)


def _read_string_tensor(raw: bytes, stored_crc, verify: bool, what: str) -> bytes:
    ln, pos = _get_varint(raw, 0)
    len_ck = raw[pos:pos + 4]
    payload = raw[pos + 4:pos + 4 + ln]
    if len(payload) != ln or pos + 4 + ln != len(raw):
        raise ValueError(f"{what}: string tensor truncated")
    if verify:
        crc = crc32c(struct.pack("<I", ln))
        if struct.pack("<I", _mask(crc)) != len_ck:
            raise ValueError(f"{what}: string length checksum mismatch")
        if stored_crc is not None and _mask(crc32c(payload, crc32c(len_ck, crc))) != stored_crc:
            raise ValueError(f"{what}: checksum mismatch")
    return payload


def read_object_graph(prefix: str, verify: bool = True):
    """Raw bytes of the ``_CHECKPOINTABLE_OBJECT_GRAPH`` string tensor (both of its checksums verified), or None."""
    num_shards, entries = read_index(prefix, verify)
    ent = entries.get(_GRAPH_KEY)
    if ent is None:
        return None
    with open(f"{prefix}.data-{ent['shard_id']:05d}-of-{num_shards:05d}", "rb") as f:
        f.seek(ent["offset"])
        raw = f.read(ent["size"])
    return _read_string_tensor(raw, ent["crc32c"], verify, _GRAPH_KEY)


def _short_successor(key: bytes) -> bytes:
    """LevelDB ``BytewiseComparator::FindShortSuccessor``: the index block's separator after the last data block."""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def write_bundle(prefix: str, tensors, graph_bytes: bytes):
    """Write ``tensors`` (mapping key -> float32 array, keys WITHOUT the attribute suffix are accepted) and the object
    graph as a single-shard tensor bundle ``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` -- byte for byte what
    ``model.save_weights`` (``charge_gn.py:462``) writes for the same values: tensor bytes in the object graph's
    variable order with the graph string last, index entries in key order (restart interval 16), empty metaindex."""
    items = {}
    for key, arr in tensors.items():
        k = key if key.endswith(_SUFFIX) else key + _SUFFIX
        items[k] = np.ascontiguousarray(arr, dtype="<f4")
    order = [k for k in object_graph_keys(graph_bytes) if k in items]
    missing = sorted(set(items) - set(order))
    if missing:
        raise ValueError(f"tensors without a node in the object graph (TensorFlow could not restore them): {missing[:3]}")
    data = bytearray()
    entries = []
    for k in order:
        raw = items[k].tobytes()
        entries.append((k.encode(), _entry_proto(_DT_FLOAT, list(items[k].shape), 0, len(data), len(raw),
                                                 _mask(crc32c(raw)))))
        data += raw
    sraw, scrc = _string_tensor(graph_bytes)
    entries.append((_GRAPH_KEY.encode(), _entry_proto(_DT_STRING, [], 0, len(data), len(sraw), scrc)))
    data += sraw
    entries.sort(key=lambda kv: kv[0])
    header = _proto_field(1, 0, 1) + _proto_field(3, 2, _proto_field(1, 0, 1))  # num_shards=1, version.producer=1
    block = _build_block([(b"", header)] + entries)

    out = bytearray()

    def emit(blk):
        off = len(out)
        out.extend(blk)
        out.append(0)
        out.extend(struct.pack("<I", _mask(crc32c(blk + b"\x00"))))
        return off, len(blk)

    doff, dsize = emit(block)
    moff, msize = emit(_build_block([]))
    ioff, isize = emit(_build_block([(_short_successor(entries[-1][0]), _put_varint(doff) + _put_varint(dsize))]))
    footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    out.extend(footer)
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))


def epnn_weight_keys(T: int, n_upd: int = 3):
    """Checkpoint key stems (no suffix) in the reference's naming, per (group, step, layer, kind)."""
    keys = {}
    g, p = "layer_with_weights-0", "layer_with_weights-1"
    for t in range(T):
        mroot = f"{g}/message_fns/{t}" if t < T - 1 else f"{g}/message_fn"
        proot = f"{p}/pass_fns/{t}" if t < T - 1 else f"{p}/pass_fn"
        for l in range(3):
            for kind in ("kernel", "bias"):
                keys[("msg", t, l, kind)] = f"{mroot}/layer_set/{l}/{kind}"
                keys[("pas", t, l, kind)] = f"{proot}/layer_set/{l}/{kind}"
    for l in range(n_upd):
        for kind in ("kernel", "bias"):
            keys[("upd", 0, l, kind)] = f"{g}/update_fn/layer_set/{l}/{kind}"
    return keys


def save_epnn_weights(prefix: str, weights, graph_bytes: bytes | None = None):
    """Inverse of :func:`load_epnn_weights` (reference ``charge_gn.py:462``)."""
    T = len(weights["msg"])
    n_upd = len(weights["upd"])
    keys = epnn_weight_keys(T, n_upd)
    tensors = {}
    for t in range(T):
        for l in range(3):
            tensors[keys[("msg", t, l, "kernel")]] = weights["msg"][t][l][0]
            tensors[keys[("msg", t, l, "bias")]] = weights["msg"][t][l][1]
            tensors[keys[("pas", t, l, "kernel")]] = weights["pas"][t][l][0]
            tensors[keys[("pas", t, l, "bias")]] = weights["pas"][t][l][1]
    for l in range(n_upd):
        tensors[keys[("upd", 0, l, "kernel")]] = weights["upd"][l][0]
        tensors[keys[("upd", 0, l, "bias")]] = weights["upd"][l][1]
    # message_fns/T-1 and message_fn are the same node of the object graph: one tensor, stored under message_fn
    write_bundle(prefix, tensors, graph_bytes if graph_bytes is not None else keras_object_graph(T, n_upd=n_upd))
