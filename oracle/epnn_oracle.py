"""CPU oracle for the EPNN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
path (epnn_amd/) never does; it fails loudly when its HIP library is missing.

This is a literal NumPy restatement of the reference's algorithm (derekmetcalf/epnn, files cited per function):
dense (B,N,N,.) tensors, materialised ``[a_i | a_j | e_ij]`` pair rows, three GEMMs per MLP sweep, 15 sweeps per
forward for T=5 -- the same arithmetic TensorFlow executes for ``charge_gn.py``, in float32 (``dtype=np.float32``,
used as the CPU baseline and for parity) or float64 (``dtype=np.float64``, used to measure float32 noise).
The only liberty taken is blocking over the first pair index ``i`` (``row_block``) so that a 2220-atom system
does not need several 3.2 GB temporaries at once; each block performs exactly the reference's operations.

Pinning: reproduces the reference's stored TensorFlow outputs
  * models/model_systems/test_pred_charges.npy  (871 systems, N=41)      max |dq| ~1.3e-6
  * data/protein.tar.gz -> protein/preds.npy      (2220 atoms, Q=+2)       max |dq| ~2.4e-6
(tests/test_oracle_golden.py).  Those goldens exercise decay_model_weights only, whose GNN output is collapsed
(every ReLU of the last update step is dead), so GNN_layer in its non-degenerate regime, padding dependence,
model_weights / model2_weights and training gradients are PARITY UNPINNED: for them this oracle is the only
reference, checked by invariants (conservation, permutation equivariance, the closed-form padding identity).
"""
from __future__ import annotations

import os

import numpy as np

# reference infer.py:13-30 (8 elements, nx = 9; matches decay_model_weights / model2_weights)
ATOM_NUM_8 = {'H': 1, 'C': 6, 'N': 7, 'O': 8, 'F': 9, 'S': 16, 'Cl': 17, 'Br': 35}
ELEM_8 = {'H': 0, 'C': 1, 'N': 2, 'O': 3, 'F': 4, 'S': 5, 'Cl': 6, 'Br': 7}
# reference charge_gn.py:9-28 (9 elements incl. P, nx = 10; matches model_weights)
ATOM_NUM_9 = {'H': 1, 'C': 6, 'N': 7, 'O': 8, 'F': 9, 'P': 15, 'S': 16, 'Cl': 17, 'Br': 35}
ELEM_9 = {'H': 0, 'C': 1, 'N': 2, 'O': 3, 'F': 4, 'P': 5, 'S': 6, 'Cl': 7, 'Br': 8}


def tables(nx):
    if nx == 9:
        return ATOM_NUM_8, ELEM_8
    if nx == 10:
        return ATOM_NUM_9, ELEM_9
    raise ValueError("nx must be 9 or 10")


# --------------------------------------------------------------------------- featurisation
def get_init_edges(xyz, num=48, cutoff=3.0, eta=2.0, row_block=256):
    """reference charge_gn.py:122-163: Gaussian-expanded distances times a cosine cutoff, float64 -> float32.

    ``xyz`` is float32 (charge_gn.py:330); SciPy's distance_matrix promotes it to float64 (:124)."""
    xyz = np.asarray(xyz, dtype=np.float32).astype(np.float64)
    n = xyz.shape[0]
    e = np.empty((n, n, num), dtype=np.float32)
    Call = np.empty((n, n), dtype=np.float64)
    for i0 in range(0, n, row_block):
        e[i0:i0 + row_block], Call[i0:i0 + row_block] = _edge_rows(xyz, i0, min(n, i0 + row_block), num, cutoff, eta)
    return e, Call


def _edge_rows(xyz64, i0, i1, num, cutoff, eta):
    """Rows i0..i1-1 of get_init_edges' output (the same operations on a block of rows)."""
    mu = np.linspace(0.1, cutoff, num=num)                              # :123
    d = xyz64[i0:i1, None, :] - xyz64[None, :, :]
    D = np.sqrt((d * d).sum(-1))                                        # :124 scipy minkowski_distance p=2
    C = (np.cos(np.pi * (D - 0.0) / cutoff) + 1.0) / 2.0                # :148
    C[D >= cutoff] = 0.0                                                # :150
    C[D <= 0.0] = 1.0                                                   # :151
    idx = np.arange(i0, i1)
    C[idx - i0, idx] = 0.0                                              # :152 fill_diagonal
    e = (C[:, :, None] * np.exp(-eta * (D[:, :, None] - mu[None, None, :]) ** 2)).astype(np.float32)   # :160-161
    return e, C


class EdgeRows:
    """The (1, n, n, num) edge tensor of ONE unpadded system, produced a block of rows at a time (for systems whose
    dense tensor does not fit: 4096 atoms = 3.2 GB in float32).  gnn_layer / epn_layer accept it in place of `e`."""

    def __init__(self, xyz, num=48, cutoff=3.0, eta=2.0):
        self.xyz = np.asarray(xyz, dtype=np.float32).astype(np.float64)
        self.num, self.cutoff, self.eta = num, cutoff, eta
        self.shape = (1, self.xyz.shape[0], self.xyz.shape[0], num)
        self._blocks = {}                  # float32 blocks are kept (the 15 sweeps of a forward ask for the same ones)

    def rows(self, i0, i1):
        key = (i0, min(i1, self.shape[1]))
        if key not in self._blocks:
            self._blocks[key] = _edge_rows(self.xyz, key[0], key[1], self.num, self.cutoff, self.eta)[0][None]
        return self._blocks[key]


def _e_rows(e, i0, i1, dtype):
    """Rows i0..i1-1 (axis 1) of the edge tensor: float32 values, and the same cast to the working dtype."""
    e32 = e.rows(i0, i1) if isinstance(e, EdgeRows) else np.asarray(e[:, i0:i1], dtype=np.float32)
    return e32, e32.astype(dtype, copy=False)


def parse_xyz(path, nx=9):
    """reference charge_gn.py:309-330: atoms = every line after the second; Q = first token of line 2."""
    atom_num, elem = tables(nx)
    with open(path, "r") as f:
        lines = f.readlines()
    Q = np.array(lines[1].strip().split()[0], dtype=np.float32)
    xyz, x = [], []
    for line in lines[2:]:
        data = line.split()
        xyz.append([data[1], data[2], data[3]])
        ohe = np.zeros(len(elem) + 1)
        ohe[0] = atom_num[data[0]]
        ohe[elem[data[0]] + 1] = 1
        x.append(ohe)
    return np.array(xyz, dtype=np.float32), np.array(x, dtype=np.float32), Q


def dense_inputs(xyz, x, Q, N, h_dim=48, e_dim=48, cutoff=3.0, eta=2.0):
    """reference charge_gn.py:331-364 for one molecule: the five (N,N,.) tensors make_model consumes.  (cutoff / eta
    are the reference's constants 3 and 2, charge_gn.py:148-161; parameters here because the C ABI's config has them.)"""
    n = x.shape[0]
    e, _ = get_init_edges(xyz, num=e_dim, cutoff=cutoff, eta=eta)
    x_p = np.zeros((N, N, x.shape[1]))
    h_p = np.zeros((N, N, h_dim))
    q_p = np.zeros((N, N, 1))
    e_p = np.zeros((N, N, e_dim))
    mask = np.zeros((N, N))
    avg_q = Q / np.float32(n)                                           # :337 float32 / int
    x_p[:n, :n] = x[None, :, :]                                         # row j*n+k of the tiled x is atom k
    q_p[:n, :n, 0] = np.float32(avg_q)
    e_p[:n, :n] = e
    mask[:n, :n] = 1
    return h_p, e_p, x_p, q_p, mask


def gen_padded_init_state(path, h_dim=48, e_dim=48, nx=9, names=None):
    """reference charge_gn.py:292-366 (directory reader).  ``names`` fixes the order (os.listdir order is
    filesystem dependent in the reference)."""
    if names is None:
        names = sorted(f[:-4] for f in os.listdir(path) if f.endswith(".xyz"))
    mols = [parse_xyz(os.path.join(path, nm + ".xyz"), nx) for nm in names]
    ys = []
    for nm, (xyz, x, Q) in zip(names, mols):
        lab = os.path.join(path, nm + ".npy")
        ys.append(np.array(np.load(lab), dtype=np.float32) if os.path.exists(lab) else np.zeros(x.shape[0]))
    N = max(len(y) for y in ys)
    out = [dense_inputs(xyz, x, Q, N, h_dim, e_dim) for xyz, x, Q in mols]
    h, e, x, q, mask = (np.stack([o[k] for o in out]) for k in range(5))
    y = np.zeros((len(names), N, 1))
    for b, yy in enumerate(ys):
        y[b, :len(yy), 0] = yy
    Q = [m[2] for m in mols]
    return x, h, q, e, Q, y, mask, np.array(names)


# --------------------------------------------------------------------------- layers
def _activation(name):
    """Keras activations by name, as keras.layers.Dense(activation=name) resolves them (reference charge_gn.py:38)."""
    if name == "relu":
        return lambda v: np.maximum(v, 0)
    if name is None or name == "linear":
        return lambda v: v
    if name == "tanh":
        return np.tanh
    if name == "sigmoid":
        return lambda v: 1.0 / (1.0 + np.exp(-v))
    raise ValueError(f"activation {name!r}")


def mlp(rows, layers, activation="relu"):
    """reference charge_gn.py:30-45 (MLP_layer): Dense(n, activation) for every layer but the last (:38), Dense(out_dim, None) (:39);
    `activation` defaults to 'relu' (:31), the only one the reference's own stacks use (:52,84,371)."""
    act = _activation(activation)
    for W, b in layers[:-1]:
        rows = act(rows @ W + b)
    W, b = layers[-1]
    return rows @ W + b


def _cast_layers(layers, dtype):
    return [(np.asarray(W, dtype=dtype), np.asarray(b, dtype=dtype)) for W, b in layers]


def gnn_layer(h, e, x, q, mask, msg, upd, dtype=np.float32, row_block=64):
    """reference charge_gn.py:56-75 (GNN_layer.call).  h (B,N,H) e (B,N,N,E) x (B,N,nx) q (B,N,1) mask (B,N,N,1)."""
    h, x, q, mask = (np.asarray(a, dtype=dtype) for a in (h, x, q, mask))
    B, N = e.shape[0], e.shape[1]
    node_mask = np.clip(mask.sum(axis=1), 0, 1)                         # :59  (B,N,1)
    upd = _cast_layers(upd, dtype)
    for layers in msg:                                                  # :60
        layers = _cast_layers(layers, dtype)
        a = np.concatenate([x, h, q], axis=-1)                          # :62
        messages = np.empty((B, N, layers[-1][0].shape[1]), dtype=dtype)
        for i0 in range(0, N, row_block):
            ai = a[:, i0:i0 + row_block]
            nb = ai.shape[1]
            inp_i = np.broadcast_to(ai[:, :, None, :], (B, nb, N, a.shape[-1]))      # :63
            inp_j = np.broadcast_to(a[:, None, :, :], (B, nb, N, a.shape[-1]))       # :64
            inp_ij = np.concatenate([inp_i, inp_j, _e_rows(e, i0, i0 + row_block, dtype)[1]], axis=-1)  # :65
            pm = mlp(inp_ij.reshape(-1, inp_ij.shape[-1]), layers)                    # :66-68
            messages[:, i0:i0 + row_block] = pm.reshape(B, nb, N, -1).sum(axis=2)     # :69-70 (ALL j)
        upd_in = np.concatenate([h, messages], axis=2) * node_mask      # :71-72
        h = mlp(upd_in.reshape(B * N, -1), upd).reshape(B, N, -1) * node_mask   # :73-74
    return h


def epn_layer(h, e, x, q, mask, pas, dtype=np.float32, row_block=64, return_transfer=False):
    """reference charge_gn.py:87-119 (EPN_layer.call)."""
    tol = np.float32(1e-5)
    h, x, q, mask = (np.asarray(a, dtype=dtype) for a in (h, x, q, mask))
    B, N = e.shape[0], e.shape[1]
    pad = mask.max(axis=-1)                                             # :116 reduce_max(mask, -1)
    transfers = []
    for layers in pas:                                                  # :98
        layers = _cast_layers(layers, dtype)
        a = np.concatenate([x, h, q], axis=-1)                          # :101
        anti = np.empty((B, N, N), dtype=dtype)
        for i0 in range(0, N, row_block):
            ai = a[:, i0:i0 + row_block]
            nb = ai.shape[1]
            inp_i = np.broadcast_to(ai[:, :, None, :], (B, nb, N, a.shape[-1]))
            inp_j = np.broadcast_to(a[:, None, :, :], (B, nb, N, a.shape[-1]))
            e32, eb = _e_rows(e, i0, i0 + row_block, dtype)
            largest = np.clip(e32, tol, np.float32(1e5)).max(axis=-1)   # :90-92 (float32 compare)
            is_near = (largest != tol).astype(dtype)                    # :93-94
            f_ij = mlp(np.concatenate([inp_i, inp_j, eb], axis=-1).reshape(-1, 2 * a.shape[-1] + eb.shape[-1]),
                       layers).reshape(B, nb, N)                        # :104,107,110,113
            f_ji = mlp(np.concatenate([inp_j, inp_i, eb], axis=-1).reshape(-1, 2 * a.shape[-1] + eb.shape[-1]),
                       layers).reshape(B, nb, N)                        # :105,108,111,114
            anti[:, i0:i0 + row_block] = (dtype(0.5) * (f_ij - f_ji) * pad[:, i0:i0 + row_block]
                                          * is_near)                             # :116
        q = q + anti.sum(axis=2)[..., None]                             # :118
        transfers.append(anti)
    if return_transfer:
        return q, transfers
    return q


def model_reduce(h_inp, x_inp, q_inp, mask_inp, dtype=np.float32):
    """reference charge_gn.py:382-384: per-atom h, x, q = sum over axis 1 / sum of mask over axis 1 (0/0 -> 0)."""
    mask_inp = np.asarray(mask_inp, dtype=dtype)
    if mask_inp.ndim == 3:
        mask_inp = mask_inp[..., None]                                  # Keras expands the rank-3 mask
    den = mask_inp.sum(axis=1)

    def red(t):
        s = np.asarray(t, dtype=dtype).sum(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(den != 0, s / np.where(den != 0, den, 1), 0).astype(dtype)

    return red(h_inp), red(x_inp), red(q_inp), mask_inp


def model_forward(h_inp, e_inp, x_inp, q_inp, mask_inp, weights, dtype=np.float32, row_block=64):
    """reference charge_gn.py:369-391 (make_model graph): model([h, e, x, q, mask]) -> (B,N,1).

    Inputs are first cast to float32 like Keras does for float64 arrays fed to float32 Inputs."""
    f32 = [np.asarray(t, dtype=np.float32) for t in (h_inp, e_inp, x_inp, q_inp, mask_inp)]
    h_inp, e_inp, x_inp, q_inp, mask_inp = f32
    h, x, q, mask = model_reduce(h_inp, x_inp, q_inp, mask_inp, dtype)
    feats = gnn_layer(h, e_inp, x, q, mask, weights["msg"], weights["upd"], dtype, row_block)   # :386
    return epn_layer(feats, e_inp, x, q, mask, weights["pas"], dtype, row_block)                 # :387


def forward_xyz(xyz, x, Q, weights, N=None, dtype=np.float32, row_block=64, cutoff=3.0, eta=2.0, h_dim=48):
    """Featurise one molecule like gen_padded_init_state and run the model; returns (N,) charges.  (h_dim: the channels of h and of
    e -- make_model gives both h_dim, charge_gn.py:376-377; infer.py:42-43 uses 48.)"""
    n = x.shape[0]
    N = n if N is None else N
    h_p, e_p, x_p, q_p, mask = dense_inputs(xyz, x, Q, N, h_dim=h_dim, e_dim=h_dim, cutoff=cutoff, eta=eta)
    out = model_forward(h_p[None], e_p[None], x_p[None], q_p[None], mask[None], weights, dtype, row_block)
    return out[0, :, 0]


def forward_xyz_large(xyz, x, Q, weights, dtype=np.float64, row_block=64, cutoff=3.0, eta=2.0):
    """forward_xyz for one UNPADDED system (N = n) too large for its dense (n,n,.) inputs: the per-atom x, h = 0 and
    q = Q/n are what make_model's reductions (:382-384) return for gen_padded_init_state's tiled arrays (:335-338), the
    mask is all ones, and the edge tensor is produced a block of rows at a time.  Same layer functions, same operations
    per block; equality with forward_xyz on small systems is a CPU test."""
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[0]
    h = np.zeros((1, n, 48), dtype=dtype)
    q = np.full((1, n, 1), np.float32(np.float32(Q) / np.float32(n)), dtype=dtype)     # :337 float32 / int
    mask = np.ones((1, n, 1, 1), dtype=dtype)              # sum over axis 1 / max over the last axis are all it is used for
    e = EdgeRows(xyz, 48, cutoff, eta)
    xx = x[None].astype(dtype)
    feats = gnn_layer(h, e, xx, q, mask, weights["msg"], weights["upd"], dtype, row_block)
    return epn_layer(feats, e, xx, q, mask, weights["pas"], dtype, row_block)[0, :, 0]
