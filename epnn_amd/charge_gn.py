"""Drop-in mirror of the reference's ``charge_gn.py`` layer / model API, backed by the MI355X HIP library.

Same names, argument meaning and error behaviour as the reference (derekmetcalf/epnn ``charge_gn.py``):
``MLP_layer`` (:30-45), ``GNN_layer`` (:47-75), ``EPN_layer`` (:77-119), ``get_init_edges`` (:122-163),
``gen_padded_init_state`` (:292-366), ``make_model`` (:369-391).  No TensorFlow: layers hold their weights as
NumPy arrays (Keras layout, Glorot-uniform kernels / zero biases like ``Dense`` defaults) and every ``call`` goes
through ``libepnn_hip.so`` (ctypes, ``include/epnn.h``).  There is no CPU fallback for the layer arithmetic; the
directory reader and ``get_init_edges`` are host code in the reference too and stay NumPy here.

Differences from the reference, all deliberate:
  * ``gen_padded_init_state`` takes an optional ``n_elems`` (9 -> the 8-element table of ``infer.py:13-30`` that the
    shipped ``decay_model_weights`` needs; 10 -> the 9-element table of ``charge_gn.py:9-28``) and lists the
    directory in sorted order (the reference uses ``os.listdir`` order, which depends on the file system).
  * the model object also offers ``predict_xyz`` -- the compact entry that never builds the (N,N,.) tensors.
"""
from __future__ import annotations

import os

import numpy as np

from . import checkpoint
from .engine import Engine, EpnnError

atom_num_dict = {'H': 1, 'C': 6, 'N': 7, 'O': 8, 'F': 9, 'P': 15, 'S': 16, 'Cl': 17, 'Br': 35}
elem_dict = {'H': 0, 'C': 1, 'N': 2, 'O': 3, 'F': 4, 'P': 5, 'S': 6, 'Cl': 7, 'Br': 8}
# infer.py:13-30
atom_num_dict_8 = {'H': 1, 'C': 6, 'N': 7, 'O': 8, 'F': 9, 'S': 16, 'Cl': 17, 'Br': 35}
elem_dict_8 = {'H': 0, 'C': 1, 'N': 2, 'O': 3, 'F': 4, 'S': 5, 'Cl': 6, 'Br': 7}

_DEVICE = int(os.environ.get("LOCAL_RANK", "0"))
_rng = np.random.default_rng(0)


def _tables(n_elems):
    if n_elems is None:
        return atom_num_dict, elem_dict
    if n_elems == 9:
        return atom_num_dict_8, elem_dict_8
    if n_elems == 10:
        return atom_num_dict, elem_dict
    raise ValueError("n_elems must be 9 (infer.py table) or 10 (charge_gn.py table)")


class _Dense:
    """Weights of one Keras Dense: kernel [in,out] Glorot uniform, bias zeros; built on first use."""

    def __init__(self, units, activation):
        self.units = units
        self.activation = activation
        self.kernel = None
        self.bias = None

    def build(self, n_in):
        if self.kernel is None:
            lim = np.sqrt(6.0 / (n_in + self.units))
            self.kernel = _rng.uniform(-lim, lim, (n_in, self.units)).astype(np.float32)
            self.bias = np.zeros((self.units,), dtype=np.float32)
        elif self.kernel.shape[0] != n_in:
            raise ValueError(f"Dense expected {self.kernel.shape[0]} input features, got {n_in}")


class MLP_layer:
    """charge_gn.py:30-45: Dense(n, activation) for n in nodes, then Dense(out_dim, None)."""

    ACTIVATIONS = ('relu', None, 'linear', 'tanh', 'sigmoid')

    def __init__(self, nodes, out_dim=1, activation='relu'):
        # charge_gn.py:31,38 hands `activation` to keras.layers.Dense.  As a stand-alone operator (`.call`) the Keras names above are
        # built; inside GNN_layer / EPN_layer / make_model only 'relu' is -- the reference's own stacks never pass anything else
        # (charge_gn.py:52,84,371) and the fused kernels are built around the ReLU (_push refuses the rest).
        if activation not in self.ACTIVATIONS:
            raise ValueError(f"activation {activation!r} is not built: one of {self.ACTIVATIONS}")
        self.nodes = list(nodes)
        self.out_dim = out_dim
        self.activation = activation
        self.layer_set = [_Dense(n, activation) for n in self.nodes] + [_Dense(out_dim, None)]

    def build(self, n_in):
        for layer in self.layer_set:
            layer.build(n_in)
            n_in = layer.units

    def get_weights(self):
        return [(l.kernel, l.bias) for l in self.layer_set]

    def set_weights(self, pairs):
        for l, (k, b) in zip(self.layer_set, pairs):
            l.kernel = np.ascontiguousarray(k, dtype=np.float32)
            l.bias = np.ascontiguousarray(b, dtype=np.float32)

    def call(self, x):
        """Row-wise MLP on the GPU (epnn_mlp_forward)."""
        x = np.asarray(x, dtype=np.float32)
        lead = x.shape[:-1]
        self.build(x.shape[-1])
        eng = _scratch_engine()
        if self.nodes == [32, 32] and self.activation == 'relu':
            out = eng.mlp_forward(x.reshape(-1, x.shape[-1]), self.get_weights())           # the matrix-pipe kernel
        else:                                                                               # any widths / activation (generic Dense stack)
            out = eng.mlp_forward_layers(x.reshape(-1, x.shape[-1]), self.get_weights(), self.activation)
        return out.reshape(lead + (self.out_dim,))

    __call__ = call


_SCRATCH = {}


def _scratch_engine():
    if "e" not in _SCRATCH:
        _SCRATCH["e"] = Engine(nx=9, T=1, device=_DEVICE)
    return _SCRATCH["e"]


class _Stack:
    """Shared by GNN_layer / EPN_layer / the model: owns one Engine and pushes layer weights to it lazily."""

    def __init__(self):
        self._engine = None
        self._pushed = None

    def _mlps(self):
        raise NotImplementedError

    def _engine_for(self, nx, T, h_dim=48):
        """h_dim: the channels of h and e (make_model gives e_inp h_dim channels, charge_gn.py:376-377); 1..48 are built -- below
        48 the library runs the model zero-padded to 48 channels (exact; include/epnn.h)."""
        if not 1 <= int(h_dim) <= 48:
            raise EpnnError(f"h_dim = {h_dim}: the HIP kernels hold 48 channels of h and e (h_dim = e_dim in 1..48 are built)")
        if self._engine is None or self._engine.nx != nx or self._engine.T != T or self._engine.h_dim != h_dim:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(nx=nx, T=T, h_dim=int(h_dim), e_dim=int(h_dim), device=_DEVICE)
            self._pushed = None
        return self._engine


def _push(eng, msg, upd, pas, nx):
    hd = eng.h_dim
    F = nx + hd + 1
    for m in list(msg or []) + ([upd] if upd is not None else []) + list(pas or []):
        if getattr(m, "activation", "relu") != "relu":
            raise EpnnError(f"an MLP_layer with activation={m.activation!r} inside GNN_layer / EPN_layer / make_model: the stacks are "
                            "built for 'relu' (the reference's own, charge_gn.py:52,84,371); other activations run in MLP_layer.call only")
    for t, m in enumerate(msg or []):
        m.build(2 * F + hd)
        for l, (k, b) in enumerate(m.get_weights()):
            eng.set_layer("msg", t, l, k, b)
    if upd is not None:
        upd.build(hd + 32)
        eng.set_update_layers(upd.nodes)          # make_model(layers, ...): any hidden widths ([32, 32]: the tuned kernels)
        for l, (k, b) in enumerate(upd.get_weights()):
            eng.set_layer("upd", 0, l, k, b)
    for t, m in enumerate(pas or []):
        m.build(2 * F + hd)
        for l, (k, b) in enumerate(m.get_weights()):
            eng.set_layer("pas", t, l, k, b)


class GNN_layer(_Stack):
    """charge_gn.py:47-75.  ``message_fn`` is the MLP class (instantiated T times with ([32,32], out_dim=32)),
    ``update_fn`` an MLP instance."""

    def __init__(self, message_fn, update_fn, T):
        super().__init__()
        self.message_fns = [message_fn([32, 32], out_dim=32) for _ in range(T)]
        self.update_fn = update_fn
        self.T = T

    def call(self, h, e, x, q, mask):
        x = np.asarray(x)
        hd = int(np.shape(h)[-1])
        if self.update_fn.out_dim != hd:
            raise EpnnError(f"GNN_layer: update_fn.out_dim must equal h's channels ({hd}: h is fed back every step, charge_gn.py:71-73), got {self.update_fn.out_dim}")
        eng = self._engine_for(x.shape[-1], self.T, hd)
        _push(eng, self.message_fns, self.update_fn, None, x.shape[-1])
        return eng.gnn_forward(h, e, x, q, mask)

    __call__ = call


class EPN_layer(_Stack):
    """charge_gn.py:77-119: charge-conserving electron passing."""

    def __init__(self, pass_fn, T=1):
        super().__init__()
        self.pass_fns = [pass_fn([32, 32]) for _ in range(T)]
        self.T = T

    def call(self, h, e, x, q, mask):
        x = np.asarray(x)
        eng = self._engine_for(x.shape[-1], self.T, int(np.shape(h)[-1]))
        _push(eng, None, None, self.pass_fns, x.shape[-1])
        return eng.epn_forward(h, e, x, q, mask)

    __call__ = call


# --------------------------------------------------------------------------------------------- featurisation
def get_init_edges(xyz, molecular_splits, num=32, cutoff=3.0, eta=2.0):
    """charge_gn.py:122-163.  Returns (e float32 (n,n,num), C float64 (n,n,num)).  ``molecular_splits`` is accepted
    and, as in the reference, does not influence the result (adj is computed there and never used); a 1-D non-empty
    array makes the reference print and exit(), which is reported here as an error instead."""
    molecular_splits = np.asarray(molecular_splits)
    if molecular_splits.ndim == 1 and molecular_splits.shape != (0,):
        raise ValueError("get_init_edges: the reference calls exit() for 1-D non-empty molecular_splits (charge_gn.py:134-145)")
    xyz = np.asarray(xyz)
    if xyz.ndim != 2 or xyz.shape[1] != 3 or xyz.shape[0] < 1:
        raise ValueError(f"get_init_edges: xyz must be (n, 3), got {xyz.shape}")
    # distances in float64 from the float32-rounded coordinates, cosine cutoff, Gaussians, float32 cast: k_edges_dense
    e, C = _scratch_engine().edges_ex(xyz, num, cutoff, eta)
    return e, np.tile(C[:, :, None], [1, 1, num])


def read_xyz(filename, n_elems=None):
    """The parsing part of charge_gn.py:309-330: (xyz float32 (n,3), x float32 (n,nx), Q float32 0-d, n_lines)."""
    atom_num, elem = _tables(n_elems)
    with open(filename, 'r') as f:
        lines = f.readlines()
    Q = np.array(lines[1].strip().split()[0], dtype=np.float32)
    xyz, this_x = [], []
    for line in lines[2:]:
        data = line.split()
        xyz.append([data[1], data[2], data[3]])
        ohe = np.zeros(len(elem) + 1)
        ohe[0] = atom_num[data[0]]
        ohe[elem[data[0]] + 1] = 1
        this_x.append(ohe)
    return np.array(xyz, dtype=np.float32), np.array(this_x, dtype=np.float32), Q, len(lines)


def gen_padded_init_state(path, h_dim, e_dim, n_elems=None):
    """charge_gn.py:292-366: read every .xyz under ``path`` (string-concatenated, so it must end in '/'), build the
    five dense tensors padded to the largest system.  Returns x, h, q, e, Q, y, mask, names with the reference's
    shapes and dtypes (float64 arrays holding float32 values; Q a list of 0-d float32 arrays)."""
    x, h, q, Q, e, y, names = [], [], [], [], [], [], []
    for filename in sorted(os.listdir(path if path else '.')):
        if not filename.endswith(".xyz"):
            continue
        xyz, this_x, Qi, nlines = read_xyz(path + filename, n_elems)
        label_file = path + filename[:-4] + '.npy'
        if os.path.exists(label_file):
            y.append(np.array(np.load(label_file), dtype=np.float32))
        else:
            print('No labels provided, y set to 0')
            y.append(np.zeros(nlines - 2))
        Q.append(Qi)
        names.append(filename[:-4])
        these_edges, _ = get_init_edges(xyz, np.array([]), num=e_dim)
        e.append(these_edges)
        x.append(this_x)
        h.append(np.zeros((this_x.shape[0], h_dim), dtype=np.float32))
        avg_q = Qi / len(this_x)
        q.append(np.array(np.ones((len(this_x), 1)) * avg_q, dtype=np.float32))
    if not names:
        raise ValueError(f"no .xyz files under {path!r}")
    B = len(Q)
    largest_system = int(np.max([yy.shape[0] for yy in y]))
    N = largest_system
    x_padded = np.zeros((B, N, N, x[0].shape[1]))
    h_padded = np.zeros((B, N, N, h_dim))
    q_padded = np.zeros((B, N, N, 1))
    e_padded = np.zeros((B, N, N, e_dim))
    y_padded = np.zeros((B, N, 1))
    mask = np.zeros((B, N, N))
    for i in range(B):
        n = x[i].shape[0]
        y_padded[i, :y[i].shape[0], 0] = y[i].reshape(-1)
        x_padded[i, :n, :n] = x[i][None, :, :]      # row j*n+k of the tiled per-atom array is atom k (:335,:360)
        h_padded[i, :n, :n] = h[i][None, :, :]
        q_padded[i, :n, :n] = q[i][None, :, :]
        e_padded[i, :n, :n] = e[i]
        mask[i, :n, :n] = 1
    return x_padded, h_padded, q_padded, e_padded, Q, y_padded, mask, np.array(names)


# --------------------------------------------------------------------------------------------- model
class EPNNModel(_Stack):
    """What ``make_model`` returns: callable on ``[h_inp, e_inp, x_inp, q_inp, mask_inp]`` like the Keras model."""

    def __init__(self, layers, h_dim, T, n_elems, natom):
        super().__init__()
        if not 1 <= int(h_dim) <= 48:
            raise EpnnError("make_model: h_dim must be in 1..48 (the HIP kernels hold 48 channels of h and e; a smaller model runs "
                            "zero-padded, exactly; the reference's own scripts use 48, charge_gn.py:413, infer.py:47)")
        self.h_dim, self.T, self.n_elems, self.natom = h_dim, T, n_elems, natom
        self.update_fn = MLP_layer(layers, out_dim=h_dim)
        self.graph_net = GNN_layer(MLP_layer, self.update_fn, T)
        self.electron_net = EPN_layer(MLP_layer, T=T)
        if not 1 <= len(list(layers)) <= 7 or any(not 1 <= int(w) <= 256 for w in layers):
            raise EpnnError("make_model: layers must be 1..7 hidden widths of 1..256 units (charge_gn.py:371; [32, 32] is the "
                            "reference's own choice and the shape the tuned kernels are built for)")
        F = n_elems + h_dim + 1
        self.update_fn.build(h_dim + 32)
        for m in self.graph_net.message_fns + self.electron_net.pass_fns:
            m.build(2 * F + h_dim)
        self._graph_bytes = None
        self._dirty = True
        self._training = False

    # ---- weights
    def _eng(self):
        eng = self._engine_for(self.n_elems, self.T, self.h_dim)
        if self._dirty or self._pushed is None:
            _push(eng, self.graph_net.message_fns, self.update_fn, self.electron_net.pass_fns, self.n_elems)
            self._pushed = True
            self._dirty = False
        return eng

    def weights_dict(self):
        _refresh_from_device(self)
        return {"msg": [m.get_weights() for m in self.graph_net.message_fns],
                "upd": self.update_fn.get_weights(),
                "pas": [m.get_weights() for m in self.electron_net.pass_fns]}

    def set_weights_dict(self, w):
        if len(w["msg"]) != self.T:
            raise EpnnError(f"weights have T={len(w['msg'])}, model has T={self.T}")
        first = w["msg"][0][0][0].shape[0]
        want = 2 * (self.n_elems + self.h_dim + 1) + self.h_dim
        if first != want:
            raise EpnnError(f"weights were trained with {(first - self.h_dim) // 2 - self.h_dim - 1} atom-feature columns (at h_dim = {self.h_dim}), "
                            f"model was built with n_elems={self.n_elems} (first kernel {first} rows, expected {want})")
        for t in range(self.T):
            self.graph_net.message_fns[t].set_weights(w["msg"][t])
            self.electron_net.pass_fns[t].set_weights(w["pas"][t])
        self.update_fn.set_weights(w["upd"])
        self._dirty = True
        self._training = False
        self._weights_version = getattr(self, "_weights_version", 0) + 1

    def load_weights(self, prefix):
        """infer.py:57 -- reads the TensorFlow tensor-bundle files directly."""
        self.set_weights_dict(checkpoint.load_epnn_weights(prefix))
        self._graph_bytes = checkpoint.read_object_graph(prefix)

    def save_weights(self, prefix):
        """charge_gn.py:462 -- writes ``<prefix>.index`` / ``<prefix>.data-00000-of-00001``."""
        checkpoint.save_epnn_weights(prefix, self.weights_dict(), self._graph_bytes)

    @property
    def trainable_variables(self):
        _refresh_from_device(self)
        return self._trainable_variables()

    def _trainable_variables(self):
        """Keras creation order (charge_gn.py:371-374): update MLP, message MLPs t=0.., pass MLPs t=0..;
        kernel then bias per Dense."""
        out = []
        for m in [self.update_fn] + self.graph_net.message_fns + self.electron_net.pass_fns:
            for k, b in m.get_weights():
                out += [k, b]
        return out

    # ---- inference
    def __call__(self, inputs, training=False):
        h_inp, e_inp, x_inp, q_inp, mask_inp = inputs
        e_inp = np.asarray(e_inp)
        if e_inp.ndim != 4 or e_inp.shape[1] != self.natom or e_inp.shape[2] != self.natom:
            raise ValueError(f"model was built for natom={self.natom}; got e_inp of shape {e_inp.shape}")
        return self._eng().model_forward_dense(h_inp, e_inp, x_inp, q_inp, mask_inp)

    predict = __call__

    def predict_xyz(self, offsets, xyz, x, Q, N=None):
        """Compact entry: flat atom arrays instead of dense tensors; N defaults to the model's natom."""
        return self._eng().forward_xyz(offsets, xyz, x, Q, self.natom if N is None else N)

    def predict_xyz_stream(self, batches, N=None, depth=8):
        """Charges of every (offsets, xyz, x, Q) batch of `batches`, in order, with `depth` batches in flight on the GPU
        (engine.Pipeline.map: the loop of infer.py:62-76 at the throughput of the compact entry)."""
        from .engine import Pipeline
        pipe = Pipeline(depth=depth, nx=self.n_elems, T=self.T, h_dim=self.h_dim, e_dim=self.h_dim, device=_DEVICE)
        try:
            pipe.set_weights(self.weights_dict())
            yield from pipe.map(batches, self.natom if N is None else N)
        finally:
            pipe.close()

    def engine(self):
        return self._eng()


def make_model(layers, h_dim, T, n_elems, natom):
    """charge_gn.py:369-391."""
    return EPNNModel(layers, h_dim, T, n_elems, natom)


def test_step(model, h, e, x, q, y, mask):
    """infer.py:32-35 (the reference closes over a global ``model``; here it is an argument)."""
    return model([h, e, x, q, mask])


test_step.__test__ = False   # not a pytest test


# --------------------------------------------------------------------------------------------- training
class Adam:
    """tf.keras.optimizers.Adam() with the Keras-2 defaults the reference uses (charge_gn.py:419)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self._bound = None

    def bind(self, model):
        """Copy the model's weights to the device as training masters and zero the moments (once per model)."""
        key = (id(model), getattr(model, "_weights_version", 0))
        if self._bound != key:
            model.engine().train_init(self.learning_rate, self.beta_1, self.beta_2, self.epsilon)
            self._bound = key
        model._training = True
        return model.engine()


class Mean:
    """tf.keras.metrics.Mean (charge_gn.py:420,422)."""

    def __init__(self, name="mean"):
        self.name = name
        self.reset_states()

    def reset_states(self):
        self._sum, self._n = 0.0, 0

    def __call__(self, values):
        v = np.asarray(values, dtype=np.float64)
        self._sum += float(v.sum())
        self._n += v.size

    def result(self):
        return self._sum / self._n if self._n else 0.0


class MeanAbsoluteError(Mean):
    """tf.keras.metrics.MeanAbsoluteError (charge_gn.py:421,423); averages over all N padded slots like Keras."""

    def __call__(self, a, b):
        d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
        self._sum += float(d.sum())
        self._n += d.size


def train_step(model, optimizer, h, e, x, q, y, mask, train_loss=None, train_acc=None):
    """charge_gn.py:393-402 with the globals made arguments: forward, loss = MSE over the size-1 last axis (a (B,N)
    tensor of (y-p)^2), gradient of its SUM, one Adam step; updates the metrics the reference updates."""
    eng = optimizer.bind(model)
    y = np.asarray(y)
    pred, _ = eng.train_step_dense(h, e, x, q, mask, y, apply=True)
    if train_loss is not None:
        train_loss((y.reshape(pred.shape) - pred) ** 2)
    if train_acc is not None:
        train_acc(pred, y.reshape(pred.shape))
    return pred


def _refresh_from_device(model):
    """After training steps the current weights live on the device: pull them into the Python layers."""
    if getattr(model, "_training", False):
        w = model.engine().get_weights()
        for t in range(model.T):
            model.graph_net.message_fns[t].set_weights(w["msg"][t])
            model.electron_net.pass_fns[t].set_weights(w["pas"][t])
        model.update_fn.set_weights(w["upd"])
