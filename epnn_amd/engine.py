"""Thin object wrapper over the C ABI: one Engine == one epnn_handle == one GPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import W_MSG, W_PAS, W_UPD, EpnnConfig, EpnnError, check, fptr, iptr

_WHICH = {"msg": W_MSG, "upd": W_UPD, "pas": W_PAS}


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class DeviceArray:
    """A device allocation owned by an Engine (used by bench.py to keep inputs resident in HBM)."""

    def __init__(self, eng, nbytes):
        self.eng = eng
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(eng.lib.epnn_dev_alloc(eng.h, self.nbytes, C.byref(p)), eng.lib)
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self.eng.lib.epnn_memcpy_h2d(self.eng.h, self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes), self.eng.lib)
        return self

    def download(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.eng.lib.epnn_memcpy_d2h(self.eng.h, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes), self.eng.lib)
        return out

    def free(self):
        if self.ptr is not None and self.eng.h:
            self.eng.lib.epnn_dev_free(self.eng.h, self.ptr)
        self.ptr = None


class Engine:
    def __init__(self, nx=9, h_dim=48, e_dim=48, T=5, hidden=32, cutoff=3.0, eta=2.0, near_tol=1e-5, device=0):
        self.lib = _lib.load()
        self.cfg = EpnnConfig(nx, h_dim, e_dim, T, hidden, cutoff, eta, near_tol)
        self.nx, self.h_dim, self.e_dim, self.T = nx, h_dim, e_dim, T
        self.device = device
        h = C.c_void_p()
        check(self.lib.epnn_create(C.byref(self.cfg), device, C.byref(h)), self.lib)
        self.h = h
        self._upd_layers = [hidden, hidden]              # hidden widths of the update MLP (set_update_layers)

    def close(self):
        if getattr(self, "h", None):
            self.lib.epnn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def weight_shape(self, group, t, layer):
        a, b = C.c_int32(), C.c_int32()
        check(self.lib.epnn_weight_shape(self.h, _WHICH[group], t, layer, C.byref(a), C.byref(b)), self.lib)
        return a.value, b.value

    def set_layer(self, group, t, layer, kernel, bias):
        shp = self.weight_shape(group, t, layer)
        kernel, bias = _f32(kernel), _f32(bias)
        if kernel.shape != shp or bias.shape != (shp[1],):
            raise EpnnError(f"{group}[{t}] layer {layer}: expected kernel {shp} / bias ({shp[1]},), "
                            f"got {kernel.shape} / {bias.shape}")
        check(self.lib.epnn_set_weights(self.h, _WHICH[group], t, layer, fptr(kernel), fptr(bias)), self.lib)

    def get_layer(self, group, t, layer):
        shp = self.weight_shape(group, t, layer)
        k = np.empty(shp, dtype=np.float32)
        b = np.empty((shp[1],), dtype=np.float32)
        check(self.lib.epnn_get_weights(self.h, _WHICH[group], t, layer, fptr(k), fptr(b)), self.lib)
        return k, b

    def set_update_layers(self, widths):
        """make_model(layers, ...): hidden widths of the update MLP (charge_gn.py:371); [32, 32] is the default and the tuned shape."""
        widths = [int(w) for w in widths]
        if widths != self._upd_layers:
            arr = np.ascontiguousarray(widths, dtype=np.int32)
            check(self.lib.epnn_set_update_layers(self.h, len(widths), iptr(arr)), self.lib)
            self._upd_layers = widths

    def set_weights(self, weights):
        """weights = {"msg": [T][3](W,b), "upd": [n_hidden + 1](W,b), "pas": [T][3](W,b)} (checkpoint.load_epnn_weights); the update
        MLP's hidden widths follow the kernels' shapes."""
        if len(weights["msg"]) != self.T or len(weights["pas"]) != self.T:
            raise EpnnError(f"checkpoint has T={len(weights['msg'])}, engine was built with T={self.T}")
        self.set_update_layers([np.asarray(k).shape[1] for k, _ in weights["upd"][:-1]])
        for t in range(self.T):
            for l in range(3):
                self.set_layer("msg", t, l, *weights["msg"][t][l])
                self.set_layer("pas", t, l, *weights["pas"][t][l])
        for l in range(len(weights["upd"])):
            self.set_layer("upd", 0, l, *weights["upd"][l])

    def get_weights(self):
        return {"msg": [[self.get_layer("msg", t, l) for l in range(3)] for t in range(self.T)],
                "upd": [self.get_layer("upd", 0, l) for l in range(len(self._upd_layers) + 1)],
                "pas": [[self.get_layer("pas", t, l) for l in range(3)] for t in range(self.T)]}

    # ------------------------------------------------------------------ compute
    def edges(self, xyz):
        xyz = _f32(xyz)
        n = xyz.shape[0]
        out = np.empty((n, n, self.e_dim), dtype=np.float32)
        check(self.lib.epnn_edges(self.h, n, fptr(xyz), fptr(out)), self.lib)
        return out

    def edges_ex(self, xyz, num, cutoff=3.0, eta=2.0):
        """get_init_edges with its own parameters: (e float32 (n,n,num), C float64 (n,n)) from the device kernel."""
        xyz = _f32(xyz)
        n = xyz.shape[0]
        e = np.empty((n, n, int(num)), dtype=np.float32)
        c = np.empty((n, n), dtype=np.float64)
        check(self.lib.epnn_edges_ex(self.h, n, fptr(xyz), int(num), float(cutoff), float(eta), fptr(e),
                                     c.ctypes.data_as(C.POINTER(C.c_double))), self.lib)
        return e, c

    def forward_xyz(self, offsets, xyz, x, Q, N):
        """Flat batch: offsets (B+1,), xyz (A,3), x (A,nx), Q (B,) -> q (A,) float32."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        xyz, x, Q = _f32(xyz), _f32(x), _f32(Q)
        B = len(offsets) - 1
        A = int(offsets[-1])
        if xyz.shape != (A, 3) or x.shape != (A, self.nx) or Q.shape != (B,):
            raise EpnnError(f"forward_xyz: shapes xyz {xyz.shape} x {x.shape} Q {Q.shape} do not match offsets (A={A}, B={B}, nx={self.nx})")
        out = np.empty((A,), dtype=np.float32)
        check(self.lib.epnn_forward_xyz(self.h, B, int(N), iptr(offsets), fptr(xyz), fptr(x), fptr(Q), fptr(out)), self.lib)
        return out

    def forward_xyz_begin(self, offsets, xyz, x, Q, N):
        """First half of forward_xyz: returns as soon as the work is queued (the arrays may be reused at once)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        xyz, x, Q = _f32(xyz), _f32(x), _f32(Q)
        B = len(offsets) - 1
        A = int(offsets[-1])
        if xyz.shape != (A, 3) or x.shape != (A, self.nx) or Q.shape != (B,):
            raise EpnnError(f"forward_xyz: shapes xyz {xyz.shape} x {x.shape} Q {Q.shape} do not match offsets (A={A}, B={B}, nx={self.nx})")
        check(self.lib.epnn_forward_xyz_begin(self.h, B, int(N), iptr(offsets), fptr(xyz), fptr(x), fptr(Q)), self.lib)
        self._begun = A

    def forward_xyz_end(self):
        """Second half: waits for the forward begun on this engine and returns its charges q (A,) float32."""
        A = getattr(self, "_begun", None)
        if A is None:
            raise EpnnError("forward_xyz_end: no forward was begun on this engine")
        self._begun = None
        out = np.empty((A,), dtype=np.float32)
        check(self.lib.epnn_forward_xyz_end(self.h, fptr(out)), self.lib)
        return out

    def _dense_args(self, B, N, tensors, chans):
        out = []
        for t, ch in zip(tensors, chans):
            t = _f32(t)
            if t.ndim == len(ch) + 1 - 1 and ch[-1] == 1 and t.shape == (B,) + ch[:-1]:
                t = t.reshape((B,) + ch)          # rank-3 mask -> rank 4 like Keras does
            if t.shape != (B,) + ch:
                raise EpnnError(f"expected shape {(B,) + ch}, got {t.shape}")
            out.append(t)
        return out

    def model_forward_dense(self, h_inp, e_inp, x_inp, q_inp, mask_inp):
        e_inp = _f32(e_inp)
        B, N = e_inp.shape[0], e_inp.shape[1]
        h_inp, e_inp, x_inp, q_inp, mask_inp = self._dense_args(
            B, N, (h_inp, e_inp, x_inp, q_inp, mask_inp),
            ((N, N, self.h_dim), (N, N, self.e_dim), (N, N, self.nx), (N, N, 1), (N, N, 1)))
        out = np.empty((B, N, 1), dtype=np.float32)
        check(self.lib.epnn_model_forward_dense(self.h, B, N, fptr(h_inp), fptr(e_inp), fptr(x_inp), fptr(q_inp),
                                                fptr(mask_inp), fptr(out)), self.lib)
        return out

    def _layer(self, fn, h, e, x, q, mask, out_ch):
        e = _f32(e)
        B, N = e.shape[0], e.shape[1]
        h, e, x, q, mask = self._dense_args(
            B, N, (h, e, x, q, mask), ((N, self.h_dim), (N, N, self.e_dim), (N, self.nx), (N, 1), (N, N, 1)))
        out = np.empty((B, N, out_ch), dtype=np.float32)
        check(fn(self.h, B, N, fptr(h), fptr(e), fptr(x), fptr(q), fptr(mask), fptr(out)), self.lib)
        return out

    def gnn_forward(self, h, e, x, q, mask):
        return self._layer(self.lib.epnn_gnn_forward, h, e, x, q, mask, self.h_dim)

    def epn_forward(self, h, e, x, q, mask):
        return self._layer(self.lib.epnn_epn_forward, h, e, x, q, mask, 1)

    ACTIVATIONS = {"relu": 0, None: 1, "linear": 1, "tanh": 2, "sigmoid": 3}

    def mlp_forward_layers(self, rows, layers, activation="relu"):
        """MLP_layer(nodes, out_dim, activation).call for any `nodes`: rows (R, n_in) through [(W,b), ...], `activation` after all
        but the last layer ('relu', None / 'linear', 'tanh', 'sigmoid': charge_gn.py:38-39)."""
        if activation not in self.ACTIVATIONS:
            raise EpnnError(f"mlp_forward_layers: activation {activation!r} is not built (relu, None / linear, tanh, sigmoid)")
        rows = _f32(rows)
        ws = [(_f32(k), _f32(b)) for k, b in layers]
        dims = [rows.shape[1]] + [k.shape[1] for k, _ in ws]
        for l, (k, b) in enumerate(ws):
            if k.shape != (dims[l], dims[l + 1]) or b.shape != (dims[l + 1],):
                raise EpnnError(f"mlp_forward_layers: layer {l} has kernel {k.shape} / bias {b.shape}, expected {(dims[l], dims[l + 1])}")
        n = len(ws)
        FP = C.POINTER(C.c_float)
        Wp = (FP * n)(*[fptr(k) for k, _ in ws])
        bp = (FP * n)(*[fptr(b) for _, b in ws])
        darr = np.ascontiguousarray(dims, dtype=np.int32)
        out = np.empty((rows.shape[0], dims[-1]), dtype=np.float32)
        check(self.lib.epnn_mlp_forward_layers(self.h, rows.shape[0], n, iptr(darr), Wp, bp, fptr(rows), fptr(out), self.ACTIVATIONS[activation]), self.lib)
        return out

    def mlp_forward(self, rows, layers):
        """MLP_layer.call: rows (R, n_in) through [(W1,b1),(W2,b2),(W3,b3)] with hidden width 32."""
        rows = _f32(rows)
        (W1, b1), (W2, b2), (W3, b3) = [(_f32(k), _f32(b)) for k, b in layers]
        if W1.shape != (rows.shape[1], 32) or W2.shape != (32, 32) or W3.shape[0] != 32:
            raise EpnnError(f"mlp_forward: unsupported layer shapes {W1.shape} {W2.shape} {W3.shape}")
        out = np.empty((rows.shape[0], W3.shape[1]), dtype=np.float32)
        check(self.lib.epnn_mlp_forward(self.h, rows.shape[0], rows.shape[1], W3.shape[1], fptr(W1), fptr(b1), fptr(W2),
                                        fptr(b2), fptr(W3), fptr(b3), fptr(rows), fptr(out)), self.lib)
        return out

    # ------------------------------------------------------------------ training (charge_gn.py:393-402)
    def train_init(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        check(self.lib.epnn_train_init(self.h, lr, beta1, beta2, eps), self.lib)

    def param_count(self):
        n = C.c_int64()
        check(self.lib.epnn_param_count(self.h, C.byref(n)), self.lib)
        return n.value

    def train_step_dense(self, h_inp, e_inp, x_inp, q_inp, mask_inp, y, apply=True):
        """Returns (predictions (B,N,1), summed loss)."""
        e_inp = _f32(e_inp)
        B, N = e_inp.shape[0], e_inp.shape[1]
        h_inp, e_inp, x_inp, q_inp, mask_inp = self._dense_args(
            B, N, (h_inp, e_inp, x_inp, q_inp, mask_inp),
            ((N, N, self.h_dim), (N, N, self.e_dim), (N, N, self.nx), (N, N, 1), (N, N, 1)))
        y = _f32(np.asarray(y).reshape(B, N, 1))
        pred = np.empty((B, N, 1), dtype=np.float32)
        loss = C.c_float()
        check(self.lib.epnn_train_step_dense(self.h, B, N, fptr(h_inp), fptr(e_inp), fptr(x_inp), fptr(q_inp), fptr(mask_inp),
                                             fptr(y), fptr(pred), C.byref(loss), int(bool(apply))), self.lib)
        return pred, loss.value

    def train_step_xyz(self, offsets, xyz, x, Q, y, N, apply=True):
        """Flat batch; y per real atom (A,). Returns (q (A,), summed loss)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        xyz, x, Q, y = _f32(xyz), _f32(x), _f32(Q), _f32(y)
        B, A = len(offsets) - 1, int(offsets[-1])
        if xyz.shape != (A, 3) or x.shape != (A, self.nx) or Q.shape != (B,) or y.shape != (A,):
            raise EpnnError("train_step_xyz: array shapes do not match offsets")
        q = np.empty((A,), dtype=np.float32)
        loss = C.c_float()
        check(self.lib.epnn_train_step_xyz(self.h, B, int(N), iptr(offsets), fptr(xyz), fptr(x), fptr(Q), fptr(y), fptr(q),
                                           C.byref(loss), int(bool(apply))), self.lib)
        return q, loss.value

    def get_gradients(self):
        g = np.empty((self.param_count(),), dtype=np.float32)
        check(self.lib.epnn_get_gradients(self.h, fptr(g), g.size), self.lib)
        return g

    def set_gradients(self, g):
        g = _f32(g)
        check(self.lib.epnn_set_gradients(self.h, fptr(g), g.size), self.lib)

    def train_apply(self):
        check(self.lib.epnn_train_apply(self.h), self.lib)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        check(self.lib.epnn_comm_init(self.h, unique_id, rank, world), self.lib)

    def debug_pairs(self, cap):
        """(pi, pj, near weight) of the pair list the last forward built outside the fused kernel (tests)"""
        pi, pj, w = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = C.c_int64(0)
        check(self.lib.epnn_debug_pairs(self.h, iptr(pi), iptr(pj), fptr(w), cap, C.byref(n)), self.lib)
        k = min(int(n.value), cap)
        return pi[:k], pj[:k], w[:k], int(n.value)

    def comm_count(self):
        """ranks that joined this engine's RCCL communicator"""
        n = C.c_int32(0)
        check(self.lib.epnn_comm_count(self.h, C.byref(n)), self.lib)
        return int(n.value)

    def comm_allreduce(self, values, op="sum"):
        """all-reduce of a few host doubles over the engine's RCCL communicator (barrier / MAX over ranks of a driver)"""
        buf = (C.c_double * len(values))(*[float(v) for v in values])
        check(self.lib.epnn_comm_allreduce(self.h, buf, len(values), {"sum": 0, "max": 1}[op]), self.lib)
        return [float(v) for v in buf]

    @staticmethod
    def comm_unique_id():
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        check(lib.epnn_comm_unique_id(buf), lib)
        return buf.raw

    # ------------------------------------------------------------------ device-resident plumbing
    def alloc(self, nbytes):
        return DeviceArray(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceArray(self, arr.nbytes).upload(arr)

    def set_partition(self, rank, world, exchange=None):
        """One large system on `world` processes: this engine computes the all-pairs sums of its own
        rows of atoms and `exchange(d_rows_ptr, row_len, n_rows, row_lo, row_hi) -> None` completes the others' after
        every GNN step (shard.make_row_exchange builds one on torch.distributed).  With exchange=None and world > 1 the
        engine's RCCL communicator (comm_init with the same rank / world) exchanges the rows on the engine's stream -- the
        multi-GPU form, no host synchronisation inside the forward.  world = 1 switches the partition off."""
        if world > 1 and exchange is not None:

            def _cb(ctx, d_rows, row_len, n_rows, row_lo, row_hi):
                try:
                    exchange(d_rows, row_len, n_rows, row_lo, row_hi)
                    return 0
                except Exception:                           # an exception must not cross the C frames: report it here,
                    import traceback                        # the forward then fails with the library's own message
                    traceback.print_exc()
                    return 1

            self._exchange_cb = _lib.EXCHANGE_FN(_cb)       # keep the thunk alive as long as the handle uses it
            fn = C.cast(self._exchange_cb, C.c_void_p)
        else:
            self._exchange_cb = None
            fn = None
        check(self.lib.epnn_set_partition(self.h, int(rank), int(world), fn, None), self.lib)

    def copy_rows_to_host(self, d_ptr, row_len, row_lo, row_hi):
        """rows [row_lo, row_hi) of a device array [..][row_len] float32 -> NumPy (used by exchange functions)."""
        out = np.empty((row_hi - row_lo, row_len), dtype=np.float32)
        if out.size:
            check(self.lib.epnn_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(d_ptr + 4 * row_len * row_lo),
                                           out.nbytes), self.lib)
        return out

    def copy_rows_to_device(self, d_ptr, row_len, row_lo, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.size:
            check(self.lib.epnn_memcpy_h2d(self.h, C.c_void_p(d_ptr + 4 * row_len * row_lo), rows.ctypes.data_as(C.c_void_p),
                                           rows.nbytes), self.lib)

    def forward_xyz_dev(self, offsets, d_xyz, d_x, d_Q, d_q, N):
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        check(self.lib.epnn_forward_xyz_dev(self.h, len(offsets) - 1, int(N), iptr(offsets), d_xyz.ptr, d_x.ptr,
                                            d_Q.ptr, d_q.ptr), self.lib)

    def sync(self):
        check(self.lib.epnn_sync(self.h), self.lib)

    def timer_begin(self):
        check(self.lib.epnn_timer_begin(self.h), self.lib)

    def timer_end(self):
        ms = C.c_float()
        check(self.lib.epnn_timer_end(self.h, C.byref(ms)), self.lib)
        return ms.value

    def set_option(self, name, value):
        check(self.lib.epnn_set_option(self.h, name.encode(), int(value)), self.lib)

    def last_timing(self):
        out = np.zeros(4, dtype=np.float32)
        check(self.lib.epnn_last_timing(self.h, fptr(out)), self.lib)
        return out

    def timing_at(self, idx):
        out = np.zeros(4, dtype=np.float32)
        check(self.lib.epnn_timing_at(self.h, int(idx), fptr(out)), self.lib)
        return out

    def last_stats(self):
        out = np.zeros(4, dtype=np.int64)
        check(self.lib.epnn_last_stats(self.h, out.ctypes.data_as(C.POINTER(C.c_int64))), self.lib)
        return out


class Pipeline:
    """Keeps `depth` batches in flight on one GPU: `depth` handles (each with its own HIP stream and workspace) take
    the calls round robin.  The fused kernel runs one wavefront per molecule and an MI355X holds 2048 of them
    (8 per CU), so one batch of 1024 molecules fills half the machine and its largest molecules finish long after
    the smallest: the wavefronts of the next batches fill those slots.  A launch lasts as long as its largest molecule
    (~3x the mean under load), so six to ten batches in flight keep every slot busy (default 8); each needs its own hardware queue
    (GPU_MAX_HW_QUEUES, raised to 16 in _lib.load(); with the runtime's default of 4 use depth 3).
    All handles carry the same weights.  Results of call k are complete after `sync()`."""

    def __init__(self, depth=8, queue_stride=None, **engine_kwargs):
        """queue_stride: hardware queues from one lane's to the next (the HIP runtime deals streams onto its hardware queues in
        creation order).  Measured on MI355X: up to eight lanes run 3 % faster on every other queue (stride 2, the default there);
        more lanes than that need every queue (stride 1)."""
        depth = max(1, int(depth))
        if queue_stride is None:
            # every other queue only while the lanes still get a queue each: with a user-set GPU_MAX_HW_QUEUES of 4 or 8 a stride of
            # 2 would fold eight lanes onto 2 or 4 queues (kernels of streams that share a queue serialise)
            _lib.load()                                       # (sets GPU_MAX_HW_QUEUES to 16 unless the caller decided otherwise)
            try:
                queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
            except ValueError:
                queues = 4
            queue_stride = 2 if depth <= 8 and 2 * depth <= queues else 1
        self.engines = []
        self._device = int(engine_kwargs.get("device", 0))
        self._skipped = False
        for k in range(depth):
            self.engines.append(Engine(**engine_kwargs))
            if queue_stride > 1 and k + 1 < depth:
                # (the placement is defined for the first pipeline of a process: later streams land where the runtime's round robin is)
                check(self.engines[0].lib.epnn_skip_hw_queues(self._device, int(queue_stride) - 1), self.engines[0].lib)
                self._skipped = True
        self._next = 0
        if len(self.engines) > 1:
            # batches side by side fill the GPU: every molecule on one wavefront (the split over two is for a lone batch)
            self.set_option("wave2", 0)

    def set_weights(self, weights):
        for e in self.engines:
            e.set_weights(weights)

    def set_option(self, name, value):
        for e in self.engines:
            e.set_option(name, value)

    def lane(self):
        e = self.engines[self._next % len(self.engines)]
        self._next += 1
        return e

    def map(self, batches, N, workers=None):
        """Charges of every batch of `batches` (an iterable of (offsets, xyz, x, Q) host arrays), in order, with up to
        `depth` batches in flight: while the GPU works on some, the host stages the next and collects the oldest.
        workers: host threads that run the first half of a call (the batch's plan, its copy into page-locked staging and the
        queueing: ~85 us for 1024 molecules, what bounds this entry, not the GPU) while this thread collects results -- the C
        entry points release the GIL and every lane is a handle of its own.  Measured on the bench batch (MI355X box, 16 cores):
        inline 184-186 M atoms/s, one worker 209-218 M, two or more no better than one (the first halves of different lanes
        do not overlap: the HIP runtime's copies and launches serialise).  Default 1 with more than one lane; 0 = everything on
        the calling thread."""
        if workers is None:
            workers = 1 if len(self.engines) > 1 else 0
        if workers <= 0:
            yield from self._map_inline(batches, N)
            return
        from concurrent.futures import ThreadPoolExecutor
        busy = []                                   # (engine, future of its forward_xyz_begin), oldest first
        pool = ThreadPoolExecutor(max_workers=int(workers))
        try:
            for offsets, xyz, x, Q in batches:
                e = self.lane()
                if busy and busy[0][0] is e:        # the lane's previous forward is collected before it takes the next
                    e0, f0 = busy.pop(0)
                    f0.result()
                    yield e0.forward_xyz_end()
                busy.append((e, pool.submit(e.forward_xyz_begin, offsets, xyz, x, Q, N)))
            while busy:
                e0, f0 = busy.pop(0)
                f0.result()
                yield e0.forward_xyz_end()
        finally:
            for e0, f0 in busy:                     # abandoned half way (error, or the caller stopped iterating): collect what
                try:                                # is in flight so that the engines can be used again
                    f0.result()
                    e0.forward_xyz_end()
                except EpnnError:
                    pass
            pool.shutdown(wait=True)

    def _map_inline(self, batches, N):
        busy = []                                   # engines with a begun forward, oldest first
        try:
            for offsets, xyz, x, Q in batches:
                e = self.lane()
                if busy and busy[0] is e:
                    yield busy.pop(0).forward_xyz_end()
                e.forward_xyz_begin(offsets, xyz, x, Q, N)
                busy.append(e)
            while busy:
                yield busy.pop(0).forward_xyz_end()
        finally:
            for e in busy:                          # abandoned half way (error, or the caller stopped iterating): collect
                try:                                # what is in flight so that the engines can be used again
                    e.forward_xyz_end()
                except EpnnError:
                    pass

    def sync(self):
        for e in self.engines:
            e.sync()

    def close(self):
        lib = self.engines[0].lib if self.engines else None
        for e in self.engines:
            e.close()
        if self._skipped and lib is not None:
            lib.epnn_skip_hw_queues(self._device, 0)          # the placeholder streams between the lanes
            self._skipped = False
