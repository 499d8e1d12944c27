#!/usr/bin/env python3
"""Literal make_model entry (five dense (B,N,N,.) tensors resident in HBM) on the QM9-like batch: the dense
front-end is the HBM-bound part of the path (428 B per atom pair)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epnn_amd import checkpoint, synth, charge_gn
from epnn_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
offsets, xyz, x, Q, N = synth.qm9_like_batch(B=B, seed=0)
eng = Engine(nx=9, T=5); eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights")))
h_inp = np.zeros((B, N, N, 48), np.float32); e_inp = np.zeros((B, N, N, 48), np.float32)
x_inp = np.zeros((B, N, N, 9), np.float32); q_inp = np.zeros((B, N, N, 1), np.float32); m_inp = np.zeros((B, N, N, 1), np.float32)
for b in range(B):
    lo, hi = offsets[b], offsets[b + 1]; n = hi - lo
    e, _ = charge_gn.get_init_edges(xyz[lo:hi], np.array([]), num=48)
    e_inp[b, :n, :n] = e; x_inp[b, :n, :n] = x[lo:hi][None]; q_inp[b, :n, :n, 0] = Q[b] / np.float32(n); m_inp[b, :n, :n, 0] = 1
d = [eng.to_device(a) for a in (h_inp, e_inp, x_inp, q_inp, m_inp)]
out = eng.alloc(B * N * 4)
fn = eng.lib.epnn_model_forward_dense_dev
def run():
    rc = fn(eng.h, B, N, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, out.ptr)
    assert rc == 0, eng.lib.epnn_last_error()
for _ in range(3): run()
eng.sync()
steps = 10
eng.timer_begin()
for _ in range(steps): run()
ms = eng.timer_end() / steps
qd = out.download((B, N))
qc = eng.forward_xyz(offsets, xyz, x, Q, N)
worst = max(np.abs(qd[b, :offsets[b+1]-offsets[b]] - qc[offsets[b]:offsets[b+1]]).max() for b in range(B))
nbytes = sum(a.nbytes for a in (h_inp, e_inp, x_inp, q_inp, m_inp))
print(f"dense make_model entry: B={B} N={N}: {ms:.3f} ms per call; inputs {nbytes/1e6:.0f} MB -> {nbytes/ms/1e6:.0f} GB/s of input per total time; "
      f"{offsets[-1]/ms*1e3:.3e} atoms/s; max |dq| vs compact entry {worst:.2e}", flush=True)
