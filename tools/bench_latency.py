#!/usr/bin/env python3
"""Latency of ONE blocking forward (the reference's infer.py pattern: one model call per molecule, infer.py:62-76) by
molecule size, and of a blocking forward of a whole batch: host arrays in, host charges out, and device-resident.
    python tools/bench_latency.py
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Engine


def timed(fn, reps):
    for _ in range(5):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    w = checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights"))
    eng = Engine(nx=9, T=5, device=0)
    eng.set_weights(w)
    for kv in sys.argv[1:]:
        name, value = kv.split("=")
        eng.set_option(name, int(value))
    offsets, xyz, x, Q, N = synth.qm9_like_batch(B=1024, seed=0, N=29)
    ns = np.diff(offsets)
    print("one molecule per call (N = 29):  n   host->host ms   device-resident ms   kernel ms")
    for n in (7, 12, 16, 17, 18, 20, 23, 25, 29):
        b = int(np.argmax(ns == n)) if (ns == n).any() else None
        if b is None:
            continue
        sl = slice(offsets[b], offsets[b + 1])
        off1 = np.array([0, n], np.int32)
        a = (xyz[sl].copy(), x[sl].copy(), Q[b:b + 1].copy())
        d = [eng.to_device(v) for v in a]
        dq = eng.alloc(n * 4)
        t_h = timed(lambda: eng.forward_xyz(off1, a[0], a[1], a[2], N), 200)

        def dev():
            eng.forward_xyz_dev(off1, d[0], d[1], d[2], dq, N)
            eng.sync()
        t_d = timed(dev, 200)
        eng.set_option("profile", 8)
        for _ in range(8):
            dev()
        k = np.mean([eng.timing_at(i)[3] for i in range(8)])
        eng.set_option("profile", 0)
        print(f"                                {n:3d}   {t_h:10.3f}   {t_d:14.3f}   {k:12.3f}", flush=True)
    print("one blocking call per batch of B molecules: B   host->host ms   device-resident ms   M atoms/s (device-resident)")
    for B in (1, 16, 64, 256, 1024, 4096):
        off, bx, bz, bQ, _ = synth.qm9_like_batch(B=B, seed=3, N=29)
        d = [eng.to_device(v) for v in (bx, bz, bQ)]
        dq = eng.alloc(int(off[-1]) * 4)
        t_h = timed(lambda: eng.forward_xyz(off, bx, bz, bQ, N), 50)

        def dev():
            eng.forward_xyz_dev(off, d[0], d[1], d[2], dq, N)
            eng.sync()
        t_d = timed(dev, 50)
        print(f"                                        {B:5d}   {t_h:10.3f}   {t_d:14.3f}   {off[-1] / t_d / 1e3:10.1f}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
