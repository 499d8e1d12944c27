#!/usr/bin/env python3
"""EPNN inference throughput on MI355X: atoms/sec on a QM9-sized batch (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one forward of the whole hot path (near-pair construction + T GNN steps + T EPN steps) over one
batch of 1024 synthetic QM9-like molecules padded to N=29, inputs (coordinates, atom features, total charges)
already resident in HBM, charges left in HBM.  With N > 1 ranks (torch.distributed.run, one rank per GPU) every
rank runs its own batch of 1024 molecules (weak scaling, molecules are independent: no data-path collective);
the only communication is the barrier and the MAX over ranks of the timed interval.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# before anything (torch included) initialises HIP in this process: one hardware queue per batch in flight (epnn_amd/_lib.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def _cpu_worker(args):
    """One CPU-baseline worker: the oracle on its share of the molecules, one molecule per call, for `budget_s` seconds."""
    mols, weights, N, budget_s = args
    from oracle import epnn_oracle as orc
    orc.forward_xyz(*mols[0], weights, N=N, dtype=np.float32)           # first call outside the clock (imports, BLAS start-up)
    atoms = calls = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        xyz, x, Q = mols[calls % len(mols)]
        orc.forward_xyz(xyz, x, Q, weights, N=N, dtype=np.float32)
        atoms += x.shape[0]
        calls += 1
    return atoms, calls, time.perf_counter() - t0


def cpu_baseline(offsets, xyz, x, Q, N, weights, budget_s=12.0):
    """The oracle (literal dense float32 restatement of charge_gn.py, one molecule per call like infer.py:62-76) timed on
    this host on a bounded sample of the same workload: `workers` single-threaded processes (spawned: nothing of this
    process's GPU state is inherited), each cycling through its share of the batch for ~12 s."""
    import multiprocessing as mp
    B = len(offsets) - 1
    workers = max(1, min(16, os.cpu_count() or 1))
    shares = [[(xyz[offsets[b]:offsets[b + 1]], x[offsets[b]:offsets[b + 1]], Q[b]) for b in range(w, B, workers)]
              for w in range(workers)]
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"                                             # inherited by the spawned workers
    try:
        with mp.get_context("spawn").Pool(workers) as pool:
            res = pool.map(_cpu_worker, [(sh, weights, N, budget_s) for sh in shares])
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    atoms = sum(r[0] for r in res)
    calls = sum(r[1] for r in res)
    dt = max(r[2] for r in res)
    return {"value": atoms / dt, "unit": "atoms/s", "cores": workers, "kind": "port",
            "sample": f"{calls} molecule calls ({atoms} atoms) of the same batch, padded to N={N}, one molecule per call like "
                      f"infer.py, NumPy float32, {workers} single-threaded worker processes, {dt:.1f} s each"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)    # 4000 x 0.09 ms = 0.35 s: the fill / drain of the six-deep pipeline
    ap.add_argument("--warmup", type=int, default=24)     # (~0.6 ms each) and the barrier do not weigh on the rate
    ap.add_argument("--molecules", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=6, help="batches in flight per GPU (handles/streams used round robin)")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (developer switch)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    dist = None
    torch = None
    device = local_rank
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        if ndev >= world:
            torch.cuda.set_device(local_rank)
            try:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))       # RCCL on ROCm
                sync_dev = torch.device("cuda", local_rank)
            except Exception:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
                sync_dev = torch.device("cpu")
        else:
            # fewer GPUs than ranks (rehearsal on a one-GPU box): ranks share devices, timing exchange over gloo
            device = local_rank % max(1, ndev)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            sync_dev = torch.device("cpu")

    from epnn_amd import checkpoint, synth
    from epnn_amd.engine import Pipeline

    weights = checkpoint.load_epnn_weights(os.path.join(ROOT, "models", "decay_model_weights"))
    pipe = Pipeline(depth=args.depth, nx=9, T=5, device=device)
    pipe.set_weights(weights)
    for kv in args.opt:
        name, value = kv.split("=")
        pipe.set_option(name, int(value))

    B = args.molecules
    offsets, xyz, x, Q, N = synth.qm9_like_batch(B=B, seed=rank, N=29)
    A = int(offsets[-1])
    # every lane keeps its own resident copy of the (identical) inputs and its own output buffer
    lanes = [(e, e.to_device(xyz), e.to_device(x), e.to_device(Q), e.alloc(A * 4)) for e in pipe.engines]

    def step(k):
        e, d_xyz, d_x, d_Q, d_q = lanes[k % len(lanes)]
        e.forward_xyz_dev(offsets, d_xyz, d_x, d_Q, d_q, N)

    def barrier():
        pipe.sync()
        if dist is not None:
            if sync_dev.type == "cuda":
                torch.cuda.synchronize()
            dist.barrier()

    for k in range(max(args.warmup, len(lanes))):
        step(k)
    pipe.sync()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    dt = time.perf_counter() - t0
    # Kernel durations for the roofline: the same steps again (up to 600), right after the timed region, this time with
    # hipEvents recorded around every stage on the stream the kernels run on (four event records per forward cost
    # ~20 us of host time per step, which would otherwise be charged to `value`).
    prof_steps = min(args.steps, 600)            # enough launches for a stable mean; four hipEvents are kept per forward
    pipe.set_option("profile", prof_steps)
    for k in range(prof_steps):
        step(k)
    pipe.sync()
    per_lane = [sum(1 for k in range(prof_steps) if k % len(lanes) == l) for l in range(len(lanes))]
    # ms per forward: front-end kernels (none: the fused kernel builds its pair lists itself), fused kernel, tiled kernels, total
    stage = np.array([lanes[l][0].timing_at(i) for l in range(len(lanes)) for i in range(per_lane[l])])
    pipe.set_option("profile", 0)

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=sync_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_max = float(tt.item())
        at = torch.tensor([float(A)], dtype=torch.float64, device=sync_dev)
        dist.all_reduce(at, op=dist.ReduceOp.SUM)
        atoms_total = float(at.item())
    else:
        dt_max, atoms_total = dt, float(A)

    eng, d_q = lanes[0][0], lanes[0][4]
    q = d_q.download((A,))
    stats = eng.last_stats()
    # sanity inside the bench: charges finite and every molecule's total charge conserved
    assert np.isfinite(q).all()
    sums = np.add.reduceat(q.astype(np.float64), offsets[:-1])
    assert np.abs(sums - Q).max() < 1e-4, np.abs(sums - Q).max()

    if rank == 0:
        ns = np.diff(offsets)
        flops = synth.algorithmic_flops(ns, int(stats[0]))
        kname = "k_wave_forward<true,true,true>"
        k_ms = float(stage[:, 1].mean())                   # duration of one launch (hipEvents on its stream)
        step_ms = dt_max / args.steps * 1e3
        # `depth` launches of the SAME kernel share the GPU (one batch is 1024 wavefronts, the machine holds 2048), so
        # the duration of one launch measures the share of the machine it got, not the kernel's rate.  achieved =
        # algorithmic flops of a launch / the time the machine spends per launch (= duration / launches in flight).
        in_flight = max(1.0, k_ms / step_ms)
        achieved = flops / (k_ms / in_flight * 1e-3) / 1e12
        per_launch = flops / (k_ms * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same
        # command (FETCH_SIZE / WRITE_SIZE in separate passes, profiles/r01_pmc_bench.json); null for other shapes
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_bench.json")) as f:
                pmc = json.load(f)
            if pmc.get("workload") == f"qm9_like_b{B}_N{N}":
                traffic = pmc["dominant"][kname]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            traffic = None
        out = {
            "metric": "atoms/sec (inference), QM9-sized batch",
            "value": atoms_total * args.steps / dt_max,
            "unit": "atoms/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"qm9_like_b{B}_N{N}", "molecules_per_gpu": B, "atoms_per_gpu": A, "N": N,
                       "near_pairs_per_gpu": int(stats[0]), "entry": "epnn_forward_xyz_dev (coordinates in HBM)",
                       "weights": "decay_model_weights", "parallelism": f"molecule-sharded x{world}",
                       "batches_in_flight_per_gpu": len(lanes)},
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": achieved,
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                         "algorithmic_gflop_per_launch": flops / 1e9, "kernel_ms_avg": k_ms,
                         "launches_in_flight": in_flight, "tflops_of_one_launch_sharing_the_gpu": per_launch,
                         "note": "kernel_ms_avg = hipEvent duration of one launch while `launches_in_flight` launches of "
                                 "the same kernel share the GPU; achieved = algorithmic flops per launch / "
                                 "(kernel_ms_avg / launches_in_flight) = flops x launches / wall time of the timed region",
                         "device_ms_per_forward_avg": float(stage[:, 3].mean())},
        }
        if not args.no_cpu_baseline and world == 1:         # the reported CPU baseline belongs to the N=1 line only
            out["cpu_baseline"] = cpu_baseline(offsets, xyz, x, Q, N, weights)
        print(json.dumps(out), flush=True)

    for lane in lanes:
        for d in lane[1:]:
            d.free()
    pipe.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
