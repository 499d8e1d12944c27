#!/usr/bin/env python3
"""Large-system timings (BASELINE.json configs[3] and [4]): the 2220-atom protein and a synthetic box.
    python tools/bench_large.py protein | box100k | box<N>k [steps]
One system on several GPUs (row-block partition of the all-pairs sweep):
    python tools/bench_large.py box100k 5 --gpus N       one process per GPU, row exchange over RCCL on the engines' streams
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_large.py box100k
                                                          (ranks sharing a GPU: the exchange is host-staged over gloo)
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epnn_amd import checkpoint, synth, charge_gn
from epnn_amd.engine import Engine

def main():
    opts = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--opt=")]          # --opt=name:value (engine option)
    sys.argv = [a for a in sys.argv if not a.startswith("--opt=")]
    argv = [a for a in sys.argv[1:] if not a.startswith("--gpus")]
    gpus = next((int(a.split("=")[1]) if "=" in a else int(sys.argv[sys.argv.index(a) + 1]) for a in sys.argv[1:] if a.startswith("--gpus")), 0)
    if gpus:
        argv = [a for a in argv if a != str(gpus)] if "--gpus" in sys.argv[1:] else argv
    if gpus > 1 and "WORLD_SIZE" not in os.environ:
        from epnn_amd.rendezvous import launch_ranks
        sys.exit(launch_ranks(__file__, sys.argv[1:], gpus))
    what = argv[0] if len(argv) > 0 else "protein"
    # (a forward of the protein lasts 0.6 ms: the GPU's clocks need some tens of milliseconds of load to come up, 200 forwards read
    #  3 % faster than 20 -- profiles/r03_large_systems.txt --, and the default takes a tenth of a second either way)
    steps = int(argv[1]) if len(argv) > 1 else (200 if what == "protein" else 5)
    w = checkpoint.load_epnn_weights(os.path.join(ROOT, "models/decay_model_weights"))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device, how = 0, ""
    from epnn_amd import _lib
    ndev = _lib.load().epnn_device_count()
    rccl = world > 1 and ndev >= world
    if world > 1:
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev)
    eng = Engine(nx=9, T=5, device=device)
    eng.set_weights(w)
    for o in opts:
        eng.set_option(o.split(":")[0], int(o.split(":")[1]))
    if rccl:
        from epnn_amd.rendezvous import Rendezvous
        rdzv = Rendezvous(rank, world)
        eng.comm_init(rdzv.broadcast(Engine.comm_unique_id() if rank == 0 else None, name="id"), rank, world)
        eng.set_partition(rank, world)                                    # rows exchanged over RCCL, in stream
        how = "RCCL in stream"
    elif world > 1:
        import torch.distributed as dist
        from epnn_amd import shard
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)      # ranks share a device: host-staged exchange
        eng.set_partition(rank, world, shard.make_row_exchange(eng, dist, rank, world))
        how = "host-staged over gloo"
    if what == "protein":
        xyz, x, Q, _ = charge_gn.read_xyz(os.path.join(ROOT, "tests/golden/protein/6qlp_capped.xyz"), 9)
        offsets = np.array([0, len(x)], dtype=np.int32)
        Q = np.array([Q], dtype=np.float32)
        N = len(x)
    else:
        n = int(what.replace("box", "").replace("k", "")) * 1000
        t0 = time.time()
        offsets, xyz, x, Q, N = synth.box_system(n_atoms=n, seed=0)
        print(f"generated {n} atoms in {time.time()-t0:.1f} s", flush=True)
    A = int(offsets[-1])
    d = [eng.to_device(a) for a in (xyz, x, Q)]
    dq = eng.alloc(A * 4)
    eng.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
    eng.sync()
    # first without the stage events (four timed event records per forward sit between one forward's last kernel and the next one's
    # first: ~20 us of an otherwise back-to-back stream), then with them
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
    eng.sync()
    dt_plain = (time.perf_counter() - t0) / steps
    eng.set_option("profile", steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward_xyz_dev(offsets, d[0], d[1], d[2], dq, N)
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    st = np.array([eng.timing_at(k) for k in range(steps)]).mean(0)
    stats = eng.last_stats()
    q = dq.download((A,))
    flops = synth.algorithmic_flops([A], int(stats[0]))
    peak, bf_share = synth.mixed_pipe_peak([A], int(stats[0]))
    if rank == 0:
      print((f"[{world} processes, rows of atoms partitioned, exchange {how}] " if world > 1 else "") + f"{what}: {A} atoms, {stats[0]} near pairs; {dt_plain*1e3:.3f} ms/forward wall without stage events, {dt*1e3:.3f} with; device stages front/fused/tiled/total ms = {np.round(st,3)}; "
          f"{A/dt:.3e} atoms/s; algorithmic {flops/1e9:.1f} Gflop -> {flops/(st[3]*1e-3)/1e12:.1f} TFLOP/s = {flops/(st[3]*1e-3)/1e12/peak:.3f} of the two matrix pipes' bound {peak:.1f} ({bf_share:.3f} of the flops on the bf16 pipe at 2500/6); sum q = {q.sum(dtype=np.float64):.6f}", flush=True)
      import json
      # the same facts as ONE JSON line in bench.py's vocabulary (last line of the output)
      print(json.dumps({"metric": "atoms/sec (inference), one large system", "value": A / dt, "unit": "atoms/s", "n_gpus": world,
                        "steps": steps, "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "f32", "data": "synthetic" if what != "protein" else "6qlp_capped.xyz",
                        "config": {"workload": what, "atoms": A, "pairs_under_cutoff": int(stats[0]), "N": int(N), "weights": "decay_model_weights",
                                   "parallelism": f"rows of atoms partitioned x{world} ({how})" if world > 1 else "one GPU"},
                        "roofline": {"bound": "mfma", "kernel": "k_lg_sweep (all-pairs sum of charge_gn.py:70) + per-step tails", "achieved": flops / (st[3] * 1e-3) / 1e12,
                                     "peak": peak, "unit": "TFLOP/s", "frac": flops / (st[3] * 1e-3) / 1e12 / peak, "traffic": None,
                                     "peak_basis": f"f32 MFMA 157.3 TFLOP/s for {1 - bf_share:.3f} of the algorithmic flops, the all-pairs Dense ({bf_share:.3f}) as six bf16 MFMAs per f32-grade product at 2500 / 6",
                                     "frac_vs_f32_mfma_peak": flops / (st[3] * 1e-3) / 1e12 / 157.3,
                                     "algorithmic_gflop_per_forward": flops / 1e9, "device_ms_per_forward": float(st[3]),
                                     "note": "whole forward (hipEvents on the handle's stream), not the sweep kernel alone"}}), flush=True)
    eng.close()

if __name__ == "__main__":
    main()
