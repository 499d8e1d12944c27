#!/usr/bin/env python3
"""EPNN inference throughput on MI355X: atoms/sec on a QM9-sized batch (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one forward of the whole hot path (near-pair construction + T GNN steps + T EPN steps) over one
batch of 1024 synthetic QM9-like molecules padded to N=29, inputs (coordinates, atom features, total charges)
already resident in HBM, charges left in HBM.  With N > 1 ranks (one per GPU) every rank runs its own batch of 1024
molecules (weak scaling, molecules are independent: no data-path collective); the only communication is the barrier
and the MAX over ranks of the timed interval.

Launch forms for N > 1: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment), or plain `python bench.py --gpus N ...`: this process then starts the N
rank processes itself, as fresh children, before it makes any GPU call of its own (it never makes one), relays rank 0's
JSON line and exits with the first non-zero exit code (a rank that dies takes the others down: rendezvous.launch_ranks).  The
barrier and the MAX over ranks of the timed interval go through the LIBRARY's own RCCL communicator (epnn_comm_init +
epnn_comm_allreduce; the 128-byte id travels over epnn_amd/rendezvous.py): the scaling run exercises the product's RCCL path,
and `ranks.rccl_ranks` (ncclCommCount) says how many ranks joined.  No torch anywhere.  With fewer GPUs than ranks (rehearsal
on a one-GPU box: RCCL refuses two ranks on one device) the ranks share devices, take fewer hardware queues each, and the
timing exchange runs over the rendezvous store.

Order of a run: one forward and its checks; the COLD measurement (`value_cold`: W warm-up + K timed steps started on a GPU that
idled for 0.3 s, what a run without anything in front of its warm-up steps reads); parity against the reference's stored outputs
and the single-launch measurement (all three skipped by --no-extras); W warm-up steps; barrier; K timed steps; barrier; the same
steps again with hipEvents for the kernel duration; host-to-host; real-data rate.  The single-launch measurement placed right before the warm-up steps also
means the timed region starts on a GPU at working clocks: K = 20 steps are 2 ms, and on a GPU that idled before the W = 5
warm-up steps (--no-extras) they read 182-185 M atoms/s instead of 194-197 M (`order` in the line; profiles/r02_warm_sweep.txt).

Prints ONE JSON line on rank 0.  Besides the contract's fields:
  roofline.*          dominant kernel k_wave_forward; `frac` is ALGORITHMIC flops of the launches per second of
                      the timed region / `peak`, the bound of the matrix pipe the kernel's products run on (every Dense layer AND the edge
                      products as six bf16 MFMAs per f32-grade product, 2500 / 6 = 416.7 TFLOP/s: `peak_basis`; until the edge products
                      moved there too the line priced them at the f32 MFMA's 157.3: `frac_with_edge_products_at_f32_peak`); `pipe_frac` = the pipes'
                      time for the MFMA flops the kernel really executes (PMC SQ_INSTS_VALU_MFMA_MOPS_F32 / _BF16 x 512) over the
                      time taken; `frac_vs_f32_mfma_peak` = against 157.3 alone; `single_launch` = a launch with the GPU to itself
                      (65536 molecules, depth 1: flops / hipEvent duration IS its fraction, reproducible from
                      profiles/r05_big_launch_kernel_stats.csv)
  parity              max |dq| of the reference's 871 validation systems against the TensorFlow predictions it stored for them
  real_data           atoms/s on that batch (real molecules of 3..38 atoms, N = 41), device-resident, pipelined
  host_to_host        the same forward from host arrays to host arrays (Pipeline.map, a DIFFERENT batch every call)
  blocking_call       one forward at a time, waited for: latency of the batch on a lone handle
  cpu_baseline        the oracle (CPU restatement of the reference's algorithm, not TensorFlow) on this host's cores
"""
import argparse
import hashlib
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# (GPU_MAX_HW_QUEUES is decided in main(), before anything initialises HIP in this process: one hardware queue per batch in
# flight, fewer when ranks share a device)

# Matrix-pipe peaks (MI355X_MICROARCH.md, dense): f32 MFMA 157.3 TFLOP/s, bf16 MFMA 2500.  The fused kernels run every Dense layer
# -- the pair MLPs' second Dense, the first Dense's atom blocks, the update MLP: 69 % of the bench batch's algorithmic flops -- and the
# edge products G = We^T e (31 %) on the bf16 pipe as SIX bf16 products of exact three-piece splits (f32-grade, DESIGN.md section
# 4): an f32-grade flop there is priced at 2500 / 6 = 416.7 TFLOP/s, and `roofline.peak` is the bound of the two pipes together,
# total flops / (f32 flops / 157.3 + bf16-pipe flops / 416.7) (synth.mixed_pipe_peak; only the EPN read-out is left in the f32 part).
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0
KNAME = "k_wave_forward<true,true,true>"
PMC_JSON = os.path.join(ROOT, "profiles", "r05_pmc_bench.json")


def kernel_source_sha():
    """Identity of the fused kernel's build: its sources (with the front-end it includes) and the compile flags.  PMC figures
    recorded for another revision or another set of flags are not quoted."""
    hsh = hashlib.sha256()
    for f in ("epnn_wave.hip.h", "epnn_common.h", "epnn_frontend.hip.h"):
        with open(os.path.join(ROOT, "epnn_amd", "csrc", f), "rb") as fh:
            hsh.update(fh.read())
    import __graft_entry__ as entry
    hsh.update(" ".join(entry.BUILD_FLAGS).encode())
    return hsh.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------ CPU baseline
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
    per_call = (atoms, calls, time.perf_counter() - t0)
    # the same algorithm with 16 molecules per call ("and also batched"): dense inputs built outside the clock
    nb = min(16, len(mols))
    dense = [orc.dense_inputs(xyz, x, Q, N) for xyz, x, Q in mols[:nb]]
    h, e, xd, q, mask = (np.stack([d[k] for d in dense]).astype(np.float32) for k in range(5))
    b_atoms = sum(m[1].shape[0] for m in mols[:nb])
    orc.model_forward(h, e, xd, q, mask, weights, dtype=np.float32)
    batoms = bcalls = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s / 3:
        orc.model_forward(h, e, xd, q, mask, weights, dtype=np.float32)
        batoms += b_atoms
        bcalls += 1
    return per_call + (batoms, bcalls, time.perf_counter() - t0, nb)


def usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(offsets, xyz, x, Q, N, weights, budget_s=12.0):
    """The oracle (literal dense float32 restatement of charge_gn.py, one molecule per call like infer.py:62-76) timed on
    this host on a bounded sample of the same workload: one single-threaded worker process per usable core (spawned:
    nothing of this process's GPU state is inherited), each cycling through its share of the batch for ~12 s."""
    import multiprocessing as mp
    B = len(offsets) - 1
    workers = min(usable_cores(), B, 256)
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
    batched = {"value": sum(r[3] for r in res) / max(r[5] for r in res), "unit": "atoms/s", "molecules_per_call": res[0][6],
               "calls": sum(r[4] for r in res), "what": "the same workers, one model call on the dense (B,N,N,.) inputs of `molecules_per_call` molecules at a time (inputs built outside the clock)"}
    return {"value": atoms / dt, "unit": "atoms/s", "cores": workers, "cores_total": os.cpu_count(), "kind": "port", "batched": batched,
            "what": "CPU restatement of the reference algorithm (oracle/epnn_oracle.py, NumPy float32), not TensorFlow",
            "sample": f"{calls} molecule calls ({atoms} atoms) of the same batch, padded to N={N}, one molecule per call like "
                      f"infer.py, {workers} single-threaded worker processes (one per usable core of {os.cpu_count()}), {dt:.1f} s each"}


# ------------------------------------------------------------------------------------------------ PMC passes (opt-in)
def collect_pmc(args):
    """`--pmc`: rocprofv3 counter passes of this same workload, each its own child run (kernel-trace only next to --pmc),
    started before this process touches the GPU.  FETCH_SIZE / WRITE_SIZE at the bench's depth (HBM bytes per launch,
    gfx950 correction: FETCH_SIZE x 2, MI355X_MICROARCH.md), the SQ counters at depth 1 (a launch alone).  Writes PMC_JSON."""
    import glob
    import pandas as pd
    base = os.path.join(ROOT, "gpurun_out", "pmc_r05")
    passes = [("fetch", "FETCH_SIZE", args.depth), ("write", "WRITE_SIZE", args.depth),
              ("mops", "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE", 1),
              ("mopsbf", "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA", 1),
              ("lds", "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY", 1)]
    vals = {}
    for name, counters, depth in passes:
        d = os.path.join(base, name)
        cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "12", "--warmup", "6", "--no-cpu-baseline", "--no-extras", "--depth", str(depth), "--molecules", str(args.molecules)]
        env = dict(os.environ, TMPDIR="/tmp")
        rc = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, cwd="/tmp")
        if rc.returncode != 0:
            print(f"[bench --pmc] pass {name} failed: {rc.stderr[-400:]}", file=sys.stderr)
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            df = pd.read_csv(f)
            df = df[df.Kernel_Name.str.contains("k_wave_forward")]
            for c, g in df.groupby("Counter_Name"):
                vals[c] = float(g.Counter_Value.mean())
                vals["launches_" + c] = int(len(g))
    out = {"workload": f"qm9_like_b{args.molecules}_N29", "kernel": KNAME, "kernel_source_sha": kernel_source_sha(),
           "command": "python bench.py --pmc  (child passes: rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 12 "
                      "--warmup 6 --no-cpu-baseline --no-extras --depth D; FETCH_SIZE / WRITE_SIZE at the bench depth, SQ_* at depth 1)",
           "counters_per_launch": vals}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # rocprofv3 reports KiB; gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes (x2), WRITE_SIZE is exact
        out["hbm_bytes_per_launch"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in vals:
        out["executed_gflop_per_launch"] = vals["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0 / 1e9
    if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in vals:
        out["executed_bf16_gflop_per_launch"] = vals["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0 / 1e9
    if "SQ_LDS_BANK_CONFLICT" in vals and vals.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_frac"] = vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and vals.get("GRBM_GUI_ACTIVE"):
        out["mfma_busy_frac_one_launch_alone"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] * 128.0)
    with open(PMC_JSON, "w") as f:
        json.dump(out, f, indent=1)
    return out


def real_data_rate(depth, device=0):
    """`--real-only` (child of the main run): atoms/s on the reference's 871-system validation split, N = 41, device-resident,
    `depth` batches in flight.  Prints one JSON object."""
    import tarfile
    import tempfile
    from epnn_amd import charge_gn, checkpoint, synth
    from epnn_amd.engine import Pipeline
    gdir = os.path.join(ROOT, "tests", "golden")
    names = [str(nm) for nm in np.load(os.path.join(gdir, "val_names.npy"), allow_pickle=True)]
    with tempfile.TemporaryDirectory() as tmp:
        with tarfile.open(os.path.join(gdir, "mixed_val.tar.gz")) as tf:
            tf.extractall(tmp)
        mols = [charge_gn.read_xyz(os.path.join(tmp, "mixed_val", nm + ".xyz"), 9) for nm in names]
    v_off = np.zeros(len(mols) + 1, np.int32)
    v_off[1:] = np.cumsum([len(m[1]) for m in mols])
    v_xyz, v_x = np.concatenate([m[0] for m in mols]), np.concatenate([m[1] for m in mols])
    v_Q = np.array([m[2] for m in mols], np.float32)
    rpipe = Pipeline(depth=depth, nx=9, T=5, device=device)
    rpipe.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models", "decay_model_weights")))
    rpipe.set_option("wave2", 0)
    vl = [(e, e.to_device(v_xyz), e.to_device(v_x), e.to_device(v_Q), e.alloc(int(v_off[-1]) * 4)) for e in rpipe.engines]
    for k in range(60 * len(vl)):                           # ~50 ms of load first (clocks)
        e, a_, b_, c_, d_ = vl[k % len(vl)]
        e.forward_xyz_dev(v_off, a_, b_, c_, d_, 41)
    rpipe.sync()
    nrep = 50 * len(vl)
    t1 = time.perf_counter()
    for k in range(nrep):
        e, a_, b_, c_, d_ = vl[k % len(vl)]
        e.forward_xyz_dev(v_off, a_, b_, c_, d_, 41)
    rpipe.sync()
    v_dt = (time.perf_counter() - t1) / nrep
    v_flops = synth.algorithmic_flops(np.diff(v_off), int(vl[0][0].last_stats()[0]))
    v_peak, _ = synth.mixed_pipe_peak(np.diff(v_off), int(vl[0][0].last_stats()[0]), chains_bf16=True)
    rpipe.close()
    print(json.dumps({"value": float(v_off[-1]) / v_dt, "unit": "atoms/s", "ms_per_batch": v_dt * 1e3,
                      "algorithmic_gflop_per_batch": v_flops / 1e9, "frac": v_flops / v_dt / 1e12 / v_peak, "peak": v_peak,
                      "workload": f"the reference's {len(mols)}-system validation split of `mixed` (3..38 atoms, {int(v_off[-1])} atoms), "
                                  f"N = 41, device-resident, {len(vl)} batches in flight, in a process of its own"}))


def committed_pmc(workload):
    try:
        with open(PMC_JSON) as f:
            pmc = json.load(f)
        if pmc.get("workload") == workload and pmc.get("kernel_source_sha") == kernel_source_sha():
            return pmc
    except (OSError, ValueError):
        pass
    return None


# ------------------------------------------------------------------------------------------------ one rank
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)    # 4000 x 0.09 ms = 0.35 s: the fill / drain of the six-deep pipeline
    ap.add_argument("--warmup", type=int, default=24)     # (~0.6 ms each) and the barrier do not weigh on the rate
    ap.add_argument("--molecules", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-launch and host-to-host measurements")
    ap.add_argument("--pmc", action="store_true", help="first run the rocprofv3 counter passes of this workload (writes profiles/r05_pmc_bench.json)")
    ap.add_argument("--depth", type=int, default=0, help="batches in flight per GPU (handles/streams used round robin); 0 = by run "
                    "length: 8 for long runs (steady state: 7-10 deep measured 213-215 M atoms/s, 6 deep 207 M on the same box), for short ones the divisor of --steps among 5, 4, 6 -- the last "
                    "round of launches then fills every lane (a launch takes ~0.3 ms whatever shares the GPU with it, so a short run "
                    "that ends with two of six lanes busy pays for it: K = 20 runs at 184 M atoms/s five deep, 175 M six deep)")
    ap.add_argument("--sleep-ms", type=float, default=0.0, help="(experiment, profiles/r02_warm_sweep.txt) idle time between the warm-up steps and the timed region; negative: the host busy-waits instead of sleeping")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (developer switch)")
    ap.add_argument("--real-depth", type=int, default=14, help="batches in flight of the real-data measurement")
    ap.add_argument("--real-only", action="store_true", help="(child of the main run) only the real-data rate, as one JSON object")
    args = ap.parse_args()

    if args.real_only:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        real_data_rate(args.real_depth, int(os.environ.get("LOCAL_RANK", "0")))
        return
    if args.depth <= 0:
        args.depth = 8 if args.steps >= 200 else next((d for d in (5, 4, 6) if args.steps % d == 0), 6)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the rank processes from here, before this process makes any GPU call (it never makes one)
        from epnn_amd import _lib as _l
        from epnn_amd.rendezvous import launch_ranks
        _l.visible_gpu_count()                               # (asked once, in a child process; the ranks inherit EPNN_NDEV)
        sys.exit(launch_ranks(__file__, sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: libraries that write to file descriptor 1 themselves (gloo announces its
    # connections there) are sent to stderr for the whole run, the line goes to a saved copy of the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.pmc and world == 1:
        collect_pmc(args)

    from epnn_amd import _lib
    device = local_rank
    rdzv = None
    backend = None
    ranks_on_device = 1
    if world > 1:
        # devices are counted WITHOUT initialising HIP: the runtime reads GPU_MAX_HW_QUEUES when it starts
        ndev = _lib.visible_gpu_count()
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if ndev and ndev < local_world:
            ranks_on_device = -(-local_world // ndev)
    want_q = _lib.queues_for_shared_device(ranks_on_device)
    if "GPU_MAX_HW_QUEUES" not in os.environ or (ranks_on_device > 1 and int(os.environ["GPU_MAX_HW_QUEUES"]) > want_q):
        os.environ["GPU_MAX_HW_QUEUES"] = str(want_q)        # (a value inherited from a parent that sized it for one rank per device is cut)
    if ranks_on_device > 1:                                 # fewer queues: fewer batches in flight
        args.depth = max(1, min(args.depth, int(os.environ["GPU_MAX_HW_QUEUES"]) - 1))
    if world > 1:
        from epnn_amd.rendezvous import Rendezvous
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rdzv = Rendezvous(rank, world, timeout=600)          # (a fresh box can take minutes to page the libraries in)
        ndev = _lib.load().epnn_device_count()
        if ndev < 1:
            raise SystemExit("bench.py: no HIP device visible")
        device = local_rank % ndev

    from epnn_amd import checkpoint, synth
    from epnn_amd.engine import Pipeline

    weights = checkpoint.load_epnn_weights(os.path.join(ROOT, "models", "decay_model_weights"))
    pipe = Pipeline(depth=args.depth, nx=9, T=5, device=device)
    pipe.set_weights(weights)
    pipe.set_option("wave2", 0)          # the throughput kernel at every depth (Pipeline does this itself above depth 1)
    for o in args.opt:                   # developer switches, e.g. --opt wave_lds=18432
        k, v = o.split("=")
        pipe.set_option(k, int(v))
    comm_eng = None
    rccl_ranks = None
    if world > 1:
        from epnn_amd.engine import Engine
        if ndev >= world:
            # the library's own communicator on lane 0's handle; the ranks agree over the rendezvous store whether every one of
            # them joined it -- if not (a node whose RCCL does not come up), all of them time over the store instead and say so
            ok, why = True, ""
            try:
                comm_eng = pipe.engines[0]
                comm_eng.comm_init(rdzv.broadcast(Engine.comm_unique_id() if rank == 0 else None, name="rccl_id"), rank, world)
                rccl_ranks = comm_eng.comm_count()
                ok = rccl_ranks == world and comm_eng.comm_allreduce([1.0], "sum")[0] == float(world)
            except Exception as exc:                        # noqa: BLE001
                ok, why = False, str(exc)
            votes = rdzv.all_gather((bool(ok), why), name="rccl_ok")
            if all(v[0] for v in votes):
                backend = "rccl (epnn_comm_allreduce on the library's communicator)"
            else:
                print(f"[bench rank {rank}] RCCL communicator unavailable ({[v[1] for v in votes if not v[0]]}); timing exchange over the rendezvous store", file=sys.stderr)
                comm_eng, rccl_ranks = None, None
                backend = "rendezvous store (the RCCL communicator did not come up)"
        else:
            backend = "rendezvous store (ranks share a device: RCCL refuses two ranks on one GPU)"

    B = args.molecules
    offsets, xyz, x, Q, N = synth.qm9_like_batch(B=B, seed=rank, N=29)
    A = int(offsets[-1])
    # every lane keeps its own resident copy of the (identical) inputs and its own output buffer
    lanes = [(e, e.to_device(xyz), e.to_device(x), e.to_device(Q), e.alloc(A * 4)) for e in pipe.engines]
    n_lanes = len(lanes)

    def step(k):
        e, d_xyz, d_x, d_Q, d_q = lanes[k % len(lanes)]
        e.forward_xyz_dev(offsets, d_xyz, d_x, d_Q, d_q, N)

    def barrier():
        pipe.sync()
        if comm_eng is not None:
            comm_eng.comm_allreduce([0.0], "sum")
        elif rdzv is not None:
            rdzv.barrier("b")

    # ---- before the timed region: the checks and the single-launch measurement (--no-extras skips them).  A rate for wrong
    # charges is worth nothing, so parity comes first; the single-launch measurement runs on EVERY rank right before the
    # warm-up steps, which also means the timed region starts on a GPU at working clocks: the driver's `--steps 20
    # --warmup 5` times 2 ms after 0.5 ms of warm-up, and from an idle GPU (--no-extras) it reads 182-185 M atoms/s where
    # the same 20 steps after 200 warm-up steps read 197-203 M; the clocks take some 20 ms of load to come up and fall
    # back within 5-20 ms of idling (profiles/r02_warm_sweep.txt), so nothing that idles the GPU (hipFree of the big
    # buffers) is done between that measurement and the timed steps.
    step(0)
    pipe.sync()
    eng, d_q = lanes[0][0], lanes[0][4]
    q = d_q.download((A,))
    stats = eng.last_stats()
    # sanity inside the bench: charges finite and every molecule's total charge conserved (the batch itself is compared
    # with the float64 oracle in tests/test_gpu_parity.py::test_bench_batch_vs_oracle)
    assert np.isfinite(q).all()
    sums = np.add.reduceat(q.astype(np.float64), offsets[:-1])
    assert np.abs(sums - Q).max() < 1e-4, np.abs(sums - Q).max()
    ns = np.diff(offsets)
    flops = synth.algorithmic_flops(ns, int(stats[0]))
    peak, bf_share = synth.mixed_pipe_peak(ns, int(stats[0]), chains_bf16=True)
    peak_edges_f32 = synth.mixed_pipe_peak(ns, int(stats[0]), chains_bf16=True, edges_bf16=False)[0]     # (the pricing of this line until the edge products moved: 277.2)

    def timed_steps():
        """W warm-up steps, barrier, K timed steps, barrier -> seconds of the timed region on this rank"""
        for k in range(max(args.warmup, len(lanes))):
            step(k)
        pipe.sync()
        barrier()
        gc.disable()                                        # (no collector pause inside a region that lasts two milliseconds)
        t_start = time.perf_counter()
        for k in range(args.steps):
            step(k)
        barrier()
        dt_ = time.perf_counter() - t_start
        gc.enable()
        return dt_

    cold_dt = None
    if not args.no_extras:
        # the same measurement from an IDLE GPU (nothing but a 0.3 s pause in front of the warm-up steps): the figure a run
        # reads whose timed region is a few milliseconds long and has no load in front of it (clocks: DESIGN.md section 5)
        pipe.sync()
        time.sleep(0.3)
        cold_dt = timed_steps()

    extras, real, held = {}, None, []
    if rank == 0 and not args.no_extras:
        # (1) the metric's second half, "max |dq| vs reference", and a REAL-data rate: the 871 systems of the reference's
        #     recorded validation split (tests/golden/mixed_val.tar.gz, 3..38 atoms, N = 41) against the TensorFlow
        #     predictions the reference stored for them (models/model_systems/test_pred_charges.npy); its rate is (4)
        try:
            import tarfile
            import tempfile
            from epnn_amd import charge_gn
            gdir = os.path.join(ROOT, "tests", "golden")
            names = [str(nm) for nm in np.load(os.path.join(gdir, "val_names.npy"), allow_pickle=True)]
            gold = np.load(os.path.join(gdir, "test_pred_charges.npy"))
            with tempfile.TemporaryDirectory() as tmp:
                with tarfile.open(os.path.join(gdir, "mixed_val.tar.gz")) as tf:
                    tf.extractall(tmp)
                mols = [charge_gn.read_xyz(os.path.join(tmp, "mixed_val", nm + ".xyz"), 9) for nm in names]
            v_off = np.zeros(len(mols) + 1, np.int32)
            v_off[1:] = np.cumsum([len(m[1]) for m in mols])
            v_xyz, v_x = np.concatenate([m[0] for m in mols]), np.concatenate([m[1] for m in mols])
            v_Q = np.array([m[2] for m in mols], np.float32)
            vq = pipe.engines[0].forward_xyz(v_off, v_xyz, v_x, v_Q, 41)
            dq = max(float(np.abs(vq[v_off[i]:v_off[i + 1]] - gold[i, :v_off[i + 1] - v_off[i]]).max()) for i in range(len(mols)))
            drift = float(np.abs(np.add.reduceat(vq.astype(np.float64), v_off[:-1]) - v_Q).max())
            extras["parity"] = {"max_abs_dq_vs_reference": dq, "tolerance": 1e-5, "max_abs_total_charge_error": drift,
                                "systems": len(mols), "atoms": int(v_off[-1]),
                                "reference": "TensorFlow predictions stored by the reference for its validation split "
                                             "(models/model_systems/test_pred_charges.npy, decay_model_weights, N = 41)"}
            real = (v_off, v_xyz, v_x, v_Q)
            assert dq <= 1e-5, dq
        except (OSError, KeyError) as exc:                  # fixtures missing: the synthetic figures stand alone
            print(f"[bench] real-data / parity extras skipped: {exc}", file=sys.stderr)

    barrier()
    if not args.no_extras:
        # (2) a launch with the GPU to itself: 64 copies of the batch in ONE launch (65536 molecules), one handle, hipEvents
        #     around each launch: algorithmic flops / duration is that kernel's rate with nothing else on the machine
        rep = max(1, 65536 // B)
        big_off = np.concatenate([[0]] + [offsets[1:] + r * A for r in range(rep)]).astype(np.int32)
        e0 = pipe.engines[0]
        big = [e0.to_device(np.tile(a, (rep,) + (1,) * (a.ndim - 1))) for a in (xyz, x, Q)]
        big_q = e0.alloc(A * rep * 4)
        for _ in range(2):
            e0.forward_xyz_dev(big_off, big[0], big[1], big[2], big_q, N)
        e0.sync()
        nbig = 10
        e0.set_option("profile", nbig)
        for _ in range(nbig):
            e0.forward_xyz_dev(big_off, big[0], big[1], big[2], big_q, N)
        e0.sync()
        big_ms = float(np.mean([e0.timing_at(i)[1] for i in range(nbig)]))
        e0.set_option("profile", 0)
        # Two further launches of the same batch keep the GPU under load while the warm-up steps are queued, and nothing that
        # idles it (hipFree, the download of the big result) happens before the timed region: see the note on clocks above.
        for _ in range(2):
            e0.forward_xyz_dev(big_off, big[0], big[1], big[2], big_q, N)
        held = big + [big_q]
        prewarm_ms = (nbig + 2) * big_ms                   # GPU time of the launches in front of the warm-up steps

    for k in range(max(args.warmup, len(lanes))):
        step(k)
    pipe.sync()
    if args.sleep_ms > 0:
        time.sleep(args.sleep_ms * 1e-3)
    elif args.sleep_ms < 0:                                  # busy wait: CPU stays hot, GPU idles
        t_end = time.perf_counter() - args.sleep_ms * 1e-3
        while time.perf_counter() < t_end:
            pass
    barrier()
    gc.disable()                                            # (see timed_steps; a collection HERE would idle the GPU for milliseconds: clocks)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    t_enq = time.perf_counter() - t0                        # (host time of the K calls; the rest of dt is the wait for the GPU)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
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

    if world > 1:
        # MAX over ranks of the timed interval and the total of the atoms: over RCCL when every rank has its own device
        if comm_eng is not None:
            mx = comm_eng.comm_allreduce([dt, cold_dt or 0.0], "max")
            atoms_total = comm_eng.comm_allreduce([float(A)], "sum")[0]
        else:
            mx = rdzv.all_reduce_max([dt, cold_dt or 0.0], name="dt")
            atoms_total = sum(rdzv.all_gather(float(A), name="atoms"))
        dt_max, cold_max = mx[0], mx[1]
        per_rank = [(float(d), float(a)) for d, a in rdzv.all_gather((dt, float(A)), name="per_rank")]    # detail for the line
    else:
        per_rank = [(dt, float(A))]
        dt_max, atoms_total, cold_max = dt, float(A), (cold_dt or 0.0)

    if held:
        qb = held[3].download((A * rep,))
        assert np.array_equal(qb[:A], q) and np.array_equal(qb[-A:], q)          # batch composition does not change the bits
        for d in held:
            d.free()
        extras["single_launch"] = {"molecules": B * rep, "launches": nbig, "kernel_ms_avg": big_ms,
                                   "algorithmic_gflop_per_launch": flops * rep / 1e9,
                                   "achieved": flops * rep / (big_ms * 1e-3) / 1e12,
                                   "frac": flops * rep / (big_ms * 1e-3) / 1e12 / peak,
                                   "atoms_per_s": A * rep / (big_ms * 1e-3)}
    q_after = d_q.download((A,))
    assert np.array_equal(q_after, q)                       # the timed steps produced the charges that were checked

    if rank == 0 and not args.no_extras:
        # (3) host to host: Pipeline.map on host arrays, a different batch every call (new plan, PCIe both ways)
        # the number of distinct batches shares no factor with the depth, so every lane really sees a different batch (and
        # builds a new plan) on every call; with 8 batches on 8 lanes each lane would keep meeting the same one
        import math
        nb = next(n for n in range(7, 40) if math.gcd(n, len(lanes)) == 1)
        batches = [synth.qm9_like_batch(B=B, seed=1000 + s, N=29)[:4] for s in range(nb)]
        ncall = 240
        stream = [batches[k % len(batches)] for k in range(ncall)]
        for _ in pipe.map(stream[:120], N):              # ~20 ms of load first (clocks), whatever ran before
            pass
        t1 = time.perf_counter()
        h_atoms = sum(qq.shape[0] for qq in pipe.map(stream, N))
        h_dt = time.perf_counter() - t1
        extras["host_to_host"] = {"value": h_atoms / h_dt, "unit": "atoms/s", "ms_per_batch": h_dt / ncall * 1e3, "calls": ncall,
                                  "what": "Pipeline.map (epnn_forward_xyz_begin/_end): host xyz/x/Q -> host q, a different "
                                          f"batch of {B} molecules (a new plan) every call on every lane, {len(lanes)} in flight; PCIe inclusive, not `value`"}

    if rank == 0 and not args.no_extras:
        # (3b) ONE blocking call at a time (what a caller without a pipeline sees): device-resident, the default mode of a
        #      lone handle (molecules of 17+ atoms split over two wavefronts) and one wavefront per molecule
        from epnn_amd.engine import Engine
        e1 = Engine(nx=9, T=5, device=device)
        e1.set_weights(weights)
        b_in = [e1.to_device(a) for a in (xyz, x, Q)]
        b_q = e1.alloc(A * 4)
        blocking = {}
        for label, opt in (("ms", -1), ("ms_one_wavefront_per_molecule", 0)):
            e1.set_option("wave2", opt)
            for _ in range(10):
                e1.forward_xyz_dev(offsets, b_in[0], b_in[1], b_in[2], b_q, N)
                e1.sync()
            t1 = time.perf_counter()
            for _ in range(100):
                e1.forward_xyz_dev(offsets, b_in[0], b_in[1], b_in[2], b_q, N)
                e1.sync()
            blocking[label] = (time.perf_counter() - t1) / 100 * 1e3
        assert np.abs(b_q.download((A,)) - q).max() <= 1e-6
        for d in b_in + [b_q]:
            d.free()
        e1.close()
        blocking["atoms_per_s"] = A / (blocking["ms"] * 1e-3)
        blocking["what"] = f"one forward of the batch of {B} molecules at a time, waited for (no pipeline), inputs resident in HBM"
        extras["blocking_call"] = blocking

    if real is not None:
        # (4) the real-data rate: the validation batch of (1), device-resident, through every lane of the pipeline
        # Fourteen batches in flight whatever the timed region used: the 22 systems of 33..38 atoms of this batch run as a launch of
        # their own (three wavefronts each, 0.12 ms of latency on 66 wavefronts) in front of the lane's main launch, so a lane is
        # busy ~0.4 ms per batch with little work in a third of it and the LANES, not the machine, limit the rate (round 4, fresh
        # process: 176 / 184 / 192 M atoms/s at 8 / 12 / 14 lanes; the 849 systems of <= 32 atoms alone run at the synthetic batch's
        # rate at any depth; a second stream per lane and a merged launch were both measured slower).
        # It runs in a CHILD process (this script with --real-only), after this process has closed its own lanes: which hardware
        # queues a pipeline's lanes land on depends on every stream the process created before (129 .. 175 M atoms/s for the same
        # eight lanes behind different histories), a fresh process has a defined one.
        for lane in lanes:
            for d in lane[1:]:
                d.free()
        pipe.close()
        pipe = None
        # (rank 0 only: `real` is set by the parity block above.)  A child that hangs or cannot be started must not cost the line its
        # already-measured `value`, nor leave the other ranks in the closing barrier: real_data is then null with the reason.
        try:
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--real-only", "--real-depth", str(args.real_depth)],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            extras["real_data"] = json.loads([l for l in child.stdout.splitlines() if l.startswith("{")][-1])
        except (subprocess.TimeoutExpired, OSError) as exc:
            extras["real_data"] = None
            extras["real_data_error"] = f"{type(exc).__name__}: {exc}"[:300]
            print(f"[bench] real-data child did not finish: {exc}", file=sys.stderr)
        except (IndexError, ValueError):
            extras["real_data"] = None
            extras["real_data_error"] = "no JSON line from the child: " + child.stderr[-200:]
            print(f"[bench] real-data child failed: {child.stderr[-400:]}", file=sys.stderr)

    if rank == 0:
        k_ms = float(stage[:, 1].mean())                   # duration of one launch (hipEvents on its stream)
        step_ms = dt_max / args.steps * 1e3
        # `depth` launches of the SAME kernel share the GPU (one batch is ~1024 wavefronts, the machine holds 2048), so
        # the duration of one launch measures the share of the machine it got, not the kernel's rate.  achieved =
        # algorithmic flops of a launch / the time the machine spends per launch (= duration / launches in flight).
        in_flight = max(1.0, k_ms / step_ms)
        achieved = flops / (k_ms / in_flight * 1e-3) / 1e12
        workload = f"qm9_like_b{B}_N{N}"
        pmc = committed_pmc(workload)
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        executed = pmc.get("executed_gflop_per_launch") if pmc else None
        executed_bf = pmc.get("executed_bf16_gflop_per_launch") if pmc else None
        # time the two matrix pipes need for what the kernel executes on them (PMC), per launch
        pipe_s = (executed * 1e9 / (FP32_MFMA_PEAK_TFLOPS * 1e12) + executed_bf * 1e9 / (BF16_MFMA_PEAK_TFLOPS * 1e12)) if executed is not None and executed_bf is not None else None
        roof = {"bound": "mfma", "kernel": KNAME, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak,
                "peak_basis": f"the bf16 matrix pipe for {bf_share:.3f} of the algorithmic flops -- every Dense layer and the edge products as six bf16 MFMAs "
                              f"per f32-grade product, {BF16_MFMA_PEAK_TFLOPS:.0f} / 6 = {BF16_MFMA_PEAK_TFLOPS / 6:.1f} TFLOP/s --, f32 MFMA {FP32_MFMA_PEAK_TFLOPS} for the rest "
                              "(the EPN read-out); peak = total / (f32 part / 157.3 + bf16-pipe part / 416.7)",
                "frac_vs_f32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                "frac_with_edge_products_at_f32_peak": achieved / peak_edges_f32,
                "frac_basis": "algorithmic flops (SURVEY section 8d); the matrix pipes' own utilisation (executed MFMA flops of either kind, PMC) is pipe_frac",
                "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                "traffic_source": (pmc["command"] + f"; kernel source {pmc['kernel_source_sha']}; profiles/r05_pmc_bench.json") if pmc else None,
                "algorithmic_gflop_per_launch": flops / 1e9,
                "executed_gflop_per_launch": executed,
                "executed_bf16_gflop_per_launch": executed_bf,
                "pipe_frac": (pipe_s / (k_ms / in_flight * 1e-3)) if pipe_s else None,
                "kernel_ms_avg": k_ms, "launches_in_flight": in_flight,
                "frac_of_one_launch_sharing_the_gpu": flops / (k_ms * 1e-3) / 1e12 / peak,
                "note": "frac = algorithmic flops per launch / (kernel_ms_avg / launches_in_flight) / peak = flops x launches / "
                        "wall time of the timed region: `launches_in_flight` launches of this kernel share the GPU, so flops / "
                        "kernel_ms_avg alone (frac_of_one_launch_sharing_the_gpu, what a rocprofv3 --stats average of THIS command "
                        "gives) is one launch's share of the machine; single_launch is the same kernel with the GPU to itself; "
                        "pipe_frac = the time the two matrix pipes need for the MFMA flops the kernel executes (PMC: f32 / 157.3 + "
                        "bf16 / 2500) over the time it takes",
                "device_ms_per_forward_avg": float(stage[:, 3].mean())}
        if "single_launch" in extras:
            roof["single_launch"] = extras["single_launch"]
        out = {
            "metric": "atoms/sec (inference), QM9-sized batch",
            "value": atoms_total * args.steps / dt_max,
            "unit": "atoms/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "value_cold": (atoms_total * args.steps / cold_max) if cold_max else None,
            "prewarm_ms": None if args.no_extras else prewarm_ms,
            "enqueue_ms": t_enq * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": ("float32 inputs, outputs and accumulation; every Dense product float32-grade: both operands split EXACTLY into "
                           "three bf16 pieces, the six largest partial products summed in f32 on the bf16 matrix pipe (what is left out is "
                           "below 2^-24 of a product; error against float64 below the f32 MFMA's own, profiles/r05_micro_bf16x6.txt); "
                           "the edge products as f32 MFMAs; `parity` below is measured with this arithmetic"),
            "data": "synthetic",
            "config": {"workload": workload, "molecules_per_gpu": B, "atoms_per_gpu": A, "N": N,
                       "pairs_under_cutoff_per_gpu": int(stats[0]), "entry": "epnn_forward_xyz_dev (coordinates in HBM)",
                       "weights": "decay_model_weights", "parallelism": f"molecule-sharded x{world}",
                       "batches_in_flight_per_gpu": n_lanes},
            "roofline": roof,
            "order": ("one forward + checks, " + ("" if args.no_extras else "0.3 s idle + warm-up + timed steps (value_cold), parity on the reference's stored outputs, single-launch measurement (+2 launches of it unmeasured, no idle gap: prewarm_ms of GPU load), ")
                      + f"{max(args.warmup, n_lanes)} warm-up steps, {args.steps} timed steps, the same steps again with hipEvents"
                      + ("" if args.no_extras else ", host-to-host, real-data rate")),
        }
        if world > 1:
            out["ranks"] = {"world_size": world, "timing_backend": backend, "rccl_ranks": rccl_ranks,
                            "ranks_per_device": ranks_on_device, "hw_queues_per_rank": int(os.environ["GPU_MAX_HW_QUEUES"]),
                            "atoms_per_s_per_rank": [a * args.steps / d for d, a in per_rank]}
        for key in ("parity", "real_data", "host_to_host", "blocking_call"):
            if key in extras:
                out[key] = extras[key]
        if not args.no_cpu_baseline and world == 1:         # the reported CPU baseline belongs to the N=1 line only
            out["cpu_baseline"] = cpu_baseline(offsets, xyz, x, Q, N, weights)
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()

    if pipe is not None:
        for lane in lanes:
            for d in lane[1:]:
                d.free()
    if rdzv is not None:
        rdzv.barrier("end")
    if pipe is not None:
        pipe.close()
    if rdzv is not None:
        rdzv.close()


if __name__ == "__main__":
    main()
