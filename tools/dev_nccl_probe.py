import os, sys, time, subprocess
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
import numpy as np
if mode != "none":
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    if mode == "nccl":
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    else:
        dist.init_process_group(backend="gloo", rank=0, world_size=1)
        t = torch.ones(4); dist.all_reduce(t)
from epnn_amd import checkpoint, synth
from epnn_amd.engine import Pipeline
w = checkpoint.load_epnn_weights("models/decay_model_weights")
DEPTH = int(os.environ.get("DEPTH", "6"))
pipe = Pipeline(depth=DEPTH, nx=9, T=5, device=0); pipe.set_weights(w)
offsets, xyz, x, Q, N = synth.qm9_like_batch(B=1024, seed=0, N=29)
A = int(offsets[-1])
lanes = [(e, e.to_device(xyz), e.to_device(x), e.to_device(Q), e.alloc(A * 4)) for e in pipe.engines]
def step(k):
    e, a, b, c, d = lanes[k % DEPTH]; e.forward_xyz_dev(offsets, a, b, c, d, N)
for k in range(12): step(k)
pipe.sync()
t0 = time.perf_counter()
for k in range(600): step(k)
pipe.sync()
dt = time.perf_counter() - t0
print(mode, "ms/step", round(dt / 600 * 1e3, 4), "M atoms/s", round(A * 600 / dt / 1e6, 1))
