#!/usr/bin/env python3
"""Kernel concurrency in a rocprofv3 --kernel-trace: busy time, sum of kernel durations, per-kernel table for the last part of the run."""
import glob, sys
import numpy as np
import pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = pd.read_csv(f).sort_values("Start_Timestamp")
d = d.iloc[len(d) // 2:]                    # steady state: second half
d["dur"] = d.End_Timestamp - d.Start_Timestamp
t0, t1 = d.Start_Timestamp.min(), d.End_Timestamp.max()
ev = sorted([(s, 1) for s in d.Start_Timestamp] + [(e, -1) for e in d.End_Timestamp])
busy = 0; depth = 0; last = t0; wsum = 0
for t, k in ev:
    if depth > 0: busy += t - last; wsum += depth * (t - last)
    depth += k; last = t
print(f"span {(t1 - t0) / 1e3:.0f} us, busy (>=1 kernel) {busy / 1e3:.0f} us, sum of durations {d.dur.sum() / 1e3:.0f} us, mean concurrency while busy {wsum / max(busy, 1):.2f}, kernels {len(d)}")
d["k"] = d.Kernel_Name.str.replace("void ", "").str.split("(").str[0].str.slice(0, 40)
print(d.groupby("k").dur.agg(["count", "mean", "sum"]).sort_values("sum", ascending=False).head(14).to_string())
print("queues used:", sorted(d.Queue_Id.unique()) if "Queue_Id" in d else "n/a")
