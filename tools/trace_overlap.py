#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel mean duration and how many fused kernels run concurrently."""
import sys, glob
import pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = pd.read_csv(f)
d["dur"] = d.End_Timestamp - d.Start_Timestamp
d = d.sort_values("Start_Timestamp")
t0 = d.Start_Timestamp.min()
print(d.groupby("Kernel_Name").dur.agg(["count", "mean", "min", "max"]).sort_values("mean", ascending=False).head(12).to_string())
w = d[d.Kernel_Name.str.contains("k_wave_forward|k_small_forward")]
w = w.iloc[len(w) // 2: len(w) // 2 + 12]
for _, r in w.iterrows():
    print(f"start {((r.Start_Timestamp - t0) / 1e3):10.1f} us  dur {r.dur / 1e3:7.1f} us  stream/queue {r.get('Queue_Id', '')}")
starts = d[d.Kernel_Name.str.contains("k_wave_forward|k_small_forward")].Start_Timestamp.values
import numpy as np
print("mean interval between fused-kernel starts (us):", float(np.diff(starts)[10:].mean() / 1e3))
