#!/usr/bin/env python3
"""Timeline of the fused kernel's launches in a rocprofv3 --kernel-trace of `bench.py --steps K`: start / end of every
launch relative to the first one of the timed region, and how many launches overlap over time (fill and drain)."""
import glob, sys
import numpy as np
import pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
K = int(sys.argv[2]); W = int(sys.argv[3])
d = pd.read_csv(f)
w = d[d.Kernel_Name.str.contains("k_wave_forward")].sort_values("Start_Timestamp").reset_index(drop=True)
t = w.iloc[W:W + K]
t0 = t.Start_Timestamp.min()
print("launches in trace", len(w), "timed region", K, "after", W, "warm-up launches")
for i, r in t.iterrows():
    print(f"  #{i - W:2d} start {(r.Start_Timestamp - t0) / 1e3:8.1f} us  end {(r.End_Timestamp - t0) / 1e3:8.1f} us  dur {(r.End_Timestamp - r.Start_Timestamp) / 1e3:6.1f}")
end = t.End_Timestamp.max()
print(f"region: {(end - t0) / 1e3:.1f} us for {K} launches = {(end - t0) / 1e3 / K:.2f} us per launch; "
      f"previous launch ended {(t0 - w.iloc[W - 1].End_Timestamp) / 1e3:.1f} us before the region's first start")
grid = np.linspace(t0, end, 41)
for a, b in zip(grid[:-1], grid[1:]):
    m = 0.5 * (a + b)
    n = int(((t.Start_Timestamp <= m) & (t.End_Timestamp >= m)).sum())
    print(f"  t = {(m - t0) / 1e3:7.1f} us: {n} launches running " + "#" * n)
