#!/usr/bin/env python3
"""Timeline of ONE forward of a large system in a rocprofv3 --kernel-trace of `tools/bench_large.py protein N`: every launch of
the forward with its start (relative to the forward's first kernel), duration and the idle gap in front of it.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/bench_large.py protein 20
    python3 tools/large_timeline.py DIR [which forward, counted from the end: default 3]"""
import glob, sys
import pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
d = d[d.Kernel_Name.str.contains("k_lg_|k_front_|k_wave_")].reset_index(drop=True)
firsts = d.index[d.Kernel_Name.str.contains("k_lg_first|k_lg_init")].tolist()
i0 = firsts[-back]
i1 = firsts[-back + 1] if back > 1 else len(d)
t = d.iloc[i0:i1]
t0 = t.Start_Timestamp.iloc[0]
prev_end = d.End_Timestamp.iloc[i0 - 1] if i0 > 0 else t0
busy = 0
for _, r in t.iterrows():
    dur = (r.End_Timestamp - r.Start_Timestamp) / 1e3
    busy += dur
    print(f"{(r.Start_Timestamp - t0) / 1e3:8.1f} us  gap {(r.Start_Timestamp - prev_end) / 1e3:6.2f}  dur {dur:7.2f}  {r.Kernel_Name.split('(')[0]}  grid {r.Grid_Size_X if 'Grid_Size_X' in r else ''} wg {r.Workgroup_Size_X if 'Workgroup_Size_X' in r else ''}")
    prev_end = r.End_Timestamp
print(f"forward: {(t.End_Timestamp.iloc[-1] - t0) / 1e3:.1f} us from first start to last end, {busy:.1f} us inside kernels, {len(t)} launches")
