#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv (largest file under the given directory) with short kernel names."""
import glob, os, sys
import pandas as pd
fs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize)
d = pd.read_csv(fs[-1])
d["Name"] = d.Name.str.replace("void ", "").str.split("(").str[0].str.slice(0, 44)
print(d[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]].head(int(sys.argv[2]) if len(sys.argv) > 2 else 30).to_string())
