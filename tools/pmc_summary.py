"""Collapse rocprofv3 --pmc output directories (one per counter pass) into profiles/<name>.json.

    python tools/pmc_summary.py --workload qm9_like_b1024_N29 --out profiles/r01_pmc_bench.json \
        gpurun_out/r1c_fetch gpurun_out/r1c_write gpurun_out/r1c_sq [--merge old.json]

Per-launch means per kernel.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  MI355X_MICROARCH.md (HBM
section): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced read and other access widths must
be calibrated.  tools/micro/fetch_calib.hip does that for the fused kernel's own patterns (16-byte pieces of scattered
64-byte pair rows; one dword per lane of contiguous weight fragments; 512 MiB read once): FETCH_SIZE reads 0.506 /
0.500 / 0.500 of the bytes, WRITE_SIZE exactly 1.000 (profiles/r01_fetch_calibration.txt).  So FETCH_SIZE is doubled,
WRITE_SIZE taken as is.  bench.py reads `dominant[<kernel>]["hbm_bytes_per_launch"]` for its roofline.traffic field.
"""
import argparse
import glob
import json
import os

import pandas as pd


def short(name):
    n = name.replace("void ", "")
    n = n.split("(")[0]
    return n.replace(", ", ",")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--source", default="")
    ap.add_argument("--merge", default=None, help="earlier summary whose `dominant` entries are kept unless re-measured")
    a = ap.parse_args()
    kernels = {}
    for d in a.dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            df = pd.read_csv(f)
            df["k"] = df.Kernel_Name.map(short)
            for (k, c), g in df.groupby(["k", "Counter_Name"]):
                kernels.setdefault(k, {})[c] = float(g.Counter_Value.mean())
                kernels[k]["launches_" + c] = int(len(g))
    dominant = {}
    if a.merge:
        with open(a.merge) as f:
            dominant = json.load(f).get("dominant", {})
    for k, v in kernels.items():
        if not k.startswith("k_wave_forward") and not k.startswith("k_lg_"):
            continue
        ent = {}
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            ent["FETCH_SIZE_KiB"] = v["FETCH_SIZE"]
            ent["WRITE_SIZE_KiB"] = v["WRITE_SIZE"]
            ent["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (GUI/8 = kernel cycles); MFMA busy cycles are summed over the
            # 1024 SIMDs: busy share of SIMD-cycles = busy / (GUI/8 * 1024)
            ent["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 128.0)
            ent["kernel_cycles"] = v["GRBM_GUI_ACTIVE"] / 8.0
        dominant[k] = ent
    out = {"source": a.source or "rocprofv3 --pmc (separate passes); MI355X; per-launch means",
           "workload": a.workload,
           "note": "hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): gfx950's FETCH_SIZE reads 1/2 of the "
                   "bytes (MI355X_MICROARCH.md HBM section), calibrated for this kernel's access patterns with "
                   "tools/micro/fetch_calib.hip (profiles/r01_fetch_calibration.txt)",
           "dominant": dominant, "kernels": kernels}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in dominant.items():
        print(k, v)


if __name__ == "__main__":
    main()
