#!/usr/bin/env python3
"""SQ counter passes over ONE large launch of the fused kernel (development aid; bench.py --pmc writes the judged summary).

    python tools/pmc_wave.py [--lib path.so] [--molecules 8192] [--out file.json]

Each pass is its own child process under `rocprofv3 --pmc ... --kernel-trace` (never combined with other trace domains); the
child is THIS script with --child, which runs a few forwards of `molecules` copies of the bench batch in one launch (every SIMD
holds its two wavefronts for most of the launch).  Prints per-launch means and a few ratios."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PASSES = [
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM",
    "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS",
    "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32",
    "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS",
    "SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_FLAT SQ_LDS_UNALIGNED_STALL",
    "GRBM_GUI_ACTIVE",
    "FETCH_SIZE",
    "WRITE_SIZE",
]


def child(lib, molecules):
    from epnn_amd import _lib
    if lib:
        _lib.LIB_PATH = os.path.abspath(lib)
    import numpy as np
    from epnn_amd import checkpoint, synth
    from epnn_amd.engine import Engine
    off, xyz, x, Q, N = synth.qm9_like_batch(1024, 0, 29)
    copies = max(1, molecules // 1024)
    ns = np.diff(off)
    offs = np.concatenate([[0], np.cumsum(np.tile(ns, copies))]).astype(np.int32)
    eng = Engine()
    eng.set_weights(checkpoint.load_epnn_weights(os.path.join(ROOT, "models", "decay_model_weights")))
    eng.set_option("wave2", 0)
    X, XX, QQ = np.tile(xyz, (copies, 1)), np.tile(x, (copies, 1)), np.tile(Q, copies)
    for _ in range(6):
        eng.forward_xyz(offs, X, XX, QQ, N)


def main():
    a = sys.argv[1:]
    lib = a[a.index("--lib") + 1] if "--lib" in a else ""
    molecules = int(a[a.index("--molecules") + 1]) if "--molecules" in a else 8192
    if "--child" in a:
        return child(lib, molecules)
    import pandas as pd
    out = a[a.index("--out") + 1] if "--out" in a else ""
    base = os.path.join(ROOT, "gpurun_out", "pmc_wave")
    vals = {}
    for i, counters in enumerate(PASSES):
        d = os.path.join(base, f"p{i}")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--child", "--molecules", str(molecules)] + (["--lib", os.path.abspath(lib)] if lib else [])
        rc = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, cwd="/tmp")
        if rc.returncode != 0:
            print(f"pass {i} failed: {rc.stderr[-300:]}", file=sys.stderr)
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            df = pd.read_csv(f)
            df = df[df.Kernel_Name.str.contains("k_wave_forward")]
            for c, g in df.groupby("Counter_Name"):
                vals[c] = float(g.Counter_Value.mean())
    mol = molecules
    print(f"per launch of {mol} molecules ({lib or 'shipped library'}); per molecule in brackets")
    for k in sorted(vals):
        print(f"  {k:34s} {vals[k]:16.0f}   [{vals[k] / mol:12.1f}]")
    g = vals.get
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        print("shares of wave cycles: wait_any %.3f  wait_inst_any %.3f  active_any %.3f | active valu %.3f lds %.3f vmem %.3f sca %.3f misc %.3f" % tuple(
            g(k, 0) / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                    "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC")))
    if g("SQ_INSTS_MFMA"):
        print("VALU (non-MFMA) per MFMA: %.3f   SALU per MFMA: %.3f  LDS per MFMA %.3f" % ((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA"),
              g("SQ_INSTS_SALU", 0) / g("SQ_INSTS_MFMA"), g("SQ_INSTS_LDS", 0) / g("SQ_INSTS_MFMA")))
    if g("GRBM_GUI_ACTIVE") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("kernel cycles %.0f; MFMA busy share of SIMD cycles %.3f" % (g("GRBM_GUI_ACTIVE") / 8, g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") * 128.0)))
    if g("SQC_ICACHE_REQ"):
        print("instruction cache: hits %.4f misses %.4f (dup %.4f) of requests" % (g("SQC_ICACHE_HITS", 0) / g("SQC_ICACHE_REQ"), g("SQC_ICACHE_MISSES", 0) / g("SQC_ICACHE_REQ"),
              g("SQC_ICACHE_MISSES_DUPLICATE", 0) / g("SQC_ICACHE_REQ")))
    if g("SQ_LDS_IDX_ACTIVE"):
        print("LDS bank conflict share %.3f" % (g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        # rocprofv3 reports KiB; gfx950's FETCH_SIZE counts half of the bytes (MI355X_MICROARCH.md, calibrated in profiles/r01_fetch_calibration.txt)
        print("HBM traffic per launch: %.2f MB (2 x FETCH_SIZE %.2f + WRITE_SIZE %.2f); per molecule %.0f B" % (
            (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e6, 2 * g("FETCH_SIZE") * 1024 / 1e6, g("WRITE_SIZE") * 1024 / 1e6,
            (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / mol))
    if out:
        with open(out, "w") as f:
            json.dump({"molecules": mol, "lib": lib, "counters_per_launch": vals}, f, indent=1)


if __name__ == "__main__":
    main()
