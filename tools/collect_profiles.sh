#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ (run on the GPU box through gpurun; results come back in gpurun_out/profiles_r05).
#   bash tools/collect_profiles.sh [1|2|3]      (three parts, each well inside one gpurun call; no argument: all)
set -o pipefail
PART=${1:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() { python3 - "$1" "$2" <<'PY'
import glob, os, shutil, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize)
shutil.copyfile(fs[-1], sys.argv[2])
PY
}
if [ "$PART" = 0 ] || [ "$PART" = 1 ]; then
# 1. ONE large launch with the GPU to itself: flops / average duration / peak IS the fraction.  The JSON line and the CSV
#    describe the SAME command (65536 molecules per launch, depth 1): roofline.algorithmic_gflop_per_launch of the line
#    / AverageNs of k_wave_forward in the CSV / 157.3 = the single-launch fraction
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/big -- python3 $R/bench.py --molecules 65536 --depth 1 --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $OUT/r05_big_launch_line.json 2> $OUT/big.err
stats $OUT/big $OUT/r05_big_launch_kernel_stats.csv
echo "big launch done"
# 2. the default bench command (eight batches in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/r05_bench_line_under_rocprof.json 2> $OUT/bench.err
stats $OUT/bench $OUT/r05_bench_kernel_stats.csv
echo "bench done"
# 3. the driver's command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/r05_bench20_line_under_rocprof.json 2> $OUT/bench20.err
stats $OUT/bench20 $OUT/r05_bench20_kernel_stats.csv
# 4. PMC passes (bench.py --pmc runs them as children before touching the GPU) + the un-profiled lines
cd $R
python3 bench.py --pmc > $OUT/r05_bench_line.json 2> $OUT/pmc.err       # (first: it rewrites profiles/r05_pmc_bench.json, which the next line quotes)
python3 bench.py --steps 20 --warmup 5 > $OUT/r05_bench20_line.json 2>> $OUT/pmc.err
cp profiles/r05_pmc_bench.json $OUT/ 2>/dev/null
echo "pmc done"
fi
if [ "$PART" = 0 ] || [ "$PART" = 2 ]; then
cd /tmp
# 5. protein / 100k box / train step / sweep microbenchmark
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prot -- python3 $R/tools/bench_large.py protein 20 > $OUT/r05_protein.txt 2> $OUT/prot.err
stats $OUT/prot $OUT/r05_protein_kernel_stats.csv
python3 $R/tools/large_timeline.py $OUT/prot 3 > $OUT/r05_protein_timeline.txt 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/protpmc -- python3 $R/tools/bench_large.py protein 5 > /dev/null 2>> $OUT/prot.err
python3 - $OUT/protpmc $OUT/r05_protein_sweep_pmc.txt <<'PY'
import glob, os, sys
import pandas as pd
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
df = pd.read_csv(f)
df = df[df.Kernel_Name.str.contains("k_lg_sweep")]
with open(sys.argv[2], "w") as out:
    out.write("rocprofv3 --pmc ... -- python3 tools/bench_large.py protein 5: k_lg_sweep, mean per launch over %d launches\n" % (len(df) // max(1, df.Counter_Name.nunique())))
    out.write(df.groupby("Counter_Name").Counter_Value.mean().to_string() + "\n")
PY
python3 $R/tools/bench_large.py protein 200 2>/dev/null | grep "protein:" | sed 's/^/[200 forwards] /' > $OUT/r05_large_systems.txt
python3 $R/tools/bench_large.py protein 50 2>/dev/null | grep "protein:" >> $OUT/r05_large_systems.txt
python3 $R/tools/bench_large.py protein 200 --opt=forward_ahead:0 2>/dev/null | grep "protein:" | sed 's/^/[200 forwards, forward_ahead=0] /' >> $OUT/r05_large_systems.txt
python3 $R/tools/bench_large.py protein 50 --opt=large_dedupe:0 2>/dev/null | grep "protein:" | sed 's/^/[large_dedupe=0] /' >> $OUT/r05_large_systems.txt
python3 $R/tools/bench_large.py protein 50 --opt=large_merge:0 2>/dev/null | grep "protein:" | sed 's/^/[large_merge=0] /' >> $OUT/r05_large_systems.txt
python3 $R/tools/bench_large.py box10k 10 2>/dev/null | grep "box10k:" >> $OUT/r05_large_systems.txt
python3 $R/tools/bench_large.py box100k 3 2>/dev/null | grep "box100k:" >> $OUT/r05_large_systems.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $R/tools/bench_train.py 1 --mode=1 > $OUT/r05_train_step.txt 2> $OUT/train.err
stats $OUT/train $OUT/r05_train_kernel_stats.csv
python3 $R/tools/bench_train.py 1 2>/dev/null | grep -v "^{" >> $OUT/r05_train_step.txt
python3 $R/tools/bench_train.py 8 2>/dev/null >> $OUT/r05_train_step.txt
python3 $R/tools/train_clocks.py 1 > $OUT/r05_train_clocks.txt 2>/dev/null
python3 $R/tools/large_clocks.py > $OUT/r05_large_clocks.txt 2>/dev/null
(cd $R/tools/micro && ./sweep_mix > $OUT/r05_micro_sweep_mix.txt 2>&1)
(cd $R/tools/micro && ./grid_barrier > $OUT/r05_micro_grid_barrier.txt 2>&1)
python3 $R/tools/bench_dense_latency.py > $OUT/r05_dense_latency.txt 2>/dev/null
fi
if [ "$PART" = 0 ] || [ "$PART" = 3 ]; then
cd /tmp
# 6. where the fused kernel's wavefronts spend their cycles, its SQ counters for one large launch, the sweep's clocks on the protein
python3 $R/tools/wave_clocks.py --copies 8 > $OUT/r05_wave_clocks.txt 2>/dev/null
python3 $R/tools/wave_clocks.py --copies 1 > $OUT/r05_wave_clocks_one_per_simd.txt 2>/dev/null
python3 $R/tools/pmc_wave.py --molecules 8192 --out $OUT/r05_pmc_wave_8192.json > $OUT/r05_pmc_wave_8192.txt 2>/dev/null
rm -rf $R/gpurun_out/pmc_wave
python3 $R/tools/large_clocks.py --forwards 5 --detail 2>/dev/null | grep -A6 "k_lg_sweep" > $OUT/r05_protein_sweep_clocks.txt
python3 $R/tools/large_clocks.py --forwards 200 --detail 2>/dev/null | grep -A6 "k_lg_sweep" >> $OUT/r05_protein_sweep_clocks.txt
(cd $R/tools/micro && ./clamp_relu > $OUT/r05_micro_clamp_relu.txt 2>&1; ./clock_cal > $OUT/r05_micro_clock_cal.txt 2>&1; ./issue_mix > $OUT/r05_micro_issue_mix.txt 2>&1; ./bf16x6 > $OUT/r05_micro_bf16x6.txt 2>&1)
python3 $R/tools/bench_mixed.py > $OUT/r05_mixed_val.txt 2>/dev/null
fi
rm -rf $OUT/big $OUT/bench $OUT/bench20 $OUT/prot $OUT/protpmc $OUT/train
ls -la $OUT
