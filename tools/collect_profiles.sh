#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ (run on the GPU box through gpurun; results come back in gpurun_out/profiles_r02).
#   bash tools/collect_profiles.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() { python3 - "$1" "$2" <<'PY'
import glob, os, shutil, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize)
shutil.copyfile(fs[-1], sys.argv[2])
PY
}
# 1. ONE large launch with the GPU to itself: flops / average duration / peak IS the fraction
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/big -- python3 $R/bench.py --molecules 65536 --depth 1 --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $OUT/r02_big_launch_line.json 2> $OUT/big.err
stats $OUT/big $OUT/r02_big_launch_kernel_stats.csv
echo "big launch done"
# 2. the default bench command (six batches in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/r02_bench_line_under_rocprof.json 2> $OUT/bench.err
stats $OUT/bench $OUT/r02_bench_kernel_stats.csv
echo "bench done"
# 3. the driver's command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/r02_bench20_line_under_rocprof.json 2> $OUT/bench20.err
stats $OUT/bench20 $OUT/r02_bench20_kernel_stats.csv
# 4. PMC passes (bench.py --pmc runs them as children before touching the GPU) + the un-profiled lines
cd $R
# (the 20-step line first: right after counter passes its 2 ms timed region once read 111 instead of 194 M atoms/s, and
#  right after the traced runs above 186 M; four runs in fresh processes on an otherwise idle box: 191-196 M)
python3 bench.py --steps 20 --warmup 5 > $OUT/r02_bench20_line.json 2> $OUT/pmc.err
python3 bench.py --pmc > $OUT/r02_bench_line.json 2>> $OUT/pmc.err
cp profiles/r02_pmc_bench.json $OUT/ 2>/dev/null
echo "pmc done"
cd /tmp
# 5. protein / 100k box / dense entry / train step
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prot -- python3 $R/tools/bench_large.py protein 20 > $OUT/r02_protein.txt 2> $OUT/prot.err
stats $OUT/prot $OUT/r02_protein_kernel_stats.csv
python3 $R/tools/bench_large.py protein 50 2>/dev/null | grep "protein:" > $OUT/r02_large_systems.txt
python3 $R/tools/bench_large.py box100k 3 2>/dev/null | grep "box100k:" >> $OUT/r02_large_systems.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dense -- python3 $R/tools/bench_dense.py > $OUT/r02_dense_entry.txt 2> $OUT/dense.err
stats $OUT/dense $OUT/r02_dense_entry_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $R/tools/bench_train.py > $OUT/r02_train_step.txt 2> $OUT/train.err
stats $OUT/train $OUT/r02_train_kernel_stats.csv
rm -rf $OUT/big $OUT/bench $OUT/bench20 $OUT/prot $OUT/dense $OUT/train
ls -la $OUT
