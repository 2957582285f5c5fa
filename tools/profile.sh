#!/bin/bash
# rocprofv3 passes of the bench command -> profiles/<prefix>_{kernel_stats.csv,pmc.json,bench.json}
#   bash tools/profile.sh r02_v1 [bench args...]
# (run on the GPU box through gpurun; separate passes for the kernel trace and
# for each PMC counter, as MI355X_MICROARCH.md prescribes)
set -e
PREFIX=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof_$PREFIX
rm -rf $OUT; mkdir -p $OUT $R/profiles
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-wallclock $@"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/ktr -o ktr -- $CMD > $OUT/short.json 2> /dev/null
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $CMD > /dev/null 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o write -- $CMD > /dev/null 2>&1
cd $R
WL=$(python -c "import json,sys; print(json.loads([l for l in open('$OUT/short.json') if l.startswith('{')][-1])['config']['workload'])")
python tools/prof_summary.py $OUT profiles/$PREFIX "$WL"
python tools/step_timeline.py $OUT/ktr profiles/${PREFIX}_step_timeline.txt || true
# the bench line LAST: its roofline.traffic comes from the PMC summary just written
# (same sources digest, same workload)
cd /tmp
python $R/bench.py --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $R
cp $OUT/bench.json profiles/${PREFIX}_bench.json
mkdir -p gpurun_out/profiles_out && cp profiles/${PREFIX}_* gpurun_out/profiles_out/
