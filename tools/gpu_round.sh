#!/bin/bash
# One GPU-box round: parity tests, bench line, rocprofv3 kernel-trace summary.  Usage: tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
tos 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
tos 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
export TMPDIR=/tmp
cd /tmp
tos 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat | head -12
