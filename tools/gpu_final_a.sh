#!/bin/bash
# First half of tools/gpu_final.sh (one gpurun call): tests, bench lines, kernel trace, PMC passes, traffic file, clocks, other shapes.
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
cd $R
tools/gpu_round.sh $TAG || exit 1
sub tools/gpu_pmc.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log
sub tools/gpu_pmc2.sh ${TAG}_pmc2 > gpurun_out/${TAG}_pmc2.log 2>&1
python3 tools/make_traffic_json.py gpurun_out/${TAG}_pmc > gpurun_out/$TAG/traffic.log 2>&1 && cp profiles/traffic_latest.json gpurun_out/$TAG/traffic_latest.json
cat gpurun_out/$TAG/traffic.log
sub tools/gpu_clock_probe.sh base > gpurun_out/$TAG/clock.log 2>&1; grep -c sclk gpurun_out/$TAG/clock.log
tos 300 python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_with_traffic.json 2>> gpurun_out/$TAG/bench.err; cat gpurun_out/$TAG/bench_with_traffic.json
tos 300 python bench.py --config cfg2 > gpurun_out/$TAG/bench_cfg2.json 2>> gpurun_out/$TAG/bench.err
tos 200 python tools/gpu_configs.py > gpurun_out/$TAG/other_shapes.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/other_shapes.txt | tail -20
tos 200 python tools/gpu_n8192.py > gpurun_out/$TAG/n8192.txt 2>&1
