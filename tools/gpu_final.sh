#!/bin/bash
# End-of-round evidence in one GPU call: parity tests, bench lines, rocprofv3 kernel trace, PMC passes, traffic file, clocks.
# usage: tools/gpu_final.sh <tag>     (outputs under gpurun_out/<tag>/ and gpurun_out/<tag>_pmc*/; traffic json -> gpurun_out/<tag>/)
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
# round 3: the constant-geometry sweep (timing + PMC, 11 points), its phase stamps (diagnostic build, if present), the fused kernel's
# in-kernel clock (diagnostic build, if present), two more seeds of the wide parity fuzz
sub tools/gpu_cg_profile.sh ${TAG}_cg > gpurun_out/${TAG}_cg.log 2>&1; python3 tools/make_cg_table.py gpurun_out/${TAG}_cg > gpurun_out/$TAG/cg_table.txt 2>&1; cat gpurun_out/$TAG/cg_table.txt
if [ -f tiny_ntt_amd/lib/libtinyntt_stamps.so ]; then TINYNTT_LIB=$R/tiny_ntt_amd/lib/libtinyntt_stamps.so tos 200 python tools/gpu_cg_stamps.py 65536 cg8_padded cg8_swizzled > gpurun_out/$TAG/cg_stamps.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/cg_stamps.txt; fi
if [ -f tiny_ntt_amd/lib/libtinyntt_fstamps.so ]; then TINYNTT_LIB=$R/tiny_ntt_amd/lib/libtinyntt_fstamps.so tos 200 python tools/gpu_fused_clock.py > gpurun_out/$TAG/fused_clock.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/fused_clock.txt; fi
tos 200 python tools/gpu_latency.py > gpurun_out/$TAG/latency.txt 2>&1
for seed in 3 4; do tos 330 python tests/dev/gpu_fuzz.py $seed 300 > gpurun_out/$TAG/fuzz_$seed.txt 2>&1; tail -1 gpurun_out/$TAG/fuzz_$seed.txt; done
