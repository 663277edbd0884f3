#!/bin/bash
# Copy the outputs of tools/gpu_final.sh <tag> from gpurun_out/ into profiles/ under the <prefix>_* names.  usage: tools/install_evidence.sh <tag> [prefix, default r3_f]
TAG=$1; P=${2:-r3_f}; F=gpurun_out/$TAG
cp $F/bench.json profiles/${P}_bench.json
cp $F/bench_with_traffic.json profiles/${P}_bench_with_traffic.json
cp $F/bench_cfg2.json profiles/${P}_bench_cfg2.json
cp $F/prof_bench.json profiles/${P}_bench_under_rocprof.json
find $F/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} profiles/${P}_kernel_stats.csv
cp gpurun_out/${TAG}_pmc/summary.txt profiles/${P}_pmc_summary_fused.txt
echo "# second set of passes (tools/gpu_pmc2.sh)" >> profiles/${P}_pmc_summary_fused.txt
cat gpurun_out/${TAG}_pmc2/summary.txt >> profiles/${P}_pmc_summary_fused.txt
grep -E "==|sclk|spin" $F/clock.log | sed 's/GPU\[0\]\t\t: //g; s/=\{5,\}//g; s/fclk.*sclk clock level: 1: /sclk /; s/Power Consumption//; s/Current Socket Graphics Package Power//' > profiles/${P}_clock_power.txt
grep -v amdgpu $F/other_shapes.txt > profiles/${P}_other_shapes.txt
grep "n=8192" $F/n8192.txt >> profiles/${P}_other_shapes.txt
cp $F/traffic_latest.json profiles/traffic_latest.json
python3 - <<PY
import json
j=json.load(open("profiles/${P}_bench.json")); t=json.load(open("profiles/${P}_bench_with_traffic.json")); c=json.load(open("profiles/${P}_bench_cfg2.json"))
print("cfg3:", j["value"], "products/s", j["ms_per_step"], "ms/step kernel", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "build", j["config"]["lib_build_id"])
print("traffic:", t["roofline"]["traffic"], t["roofline"]["traffic"]/t["roofline"]["algorithmic_bytes_per_launch"])
print("cpu:", j["cpu_baseline"]["single_thread_value"], j["cpu_baseline"]["value"], j["cpu_baseline"]["cpu_model"])
print("cfg2:", c["value"], c["ms_per_step"], c["roofline"]["frac"], c["cpu_baseline"]["single_thread_value"], c["cpu_baseline"]["value"])
PY
head -3 profiles/${P}_kernel_stats.csv | cut -c1-60,250-330
