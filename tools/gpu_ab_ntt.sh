R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in base pin99 base pin99; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  echo "== $n"
  TINYNTT_LIB=$L timeout -k 10 120 python $R/tools/gpu_n8192.py > $R/gpurun_out/abn_$n.txt 2>&1; grep "fused" $R/gpurun_out/abn_$n.txt
  TINYNTT_LIB=$L timeout -k 10 120 python $R/tools/gpu_configs.py > $R/gpurun_out/abc_$n.txt 2>&1; grep "fused  \|ntt fused\|intt fused" $R/gpurun_out/abc_$n.txt
done
