#!/bin/bash
# Same-box A/B of the last wave-local transpose through DPP lane permutes (TN_SHUFFLE_LAST=1) against LDS (shipped): timing, then the
# instruction counters of both in separate rocprofv3 --pmc passes.  usage: tools/gpu_shuffle_ab.sh  (needs libtinyntt_shuffle.so: build_variant.sh shuffle -DTN_SHUFFLE_LAST=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
. "$R/tools/gpu_lib.sh"
OUT=$R/gpurun_out/shuffle_ab; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for n in base shuffle base4 shuffle4 base shuffle base4 shuffle4; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  TINYNTT_LIB=$L tos 120 python3 $R/tests/dev/gpu_speed.py 65536 fused > $OUT/time_$n.txt 2>&1
  echo "== $n"; grep -E "fused:|checksum|oracle" $OUT/time_$n.txt | tr "\n" " "; echo
done
for n in base shuffle shuffle4; do
  if [ "$n" = base ]; then L=$R/tiny_ntt_amd/lib/libtinyntt.so; else L=$R/tiny_ntt_amd/lib/libtinyntt_$n.so; fi
  TINYNTT_LIB=$L tos 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_$n -- python3 $R/tests/dev/gpu_speed.py 65536 fused > $OUT/pmc_$n.log 2>&1
  echo "== pmc $n"; python3 $R/tools/pmc_summary.py $OUT/pmc_$n | grep polymul
done
