#!/bin/bash
# End-of-round evidence: tools/gpu_final_a.sh <tag> then tools/gpu_final_b.sh <tag> (each sized for one gpurun call of <= 20 minutes).
# Outputs under gpurun_out/<tag>/ and gpurun_out/<tag>_{pmc,pmc2,cg}/; tools/install_evidence.sh copies them into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
"$R/tools/gpu_final_a.sh" "$@" && "$R/tools/gpu_final_b.sh" "$@"
