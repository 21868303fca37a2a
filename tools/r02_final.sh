#!/bin/bash
# Round-2 closing batch on the GPU box: the measurement batch, then the PMC passes of c5 (its smoother records changed).
R=${GRAFT_REPO_ROOT:-$PWD}
bash $R/tools/r02_measure.sh && bash $R/tools/pmc.sh pmc_c5_final --kernel c5 && cd $R && \
  cp profiles/r02_traffic.json gpurun_out/r02_final/r02_traffic.json && \
  python3 tools/pmc_traffic.py gpurun_out/pmc_c5_final c5_f64_log2n20 gpurun_out/r02_final/r02_traffic.json && \
  cp gpurun_out/pmc_c5_final/summary.txt gpurun_out/r02_final/c5_pmc_summary.txt && echo final-done
