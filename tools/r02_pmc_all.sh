#!/bin/bash
# PMC passes of the three single-GPU configs and the traffic file bench.py reads (run on the GPU box).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/r02_pmc && cp profiles/r02_traffic.json gpurun_out/r02_pmc/r02_traffic.json
bash tools/pmc.sh r02_pmc/c2 && python3 tools/pmc_traffic.py gpurun_out/r02_pmc/c2 matern32_f64_log2n20 gpurun_out/r02_pmc/r02_traffic.json 2.0 > /dev/null && \
bash tools/pmc.sh r02_pmc/c3 --kernel rbf6 --dtype f32 && python3 tools/pmc_traffic.py gpurun_out/r02_pmc/c3 rbf6_f32_log2n20 gpurun_out/r02_pmc/r02_traffic.json 1.22 > /dev/null && \
bash tools/pmc.sh r02_pmc/c5 --kernel c5 && python3 tools/pmc_traffic.py gpurun_out/r02_pmc/c5 c5_f64_log2n20 gpurun_out/r02_pmc/r02_traffic.json 2.0 && echo pmc-done
