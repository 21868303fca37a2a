#!/bin/bash
# Round-2 closing batch (GPU box): bench lines + rocprofv3 kernel stats (r02_measure.sh), then the PMC passes of c2 / c3 / c5.
R=${GRAFT_REPO_ROOT:-$PWD}
bash $R/tools/r02_measure.sh && bash $R/tools/r02_pmc_all.sh
