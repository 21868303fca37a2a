#!/bin/bash
# two-rows vs LDS-tile level-1 kernels per kernel and state dimension (PGPS_WC_ROWS2 mask: 1 reduce, 2 apply without and
# 4 with the smoothing total, 8 smoother), f64 and f32, RBF orders 24..32 at 2^16 steps
mkdir -p gpurun_out/cross
for dt in f64 f32; do for d in 18 20 22 24 28 32; do for mask in 15 11 0; do
  PGPS_WC_ROWS2=$mask timeout -k 10 120 python3 bench.py --kernel rbf$d --log2n 16 --dtype $dt --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/cross/r_${dt}_${d}_$mask.json 2>/dev/null || { echo fail $dt $d $mask; exit 1; }
  python3 - gpurun_out/cross/r_${dt}_${d}_$mask.json $dt $d $mask <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); k=r["kernel_ms_per_pass"]
print(sys.argv[2], "d=%s mask=%2s"%(sys.argv[3],sys.argv[4]), "%7.3f ms"%r["ms_per_step"], {a:round(b,3) for a,b in k.items() if a!="k_ll_finalize"})
PY
done; done; done
