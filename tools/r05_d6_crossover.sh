#!/bin/bash
# d = 6 float32, whole series: lane-chunk (family 1, with the forgetting shortcut of round 5) against quad-cooperative
# (family 4) kernels per series length -- the measurement behind choose_family's rule for d = 6 (csrc/pgps_core.hip).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for L in 14 16 17 18 19 20 21 22; do
  line="2^$L"
  for F in 1 4; do
    ms=$(python bench.py --kernel rbf6 --dtype f32 --f32-policy 1 --log2n $L --family $F --no-cpu-baseline --main-only --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    line="$line  family $F: $ms ms"
  done
  echo "$line"
done
