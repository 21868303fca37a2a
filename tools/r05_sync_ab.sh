#!/bin/bash
# Effect of the wavefront-scope LDS syncs (round 5) on the configs that run the lane-chunk / row- / quad-cooperative kernels.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() { python bench.py "$@" --no-cpu-baseline --main-only 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('%-60s %.4f ms/pass (gpu %.4f)  %s' % (' '.join(sys.argv[1:]), j['ms_per_step'], j['gpu_event_ms_per_step'], {k: round(v*1e3,1) for k,v in j['kernel_ms_per_pass'].items()}))" "$@"; }
for i in 1 2; do
run --kernel rbf6 --dtype f32 --f32-policy 1 --steps 100 --warmup 10
run --kernel c5 --steps 50 --warmup 10
run --kernel rbf8 --dtype f32 --f32-policy 1 --steps 50 --warmup 10
run --resident 0 --steps 200 --warmup 20
run --kernel matern52 --steps 100 --warmup 10
run --log2n 24 --steps 30 --warmup 5
done
