#!/bin/bash
# c3 (RBF order 6, float32, 2^20 steps): the automatic float32 policy (device probe in front of every smoother call) against
# policy 1 (no probe), interleaved on one box; and the promoted pass at 2^15 steps on the reference's dense grid.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for i in 1 2 3; do
  for pol in 0 1; do
    python bench.py --kernel rbf6 --dtype f32 --f32-policy $pol --no-cpu-baseline --main-only --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('c3 policy $pol round $i: %.4f ms/pass (gpu %.4f) promoted=%s' % (j['ms_per_step'], j['gpu_event_ms_per_step'], j['f32_promoted']))"
  done
done
for i in 1 2; do
  python bench.py --kernel rbf6 --dtype f32 --grid reference --log2n 15 --no-cpu-baseline --main-only --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('rbf6 f32 2^15 reference grid (automatic): %.4f ms/pass promoted=%s' % (j['ms_per_step'], j['f32_promoted']))"
  python bench.py --kernel rbf6 --dtype f32 --grid reference --log2n 15 --f32-policy 2 --no-cpu-baseline --main-only --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('rbf6 f32 2^15 reference grid (policy 2): %.4f ms/pass promoted=%s' % (j['ms_per_step'], j['f32_promoted']))"
done
