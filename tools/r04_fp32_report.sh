#!/bin/bash
# Round-4: float32 on the reference's dense grid -- errors (tools/fp32_grid_errors.py: policy native vs automatic) and what the
# promoted calls cost (bench.py --grid reference with the three policies), and c3 on BASELINE's grid with the probe in place.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-r04_fp32}
mkdir -p $O
cd $R
python tools/fp32_grid_errors.py m32 m52 rbf6 > $O/fp32_grid_errors.txt 2>&1
for n in 15 20; do
  for pol in 1 0 2; do
    python bench.py --kernel rbf6 --dtype f32 --log2n $n --grid reference --f32-policy $pol --no-cpu-baseline --main-only --steps 50 --warmup 10 > $O/bench_rbf6_f32_2p${n}_reference_policy$pol.json 2> /dev/null
  done
done
python bench.py --kernel rbf6 --dtype f32 --no-cpu-baseline --main-only > $O/bench_c3_auto.json 2> /dev/null
python bench.py --kernel rbf6 --dtype f32 --no-cpu-baseline --main-only --f32-policy 1 > $O/bench_c3_native.json 2> /dev/null
python bench.py --kernel rbf6 --dtype f32 --no-cpu-baseline --main-only > $O/bench_c3_auto_b.json 2> /dev/null
python bench.py --kernel rbf6 --dtype f32 --no-cpu-baseline --main-only --f32-policy 1 > $O/bench_c3_native_b.json 2> /dev/null
python3 - $O <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms_per_step %.4f"%j["ms_per_step"], "promoted", j.get("f32_promoted"))
    except Exception as e: print(f, "unreadable", e)
PY
echo done
