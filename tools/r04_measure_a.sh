#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_a
mkdir -p $O
cd $R
python tools/grad_cost.py > $O/grad_cost.txt 2>&1
python tools/small_n_latency.py > $O/small_n_latency.txt 2>&1
for n in 20 21 24; do
  st=100; [ $n = 24 ] && st=30
  python bench.py --log2n $n --steps $st --warmup 10 --no-cpu-baseline --main-only > $O/bench_c2_2p${n}.json 2>/dev/null
  python bench.py --log2n $n --steps $st --warmup 10 --no-cpu-baseline --main-only --single-pass 1 > $O/bench_c2_2p${n}_single.json 2>/dev/null
  python bench.py --log2n $n --steps $st --warmup 10 --no-cpu-baseline --main-only > $O/bench_c2_2p${n}_b.json 2>/dev/null
  python bench.py --log2n $n --steps $st --warmup 10 --no-cpu-baseline --main-only --single-pass 1 > $O/bench_c2_2p${n}_single_b.json 2>/dev/null
done
python3 - $O <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms_per_step %.4f"%j["ms_per_step"], "whole_path_frac %.3f"%j["roofline"]["whole_path_frac"], {k: round(v["ms_per_pass"]*1e3,1) for k,v in j["roofline"]["slots"].items()})
    except Exception as e: print(f, "unreadable", e)
PY
echo done
