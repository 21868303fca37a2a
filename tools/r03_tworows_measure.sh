#!/bin/bash
# Round-3 measurements of the d = 17..32 family after the two-rows level-1 kernels (csrc/pgps_rc2.hip.h): the same
# bench lines through the two-rows kernels (default) and through the LDS-tile kernels (PGPS_WC_ROWS2=0), the rocprofv3
# kernel statistics of the CO2 (d = 18) pass, and the CO2 experiment's own call sizes.  Run on the GPU box:
#   bash tools/r03_tworows_measure.sh        -> gpurun_out/r03_tworows/   (copy what is to be judged into profiles/)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_tworows
mkdir -p $O
export TMPDIR=/tmp
run() { # name rows2 args...
  local name=$1 rows=$2; shift 2
  PGPS_WC_ROWS2=$rows timeout -k 10 280 python3 $R/bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "bench $name failed"; tail -n 3 $O/bench_$name.err; return 1; }
  python3 - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); k=r["kernel_ms_per_pass"]
print("%-34s %8.3f ms  %s  ll_rel %.1e" % (sys.argv[2], r["ms_per_step"], {a:round(b,3) for a,b in k.items()}, r.get("parity_vs_cpu_oracle",{}).get("ll_rel", float("nan"))))
PY
}
run co2_d18_2p17_tworows 15 --kernel co2 --log2n 17 || exit 1
run co2_d18_2p17_ldstiles 0 --kernel co2 --log2n 17 --no-cpu-baseline || exit 1
run co2_d18_2p17_f32_tworows 15 --kernel co2 --log2n 17 --dtype f32 --no-cpu-baseline || exit 1
run co2_d18_2p17_f32_ldstiles 0 --kernel co2 --log2n 17 --dtype f32 --no-cpu-baseline || exit 1
run periodic10_d22_2p17_tworows 15 --kernel periodic10 --log2n 17 --no-cpu-baseline || exit 1
run periodic10_d22_2p17_ldstiles 0 --kernel periodic10 --log2n 17 --no-cpu-baseline || exit 1
run rbf32_d32_2p15_tworows 15 --kernel rbf32 --log2n 15 --no-cpu-baseline || exit 1
run rbf32_d32_2p15_ldstiles 0 --kernel rbf32 --log2n 15 --no-cpu-baseline || exit 1
run co2_d18_2p20_tworows 15 --kernel co2 --log2n 20 --no-cpu-baseline --steps 10 --warmup 2 || exit 1
# per-kernel durations of the CO2 pass
rm -rf $O/kt_co2
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_co2 -- python3 $R/bench.py --kernel co2 --log2n 17 --no-cpu-baseline --steps 50 --warmup 10 > $O/bench_co2_under_rocprof.json 2> /dev/null
python3 - $O <<'PY'
import csv,glob,sys,os
O=sys.argv[1]
fs=glob.glob(os.path.join(O,"kt_co2","**","*kernel_stats.csv"),recursive=True)
if fs:
    rows=list(csv.DictReader(open(fs[0])))
    with open(os.path.join(O,"co2_d18_tworows_kernel_stats.txt"),"w") as out:
        out.write("%-96s %7s %12s %12s %12s %8s\n"%("kernel (rocprofv3 --kernel-trace --stats)","calls","avg_us","min_us","max_us","pct"))
        for r in rows[:18]:
            out.write("%-96s %7s %12.1f %12.1f %12.1f %8s\n"%(r["Name"][:96],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3,r["Percentage"]))
    print(open(os.path.join(O,"co2_d18_tworows_kernel_stats.txt")).read())
PY
# the CO2 experiment's own sizes (3 192 points) and 2^17
timeout -k 10 280 python3 $R/tools/co2_d18_timing.py > $O/co2_timing.txt 2>&1; tail -n 6 $O/co2_timing.txt
echo done
