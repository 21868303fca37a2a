#!/bin/bash
# Round-4: kernel traces of single evaluations at the reference's sizes (tools/trace_small.py), one rocprofv3 run per scenario.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-r04_trace}
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sc in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$sc -- python3 $R/tools/trace_small.py $sc 20 > $O/$sc.txt 2> $O/$sc.err
  python3 - "$O/kt_$sc" "$O/${sc}_kernel_stats.txt" <<'PY'
import csv,glob,sys
fs=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)
with open(sys.argv[2],"w") as out:
    out.write("%-100s %7s %10s %10s %8s\n"%("kernel (rocprofv3 --kernel-trace --stats; 23 calls of the scenario)","calls","avg_us","max_us","pct"))
    if fs:
        for r in csv.DictReader(open(fs[0])):
            out.write("%-100s %7s %10.1f %10.1f %8s\n"%(r["Name"][:100],r["Calls"],float(r["AverageNs"])/1e3,float(r["MaxNs"])/1e3,r["Percentage"]))
PY
  cat $O/$sc.txt
done
echo done
