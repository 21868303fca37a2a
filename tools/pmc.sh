#!/bin/bash
# PMC passes for the bench workload (run on the GPU box via gpurun). Usage: tools/pmc.sh <outdir> [bench args...]
# Counters in their own runs, --kernel-trace only (never with --sys-trace / hip / marker domains: refused on this pool).
out=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$out
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/$out/p$i -- python3 $R/bench.py --steps 5 --warmup 2 --rewarm-ms 0 --no-cpu-baseline --main-only --event-every 0 "$@" > $R/gpurun_out/$out/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - "$R/gpurun_out/$out" <<'PY'
import csv, glob, sys, collections
root=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "pgps" not in k: continue
        k=k.split("(")[0].replace("void pgps::","")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(root+"/summary.txt","w") as out:
    for k,v in agg.items():
        line=k+"\n"+"\n".join("   %-32s mean %.6g  (n=%d)"%(c,sum(x)/len(x),len(x)) for c,x in sorted(v.items()))
        out.write(line+"\n")
print("summary written")
PY
