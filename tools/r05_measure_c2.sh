#!/bin/bash
# Round-5 c2 batch (GPU box, through gpurun): PMC passes of the array road and of the fused road (traffic stamped with the
# kernel sources' hash), bench lines under the driver's protocol and the 200-step one, rocprofv3 kernel stats.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_c2
mkdir -p $O
cd $R
[ -f $R/profiles/r05_traffic.json ] || echo "{}" > $R/profiles/r05_traffic.json
bash tools/pmc.sh pmc_c2_r05 && python3 tools/pmc_traffic.py gpurun_out/pmc_c2_r05 matern32_f64_log2n20 $R/profiles/r05_traffic.json && cp gpurun_out/pmc_c2_r05/summary.txt $O/c2_pmc_summary.txt
bash tools/pmc.sh pmc_c2f_r05 --path fused && python3 tools/pmc_traffic.py gpurun_out/pmc_c2f_r05 matern32_f64_log2n20 $R/profiles/r05_traffic.json 2.0 k_pkfs_resident=fused_path && cp gpurun_out/pmc_c2f_r05/summary.txt $O/c2_fused_pmc_summary.txt
cp $R/profiles/r05_traffic.json $O/r05_traffic.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_protocol.json 2> $O/bench_c2.err
python bench.py > $O/bench_c2.json 2>> $O/bench_c2.err
python bench.py --resident 0 --no-cpu-baseline > $O/bench_c2_three_launches.json 2>> $O/bench_c2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2 -- python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 > $O/bench_c2_under_rocprof.json 2> /dev/null
python3 - "$O/kt_c2" "$O/c2_kernel_stats.txt" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
with open(sys.argv[2],"w") as out:
    out.write("%-96s %7s %12s %12s %12s %8s\n"%("kernel (rocprofv3 --kernel-trace --stats)","calls","avg_us","min_us","max_us","pct"))
    for r in csv.DictReader(open(f)):
        out.write("%-96s %7s %12.1f %12.1f %12.1f %8s\n"%(r["Name"][:96],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3,r["Percentage"]))
PY
rm -rf $O/kt_c2
echo done
