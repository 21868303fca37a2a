#!/bin/bash
# Round-3 measurement batch (run on the GPU box through gpurun): bench lines and rocprofv3 kernel stats per config, then the
# PMC passes of the default command (c2).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03_final
mkdir -p $O
cd $R
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --kernel rbf6 --dtype f32 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --kernel rbf6 --dtype f32 --family 4 --no-cpu-baseline > $O/bench_c3_quad.json 2> /dev/null
python bench.py --kernel c5 --steps 50 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --log2n 24 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_c2_2p24.json 2> /dev/null
python bench.py --log2n 21 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c2_2p21.json 2> /dev/null
python bench.py --force-segments --no-cpu-baseline > $O/bench_c2_segments_rccl_world1.json 2> $O/bench_seg.err
python bench.py --kernel rbf8 --dtype f32 --no-cpu-baseline > $O/bench_rbf8_f32_quad_auto.json 2> /dev/null
python bench.py --kernel rbf8 --dtype f32 --family 3 --no-cpu-baseline > $O/bench_rbf8_f32_rowcoop.json 2> /dev/null
python bench.py --kernel rbf6 --dtype f32 --log2n 18 --no-cpu-baseline > $O/bench_rbf6_f32_2p18_quad_auto.json 2> /dev/null
python bench.py --kernel rbf6 --dtype f32 --log2n 22 --steps 50 --no-cpu-baseline > $O/bench_rbf6_f32_2p22_quad_auto.json 2> /dev/null
python bench.py --gpus 2 --all-on-gpu0 --steps 10 --warmup 3 > $O/bench_gpus2_dryrun_gloo.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" >> $O/bench_gpus2.err
cd /tmp && export TMPDIR=/tmp
for cfg in "c2:" "c3:--kernel rbf6 --dtype f32" "c5:--kernel c5"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python3 $R/bench.py $args --no-cpu-baseline --steps 50 --warmup 10 > $O/bench_${name}_under_rocprof.json 2> /dev/null
  python3 - "$O/kt_$name" "$O/${name}_kernel_stats.txt" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
with open(sys.argv[2],"w") as out:
    out.write("%-96s %7s %12s %12s %12s %8s\n"%("kernel (rocprofv3 --kernel-trace --stats)","calls","avg_us","min_us","max_us","pct"))
    for r in csv.DictReader(open(f)):
        out.write("%-96s %7s %12.1f %12.1f %12.1f %8s\n"%(r["Name"][:96],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3,r["Percentage"]))
PY
done
cd $R && bash tools/pmc.sh pmc_c2_r03 && cp profiles/r02_traffic.json $O/r03_traffic.json && \
  python3 tools/pmc_traffic.py gpurun_out/pmc_c2_r03 matern32_f64_log2n20 $O/r03_traffic.json && cp gpurun_out/pmc_c2_r03/summary.txt $O/c2_pmc_summary.txt
echo done
