#!/bin/bash
# Round-4 measurement batch (GPU box, through gpurun): bench lines and rocprofv3 kernel stats per BASELINE config, PMC passes
# (traffic stamped with the kernel sources' hash), the d = 24 .. 32 fp64 configs that moved to the LDS-tile kernels, the
# gradient / latency tables.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
# PMC first: the bench lines below then carry traffic measured on THIS tree's kernels (traffic_stale = false)
cd $R
echo "{}" > $R/profiles/r04_traffic.json
bash tools/pmc.sh pmc_c2_r04 && python3 tools/pmc_traffic.py gpurun_out/pmc_c2_r04 matern32_f64_log2n20 $R/profiles/r04_traffic.json && cp gpurun_out/pmc_c2_r04/summary.txt $O/c2_pmc_summary.txt
bash tools/pmc.sh pmc_c3_r04 --kernel rbf6 --dtype f32 && python3 tools/pmc_traffic.py gpurun_out/pmc_c3_r04 rbf6_f32_log2n20 $R/profiles/r04_traffic.json 1.22 && cp gpurun_out/pmc_c3_r04/summary.txt $O/c3_pmc_summary.txt
bash tools/pmc.sh pmc_c5_r04 --kernel c5 && python3 tools/pmc_traffic.py gpurun_out/pmc_c5_r04 c5_f64_log2n20 $R/profiles/r04_traffic.json && cp gpurun_out/pmc_c5_r04/summary.txt $O/c5_pmc_summary.txt
cp $R/profiles/r04_traffic.json $O/r04_traffic.json
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --kernel rbf6 --dtype f32 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --kernel c5 --steps 50 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --log2n 24 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_c2_2p24.json 2> /dev/null
python bench.py --log2n 21 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c2_2p21.json 2> /dev/null
python bench.py --force-segments --no-cpu-baseline > $O/bench_c2_segments_rccl_world1.json 2> $O/bench_seg.err
python bench.py --kernel co2 --log2n 17 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_co2_d18_2p17.json 2> /dev/null
python bench.py --kernel rbf32 --log2n 15 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_rbf32_d32_2p15_ldstiles_default.json 2> /dev/null
python bench.py --gpus 2 --all-on-gpu0 --steps 10 --warmup 3 > $O/bench_gpus2_dryrun_gloo.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" >> $O/bench_gpus2.err
python tools/grad_cost.py > $O/grad_cost.txt 2>&1
python tools/small_n_latency.py > $O/small_n_latency.txt 2>&1
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
echo done
