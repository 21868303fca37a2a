#!/bin/bash
# Round-5 measurement batch (GPU box, through gpurun): PMC passes first (traffic stamped with the kernel sources' hash), then
# the bench lines of every BASELINE config, rocprofv3 kernel stats, the resident launch's phase stamps.  Progress goes to
# gpurun_out/r05_final/progress.txt (a line per item: the box's silence watchdog reads it as life).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
say() { echo "$(date +%H:%M:%S) $*" | tee -a $O/progress.txt; }
echo "{}" > $R/profiles/r05_traffic.json
say pmc c2;  bash tools/pmc.sh pmc_c2_r05 && python3 tools/pmc_traffic.py gpurun_out/pmc_c2_r05 matern32_f64_log2n20 $R/profiles/r05_traffic.json && cp gpurun_out/pmc_c2_r05/summary.txt $O/c2_pmc_summary.txt
say pmc c2 fused; bash tools/pmc.sh pmc_c2f_r05 --path fused && python3 tools/pmc_traffic.py gpurun_out/pmc_c2f_r05 matern32_f64_log2n20 $R/profiles/r05_traffic.json 2.0 k_pkfs_resident=fused_path && cp gpurun_out/pmc_c2f_r05/summary.txt $O/c2_fused_pmc_summary.txt
say pmc c3;  bash tools/pmc.sh pmc_c3_r05 --kernel rbf6 --dtype f32 && python3 tools/pmc_traffic.py gpurun_out/pmc_c3_r05 rbf6_f32_log2n20 $R/profiles/r05_traffic.json 1.22 && cp gpurun_out/pmc_c3_r05/summary.txt $O/c3_pmc_summary.txt
say pmc c5;  bash tools/pmc.sh pmc_c5_r05 --kernel c5 && python3 tools/pmc_traffic.py gpurun_out/pmc_c5_r05 c5_f64_log2n20 $R/profiles/r05_traffic.json && cp gpurun_out/pmc_c5_r05/summary.txt $O/c5_pmc_summary.txt
cp $R/profiles/r05_traffic.json $O/r05_traffic.json
say bench c2 driver protocol; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c2_driver_protocol.json 2> $O/bench_c2.err
say bench c2;            python bench.py > $O/bench_c2.json 2>> $O/bench_c2.err
say bench c2 three;      python bench.py --resident 0 --no-cpu-baseline > $O/bench_c2_three_launches.json 2>> $O/bench_c2.err
say bench c3;            python bench.py --kernel rbf6 --dtype f32 > $O/bench_c3.json 2> $O/bench_c3.err
say bench c5;            python bench.py --kernel c5 --steps 50 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err
say bench 2p24;          python bench.py --log2n 24 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_c2_2p24.json 2> /dev/null
say bench 2p21;          python bench.py --log2n 21 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c2_2p21.json 2> /dev/null
say bench 2p19;          python bench.py --log2n 19 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c2_2p19.json 2> /dev/null
say bench segments;      python bench.py --force-segments --no-cpu-baseline > $O/bench_c2_segments_rccl_world1.json 2> $O/bench_seg.err
say bench co2;           python bench.py --kernel co2 --log2n 17 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_co2_d18_2p17.json 2> /dev/null
say bench gpus2 dry run; python bench.py --gpus 2 --all-on-gpu0 --steps 10 --warmup 3 > $O/bench_gpus2_dryrun_gloo.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" >> $O/bench_gpus2.err
say stamps;              python tools/res_check.py --time > $O/resident_phase_stamps.txt 2>&1
say small n latency;      python tools/small_n_latency.py > $O/small_n_latency.txt 2>&1
say grad cost;            python tools/grad_cost.py > $O/grad_cost.txt 2>&1
say probe ab;            bash tools/r05_probe_ab.sh > $O/probe_ab.txt 2>&1
say d6 crossover;        bash tools/r05_d6_crossover.sh > $O/d6_crossover.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for cfg in "c2:" "c3:--kernel rbf6 --dtype f32" "c5:--kernel c5"; do
  name=${cfg%%:*}; args=${cfg#*:}
  say kernel trace $name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python3 $R/bench.py $args --no-cpu-baseline --steps 50 --warmup 10 > $O/bench_${name}_under_rocprof.json 2> /dev/null
  python3 - "$O/kt_$name" "$O/${name}_kernel_stats.txt" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
with open(sys.argv[2],"w") as out:
    out.write("%-96s %7s %12s %12s %12s %8s\n"%("kernel (rocprofv3 --kernel-trace --stats)","calls","avg_us","min_us","max_us","pct"))
    for r in csv.DictReader(open(f)):
        out.write("%-96s %7s %12.1f %12.1f %12.1f %8s\n"%(r["Name"][:96],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3,r["Percentage"]))
PY
  rm -rf $O/kt_$name
done
say done
