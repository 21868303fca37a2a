#!/bin/bash
# Same-box A/B of the round-5 staged-loop order (no load in flight across the back edge; packed float32 algebra) against the
# library as it was before (libpgps_nopk.so = the old loops, scalar float32), interleaved; one line per run.  The rewritten
# loops were NOT kept (profiles/r05_experiments.txt item 9): this script documents how profiles/r05_loop_ab.txt was made.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
A=$R/parallel-gps_amd/pssgp/libpgps_nopk.so
B=$R/parallel-gps_amd/pssgp/libpgps.so
C="--no-cpu-baseline --main-only"
one() { lib=$1; shift; PGPS_LIB=$lib python bench.py "$@" $C 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('  %-8s %.4f ms/pass  %s' % (sys.argv[1], j['ms_per_step'], {k: round(v*1e3,1) for k,v in j['kernel_ms_per_pass'].items()}))" $(basename $lib .so | sed 's/libpgps_nopk/before/;s/libpgps/shipped/'); }
ab() { n=$1; shift; echo "== $*"; for i in $(seq 1 $n); do one $A "$@"; one $B "$@"; done; }
ab 2 --kernel rbf6 --dtype f32 --f32-policy 1 --steps 100 --warmup 10
ab 2 --kernel rbf4 --dtype f32 --f32-policy 1 --steps 100 --warmup 10
ab 2 --kernel matern52 --steps 100 --warmup 10
ab 2 --kernel rbf4 --steps 50 --warmup 10
ab 2 --resident 0 --steps 200 --warmup 20
ab 2 --log2n 24 --steps 30 --warmup 5
ab 1 --kernel rbf6 --steps 30 --warmup 5
