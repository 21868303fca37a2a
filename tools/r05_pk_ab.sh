#!/bin/bash
# Same-box A/B of the packed float32 algebra (round 5): libpgps_nopk.so (-DPGPS_PK_F32=0 on the float32 lane-chunk units) against
# the shipped library, interleaved; one line per run.  The variant:
#   make -C parallel-gps_amd/csrc variant NAME=nopk UNIT="inst_f32_2 ... inst_f32_6 instn_f32_2 ... instn_f32_6" EXTRA=-DPGPS_PK_F32=0
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
A=$R/parallel-gps_amd/pssgp/libpgps_nopk.so
B=$R/parallel-gps_amd/pssgp/libpgps.so
C="--no-cpu-baseline --main-only"
one() { lib=$1; shift; PGPS_LIB=$lib python bench.py "$@" $C 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('  %-8s %.4f ms/pass  %s' % (sys.argv[1], j['ms_per_step'], {k: round(v*1e3,1) for k,v in j['kernel_ms_per_pass'].items()}))" $(basename $lib .so | sed 's/libpgps_//;s/libpgps/shipped/'); }
ab() { n=$1; shift; echo "== $*"; for i in $(seq 1 $n); do one $A "$@"; one $B "$@"; done; }
ab ${1:-3} --kernel rbf6 --dtype f32 --f32-policy 1 --steps 100 --warmup 10
ab 2 --kernel rbf4 --dtype f32 --f32-policy 1 --steps 100 --warmup 10
ab 2 --kernel matern52 --dtype f32 --f32-policy 1 --steps 100 --warmup 10
