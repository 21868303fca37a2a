# A/B of libpgps builds on fp64 RBF models of order 8..10 (row-cooperative kernels): tools/rc_occ_ab.sh <lib> [<lib> ...]
for k in rbf8 rbf9 rbf10; do
for lib in "$@"; do
PGPS_LIB=$PWD/parallel-gps_amd/pssgp/$lib timeout -k 10 120 python bench.py --kernel $k --dtype f64 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('$lib $k', 'ms=%.4f'%j['gpu_event_ms_per_step'], 'chunk', j['chunk'], {k[2:]:round(v,4) for k,v in j['kernel_ms_per_pass'].items() if 'final' not in k})
"
done; done
