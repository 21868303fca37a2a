# A/B of builds of libpgps on the row-cooperative configs: tools/rc_ab.sh <lib> [<lib> ...]
for cfg in "--kernel rbf6 --dtype f64" "--kernel rbf6 --dtype f32" "--kernel c5 --dtype f64"; do
for lib in "$@"; do
PGPS_LIB=$PWD/parallel-gps_amd/pssgp/$lib timeout -k 10 120 python bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('$lib', j['config']['workload'][:34], 'ms=%.4f'%j['gpu_event_ms_per_step'], {k[2:]:round(v,4) for k,v in j['kernel_ms_per_pass'].items() if 'final' not in k})
"
done; done
