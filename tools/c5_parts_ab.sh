# timing-only variants of rc_apply1 (results are not valid): tools/c5_parts_ab.sh <lib> [<lib> ...]
for lib in "$@"; do
PGPS_LIB=$PWD/parallel-gps_amd/pssgp/$lib timeout -k 10 120 python bench.py --kernel c5 --dtype f64 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('$lib', 'ms=%.4f'%j['gpu_event_ms_per_step'], {k[2:]:round(v,4) for k,v in j['kernel_ms_per_pass'].items() if 'final' not in k})
"
done
