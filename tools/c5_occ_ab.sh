# A/B of libpgps builds on c5 (d = 11 fp64) at several chain lengths: tools/c5_occ_ab.sh <lib> [<lib> ...]
for chunk in 0 128 64; do
for lib in "$@"; do
PGPS_LIB=$PWD/parallel-gps_amd/pssgp/$lib timeout -k 10 120 python bench.py --kernel c5 --dtype f64 --chunk $chunk --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('$lib chunk=$chunk', 'ms=%.4f'%j['gpu_event_ms_per_step'], {k[2:]:round(v,4) for k,v in j['kernel_ms_per_pass'].items() if 'final' not in k})
"
done; done
