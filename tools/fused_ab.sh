# compares the side legs of the default bench line between builds: tools/fused_ab.sh <lib> [<lib> ...]
for lib in "$@"; do
PGPS_LIB=$PWD/parallel-gps_amd/pssgp/$lib python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        j=json.loads(ln); print('$lib', 'ms=%.4f'%j['gpu_event_ms_per_step'], {k[:40]:round(v['ms_per_step'],4) for k,v in j.get('fused_path',{}).items() if isinstance(v,dict)})
"
done
