# A/B of libpgps builds on lane-chunk configs: tools/lane_ab.sh "<libs>" "<bench args>" "<chunk list>"
for ch in $3; do for lib in $1; do
PGPS_LIB=$PWD/parallel-gps_amd/pssgp/$lib python bench.py $2 --chunk $ch --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        j=json.loads(ln); print('$lib $2 chunk=$ch', 'ms=%.4f'%j['gpu_event_ms_per_step'], {k[2:]:round(v,4) for k,v in j['kernel_ms_per_pass'].items()}, 'whole=%.3f'%j['roofline']['whole_path_frac'])
"
done; done
