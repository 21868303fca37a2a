# the fused entry point (pgps_gp_dev: ts, ys in) at several steps per lane, per build: tools/fusedpath_ab.sh "<libs>" "<paths>" "<chunks>" "<log2n list>"
for l in $4; do for pth in $2; do for ch in $3; do for lib in $1; do
PGPS_LIB=$PWD/parallel-gps_amd/pssgp/$lib python bench.py --path $pth --log2n $l --chunk $ch --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        j=json.loads(ln); print('$lib $pth log2n=$l chunk=$ch', 'ms=%.4f'%j['gpu_event_ms_per_step'], {k[2:]:round(v,4) for k,v in j['kernel_ms_per_pass'].items()})
"
done; done; done; done
