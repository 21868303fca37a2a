# usage: tools/c2_matrix.sh "<log2n list>" "<stage list>" "<chunk list>" [extra bench args]
L=${1:-"20 22 24"}; S=${2:-"4"}; C=${3:-"8 16 32"}; shift 3
for l in $L; do for st in $S; do for ch in $C; do
python bench.py --log2n $l --stage $st --chunk $ch --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        j=json.loads(ln); print('log2n=$l stage=$st chunk=$ch', 'ms=%.4f'%j['gpu_event_ms_per_step'], {k:round(v,4) for k,v in j['kernel_ms_per_pass'].items()}, 'whole_frac=%.3f'%j['roofline']['whole_path_frac'], 'll=%.6f'%j['log_likelihood'])
"
done; done; done
