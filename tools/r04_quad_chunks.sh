cd $GRAFT_REPO_ROOT
for c in 0 32 16; do
  python bench.py --kernel rbf6 --dtype f32 --family 4 --chunk $c --f32-policy 1 --no-cpu-baseline --main-only --steps 50 --warmup 10 2>/dev/null | python3 -c "
import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('quad chunk $c: ms %.4f'%j['ms_per_step'], {k: round(v*1e3,1) for k,v in j['kernel_ms_per_pass'].items()}, j['chunk'])"
done
python bench.py --kernel rbf6 --dtype f32 --f32-policy 1 --no-cpu-baseline --main-only --steps 50 --warmup 10 2>/dev/null | python3 -c "
import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('lane-chunk: ms %.4f'%j['ms_per_step'], {k: round(v*1e3,1) for k,v in j['kernel_ms_per_pass'].items()})"
