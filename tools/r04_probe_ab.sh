# c3 with the dense-grid probe (automatic policy) against float32 arithmetic without it (policy 1), interleaved on one box
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
for pol in 0 1; do
  python bench.py --kernel rbf6 --dtype f32 --f32-policy $pol --no-cpu-baseline --main-only 2>/dev/null | python3 -c "
import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('round $i policy $pol: ms %.4f'%j['ms_per_step'], {k: round(v*1e3,1) for k,v in j['kernel_ms_per_pass'].items()})"
done; done
