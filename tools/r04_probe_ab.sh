# c3 with the dense-grid probe (automatic policy) against float32 arithmetic without it (policy 1), interleaved on one box.
# (the third run of profiles/r04_experiments.txt item 2 also set PGPS_PROBE_TEST = 1 / 2 / 3 -- events without the system-scope
# fence, 512 samples, both -- through a switch that existed in pgps_core.hip for that run only; the library ignores it now)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
for v in "1:0" "0:0" "0:1" "0:2" "0:3"; do
  pol=${v%%:*}; pt=${v#*:}
  PGPS_PROBE_TEST=$pt python bench.py --kernel rbf6 --dtype f32 --f32-policy $pol --no-cpu-baseline --main-only --steps 100 2>/dev/null | python3 -c "
import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('round $i policy $pol probe-test $pt: ms %.4f'%j['ms_per_step'], {k: round(v*1e3,1) for k,v in j['kernel_ms_per_pass'].items()})"
done; done
