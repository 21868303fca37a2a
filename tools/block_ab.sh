# same-box A/B of the lane-chunk workgroup size (256 lanes forced against the automatic choice): tools/block_ab.sh
for cfg in "--log2n 14" "--log2n 16" "--log2n 18" "--log2n 19" "--log2n 20" "--log2n 21" "--log2n 22" "--kernel matern52" "--kernel rbf4" "--kernel rbf6 --dtype f32" "--kernel rbf6 --dtype f32 --log2n 18" "--kernel rbf5 --dtype f32"; do
for blk in 256 0 256 0; do
python bench.py $cfg --block $blk --steps 30 --warmup 5 --no-cpu-baseline --main-only 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        j=json.loads(ln); print('$cfg block=$blk', 'ms=%.4f'%j['gpu_event_ms_per_step'], {k[2:]:round(v,4) for k,v in j['kernel_ms_per_pass'].items()}, 'whole=%.3f'%j['roofline']['whole_path_frac'])
"
done; done
