#!/bin/bash
# times the CO2 (d = 18) pass with each diagnostic build of the wave-cooperative kernels (csrc/build/x/libpgps_x*.so)
out=gpurun_out/wc_x; mkdir -p $out
for lib in "" $(ls parallel-gps_amd/csrc/build/x/libpgps_x*.so 2>/dev/null); do
  tag=$(basename "${lib:-base}" .so)
  if [ -n "$lib" ]; then export PGPS_LIB=$PWD/$lib; else unset PGPS_LIB; fi
  timeout -k 10 200 python bench.py --kernel co2 --log2n 17 --steps 40 --warmup 5 > $out/$tag.json 2> $out/$tag.err || { echo "$tag failed"; exit 1; }
  python - "$out/$tag.json" "$tag" <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); k=r["kernel_ms_per_pass"]
print(sys.argv[2], round(r["ms_per_step"],3), {a:round(b,3) for a,b in k.items()})
PY
done
