#!/bin/bash
# A/B of two builds of libpgps on ONE box (one gpurun lease), interleaved:  tools/ab.sh <base.so|-> <new.so> <rounds> -- cmd ...
# Runs `cmd` with PGPS_LIB=<base> and PGPS_LIB=<new> alternately (base, new, base, new, ...), takes "ms_per_step" (or, with
# AB_KEY=<json key>, another number) from the JSON line each run prints, and reports every value, the medians and the
# libraries' sha256 (and a variant's manifest, csrc/Makefile `variant`).  `-` for base = parallel-gps_amd/pssgp/libpgps.so.
R=${GRAFT_REPO_ROOT:-$PWD}
base=$1; new=$2; rounds=$3; shift 3
[ "$1" = "--" ] && shift
[ "$base" = "-" ] && base=$R/parallel-gps_amd/pssgp/libpgps.so
key=${AB_KEY:-ms_per_step}
for f in "$base" "$new"; do
  echo "library $(sha256sum $f | cut -c1-16)  $f"
  [ -f "$f.manifest" ] && sed 's/^/    /' "$f.manifest" | head -2
done
va=(); vb=()
for i in $(seq 1 $rounds); do
  a=$(PGPS_LIB=$base "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['$key'])")
  b=$(PGPS_LIB=$new "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['$key'])")
  echo "round $i: base $a   new $b"
  va+=($a); vb+=($b)
done
python3 - "${va[*]}" "${vb[*]}" "$key" <<'PY'
import sys, statistics
a=[float(x) for x in sys.argv[1].split()]; b=[float(x) for x in sys.argv[2].split()]
ma, mb = statistics.median(a), statistics.median(b)
print(f"median {sys.argv[3]}: base {ma:.6g}   new {mb:.6g}   new/base {mb/ma:.4f}")
PY
