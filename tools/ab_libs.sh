#!/bin/bash
# A/B of prebuilt library variants on the GPU box (no rebuild there): bench.py with the shipped libcals_hip.so,
# then with every cp-cals_amd/build/variants/libcals_hip_<name>.so, then the shipped one again.  A variant is
# selected per process (CALS_HIP_LIB, cp_cals_amd.load_library); the shipped library is never overwritten.
#   bash tools/ab_libs.sh <workload> <out_dir> <name> [<name> ...]
set -e
W="${1:-c3}"; OUT="${2:-gpurun_out/ab}"; shift 2
mkdir -p "$OUT"
run() { CALS_HIP_LIB="$2" python bench.py --workload "$W" --no-cpu-baseline --no-strong-leg --steady-steps 0 > "$OUT/$1.json" 2> "$OUT/$1.err"; }
run prod_1
for v in "$@"; do
  L="$PWD/cp-cals_amd/build/variants/libcals_hip_$v.so"
  run "${v}_1" "$L"
  run "${v}_2" "$L"
done
run prod_2
python - "$OUT" prod_1 $(for v in "$@"; do echo ${v}_1 ${v}_2; done) prod_2 <<'PY'
import json, sys
for n in sys.argv[2:]:
    d = json.loads(open("%s/%s.json" % (sys.argv[1], n)).readline())
    r = d["roofline"]
    print("%-12s %8.2f it/s   dominant %.4f ms  frac %.4f  rest %.4f" % (n, d["value"], r["avg_launch_ms"], r["frac"], r.get("rest_ms_per_step", 0)))
PY
