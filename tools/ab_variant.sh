#!/bin/bash
# A/B of a compile-time kernel variant on the GPU box: production build first, then the variant given as
# CALS_EXTRA_DEFS, same session, `bench.py --workload W` without the CPU baseline.  Usage (inside gpurun):
#   bash tools/ab_variant.sh "-DCALS_TTM_EXP_NOCLEAR=1" c3 out_dir
set -e
DEFS="$1"; W="${2:-c3}"; OUT="${3:-gpurun_out/ab}"
mkdir -p "$OUT"
python bench.py --workload "$W" --no-cpu-baseline --steady-steps 0 > "$OUT/base_1.json"
CALS_EXTRA_DEFS="$DEFS" python -c "
import sys; sys.path.insert(0,'.')
import importlib; importlib.import_module('cp-cals_amd.build').build()"
python bench.py --workload "$W" --no-cpu-baseline --steady-steps 0 > "$OUT/variant_1.json"
python bench.py --workload "$W" --no-cpu-baseline --steady-steps 0 > "$OUT/variant_2.json"
python -c "
import sys; sys.path.insert(0,'.')
import importlib; importlib.import_module('cp-cals_amd.build').build()"
python bench.py --workload "$W" --no-cpu-baseline --steady-steps 0 > "$OUT/base_2.json"
python - "$OUT" <<'PY'
import json, sys
for n in ("base_1", "variant_1", "variant_2", "base_2"):
    d = json.load(open("%s/%s.json" % (sys.argv[1], n)))
    r = d["roofline"]
    print("%-10s %8.2f it/s   dominant %.4f ms  frac %.4f" % (n, d["value"], r["avg_launch_ms"], r["frac"]))
PY
