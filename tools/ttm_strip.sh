#!/bin/bash
# Timing-only stripping ladder of ttm_kernel at C3: variants compiled with -DCALS_TTM_STRIP=<mask> and prebuilt as
# cp-cals_amd/build/variants/libcals_hip_strip<mask>.so (bits: ttm_kernel.hip; results of stripped runs are garbage
# by design -- only the TTM's launch time is read).   bash tools/ttm_strip.sh <out_dir> <mask> [<mask> ...]
# Every run points its own process at its variant (CALS_HIP_LIB, cp_cals_amd.load_library): the shipped library is
# never overwritten, so an interrupted ladder leaves nothing behind.
OUT="${1:-gpurun_out/strip}"; shift
W="${CALS_STRIP_WORKLOAD:-c3}"   # CALS_STRIP_WORKLOAD=c4: the fp32 kernel
mkdir -p "$OUT"
run() {  # run <mask> [<library>]
  CALS_HIP_LIB="$2" python bench.py --workload "$W" --no-cpu-baseline --no-strong-leg --steady-steps 0 --steps 20 > "$OUT/strip_$1.json" 2> "$OUT/strip_$1.err"
  python - "$OUT/strip_$1.json" $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).readline())
    r = d["roofline"]
    ks = {k.split(" ")[0]: v["avg_launch_ms"] for k, v in r["mfma_kernels"].items()}
    print("strip %5s: %s  (%.1f it/s)" % (sys.argv[2], "  ".join("%s %.4f ms" % kv for kv in sorted(ks.items())), d["value"]))
except Exception as ex:
    print("strip %5s: failed (%s)" % (sys.argv[2], ex))
PY
}
run 0
for m in "$@"; do
  run $m "$PWD/cp-cals_amd/build/variants/libcals_hip_strip$m.so"
done
run 0
