#!/bin/bash
# Round profile of bench.py on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats, then the two HBM-side PMC passes (separate runs, as the guide prescribes),
#   condensed by tools/rocprof_summary.py into gpurun_out/prof_<workload>_*.txt
# usage: tools/profile_round.sh <workload> [steps]
set -e
W=${1:-c3}
K=${2:-50}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$W
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o $W --output-format csv -- python3 bench.py --workload $W --steps $K --warmup 5 --steady-steps 0 --no-cpu-baseline --no-strong-leg > $OUT/bench_kt.log 2>&1
python3 tools/rocprof_summary.py $OUT/kt "bench.py ($W), rocprofv3 --kernel-trace --stats" > gpurun_out/prof_${W}_kernel_trace_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/$C -o $W --output-format csv -- python3 bench.py --workload $W --steps $K --warmup 5 --steady-steps 0 --no-cpu-baseline --no-strong-leg > $OUT/bench_$C.log 2>&1
  python3 tools/rocprof_summary.py $OUT/$C "bench.py ($W), rocprofv3 --pmc $C (KiB)" > gpurun_out/prof_${W}_pmc_$C.txt
done
# matrix-core busy cycles / LDS conflicts of the same command (its own pass, counters only)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d $OUT/SQ -o $W --output-format csv -- python3 bench.py --workload $W --steps 10 --warmup 5 --steady-steps 0 --no-cpu-baseline --no-strong-leg > $OUT/bench_SQ.log 2>&1 && python3 tools/rocprof_summary.py $OUT/SQ "bench.py ($W), rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" > gpurun_out/prof_${W}_pmc_SQ.txt || echo "SQ pass failed (see $OUT/bench_SQ.log)"
tail -1 $OUT/bench_kt.log | cut -c1-300
rm -rf $OUT
