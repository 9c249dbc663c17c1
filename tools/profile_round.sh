#!/bin/bash
# Profiling recipe of a round, run ON THE GPU BOX (through gpurun) from the repository root:
#     bash tools/profile_round.sh r03 [commit]
#   1. rocprofv3 --kernel-trace --stats of the bench command          -> gpurun_out/<tag>_stats/  (+ profiles/<tag>_kernel_stats.csv)
#   2. rocprofv3 --pmc FETCH_SIZE, its own run (kernel trace only)    -> gpurun_out/<tag>_pmc_fetch/
#   3. rocprofv3 --pmc WRITE_SIZE, its own run                        -> gpurun_out/<tag>_pmc_write/
#   4. tools/pmc_summary.py (fails when a pass left no rows)          -> gpurun_out/<tag>_pmc_traffic.json
#   5. rocprofv3 --pmc SQ_* (wave-cycle buckets + MFMA busy), own run  -> gpurun_out/<tag>_sq_summary.txt (tools/pmc_sq_summary.py)
# The PMC passes run bench.py with FW_KEY_STREAM=0: under per-dispatch counter interception the round-1 WRITE_SIZE pass died with
# SIGSEGV in a profiler thread at the first replay of the graph that forks the key-encoder branch onto a second HIP stream
# (gpurun_out/pmc_write3.log of round 1); a single-stream graph has the same kernels and the same bytes per launch.
# Steps are chained with && : after a failure no further GPU step runs.
set -o pipefail
TAG=${1:-r03}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $BENCH > $OUT/${TAG}_stats.log 2> $OUT/${TAG}_stats.err &&
  cp "$(ls $OUT/${TAG}_stats/*/*_kernel_stats.csv | head -1)" $OUT/${TAG}_kernel_stats.csv &&
  python3 tools/prof_summary.py $OUT/${TAG}_stats 30 > $OUT/${TAG}_prof_summary.txt &&
  FW_KEY_STREAM=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/${TAG}_pmc_fetch.log 2>&1 &&
  FW_KEY_STREAM=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/${TAG}_pmc_write.log 2>&1 &&
  python3 tools/pmc_summary.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_traffic.json "$COMMIT" "$(date -u +%Y-%m-%dT%H:%MZ)" > $OUT/${TAG}_pmc_summary.txt &&
  FW_KEY_STREAM=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/${TAG}_pmc_sq.log 2>&1 &&
  python3 tools/pmc_sq_summary.py $OUT/${TAG}_pmc_sq > $OUT/${TAG}_sq_summary.txt &&
  rm -rf $OUT/${TAG}_pmc_sq/*/*_kernel_trace.csv $OUT/${TAG}_pmc_fetch/*/*_kernel_trace.csv $OUT/${TAG}_pmc_write/*/*_kernel_trace.csv $OUT/${TAG}_stats/*/*_kernel_trace.csv
rc=$?
echo "profile_round: rc=$rc"
tail -3 $OUT/${TAG}_pmc_write.log 2>/dev/null
exit $rc
