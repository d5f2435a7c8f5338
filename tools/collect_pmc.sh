#!/bin/bash
# HBM traffic counters of the benchmark's timed region, one rocprofv3 --pmc pass per counter (FETCH_SIZE takes three of
# the four TCC slots and WRITE_SIZE two, so they cannot share a pass), plus the same two passes over tools/gather_probe,
# whose kernels move known byte counts, and a kernel-trace pass for the average durations.  Run on the GPU box from the
# repository root:  bash tools/collect_pmc.sh gpurun_out/r02/pmc
# then summarise with:  python tools/pmc_to_json.py gpurun_out/r02/pmc profiles/pmc_traffic.json
set -e
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --skip-cpu --skip-extras --steps 12 --warmup 2"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/bench_$C" -- $BENCH > "$OUT/bench_$C.log" 2>&1
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$OUT/probe_$C" -- ./tools/gather_probe > "$OUT/probe_$C.log" 2>&1
done
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- $BENCH > "$OUT/bench_trace.log" 2>&1
grep "^{" "$OUT/bench_trace.log" > "$OUT/bench_line_under_rocprof.json"
echo done
