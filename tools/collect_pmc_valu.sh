#!/bin/bash
# Instruction-issue counters of the three hot kernels as they are now: rocprofv3 --pmc, one pass per counter group (at
# most eight SQ counters a pass; FETCH_SIZE and WRITE_SIZE cannot share one), the program itself after "--".
#   A: bench.py's timed region (accumulate_kernel, ntt_tile_kernel)      B: tools/evalh_bench.py (evalh_interp_kernel)
# plus a kernel-trace pass of each for the average durations the counters are divided by.  Run on the GPU box from the
# repository root:  bash tools/collect_pmc_valu.sh gpurun_out/r03/pmc_valu
# then:  python tools/pmc_valu_to_json.py gpurun_out/r03/pmc_valu profiles/r03_pmc_valu.json
set -e
OUT=${1:-gpurun_out/pmc_valu}
mkdir -p "$OUT"
export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE"
SQ2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
run() {  # name, counters..., then the program
  local name=$1; shift
  local counters=$1; shift
  echo "pass $name"
  timeout -s KILL 300 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1
}
run bench_SQ1 "$SQ1" python3 bench.py --skip-cpu --skip-extras --steps 12 --warmup 2
run bench_SQ2 "$SQ2" python3 bench.py --skip-cpu --skip-extras --steps 12 --warmup 2
run evalh_SQ1 "$SQ1" python3 tools/evalh_bench.py
run evalh_SQ2 "$SQ2" python3 tools/evalh_bench.py
run evalh_FETCH "FETCH_SIZE" python3 tools/evalh_bench.py
run evalh_WRITE "WRITE_SIZE" python3 tools/evalh_bench.py
echo "trace passes"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 bench.py --skip-cpu --skip-extras --steps 12 --warmup 2 > "$OUT/bench_trace.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/evalh_trace" -- python3 tools/evalh_bench.py > "$OUT/evalh_trace.log" 2>&1
echo done
