#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "occurrence or paralog" 2>&1 | tail -2
KAPTIVE_AMD_SWEEP=256 KAPTIVE_AMD_SWEEP_CONFIGS=paralog,kpsc timeout 1500 python -m pytest tests -m gpu -q -x -k parity_sweep 2>&1 | tail -2
BENCH_EXTRA="--background paralog --no-secondary" bash tools/gpu_job.sh r6q3 trace > /dev/null 2>&1
f=$(ls gpurun_out/r6q3_trace/*/*kernel_stats.csv | head -1); grep "occ_\|kp_chain_kernel" $f | cut -c1-50,120-300
timeout 900 python bench.py --background paralog --no-cli --no-cpu-baseline --no-secondary --no-e2e > $OUT/r6q3_paralog.log 2> $OUT/r6q3_paralog.err
grep '^{' $OUT/r6q3_paralog.log | cut -c1-200
