#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
bash tools/gpu_job.sh r6t tests full trace trace_full pmc_sq pmc_clk pmc_mem
timeout 900 python bench.py --background paralog --no-cli --no-cpu-baseline --no-secondary > $OUT/r6t_paralog.log 2> $OUT/r6t_paralog.err
grep '^{' $OUT/r6t_paralog.log | cut -c1-200
