#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
bash tools/gpu_job.sh r6u tests full trace trace_full pmc_sq pmc_clk pmc_mem
timeout 900 python bench.py --mix joins --no-cli --no-cpu-baseline --no-secondary --no-e2e > $OUT/r6u_joins.log 2> $OUT/r6u_joins.err
grep '^{' $OUT/r6u_joins.log | cut -c1-200
timeout 900 python bench.py --background paralog --no-cli --no-cpu-baseline --no-secondary > $OUT/r6u_paralog.log 2> $OUT/r6u_paralog.err
grep '^{' $OUT/r6u_paralog.log | cut -c1-200
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6u_jtrace -- python $GRAFT_REPO_ROOT/bench.py --assemblies 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --workers 16 --mix joins --no-secondary > $OUT/r6u_jtrace.log 2>&1)
