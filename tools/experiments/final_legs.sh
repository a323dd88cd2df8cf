#!/bin/bash
# round 6, final tree: the secondary workloads at their full size (round 5's commands), and the 8-rank rehearsal of config 5
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python bench.py --background paralog --no-cli --no-cpu-baseline --no-secondary > $OUT/r6_paralog.log 2> $OUT/r6_paralog.err
grep '^{' $OUT/r6_paralog.log | cut -c1-260
timeout 900 python bench.py --db ab_k --assemblies 5000 --no-cpu-baseline --no-cli --no-secondary > $OUT/r6_config4.log 2> $OUT/r6_config4.err
grep '^{' $OUT/r6_config4.log | cut -c1-260
timeout 900 python bench.py --mix joins --no-cli --no-cpu-baseline --no-secondary --no-e2e > $OUT/r6_joins.log 2> $OUT/r6_joins.err
grep '^{' $OUT/r6_joins.log | cut -c1-260
timeout 1200 bash tools/gpu_config5_rehearsal.sh > $OUT/r6_rehearsal.log 2>&1; tail -3 $OUT/r6_rehearsal.log | cut -c1-400
