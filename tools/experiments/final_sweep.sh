#!/bin/bash
# round 6, final tree: the parity sweep at 1024 assemblies per configuration and a 150-step soak
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
KAPTIVE_AMD_SWEEP=${SWEEP_N:-1024} KAPTIVE_AMD_SWEEP_OUT=$OUT/r6_sweep.json timeout 3300 python -m pytest tests -m gpu -q -x -k parity_sweep 2>&1 | tail -3
cat $OUT/r6_sweep.json
timeout 900 python bench.py --steps 150 --warmup 2 --no-cpu-baseline --no-e2e --no-cli --no-secondary > $OUT/r6_soak.log 2> $OUT/r6_soak.err
grep '^{' $OUT/r6_soak.log | cut -c1-300
