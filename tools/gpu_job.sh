#!/bin/bash
# One GPU-box job: tests, a short bench, and an SQ-counter pass over the SW kernel.  Usage: tools/gpu_job.sh TAG
TAG=${1:-job}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1
tail -5 $OUT/${TAG}_tests.log
timeout 900 python bench.py --assemblies 3000 --steps 2 --warmup 1 > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
tail -c 3000 $OUT/${TAG}_bench.log; tail -5 $OUT/${TAG}_bench.err
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $OUT/${TAG}_pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --assemblies 1000 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --workers 16 > $OUT/${TAG}_pmc_sq.log 2>&1
tail -3 $OUT/${TAG}_pmc_sq.log
ls -la $OUT/${TAG}_pmc_sq/* | head
