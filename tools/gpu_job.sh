#!/bin/bash
# One GPU-box job.  Usage: tools/gpu_job.sh TAG [tests] [bench] [trace] [pmc_sq] [pmc_mem] [full]
TAG=${1:-job}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SMALL="--assemblies 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --workers 16 $BENCH_EXTRA"
for what in "$@"; do
  case $what in
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1; tail -5 $OUT/${TAG}_tests.log;;
    bench)
      timeout 900 python bench.py --assemblies 3000 --steps 4 --warmup 1 --no-cpu-baseline $BENCH_EXTRA > $OUT/${TAG}_bench.log 2> $OUT/${TAG}_bench.err
      tail -c 2500 $OUT/${TAG}_bench.log; tail -3 $OUT/${TAG}_bench.err;;
    full)
      timeout 1500 python bench.py > $OUT/${TAG}_full.log 2> $OUT/${TAG}_full.err
      tail -c 4000 $OUT/${TAG}_full.log; tail -3 $OUT/${TAG}_full.err;;
    trace)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python $GRAFT_REPO_ROOT/bench.py $SMALL > $OUT/${TAG}_trace.log 2>&1)
      tail -c 1500 $OUT/${TAG}_trace.log | head -c 1200; f=$(ls $OUT/${TAG}_trace/*/*kernel_stats.csv | head -1); head -25 $f;;
    trace_full)  # the default bench command itself under the kernel trace (the summary profiles/ holds for the bench line)
      (cd /tmp && timeout 1700 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_full -- python $GRAFT_REPO_ROOT/bench.py > $OUT/${TAG}_trace_full.log 2>&1)
      grep '^{' $OUT/${TAG}_trace_full.log | cut -c1-400; f=$(ls $OUT/${TAG}_trace_full/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200;;
    pmc_sq)
      (cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        --output-format csv -d $OUT/${TAG}_pmc_sq -- python $GRAFT_REPO_ROOT/bench.py $SMALL > $OUT/${TAG}_pmc_sq.log 2>&1); tail -2 $OUT/${TAG}_pmc_sq.log;;
    pmc_sq2)
      (cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU \
        --output-format csv -d $OUT/${TAG}_pmc_sq2 -- python $GRAFT_REPO_ROOT/bench.py $SMALL > $OUT/${TAG}_pmc_sq2.log 2>&1); tail -2 $OUT/${TAG}_pmc_sq2.log
      (cd /tmp && rocprofv3 -L > $OUT/${TAG}_counters.txt 2>&1); wc -l $OUT/${TAG}_counters.txt;;
    pmc_clk)
      (cd /tmp && timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}_pmc_clk -- python $GRAFT_REPO_ROOT/bench.py $SMALL > $OUT/${TAG}_pmc_clk.log 2>&1); tail -2 $OUT/${TAG}_pmc_clk.log | cut -c1-200;;
    pmc_prot)  # the latency-bound kernels after the fill: instruction counts and wave cycles
      (cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY \
        --output-format csv -d $OUT/${TAG}_pmc_prot -- python $GRAFT_REPO_ROOT/bench.py $SMALL > $OUT/${TAG}_pmc_prot.log 2>&1); tail -2 $OUT/${TAG}_pmc_prot.log | cut -c1-200;;
    pmc_mem)
      (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py $SMALL > $OUT/${TAG}_pmc_fetch.log 2>&1)
      (cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python $GRAFT_REPO_ROOT/bench.py $SMALL > $OUT/${TAG}_pmc_write.log 2>&1)
      (cd /tmp && timeout 900 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_pmc_tcc -- python $GRAFT_REPO_ROOT/bench.py $SMALL > $OUT/${TAG}_pmc_tcc.log 2>&1)
      tail -2 $OUT/${TAG}_pmc_tcc.log;;
  esac
done
