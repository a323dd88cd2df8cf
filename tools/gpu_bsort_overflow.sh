#!/bin/bash
# On the GPU box: the anchor sort built with list lengths of 2 and 1, so that an ordinary assembly overflows both the list
# of wave-sized buckets (sorted where they are met instead) and the list of block-sized ones (ranked in place by one lane),
# then the parity tests that compare sorted anchors, band tasks and hit tables with the oracle's.
#   gpurun -- 'bash tools/gpu_bsort_overflow.sh > gpurun_out/bsort_overflow.txt 2>&1'
set -e
cd "$(dirname "$0")/.."
cp kaptive_amd/libkaptive_amd.so /tmp/libkaptive_amd.keep.so
touch kaptive_amd/csrc/kp_bsort.hip
KAPTIVE_AMD_EXTRA_FLAGS="-DKP_BS_BIG_LIST=2 -DKP_BS_HUGE_LIST=1" python -m kaptive_amd.build > /dev/null
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "anchor or bucket or config or stress or many or gene" 2>&1 | tail -5
cp /tmp/libkaptive_amd.keep.so kaptive_amd/libkaptive_amd.so
