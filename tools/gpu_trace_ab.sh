#!/bin/bash
# Kernel-trace A/B on the GPU box: tools/gpu_trace_ab.sh TAG PATTERN "name=flags" ...  -> average microseconds of kernels matching PATTERN
TAG=${1:-tab}; PAT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  KAPTIVE_AMD_EXTRA_FLAGS="$flags" python -m kaptive_amd.build --force > $OUT/${TAG}_${name}_build.log 2>&1 || { tail -5 $OUT/${TAG}_${name}_build.log; continue; }
  rm -rf $OUT/${TAG}_${name}_trace
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${name}_trace -- python $GRAFT_REPO_ROOT/bench.py --assemblies 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --workers 16 $AB_EXTRA > $OUT/${TAG}_${name}_trace.log 2>&1)
  f=$(ls $OUT/${TAG}_${name}_trace/*/*kernel_stats.csv | head -1)
  echo "== $name"; grep -E "$PAT" $f | awk -F'","|",|,"' '{n=split($0,a,","); print substr($1,1,60), a[n-6], a[n-4]}' | head -8
done
