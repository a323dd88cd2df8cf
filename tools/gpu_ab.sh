#!/bin/bash
# A/B builds on the GPU box.  Usage: tools/gpu_ab.sh TAG "name1=flags1" "name2=flags2" ...   (flags may be empty)
# Each variant: rebuild the library with KAPTIVE_AMD_EXTRA_FLAGS=flags, run a one-batch-per-step bench (kernels alone) and a
# 3000-assembly bench (overlapped), print the figures that matter.
TAG=${1:-ab}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  KAPTIVE_AMD_EXTRA_FLAGS="$flags" python -m kaptive_amd.build --force > $OUT/${TAG}_${name}_build.log 2>&1 || { tail -5 $OUT/${TAG}_${name}_build.log; continue; }
  python bench.py --assemblies ${AB_N:-3000} --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --workers 16 $AB_EXTRA > $OUT/${TAG}_${name}.log 2> $OUT/${TAG}_${name}.err
  python - $OUT/${TAG}_${name}.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line:
    print(sys.argv[2], "no line"); sys.exit()
d = json.loads(line[-1])
print(f"{sys.argv[2]:>14}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms  fill alone {d['dp']['fill_issue_model']['ms_per_launch_alone']:.3f}  "
      f"scan alone {d['roofline']['alone']['ms_per_launch']:.3f}  sha {d['config']['tsv_rows_sha1'][:10]}  kernels {d['kernel_ms_per_step']}")
PY
done
