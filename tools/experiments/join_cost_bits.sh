#!/bin/bash
# round 6: which of the join kernels costs the headline workload what (same box, same build): KAPTIVE_AMD_SKIP_JOINS bits
# 1 = no chaining of groups, 2 = no joined fill, 4 = no walk-back; twice each, interleaved
cd $GRAFT_REPO_ROOT; OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --assemblies 10000 --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --workers 16 > $OUT/jbits_$name.log 2> $OUT/jbits_$name.err
  python - $OUT/jbits_$name.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
d = json.loads(line[-1])
print(f"{sys.argv[2]:>12}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms {d['ms_each_step']}  kernels {d['kernel_ms_per_step']}")
PY
}
for rep in a b; do
  run all_$rep X=1
  run notrace_$rep KAPTIVE_AMD_SKIP_JOINS=4
  run chainonly_$rep KAPTIVE_AMD_SKIP_JOINS=6
  run none_$rep KAPTIVE_AMD_SKIP_JOINS=7
done
