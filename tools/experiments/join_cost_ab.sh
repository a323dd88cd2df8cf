#!/bin/bash
# round 6: what the join kernels cost on the headline workload (same box, same build): with them / without them, twice
cd $GRAFT_REPO_ROOT; OUT=gpurun_out; mkdir -p $OUT
N=${1:-10000}
run() { name=$1; shift
  env "$@" python bench.py --assemblies $N --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --workers 16 $EXTRA > $OUT/jab_$name.log 2> $OUT/jab_$name.err
  python - $OUT/jab_$name.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
d = json.loads(line[-1])
print(f"{sys.argv[2]:>12}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms  sha {d['config']['tsv_rows_sha1'][:10]}  kernels {d['kernel_ms_per_step']}")
PY
}
run none KAPTIVE_AMD_SKIP_JOINS=7
run default X=1
run none2 KAPTIVE_AMD_SKIP_JOINS=7
run default2 X=1
