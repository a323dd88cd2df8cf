#!/bin/bash
# round 6: what about the joined fill costs the headline workload (20 joins per batch): grid, wave priority, the launch itself
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --workers 16 > $OUT/jf_$name.log 2> $OUT/jf_$name.err
  python - $OUT/jf_$name.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line: print(sys.argv[2], "no line"); sys.exit()
d = json.loads(line[-1])
print(f"{sys.argv[2]:>12}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms {d['ms_each_step']}")
PY
}
for rep in a b; do
  run default_$rep X=1
  run smallgrid_$rep KAPTIVE_AMD_JOIN_GRID=32,8,64,8
  run noprio_$rep KAPTIVE_AMD_JOIN_PRIO=0
  run small_noprio_$rep KAPTIVE_AMD_JOIN_GRID=32,8,64,8 KAPTIVE_AMD_JOIN_PRIO=0
  run nofill_$rep KAPTIVE_AMD_SKIP_JOINS=6
  run d16_$rep KAPTIVE_AMD_DUMMY_LAUNCHES=16
done
