#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --workers 16 > $OUT/jf4_$name.log 2> $OUT/jf4_$name.err
  python - $OUT/jf4_$name.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line: print(sys.argv[2], "no line"); sys.exit()
d = json.loads(line[-1])
print(f"{sys.argv[2]:>14}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms {d['ms_each_step']} sha {d['config']['tsv_rows_sha1'][:10]} {d['kernel_ms_per_step'].get('join_fill_and_walk')}")
PY
}
for rep in a b; do
  run default_$rep X=1
  run afterfill_$rep KAPTIVE_AMD_JOIN_AFTER_FILL=1
  run afterfill_small_$rep KAPTIVE_AMD_JOIN_AFTER_FILL=1 KAPTIVE_AMD_JOIN_GRID=64,16,512,16
  run none_$rep KAPTIVE_AMD_SKIP_JOINS=7
done
