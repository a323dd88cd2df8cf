#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "join or long or beyond or occurrence" 2>&1 | tail -2
run() { name=$1; shift; extra=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --workers 16 $extra > $OUT/jf6_$name.log 2> $OUT/jf6_$name.err
  python - $OUT/jf6_$name.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line: print(sys.argv[2], "no line"); sys.exit()
d = json.loads(line[-1])
print(f"{sys.argv[2]:>14}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms {d['ms_each_step']} sha {d['config']['tsv_rows_sha1'][:10]} {d['kernel_ms_per_step'].get('join_fill_and_walk')}")
PY
}
for rep in a b c; do
  run default_$rep "" X=1
  run none_$rep "" KAPTIVE_AMD_SKIP_JOINS=7
done
