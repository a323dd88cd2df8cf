#!/bin/bash
# round 6: what an (empty) kernel launch per alignment pass costs the overlapped step: 0 / 8 / 32 dummy launches, interleaved
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --workers 16 > $OUT/lc_$name.log 2> $OUT/lc_$name.err
  python - $OUT/lc_$name.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line: print(sys.argv[2], "no line"); sys.exit()
d = json.loads(line[-1])
print(f"{sys.argv[2]:>10}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms {d['ms_each_step']}")
PY
}
for rep in a b; do
  run d0_$rep X=1
  run d8_$rep KAPTIVE_AMD_DUMMY_LAUNCHES=8
  run d32_$rep KAPTIVE_AMD_DUMMY_LAUNCHES=32
done
