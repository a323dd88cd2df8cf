#!/bin/bash
# round 6: the working tree against the round-5 tree (build/old_tree, exported from git by the caller) and against itself without
# join kernels, same box, interleaved
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
(cd build/old_tree && python -m kaptive_amd.build > $OUT/fab2_old_build.log 2>&1)
run() { name=$1; dir=$2; shift 2
  (cd $dir && env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --workers 16 > $OUT/fab2_$name.log 2> $OUT/fab2_$name.err)
  python - $OUT/fab2_$name.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line: print(sys.argv[2], "no line"); sys.exit()
d = json.loads(line[-1])
print(f"{sys.argv[2]:>10}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms {d['ms_each_step']} sha {d['config']['tsv_rows_sha1'][:10]}  kernels {d['kernel_ms_per_step']}")
PY
}
for rep in a b c; do
  run old_$rep build/old_tree X=1
  run new_$rep . X=1
  run nojoin_$rep . KAPTIVE_AMD_SKIP_JOINS=7
done
