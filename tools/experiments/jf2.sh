#!/bin/bash
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_job.sh r6x trace > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r6x_trace/runc/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'kp_scan_dense' in r['Kernel_Name']]
i0=idx[-2]; t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+22]:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:40]
    if 'join' in n or 'kp_sw' in n:
        s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
        print(f"{s:10.1f} {e:10.1f} {e-s:9.1f} {n}")
PY
run() { name=$1; shift
  env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary --workers 16 > $OUT/jf2_$name.log 2> $OUT/jf2_$name.err
  python - $OUT/jf2_$name.log $name <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line: print(sys.argv[2], "no line"); sys.exit()
d = json.loads(line[-1])
print(f"{sys.argv[2]:>12}: {d['value']:9.0f} asm/s  step {d['ms_per_step']:.1f} ms {d['ms_each_step']} sha {d['config']['tsv_rows_sha1'][:10]}")
PY
}
for rep in a b; do
  run default_$rep X=1
  run nofill_$rep KAPTIVE_AMD_SKIP_JOINS=6
  run none_$rep KAPTIVE_AMD_SKIP_JOINS=7
done
