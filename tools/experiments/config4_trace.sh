# per-kernel times of BASELINE config 4 (ab_k, ~1500 contigs per assembly), one batch of 1000 per step
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ct_c4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ct_c4 -- python $GRAFT_REPO_ROOT/bench.py --db ab_k --assemblies 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-cli --workers 16 > $GRAFT_REPO_ROOT/gpurun_out/ct_c4.log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/ct_c4/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:22]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{n:28s} calls {r['Calls']:>3} avg {float(r['AverageNs'])/1e3:9.1f} us  max {float(r['MaxNs'])/1e3:9.1f}  {r['Percentage']}%")
PY
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/ct_c4.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['dp']['tasks_per_step'], d['dp']['anchors_per_step'], d['dp']['cells_per_step'], d['dp']['hits_per_step'])"
