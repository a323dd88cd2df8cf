# kp_chain_kernel and its neighbours alone, on the headline and the paralog workload (one batch per step)
cd /tmp; export TMPDIR=/tmp
for bg in iid paralog; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/ct_$bg
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ct_$bg -- python $GRAFT_REPO_ROOT/bench.py --background $bg --assemblies 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-cli --workers 16 > $GRAFT_REPO_ROOT/gpurun_out/ct_$bg.log 2>&1
  python - $bg <<'PY'
import csv, glob, os, sys
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + f"/gpurun_out/ct_{sys.argv[1]}/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith(("kp_chain", "kp_sw_kernel", "kp_occ", "kp_task", "kp_sw_trace", "kp_scan_dense", "kp_protein", "kp_hit_sort", "kp_reduce")):
        print(f"{sys.argv[1]:8s} {n:28s} avg {float(r['AverageNs'])/1e3:9.1f} us  max {float(r['MaxNs'])/1e3:9.1f}")
PY
done
