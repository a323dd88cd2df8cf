# SQ_INSTS_VALU of the traceback kernel with and without the 16-step fast path
cd $GRAFT_REPO_ROOT
for v in "tb16=" "tb8=-DKP_TB_NO16"; do
  name=${v%%=*}; flags=${v#*=}
  KAPTIVE_AMD_EXTRA_FLAGS="$flags" python -m kaptive_amd.build --force > gpurun_out/tbv_${name}_build.log 2>&1
  (cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/tbv_$name && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tbv_$name -- python $GRAFT_REPO_ROOT/bench.py --assemblies 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-cli --workers 16 > $GRAFT_REPO_ROOT/gpurun_out/tbv_$name.log 2>&1)
  python - $name <<'PY'
import csv, glob, os, sys, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + f"/gpurun_out/tbv_{sys.argv[1]}/*/*counter_collection.csv")[0]
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if "traceback" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
        tot["valu"] += float(r["Counter_Value"]); n["d"] += 1
print(sys.argv[1], "traceback SQ_INSTS_VALU per launch", tot["valu"] / max(n["d"], 1) / 1e9, "G over", n["d"], "dispatch rows")
PY
done
KAPTIVE_AMD_EXTRA_FLAGS="" python -m kaptive_amd.build --force > /dev/null 2>&1
