cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/e2e_trace -- python $GRAFT_REPO_ROOT/bench.py --assemblies 10000 --steps 1 --warmup 1 --e2e-steps 2 --no-cpu-baseline --no-cli > $OUT/e2e_trace.log 2>&1)
grep '^{' $OUT/e2e_trace.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['e2e']['from_host_shards'], d['e2e']['with_tsv'], d['e2e']['h2d_alone'])"
ls $OUT/e2e_trace/*/
