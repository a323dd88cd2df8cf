#!/bin/bash
# BASELINE.json config 5 rehearsed on one GPU: 8 ranks (processes, HIP contexts, driving threads) share device 0, each
# types its 1000-assembly share of 8000 full-size (5 Mbp) assemblies in batches of 500 (8 ranks x 3 work sets of
# direction bits must fit one GPU's HBM here; on 8 GPUs a rank has 288 GB to itself and runs batches of 1000).
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
free -g | head -2; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29533 bench.py \
  --gpus 8 --assemblies-total 8000 --batch 500 --steps 1 --warmup 1 --no-cpu-baseline --dist-backend gloo --share-gpu --workers 2 \
  > $OUT/config5_rehearsal.log 2> $OUT/config5_rehearsal.err
tail -3 $OUT/config5_rehearsal.err | cut -c1-300
grep '^{' $OUT/config5_rehearsal.log | tail -1 > $OUT/config5_rehearsal.json
python - <<PY
import json
d=json.load(open("$OUT/config5_rehearsal.json"))
print(d["value"], d["ms_per_step"], d["n_gpus"])
for h in d["config"]["host_per_rank"]: print(h)
PY
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
