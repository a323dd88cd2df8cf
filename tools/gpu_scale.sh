#!/bin/bash
# BASELINE.json config 5 in one command, for whoever has an 8-GPU node: the scaling curve of bench.py (100 000 synthetic
# assemblies over 8 GPUs = 12 500 per GPU; the same shard per GPU at 1, 2 and 4: one process per GPU, no collective on the data path) and `kaptive assembly
# --devices all` over 8 x 2304 FASTA files.  Prints one JSON line per run and keeps everything under gpurun_out/scale/.
#
#     bash tools/gpu_scale.sh [GPUS="1 2 4 8"] [TOTAL=100000]
#
# Each bench line is the driver's contract (python -m torch.distributed.run ... bench.py --gpus N): `value` = assemblies
# typed per second by all ranks together, max over ranks of the timed region; scaling efficiency is value(N) / (N * value(1)).
set -u
cd "$(dirname "$0")/.."
GPUS=${1:-"1 2 4 8"}
TOTAL=${2:-100000}
PER=$((TOTAL / 8))  # config 5's shard per GPU (12 500): weak scaling, the 8-GPU run types all 100 000
OUT=gpurun_out/scale
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29511
for N in $GPUS; do
  HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
  if [ "$HAVE" -lt "$N" ]; then echo "{\"n_gpus\": $N, \"skipped\": \"only $HAVE device(s) visible\"}"; continue; fi
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --assemblies $PER --no-cli > $OUT/bench_n$N.json 2> $OUT/bench_n$N.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + N)) \
      bench.py --gpus $N --assemblies $PER --no-cli > $OUT/bench_n$N.json 2> $OUT/bench_n$N.err
  fi
  grep '^{' $OUT/bench_n$N.json | python -c "
import json, sys
for line in sys.stdin:
    d = json.loads(line)
    print(json.dumps({k: d.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'ms_per_step', 'scaling')} | {'assemblies_total': d['config'].get('assemblies_total')}))
"
done
# the command-line tool over all devices: 8 x 2304 files (192 distinct assemblies on tmpfs, listed 12 times per device)
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
python tools/cli_probe.py --devices all --files 192 --repeats $((12 * NDEV)) > $OUT/cli_all_devices.json 2> $OUT/cli_all_devices.err || true
tail -1 $OUT/cli_all_devices.json
