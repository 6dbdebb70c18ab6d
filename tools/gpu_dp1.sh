#!/bin/bash
# one-rank RCCL run of the bench step (exercises init / async all-reduce / barrier on the 1-GPU box)
export TMPDIR=/tmp
run() {
  "$@" 2>/tmp/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'n_gpus', d['n_gpus'])
" || tail -5 /tmp/err.txt
}
echo "plain"; run timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-hamilton-gemm --no-kernel-timing
echo "forced one-rank collectives"; QK_DP_FORCE_COLLECTIVES=1 run timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-hamilton-gemm --no-kernel-timing
echo "forced, graph-multi"; QK_DP_FORCE_COLLECTIVES=1 run timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-hamilton-gemm --no-kernel-timing --graph --graph-multi
echo "forced, model"; QK_DP_FORCE_COLLECTIVES=1 run timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --workload cfg3_qcnn_timit_b256_bf16
