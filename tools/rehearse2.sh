#!/bin/bash
# Run ON THE GPU BOX: two ranks on one GPU over gloo (real engine) next to one rank
# given the same 2048 walkers; the chi2 checksums must be bitwise equal.
# Output: gpurun_out/rehearse2_{n1,n2}.json
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python bench.py --gpus 1 --walkers 2048 --steps 5 --warmup 1 --cpu-evals 0 > gpurun_out/rehearse2_n1.json 2> gpurun_out/rehearse2_n1.err || exit 1
SLAM_BENCH_WATCHDOG=90 timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --walkers 1024 --steps 5 --warmup 1 --cpu-evals 0 > gpurun_out/rehearse2_n2.json 2> gpurun_out/rehearse2_n2.err || exit 2
python - <<'PY'
import json
a = json.loads(open('gpurun_out/rehearse2_n1.json').read().strip().splitlines()[-1])
b = json.loads(open('gpurun_out/rehearse2_n2.json').read().strip().splitlines()[-1])
ca, cb = a['config']['chi2_checksum'], b['config']['chi2_checksum']
print('one rank  :', a['value'], 'evals/s, checksum', repr(ca), 'startup', a['config']['startup_s_per_rank'])
print('two ranks :', b['value'], 'evals/s, checksum', repr(cb), 'startup', b['config']['startup_s_per_rank'])
assert ca == cb, 'checksums differ'
print('checksums bitwise equal')
PY
