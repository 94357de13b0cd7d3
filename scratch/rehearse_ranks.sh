#!/bin/bash
# two ranks on one GPU over the host channel, both workloads, against one rank:
# bash scratch/rehearse_ranks.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ranks
quiet="--cpu-frames 0 --literal-frames 0 --skip-host-input --api-frames 0 --gk-frames 0 --flux-atoms 0"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['n_gpus'], round(d['value'],1), d['transport'], d['pairs_counted'])"; }
MDX_COMM_TRANSPORT=host timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 $quiet 2>gpurun_out/ranks/n2.err | show nacl || exit 1
timeout -k 10 300 python bench.py --steps 4 --warmup 1 $quiet 2>>gpurun_out/ranks/n2.err | show nacl || exit 1
MDX_COMM_TRANSPORT=host timeout -k 10 300 python bench.py --gpus 2 --workload water --steps 4 --warmup 1 $quiet 2>>gpurun_out/ranks/n2.err | show water || exit 1
timeout -k 10 300 python bench.py --workload water --steps 4 --warmup 1 $quiet 2>>gpurun_out/ranks/n2.err | show water || exit 1
