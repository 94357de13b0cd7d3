#!/bin/bash
# time of the pack slot (pack + sort kernels) per 5000-frame launch with parts of the sort
# kernel switched off (timing build, wrong counts): bash scratch/time_sort_parts.sh lib.so
cd "$(dirname "$0")/.."
lib=${1:?library}
run() {
  ALT=$lib timeout -k 10 200 python scratch/ab.py bench.py --steps 6 --warmup 2 --frames-per-step 5000 --cpu-frames 0 --literal-frames 0 --skip-host-input --api-frames 0 --gk-frames 0 --flux-atoms 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'pack+sort ms per launch', round(d['valu']['pack_kernel_ms_total']/8,2))"
}
run whole_sort || exit 1
MDX_RDF_TIMING_MASK=100 run no_rank_loop || exit 1
MDX_RDF_TIMING_MASK=101 run no_rank_loop_no_boxes || exit 1
MDX_RDF_UNFOLD=0 run no_sort || exit 1
