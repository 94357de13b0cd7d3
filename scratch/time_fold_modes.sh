#!/bin/bash
# headline step rate of a -DMDX_TIMING_SWITCHES build with the sorted-frame loop forced to
# one variant (counts wrong with a forced mask): bash scratch/time_fold_modes.sh lib.so
cd "$(dirname "$0")/.."
lib=${1:?library}
run() {
  ALT=$lib timeout -k 10 200 python scratch/ab.py bench.py --steps 10 --warmup 3 --frames-per-step 5000 --cpu-frames 0 --literal-frames 0 --skip-host-input --api-frames 0 --gk-frames 0 --flux-atoms 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['roofline']['avg_launch_ms'],2), round(d['valu']['pack_kernel_ms_total']/13,2))"
}
run sorted_masks_from_boxes || exit 1
MDX_RDF_UNFOLD=0 run unsorted || exit 1
MDX_RDF_TIMING_MASK=7 run sorted_all_folded || exit 1
MDX_RDF_TIMING_MASK=0 run sorted_nothing_folded || exit 1
MDX_RDF_FAST_NO_EDGE=1 run sorted_no_edge_band || exit 1
