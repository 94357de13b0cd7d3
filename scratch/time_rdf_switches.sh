#!/bin/bash
# headline step rate of a -DMDX_TIMING_SWITCHES build with the near-edge path switched off
# in parts (counts are wrong on purpose with either switch): bash scratch/time_rdf_switches.sh lib.so
cd "$(dirname "$0")/.."
lib=${1:?library}
run() {
  ALT=$lib timeout -k 10 200 python scratch/ab.py bench.py --steps 10 --warmup 3 --frames-per-step 5000 --cpu-frames 0 --literal-frames 0 --skip-host-input --api-frames 0 --gk-frames 0 --flux-atoms 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['roofline']['avg_launch_ms'],2))"
}
run exact
MDX_RDF_TIMING_MASK=200 run noted_pairs_never_resolved
MDX_RDF_FAST_NO_EDGE=1 run near_edge_band_of_width_zero
