#!/bin/bash
# the RDF parity tests and a short headline run: bash scratch/quick_rdf.sh <tag>
cd "$(dirname "$0")/.."
tag=${1:?tag}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -u -m pytest tests/test_gpu_rdf_guards.py tests/test_gpu_parity_at_size.py tests/test_gpu_rdf_parity.py tests/test_gpu_full_size.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$tag/tests.log 2>&1
rc=$?
tail -3 gpurun_out/$tag/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --frames-per-step 5000 --literal-frames 0 --skip-host-input --api-frames 0 --gk-frames 0 --flux-atoms 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/$tag/bench.json')); print(round(d['value']), round(d['roofline']['avg_launch_ms'],2), round(d['valu']['pack_kernel_ms_total']/13,2), d['cpu_baseline']['gpu_matches_bit_exact'], round(d['valu']['frac_of_issue_bound'],3))"
