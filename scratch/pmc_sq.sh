#!/bin/bash
# The two SQ counter passes over one python script (no HBM passes):
# bash scratch/pmc_sq.sh <tag> <script.py> [args]; exported variables reach the script
set -e -o pipefail
tag=${1:?tag}; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$repo
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$out/$name" -- \
      python3 "$@" > "$out/$name.out" 2> "$out/$name.err"
  echo "$name done"
}
PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES" pass sq1 "$@"
PMC="SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" pass sq2 "$@"
cd "$repo"
python3 profiles/summarize_pmc.py "$out/sq1" "$out/sq2" > "$out/pmc.json"
find "$out" -name '*kernel_trace.csv' -delete
find "$out" -name '*counter_collection.csv' -delete
