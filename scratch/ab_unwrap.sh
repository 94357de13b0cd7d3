#!/bin/bash
# times scratch/time_unwrap.py against the in-tree build and every alternative build under scratch/alt/
set -e
cd "$(dirname "$0")/.."
echo "== in-tree build"
timeout -k 10 120 python -u scratch/time_unwrap.py "$@" 2>&1
for lib in scratch/alt/libmdx_q*.so; do
  echo "== $lib"
  ALT=$lib timeout -k 10 120 python -u scratch/ab.py scratch/time_unwrap.py "$@" 2>&1 | grep -v "block-per-atom"
done
