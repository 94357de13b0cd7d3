#!/bin/bash
# Collects one round's profile set on a 1-GPU box: the bench line under the flags the
# round-end driver uses, the kernel statistics of the same command and four counter
# passes.  usage: bash profiles/collect.sh <tag>
# Results land under gpurun_out/<tag>/; profiles/summarize_pmc.py and
# profiles/group_launches.py turn them into the files kept under profiles/.
set -e -o pipefail
tag=${1:?tag}
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
flags="--steps 20 --warmup 5"
frames_per_launch=5000   # 100 000 frames over 20 timed steps, one RDF launch each
timeout -k 10 600 python3 "$repo/bench.py" $flags > "$out/bench_n1.json" 2> "$out/bench_n1.err"
echo bench done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- \
    python3 "$repo/bench.py" $flags > "$out/bench_prof.json" 2> "$out/stats.err"
echo stats done
# the counter passes leave out what launches RDF kernels of other shapes (CPU-baseline
# checks, host-input blocks, the API run) so that every recorded launch is a full step
quick="$flags --cpu-frames 0 --literal-frames 0 --skip-host-input --api-frames 0"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv \
    -d "$out/pmc_fetch" -- python3 "$repo/bench.py" $quick > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err"
echo fetch done
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv \
    -d "$out/pmc_write" -- python3 "$repo/bench.py" $quick > "$out/pmc_write.json" 2> "$out/pmc_write.err"
echo write done
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
    SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv \
    -d "$out/pmc_sq1" -- python3 "$repo/bench.py" $quick --gk-frames 0 > "$out/pmc_sq1.json" 2> "$out/pmc_sq1.err"
echo sq1 done
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS \
    SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace \
    --output-format csv -d "$out/pmc_sq2" -- python3 "$repo/bench.py" $quick --gk-frames 0 \
    > "$out/pmc_sq2.json" 2> "$out/pmc_sq2.err"
echo sq2 done
cd "$repo"
python3 profiles/summarize_pmc.py --meta rdf_frames_per_launch=$frames_per_launch \
    --meta "bench_flags=$quick" \
    "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq1" "$out/pmc_sq2" > "$out/pmc.json"
python3 profiles/group_launches.py "$out/stats" > "$out/rdf_by_launch_ms.csv"
# the raw traces are large; keep the summaries only
find "$out" -name '*kernel_trace.csv' -delete
find "$out" -name '*counter_collection.csv' -delete
ls "$out"
