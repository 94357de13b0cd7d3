#!/bin/bash
# FETCH_SIZE and launch time of the fast RDF kernel for several chunk sizes (frames per chunk)
cd /tmp && export TMPDIR=/tmp
repo=${GRAFT_REPO_ROOT:?}
mkdir -p $repo/gpurun_out/chunk_scan
for c in 2 4 8 16; do
  export MDX_RDF_FRAMES_PER_CHUNK=$c
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $repo/gpurun_out/chunk_scan/c$c -- \
    python3 $repo/bench.py --steps 6 --warmup 2 --frames-per-step 5000 --cpu-frames 0 --literal-frames 0 --skip-host-input --api-frames 0 --gk-frames 0 --flux-atoms 0 > $repo/gpurun_out/chunk_scan/c$c.json 2> $repo/gpurun_out/chunk_scan/c$c.err
  python3 - <<PY
import csv, glob, json
rows=[r for p in glob.glob("$repo/gpurun_out/chunk_scan/c$c/*/*counter_collection.csv") for r in csv.DictReader(open(p)) if "rdf_fast_kernel" in r["Kernel_Name"]]
full=[r for r in rows]
f=[float(r["Counter_Value"]) for r in full]; d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in full]
line=json.loads(open("$repo/gpurun_out/chunk_scan/c$c.json").read().strip().splitlines()[-1])
print("frames per chunk $c: launches", len(f), "mean FETCH_SIZE %.3f GB raw" % (sum(f)/len(f)*1024/1e9), "mean launch %.2f ms" % (sum(d)/len(d)), "bench value %.0f" % line["value"])
PY
done
find $repo/gpurun_out/chunk_scan -name '*.csv' -delete
