"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs.

usage: python profiles/summarize_pmc.py [--meta key=value ...] gpurun_out/pmc_fetch \
           gpurun_out/pmc_write ... > out.json

``--meta`` entries go into the table's ``__meta__`` object; ``rdf_frames_per_launch`` says
how many frames one recorded launch of the RDF pair kernel processed, so that readers of
the table (bench.py) can scale its counters to launches of another size.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts
128-byte read requests as 64 bytes, so wide streaming reads show half their bytes;
the table keeps the raw counter and adds the doubled figure as `fetch_bytes_x2`.
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def main(argv):
    meta, dirs = {}, []
    it = iter(argv)
    for arg in it:
        if arg == "--meta":
            key, _, value = next(it).partition("=")
            meta[key] = float(value) if value.replace(".", "", 1).isdigit() else value
        else:
            dirs.append(arg)
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for d in dirs:
        for path in glob.glob(f"{d}/*/*counter_collection.csv"):
            for row in csv.DictReader(open(path)):
                key = f'{short(row["Kernel_Name"])} grid={row["Grid_Size"]}'
                table[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                durations[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out = {"__meta__": meta}
    for key, counters in sorted(table.items()):
        entry = {"launches_seen": max(len(v) for v in counters.values()),
                 "mean_duration_us_under_pmc": sum(durations[key]) / len(durations[key]) / 1e3}
        for name, vals in sorted(counters.items()):
            entry[name] = sum(vals) / len(vals)
        if "FETCH_SIZE" in entry:
            entry["fetch_bytes_raw"] = entry["FETCH_SIZE"] * 1024
            entry["fetch_bytes_x2"] = entry["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in entry:
            entry["write_bytes"] = entry["WRITE_SIZE"] * 1024
        out[key] = entry
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
