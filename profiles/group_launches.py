"""RDF kernel launches of a rocprofv3 kernel trace, grouped by kernel and grid.

usage: python profiles/group_launches.py <dir with *_kernel_trace.csv> > out.csv
"""
import collections
import csv
import glob
import re
import sys


def main(directory):
    groups = collections.defaultdict(list)
    for path in glob.glob(f"{directory}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("(anonymous namespace)::", "")
            name = name.split("(")[0]
            if "rdf_" not in name:
                continue
            key = (name, row["Grid_Size_X"], row["Grid_Size_Y"])
            groups[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "grid_x", "grid_y", "launches", "mean_ms", "min_ms", "max_ms"])
    for (name, gx, gy), ms in sorted(groups.items()):
        out.writerow([name, gx, gy, len(ms), f"{sum(ms) / len(ms):.4f}", f"{min(ms):.4f}",
                      f"{max(ms):.4f}"])


if __name__ == "__main__":
    main(sys.argv[1])
