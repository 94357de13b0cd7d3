"""Prints selected kernels of a pmc.json: python scratch/show_pmc.py file.json substr [substr...]"""
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if k == "__meta__" or not any(s in k for s in sys.argv[2:]):
        continue
    print(k)
    for name, val in v.items():
        print(f"   {name:32s} {val:.6g}")
