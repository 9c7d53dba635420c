#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per launch.

usage: pmc_summary.py DIR [DIR ...]   (each DIR is one --pmc pass)
HBM bytes: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE reads exactly half
of the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM section), so the read side is
doubled:  traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.
"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if "vslam" not in name:
                continue
            name = name.replace("void ", "").replace("vslam::", "").split("<")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    row = {c: sum(v) / len(v) for c, v in cs.items()}          # mean per launch
    row["launches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        # totals over every launch of the run (a kernel may be launched once per octave)
        row["hbm_bytes_total"] = 2 * sum(cs["FETCH_SIZE"]) * 1024 + sum(cs["WRITE_SIZE"]) * 1024
    out[k] = row
print(json.dumps(out, indent=1))
