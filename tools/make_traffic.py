#!/usr/bin/env python3
"""profiles/traffic.json from a pmc_summary.py file of a `bench.py --frames F --steps 1 --warmup 1` run.

    python tools/make_traffic.py gpurun_out/r02/pmc_f64.json 64 2 "r02 ..." > profiles/traffic.json

bytes per frame = hbm_bytes_total / (frames * steps_run); steps_run counts the warm-up step too.
"""
import json
import sys

d = json.load(open(sys.argv[1]))
frames, steps = int(sys.argv[2]), int(sys.argv[3])
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --frames %d --steps 1 --warmup 1 --cpu-sample 0 --modes 0`; "
               "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over all launches, divided by frames*steps; FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)" % frames,
       "_round": sys.argv[4] if len(sys.argv) > 4 else ""}
tot = 0.0
for k, v in sorted(d.items()):
    if k == "k_queue_probe":  # a one-off probe kernel an earlier build of the library ran once per context (not per frame)
        continue
    if "hbm_bytes_total" in v:
        per = v["hbm_bytes_total"] / (frames * steps)
        out[k] = {"hbm_bytes_per_frame": per, "launches_per_step": v["launches"] // steps,
                  "valu_wave_instr_per_frame": v.get("SQ_INSTS_VALU", 0) * v["launches"] / (frames * steps)}
        tot += per
out["_total_bytes_per_frame"] = tot
print(json.dumps(out, indent=1))
