"""Print the kernel timeline of the last complete step in a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "vslam" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_resize_linear2x" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    print("%8.1f %8.1f  q%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                  r.get("Queue_Id", "?"), r["Kernel_Name"][:80]))
