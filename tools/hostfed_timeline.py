"""Timeline of one steady-state batch of the host-fed pipeline from a rocprofv3 kernel + memory-copy trace:
    python tools/hostfed_timeline.py <trace dir>
prints, for the third batch from the end, the H2D / D2H copies and the main kernels with start / end in ms."""
import csv, glob, sys
d = sys.argv[1]
k = [r for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
m = [r for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
ev = []
for r in k:
    n = r["Kernel_Name"]
    if "vslam" not in n:
        continue
    short = n.split("(")[0].replace("void ", "").replace("vslam::", "")[:46]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r["Queue_Id"], short)))
for r in m:  # this rocprofv3 reports no sizes: keep the copies that take longer than 0.2 ms
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e_ - s_ < 200000:
        continue
    ev.append((s_, e_, "C %s (stream %s)" % (r["Direction"].replace("MEMORY_COPY_", ""), r["Stream_Id"])))
ev.sort()
ups = [i for i, e in enumerate(ev) if e[2].startswith("K") and "k_resize_linear2x_slide" in e[2]]
# batches start at every second upsample launch (two halves per batch)
starts = ups[::2]
if len(starts) < 5:
    print("too few batches in the trace", len(starts)); sys.exit(0)
i0, i1 = starts[-4], starts[-3]
t0 = ev[i0][0]
print("batch period %.3f ms" % ((ev[i1][0] - t0) / 1e6))
for s, e, n in ev:
    if s >= t0 - 3e6 and s < ev[i1][0] and ("C " in n[:2] or any(x in n for x in ("k_pyr_octave", "k_resize", "k_extrema_w3", "k_harris", "k_gauss", "k_pack"))):
        print("%9.3f %9.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, n))
