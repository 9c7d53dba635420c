"""Kernel timeline of the last full batch of a matrix-path run in a rocprofv3 kernel trace (batches start at octave 0's
k_pyr_octave_mx, the kernel with the fused upsample).

    python tools/timeline_mx.py <..._kernel_trace.csv>
"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_pyr_octave_mx' in r['Kernel_Name'] and 'Li9ELi13E' in r['Kernel_Name'].replace(' ', '').replace('9,13', 'Li9ELi13E')]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
first = min(i0, next(i for i in range(i0, -1, -1) if int(rows[i]['End_Timestamp']) < t0 - 2_000_000 or i == 0))
for r in rows[i0 - 6:i1]:
    if 'at::native' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']:
        continue
    n = r['Kernel_Name']
    n = n.replace('void vslam::', '').replace('vslam::', '')
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:8.3f} {(int(r['End_Timestamp'])-t0)/1e6:8.3f}  q{r['Queue_Id']:>2} {n[:70]}")
