"""Print the kernel timeline of the last full batch in a rocprofv3 kernel trace.

    python tools/timeline.py <..._kernel_trace.csv>     (the CSV file itself, not its directory)
"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mainq = next(r['Queue_Id'] for r in rows if 'k_pyr_octave' in r['Kernel_Name'])  # the second half's upsample runs on a side queue
idx = [i for i, r in enumerate(rows) if 'k_resize_linear2x_slide' in r['Kernel_Name'] and r['Queue_Id'] == mainq]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    if 'at::native' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']:
        continue
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:8.3f} {(int(r['End_Timestamp'])-t0)/1e6:8.3f}  q{r['Queue_Id']:>2} {r['Kernel_Name'][:64]}")
