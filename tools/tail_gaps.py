"""Scratch: from a rocprofv3 kernel trace of bench.py, the timeline of the END of a step (from the last neck convolution to the first
kernel of the next step): where the head / post-processing stage spends its time."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# find the launches of k_pack_detections (one per step); print the 30 kernels before and 3 after for a late step
idx = [i for i, r in enumerate(rows) if "k_pack_detections" in r["Kernel_Name"]]
i = idx[-3]
t0 = int(rows[i - 34]["Start_Timestamp"])
prev_end = None
for r in rows[i - 34:i + 6]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {gap:7.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
