"""Scratch: from a rocprofv3 kernel trace of tools/bench_train.py, where the GPU idles inside one training step: idle time by the kernel that
follows the gap (top 25), and the step's busy / idle totals."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one K4 backward launch per step marks the steps
marks = [i for i, r in enumerate(rows) if "k_ray_stats_packed<true>" in r["Kernel_Name"]]
a, b = marks[-3], marks[-2]
seg = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
span = (int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
print(f"step span {span:.2f} ms, kernel time {busy:.2f} ms, launches {len(seg)}")
gaps = collections.defaultdict(lambda: [0.0, 0])
big = []
prev_end = int(seg[0]["End_Timestamp"])
for r in seg[1:] + [rows[b]]:
    s = int(r["Start_Timestamp"])
    g = max(0, s - prev_end) / 1e3
    k = r["Kernel_Name"][:60]
    gaps[k][0] += g; gaps[k][1] += 1
    if g > 100:
        big.append((g, k))
    prev_end = max(prev_end, int(r["End_Timestamp"]))
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{g / 1e3:7.2f} ms idle before {n:4d} x {k}")
seq = seg + [rows[b]]
pe = int(seq[0]["End_Timestamp"])
for i, r in enumerate(seq[1:], 1):
    g = (int(r["Start_Timestamp"]) - pe) / 1e3
    if g > 300:
        print(f"--- gap {g:.0f} us at launch {i} of {len(seq)}; before: " + " | ".join(x["Kernel_Name"][:45] for x in seq[max(0, i - 4):i]))
        print("    after: " + " | ".join(x["Kernel_Name"][:45] for x in seq[i:i + 4]))
    pe = max(pe, int(r["End_Timestamp"]))
print("gaps > 100 us:", [(round(g), k[:40]) for g, k in big])
