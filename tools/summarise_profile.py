"""Turn rocprofv3 output directories into the summaries committed under profiles/.

  python tools/summarise_profile.py stats  <kernel-trace dir> <out.csv> [--last-steps K --steps-total T]
  python tools/summarise_profile.py pmc    <FETCH_SIZE dir> <WRITE_SIZE dir> <out.json>

`stats` groups the kernel trace by (shortened) kernel name: launches, average / total duration, share of GPU time.
`pmc` averages FETCH_SIZE / WRITE_SIZE (KB) per kernel and derives the projection kernel's HBM traffic the way
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE doubled for wide coalesced reads, WRITE_SIZE exact).
"""
import csv
import glob
import json
import os
import sys
from collections import OrderedDict


def short(name: str) -> str:
    name = name.strip('"')
    for cut in ("(", ):
        if cut in name and not name.startswith("void k_") and not name.startswith("k_"):
            name = name.split(cut)[0]
    return name[:96]


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[0]


def stats(trace_dir, out_csv, header="", exclude=()):
    rows = [r for r in csv.DictReader(open(find(trace_dir, "*kernel_trace.csv"))) if not any(e in r["Kernel_Name"] for e in exclude)]
    agg = OrderedDict()
    for r in rows:
        k = short(r["Kernel_Name"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    with open(out_csv, "w") as f:
        if header:
            f.write(f"# {header}\n")
        f.write("kernel,launches,avg_us,min_us,max_us,total_ms,percent\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{k}\",{a[0]},{a[1] / a[0]:.2f},{a[2]:.2f},{a[3]:.2f},{a[1] / 1e3:.3f},{100 * a[1] / total:.2f}\n")
    print(f"wrote {out_csv}: {len(agg)} kernels, {total / 1e3:.1f} ms of GPU time")


def counter(d, name):
    agg = OrderedDict()
    for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        if r["Counter_Name"] != name:
            continue
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: {"launches": a[0], "avg_kb": a[1] / a[0]} for k, a in agg.items()}


def pmc(fetch_dir, write_dir, out_json, command="", workload="cfg2", algorithmic_bytes=None):
    f = counter(fetch_dir, "FETCH_SIZE")
    w = counter(write_dir, "WRITE_SIZE")
    k1 = next((k for k in f if "k_backproject_aggregate" in k and "bwd" not in k), None)
    if k1 is None:      # a profile without the inference path (the training bench): per-kernel table only
        k1 = next(iter(f))
    traffic = int(round((2.0 * f[k1]["avg_kb"] + w[k1]["avg_kb"]) * 1024))
    ours = {}
    for k in f:
        if not (k.startswith("void k_") or k.startswith("k_")):
            continue
        name = k[5:] if k.startswith("void ") else k
        name = name.split("(")[0].replace(", ", ",")
        ours[name] = {"fetch_kb_raw": f[k]["avg_kb"], "write_kb": w.get(k, {"avg_kb": 0.0})["avg_kb"], "launches": f[k]["launches"],
                      "traffic_bytes": int(round((2.0 * f[k]["avg_kb"] + w.get(k, {"avg_kb": 0.0})["avg_kb"]) * 1024))}
    out = {
        "command": command,
        "note": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide (16 B/lane) coalesced reads -> doubled for the projection "
                "kernel (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact. Counters are L2 memory-side (fabric) requests; "
                "Infinity-Cache hits are included.",
        "workload": workload,
        "k1": {"kernel": k1, "fetch_kb_raw": f[k1]["avg_kb"], "write_kb": w[k1]["avg_kb"], "traffic_bytes": traffic,
               "algorithmic_bytes": algorithmic_bytes},
        "kernels": ours,
        "per_kernel_raw_kb": {"FETCH_SIZE": f, "WRITE_SIZE": w},
    }
    json.dump(out, open(out_json, "w"), indent=1)
    print(f"wrote {out_json}: K1 traffic {traffic / 1e6:.1f} MB per launch")


def counters(d, out_json, command=""):
    """Average of every collected counter per kernel (one rocprofv3 --pmc pass), plus the dispatch duration: the MFMA-busy / clock
    evidence for the convolution kernels (MI355X_MICROARCH.md: effective clock = GRBM_GUI_ACTIVE / 8 / wall time)."""
    agg = OrderedDict()
    for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        if not (k.startswith("void k_") or k.startswith("k_")):
            continue
        a = agg.setdefault(k, {"launches": set(), "dur_us": []})
        a.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        did = r.get("Dispatch_Id", len(a["dur_us"]))
        if did not in a["launches"]:
            a["launches"].add(did)
            if "End_Timestamp" in r and "Start_Timestamp" in r:
                a["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {"command": command, "kernels": {}}
    for k, a in agg.items():
        row = {"launches": len(a["launches"]), "avg_dur_us": sum(a["dur_us"]) / max(1, len(a["dur_us"]))}
        for name, vals in a.items():
            if name not in ("launches", "dur_us"):
                row[name] = sum(vals) / len(vals)
        if "GRBM_GUI_ACTIVE" in row and row["avg_dur_us"] > 0:
            row["effective_clock_GHz"] = row["GRBM_GUI_ACTIVE"] / 8.0 / (row["avg_dur_us"] * 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "GRBM_GUI_ACTIVE" in row and row["GRBM_GUI_ACTIVE"] > 0:
            # MFMA-busy cycles summed over the 1024 SIMDs of the chip / (cycles of the dispatch x 1024): fraction of the pipe in use
            row["mfma_pipe_busy_frac"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (row["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        out["kernels"][k] = row
    json.dump(out, open(out_json, "w"), indent=1)
    print(f"wrote {out_json}: {len(out['kernels'])} kernels")


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        # --exclude=a,b drops kernels whose name contains a or b (e.g. MIOpen's find-mode trial kernels "naive_conv" of the warm-up)
        ex = [a.split("=", 1)[1].split(",") for a in sys.argv[4:] if a.startswith("--exclude=")]
        stats(sys.argv[2], sys.argv[3], header=" ".join(a for a in sys.argv[4:] if not a.startswith("--exclude=")), exclude=ex[0] if ex else ())
    elif mode == "counters":
        counters(sys.argv[2], sys.argv[3], command=" ".join(sys.argv[4:]))
    elif mode == "pmc":
        wl = [a.split("=", 1)[1] for a in sys.argv[5:] if a.startswith("--workload=")]
        extra = [a for a in sys.argv[5:] if not a.startswith("--workload=")]
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], command=extra[0] if extra else "", workload=wl[0] if wl else "cfg2",
            algorithmic_bytes=int(extra[1]) if len(extra) > 1 else None)
    else:
        raise SystemExit(__doc__)
