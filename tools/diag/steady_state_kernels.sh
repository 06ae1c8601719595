# GPU box: which kernels does ONE steady-state inference step launch?  Two rocprofv3 kernel traces of bench.py that differ only in the number of timed
# steps (20 and 60); per kernel (launches_60 - launches_20) / 40 and the same for the time.  One-time work (weight packs, warm-up, the probes after
# the timed region) cancels.
#   bash tools/diag/steady_state_kernels.sh > gpurun_out/steady_state_kernels.txt
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ss20 /tmp/ss60
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ss20 -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving --no-train-probe > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ss60 -o run -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-serving --no-train-probe > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        out[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return out
a, b = load("/tmp/ss20"), load("/tmp/ss60")
rows = []
for k, (n60, t60) in b.items():
    n20, t20 = a.get(k, (0, 0.0))
    dn, dt = (n60 - n20) / 40.0, (t60 - t20) / 40.0 / 1e3
    if dn > 0.01:
        rows.append((dt, dn, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"steady-state step: {sum(r[1] for r in rows):.1f} launches, {tot / 1e3:.3f} ms of kernel time")
for dt, dn, k in rows:
    print(f"{dn:7.2f} x {dt / max(dn, 1e-9):8.1f} us = {dt:8.1f} us/step  {k[:150]}")
PY
