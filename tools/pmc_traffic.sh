# scratch (GPU box): FETCH_SIZE / WRITE_SIZE of one 3D convolution shape.  usage: bash tools/pmc_traffic.sh <lib.so> "cin cout X Y Z k tile splits"
export TMPDIR=/tmp
cp $1 nerf-det_amd/lib/libnerfdet_hip.so
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$C
  timeout -k 5 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pt_$C -o run -- python3 tools/run_conv3d_once.py $2 > /tmp/ptlog.txt 2>&1 || tail -3 /tmp/ptlog.txt
  python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pt_$C/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if 'k_conv_split' in r['Kernel_Name']: agg[r['Kernel_Name'][:40]+' '+r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, f"{sum(v)/len(v):.4g}", len(v))
PY
done
