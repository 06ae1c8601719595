export TMPDIR=/tmp
PMC_SET=${PMC_SET:-"GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"}
timeout -k 5 120 rocprofv3 --pmc $PMC_SET --kernel-trace --output-format csv -d /tmp/pc -o run -- python3 tools/time_conv_tile.py 3257 > gpurun_out/pmc_halo.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pc/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(list); dur=[]
for r in csv.DictReader(open(f[0])):
    if 'halo' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value'])); 
        if r['Counter_Name']=='GRBM_GUI_ACTIVE': dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
print('dur_us', sum(dur)/len(dur))
PY
