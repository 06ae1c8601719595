# Scratch (GPU box): SQ / LDS counters of the fp16-pair halo tiles on the neck's 256->256 3x3x3 layer (40x40x16), separate rocprofv3 --pmc passes.
# usage: bash tools/diag/pmc_halo_f16.sh <tile>
export TMPDIR=/tmp
T=${1:-3257}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU"; do
  i=$((i+1))
  rm -rf /tmp/ph$i
  timeout -k 5 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/ph$i -o run -- python3 tools/run_conv3d_once.py 256 256 40 40 16 3 $T 1 f16x2 > /tmp/phlog$i.txt 2>&1 || tail -3 /tmp/phlog$i.txt
  python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/ph$i/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if 'k_conv_split_halo' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print("tile $T", k, f"{sum(v)/len(v):.4g}", len(v))
PY
done
