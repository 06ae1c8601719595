# Scratch (GPU box): SQ / LDS counters of the bf16x3 convolution kernel on the 849-GFLOP layer, four rocprofv3 --pmc passes.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcconv
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pc$i -o run -- python3 tools/run_conv_once.py bf16x3 fpn 128 1 > gpurun_out/pmcconv/log$i.txt 2>&1 || { tail -5 gpurun_out/pmcconv/log$i.txt; }
  python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pc$i/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if 'k_conv_split' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, sum(v)/len(v), len(v))
PY
done
