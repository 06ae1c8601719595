# GPU box: the rocprofv3 evidence of one round, summarised into gpurun_out/prof_<tag>_* (copy what is to be judged into profiles/).
#   bash tools/profile_round.sh r02_a
set -e
TAG=${1:-r02_a}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write /tmp/p_mfma /tmp/p_tstats /tmp/p_tfetch /tmp/p_twrite
echo "== kernel trace, inference bench"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving --no-train-probe > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}_bench.err
python3 $R/tools/summarise_profile.py stats /tmp/p_stats $R/gpurun_out/prof_${TAG}_bench_cfg2_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving --no-train-probe (all launches incl. warm-up)"
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_${TAG}_bench_cfg2_rocprofv3_kernel_stats_raw.csv
echo "== PMC FETCH_SIZE / WRITE_SIZE, separate passes"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-serving --no-train-probe > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-serving --no-train-probe > /dev/null 2>&1
python3 $R/tools/summarise_profile.py pmc /tmp/p_fetch /tmp/p_write $R/gpurun_out/prof_${TAG}_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-serving --no-train-probe (two separate passes)" 272179200
echo "== PMC MFMA busy / clock"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p_mfma -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-serving --no-train-probe > /dev/null 2>&1
python3 $R/tools/summarise_profile.py counters /tmp/p_mfma $R/gpurun_out/prof_${TAG}_pmc_mfma_busy.json "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-serving --no-train-probe"
echo "== training bench: kernel trace + PMC traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tstats -o run -- python3 $R/tools/bench_train.py --steps 20 --warmup 3 > $R/gpurun_out/prof_${TAG}_train.json 2> $R/gpurun_out/prof_${TAG}_train.err
python3 $R/tools/summarise_profile.py stats /tmp/p_tstats $R/gpurun_out/prof_${TAG}_train_cfg3_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 tools/bench_train.py --steps 20 --warmup 3 (23 steps; MIOpen find-mode trial kernels of the warm-up excluded)" --exclude=naive_conv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_tfetch -o run -- python3 $R/tools/bench_train.py --steps 3 --warmup 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_twrite -o run -- python3 $R/tools/bench_train.py --steps 3 --warmup 3 > /dev/null 2>&1
python3 $R/tools/summarise_profile.py pmc /tmp/p_tfetch /tmp/p_twrite $R/gpurun_out/prof_${TAG}_train_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/bench_train.py --steps 3 --warmup 3 (two separate passes)" --workload=cfg3-train
echo done
