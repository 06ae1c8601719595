#!/bin/bash
# same-box A/B of two builds of the library on the whole cfg2 step: bench.py medians, alternating.  usage: ab_bench.sh <variant.so> [rounds]
L=nerf-det_amd/lib
cp $L/libnerfdet_hip.so /tmp/base.so
for r in $(seq 1 ${2:-3}); do
  cp /tmp/base.so $L/libnerfdet_hip.so; python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-serving > gpurun_out/abb_base_$r.json 2>gpurun_out/abb_err.log
  echo "base    $r: $(python -c "import json,sys; d=json.loads(open('gpurun_out/abb_base_$r.json').read().strip().splitlines()[-1]); print(d['median_ms'], d['p10_ms'], d['p90_ms'])")"
  cp $1 $L/libnerfdet_hip.so; python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-serving > gpurun_out/abb_var_$r.json 2>gpurun_out/abb_err.log
  echo "variant $r: $(python -c "import json,sys; d=json.loads(open('gpurun_out/abb_var_$r.json').read().strip().splitlines()[-1]); print(d['median_ms'], d['p10_ms'], d['p90_ms'])")"
done
cp /tmp/base.so $L/libnerfdet_hip.so
