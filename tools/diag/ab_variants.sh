#!/bin/bash
# same-box comparison of several builds of the library: sum of kernel spans of a cfg2 step (tools/layer_times.py), alternating, N rounds.
#   usage: ab_variants.sh <rounds> name=path[:ENV=VAL] ...     (the first entry should be the base build)
L=nerf-det_amd/lib
cp $L/libnerfdet_hip.so /tmp/keep.so
R=$1; shift
for r in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%=*}; rest=${spec#*=}; path=${rest%%:*}; envs=""; [ "$rest" != "$path" ] && envs=${rest#*:}
    cp $path $L/libnerfdet_hip.so
    env $envs PYTHONPATH=. python tools/layer_times.py cfg2 f16x2 > gpurun_out/abv_${name}_$r.log 2>&1
    echo "$name round $r: $(grep 'sum of spans' gpurun_out/abv_${name}_$r.log)"
  done
done
cp /tmp/keep.so $L/libnerfdet_hip.so
