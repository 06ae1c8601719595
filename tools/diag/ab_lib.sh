#!/bin/bash
# same-box A/B of two builds of the library: layer times of a cfg2 step with each, two rounds.  usage: ab_lib.sh <variant.so>
L=nerf-det_amd/lib
cp $L/libnerfdet_hip.so /tmp/base.so
for r in 1 2; do
  cp /tmp/base.so $L/libnerfdet_hip.so; python tools/layer_times.py cfg2 f16x2 > gpurun_out/ab_base_$r.log 2>&1; echo "base round $r: $(grep 'sum of spans' gpurun_out/ab_base_$r.log)"
  cp $1 $L/libnerfdet_hip.so; python tools/layer_times.py cfg2 f16x2 > gpurun_out/ab_var_$r.log 2>&1; echo "variant round $r: $(grep 'sum of spans' gpurun_out/ab_var_$r.log)"
done
cp /tmp/base.so $L/libnerfdet_hip.so
paste <(awk 'NR>2{printf "%-40s %7s\n",$2,$3}' gpurun_out/ab_base_2.log) <(awk 'NR>2{printf "%7s\n",$3}' gpurun_out/ab_var_2.log) | awk '{d=$3-$2; if (d>0.004||d<-0.004) printf "%2d %s %+0.3f\n", NR-1, $0, d}'
