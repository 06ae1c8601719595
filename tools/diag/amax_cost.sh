#!/bin/bash
# What the amax commits cost with the operands unchanged: the convolution spans of a cfg2 step with the commits (production) and without them
# (NDET_NO_AMAX_COMMIT=1: every fp16-pair launch takes its own ndet_amax_f32 pass, which is not a convolution span).  Same box, alternating.
for r in 1 2 3; do
  PYTHONPATH=. python tools/layer_times.py cfg2 f16x2 > gpurun_out/amaxcost_with_$r.log 2>&1; echo "with commits    $r: $(grep 'sum of spans' gpurun_out/amaxcost_with_$r.log)"
  NDET_NO_AMAX_COMMIT=1 PYTHONPATH=. python tools/layer_times.py cfg2 f16x2 > gpurun_out/amaxcost_without_$r.log 2>&1; echo "without commits $r: $(grep 'sum of spans' gpurun_out/amaxcost_without_$r.log) $(grep 'fallbacks' gpurun_out/amaxcost_without_$r.log)"
done
