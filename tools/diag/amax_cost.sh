#!/bin/bash
# What the amax commits cost with the operands unchanged: the convolution spans of a cfg2 step with the commits (production) and without them
# (layer_times.py ... no_amax_commit: every fp16-pair launch takes its own ndet_amax_f32 pass, which is not a convolution span).  Same box, alternating.
for r in 1 2 3; do
  PYTHONPATH=. python tools/layer_times.py cfg2 f16x2 > gpurun_out/amaxcost_with_$r.log 2>&1; echo "with commits    $r: $(grep 'sum of spans' gpurun_out/amaxcost_with_$r.log)"
  PYTHONPATH=. python tools/layer_times.py cfg2 f16x2 no_amax_commit > gpurun_out/amaxcost_without_$r.log 2>&1; echo "without commits $r: $(grep 'sum of spans' gpurun_out/amaxcost_without_$r.log) $(grep 'fallbacks' gpurun_out/amaxcost_without_$r.log)"
done
