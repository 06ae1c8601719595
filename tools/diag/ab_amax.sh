#!/bin/bash
# same-box comparison: sum of conv spans of a cfg2 step, f16x2 vs bf16x3, two rounds
for r in 1 2; do for a in f16x2 bf16x3; do
  python tools/layer_times.py cfg2 $a > gpurun_out/ab_${a}_${r}.log 2>&1
  echo "$a round $r: $(grep 'sum of spans' gpurun_out/ab_${a}_${r}.log)"
done; done
