#!/bin/bash
# GPU box: the fp16-pair halo tiles with one ingredient of the K step removed at a time (diagnostic builds: wrong results, same instruction
# The three libdiag_*.so are builds of csrc with tools/diag/halo_diag.patch applied and -DHALO_DIAG_NO_DMA / _NO_BARRIER / _NO_READS (one each), linked like the Makefile links libnerfdet_hip.so.
# streams otherwise): which one sets the step?  libdiag_NO_DMA: the producers issue no weight DMA after the prologue; libdiag_NO_BARRIER: no
# per-step barrier on either side; libdiag_NO_READS: the consumers multiply stale registers (no fragment reads after step 0).
L=nerf-det_amd/lib
cp $L/libnerfdet_hip.so /tmp/base.so
for v in base NO_DMA NO_BARRIER NO_READS; do
  if [ $v = base ]; then cp /tmp/base.so $L/libnerfdet_hip.so; else cp $L/libdiag_$v.so $L/libnerfdet_hip.so; fi
  echo "== $v"
  TUNE_TILES=3257,3256,3128 TUNE_LAYERS=fpn.out0 TUNE_VERBOSE=1 timeout -k 10 120 python tools/tune_conv2d.py f16x2 2>&1 | grep "s= 1 "
  TUNE_TILES=3257,3256,3128 TUNE_LAYERS=down0 timeout -k 10 120 python tools/tune_conv3d.py f16x2 2>&1 | grep "splits=1 "
done
cp /tmp/base.so $L/libnerfdet_hip.so
