"""Measured tile / split-K choices of the MFMA convolution on MI355X for the layer shapes of the nerfdet configs
(cfg2: 50 views 240x320, 40x40x16 voxels).  Produced by tools/tune_conv3d.py and tools/tune_conv2d.py (interleaved
sweeps over tile in (64, 128) x splits in (1..8), median of 5 launches); key = (GEMM rows M, Cout, K steps of 32
channels, transposed).  Shapes not listed fall back to the heuristic in conv3d.choose_tiling."""

TUNED = {
    (240000, 256, 216, 0): (128, 1),  # fpn-like 256->256 @(50*60)x80x1 3x3x1?: 6892 us
    (25600, 256, 216, 0): (128, 3),  # down0.conv 256->256 @40x40x16: 815 us
    (25600, 128, 216, 0): (128, 6),  # out0 256->128 @40x40x16: 426 us
    (3200, 512, 216, 0): (64, 8),  # down1.conv1 256->512 s2: 261 us
    (3200, 512, 432, 0): (64, 8),  # down1.conv2 512->512 @20x20x8: 465 us
    (3200, 512, 8, 0): (64, 1),  # down1.ds 1x1 s2 256->512: 26 us
    (3200, 128, 432, 0): (128, 8),  # out1 512->128 @20x20x8: 150 us
    (400, 1024, 432, 0): (128, 8),  # down2.conv1 512->1024 s2: 152 us
    (400, 1024, 864, 0): (128, 8),  # down2.conv2 1024->1024 @10x10x4: 272 us
    (400, 128, 864, 0): (64, 8),  # out2 1024->128 @10x10x4: 115 us
    (400, 512, 32, 1): (64, 1),  # up2.convT 1024->512: 61 us
    (3200, 256, 16, 1): (128, 1),  # up1.convT 512->256: 96 us
    (25600, 25, 108, 0): (64, 6),  # head 128->25 @40x40x16: 147 us
    (240000, 64, 8, 0): (64, 1),  # l1.conv1 1x1 256->64: 112 us
    (240000, 64, 18, 0): (64, 1),  # l1.conv2 3x3 64->64: 228 us
    (240000, 256, 2, 0): (128, 1),  # l1.conv3 1x1 64->256 +res: 196 us
    (60000, 128, 16, 0): (128, 1),  # l2.conv1 1x1 512->128: 91 us
    (60000, 128, 36, 0): (128, 1),  # l2.conv2 3x3 128->128: 182 us
    (60000, 512, 4, 0): (64, 1),  # l2.conv3 1x1 128->512 +res: 127 us
    (15000, 256, 32, 0): (128, 1),  # l3.conv1 1x1 1024->256: 90 us
    (15000, 256, 72, 0): (128, 1),  # l3.conv2 3x3 256->256: 181 us
    (15000, 1024, 8, 0): (128, 1),  # l3.conv3 1x1 256->1024 +res: 105 us
    (4000, 512, 64, 0): (128, 2),  # l4.conv1 1x1 2048->512: 97 us
    (4000, 512, 144, 0): (128, 2),  # l4.conv2 3x3 512->512: 194 us
    (4000, 2048, 16, 0): (128, 1),  # l4.conv3 1x1 512->2048 +res: 93 us
    (240000, 256, 8, 0): (128, 1),  # fpn.lat0 1x1 256->256: 327 us
    (240000, 256, 72, 0): (128, 1),  # fpn.out0 3x3 256->256: 2337 us
}


# (M, Cout, K-steps, transposed) -> (tile, splits) for the bf16x3 kernel (csrc/conv_split_kernels.hip); filled from
# tools/tune_conv_split.py sweeps on MI355X
TUNED_SPLIT = {}
