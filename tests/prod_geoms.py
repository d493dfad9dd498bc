"""Convolution geometries that select the PRODUCTION tile instantiations (the ones the BASELINE configs launch at full
batch), shared by the GPU parity test (test_gpu_production_tiles.py) and the CPU coverage proof (test_kernel_coverage.py).

Tuple: (N, H, W, C, K, k, stride, pad).  Sizes are the real layer shapes of WRN-28-10 b128 / WRN-50-2 (spec B) at a
batch that keeps the same kernel choice as b256 but lets the torch-CPU reference finish in a few seconds."""

PROD_GEOMS = [
    # ---- WRN-28-10, CIFAR, batch 128 (config 3) ----
    (128, 32, 32, 160, 160, 3, 1, 1),     # stage 1: igemm8r<256x160> (512 tiles; round 3: igemm_patch<128x160>, 1,024 tiles), wgrad<160x160> split 56 ways
    (128, 32, 32, 160, 320, 3, 2, 1),     # stage 2 entry: stride-2 forward + 4-class dgrad
    (128, 32, 32, 160, 320, 1, 2, 0),     # projection shortcut 160 -> 320
    (128, 16, 16, 320, 320, 3, 1, 1),     # stage 2: 512 tiles
    (128, 16, 16, 320, 640, 3, 2, 1),     # stage 3 entry
    (128, 16, 16, 320, 640, 1, 2, 0),
    (128, 8, 8, 640, 640, 3, 1, 1),       # stage 3: 256 tiles -> wave-specialised kernel
    (128, 32, 32, 0, 160, 3, 1, 1),       # the stem on the MFMA route (C = 0: one 16-byte chunk of the dtype; im2col wgrad)
    # ---- WRN-50-2 spec B, 224 x 224 (config 5) at batch 16 / 64 ----
    (16, 56, 56, 128, 128, 3, 1, 1),      # 3x3 @56: igemm_dma<128x128>
    (16, 56, 56, 512, 128, 1, 1, 0),      # 1x1 reduce, dense_src, > 256 tiles
    (16, 56, 56, 128, 512, 1, 1, 0),      # 1x1 expand
    (16, 56, 56, 512, 256, 1, 1, 0),      # stage-2 entry 1x1 (Cb = C/2 when downsampling)
    (16, 56, 56, 256, 256, 3, 2, 1),      # stride-2 3x3 of a bottleneck
    (16, 56, 56, 512, 1024, 1, 2, 0),     # projection 512 -> 1024 stride 2
    (64, 14, 14, 512, 512, 3, 1, 1),      # 3x3 @14
    (64, 14, 14, 2048, 512, 1, 1, 0),
    (64, 7, 7, 1024, 1024, 3, 1, 1),      # 3x3 @7
    (64, 7, 7, 1024, 4096, 1, 1, 0),
    (8, 224, 224, 0, 512, 7, 2, 3),       # 7x7 stride-2 stem, 512 output channels, im2col wgrad
    (8, 224, 224, 0, 256, 7, 2, 3),       # spec A's stem
    # ---- thin nets at batch 128 (configs 2, 4) ----
    (128, 32, 32, 0, 16, 3, 1, 1),        # ResNet-20 stem
    (128, 32, 32, 0, 64, 3, 1, 1),        # ResNet-v2-164 stem
    (128, 32, 32, 16, 16, 3, 1, 1),       # ResNet-20 stage 1: igemm_dma<256x32>
    (128, 32, 32, 16, 32, 3, 2, 1),
    (128, 16, 16, 32, 32, 3, 1, 1),
    (128, 8, 8, 64, 64, 3, 1, 1),
    (128, 32, 32, 64, 16, 1, 1, 0),       # v2-164 bottleneck 1x1 reduce / expand
    (128, 32, 32, 16, 64, 1, 1, 0),
    (128, 32, 32, 64, 32, 1, 1, 0),
    (128, 32, 32, 32, 32, 3, 2, 1),
    (128, 16, 16, 32, 128, 1, 1, 0),
    (128, 32, 32, 64, 128, 1, 2, 0),
    (128, 8, 8, 64, 256, 1, 1, 0),
    (128, 8, 8, 256, 64, 1, 1, 0),
    (128, 16, 16, 128, 256, 1, 2, 0),
    (128, 16, 16, 128, 64, 1, 1, 0),
    (128, 16, 16, 128, 32, 1, 1, 0),
    (128, 16, 16, 64, 64, 3, 2, 1),
]

# WRN-50-2 layer shapes at a batch that keeps >= 192 tiles of 256 x 256, i.e. the eight-phase kernels (conv_igemm8.hip) the full-batch configuration
# launches; run with every operand set those kernels are specialised for (tests/test_gpu_production_tiles.py::test_igemm8_production_operand_sets)
IGEMM8_GEOMS = [
    (128, 14, 14, 512, 512, 3, 1, 1),     # 3x3 @14, forward and data gradient: 196 tiles, 72 K tiles
    (64, 28, 28, 256, 1024, 1, 1, 0),     # 1x1 expand: 784 tiles of 4 K tiles forward; data gradient 196 tiles of 16
    (64, 28, 28, 1024, 256, 1, 1, 0),     # 1x1 reduce
    (128, 28, 28, 512, 512, 3, 2, 1),     # stride-2 3x3 forward (its data gradient: four parity classes on the 128-row kernels)
    (64, 56, 56, 512, 1024, 1, 2, 0),     # stride-2 projection shortcut, forward
    (16, 56, 56, 128, 128, 3, 1, 1),      # stage 1 of WRN-50-2: 128 output channels -> column tiles of 128 (igemm8<256x128>), 196 tiles
    (16, 56, 56, 512, 128, 1, 1, 0),      # 1x1 reduce to 128 channels (forward 256x128 tiles; data gradient 256x256)
    (64, 56, 56, 128, 128, 3, 2, 1),      # WRN-50-2 spec A: stride-2 3x3 with 128 channels (parity classes on 256x128 tiles, strided epilogue)
]

# WRN-28-10's 3x3 stride-1 layers at batch 128 on the row-patch 256 x 160 kernel (conv_igemm8r.hip) by the shipped rule, run with every operand set the
# block forward / backward launches them with (tests/test_gpu_production_tiles.py::test_igemm8r_production_operand_sets)
R8_GEOMS = [
    (128, 32, 32, 160, 160, 3, 1, 1),     # stage 1: 512 tiles, 8 groups of 3 K tiles
    (128, 16, 16, 320, 320, 3, 1, 1),     # stage 2: 256 tiles, 15 groups
]

# the ImageNet stems on the eight-phase kernel's tap-chunk mode (bias + statistics, no residual): tests/test_gpu_production_tiles.py::test_igemm8_stem
STEM8_GEOMS = [(8, 224, 224, 0, 512, 7, 2, 3), (8, 224, 224, 0, 256, 7, 2, 3)]

# the full-batch configurations whose every convolution must select a tile that PROD_GEOMS (or the small geometries of
# test_gpu_kernels.py) also selects
CONFIGS = {
    'rn20':      dict(spec='c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10', preact=False, use_proj=False, hw=32, batch=128),
    'wrn-28-10': dict(spec='c3,160,3,1,1 r4 r4 r4 n a ap8,1,0 fc640,10', preact=True, use_proj=True, hw=32, batch=128),
    'v2-164':    dict(spec='c3,64,3,1,1 b18 b18 b18 n a ap8,1,0 fc256,100', preact=True, use_proj=True, hw=32, batch=128),
    'wrn-50-2a': dict(spec='c3,256,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc2048,1000', preact=False, use_proj=True, hw=224, batch=256),
    'wrn-50-2b': dict(spec='c3,512,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc4096,1000', preact=False, use_proj=True, hw=224, batch=256),
}


def resolve(g, fp32):
    """C == 0 marks a stem: the 3 image channels zero-padded to one 16-byte chunk (4 fp32 / 8 16-bit elements)."""
    g = list(g)
    if g[3] == 0:
        g[3] = 4 if fp32 else 8
    return tuple(g)


def geom(N, H, W, C, K, k, stride, pad):
    P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return dict(N=N, H=H, W=W, C=C, P=P, Q=Q, K=K, R=k, S=k, stride=stride, pad=pad)
