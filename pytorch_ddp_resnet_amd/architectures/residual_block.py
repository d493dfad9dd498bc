"""
Parameter containers for the residual blocks.

Interface mirrored from /root/reference/resnet/architectures/residual_block.py: class names, constructor signature
``(channels, downsample, preact, use_proj, dropout_prob)`` (:8-16, :102-110), attribute names ``_conv{1,2,3}``,
``_proj``, ``_norm{1,2,3}`` and their registration order (so ``state_dict()`` / ``named_parameters()`` match key for
key; pinned by tests/golden/g5_grammar.npz), widths and strides (:26-61, :120-165; ``block_convs`` in spec.py).

Unlike the reference these modules do no arithmetic: the blocks are executed by the HIP engine from the plan that
``engine/lowering.py`` derives from the spec.  Convolution weights are stored channels_last, i.e. KRSC in HBM, which
is the operand layout of the implicit-GEMM kernels; their logical shape stays the reference's [K, C, R, S].
"""
import torch
from torch import nn

from .spec import block_convs


def _conv_holder(cin, cout, k, stride, pad, bias=False):
    m = nn.Conv2d(cin, cout, (k, k), (stride, stride), (pad, pad), bias=bias)     # torch default init, as the reference
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


class _Block(nn.Module):
    KIND = None

    def __init__(self, channels: int, downsample: bool, preact: bool, use_proj: bool, dropout_prob: float):
        super().__init__()
        convs, norms, cout = block_convs(self.KIND, channels, downsample, preact)
        self._in_channels, self._out_channels = channels, cout
        self._downsample, self._preact, self._use_proj, self._dropout_prob = downsample, preact, use_proj, dropout_prob
        for j, (ci, co, k, s, p) in enumerate(convs, 1):
            setattr(self, f'_conv{j}', _conv_holder(ci, co, k, s, p))
        if downsample and use_proj:
            self._proj = _conv_holder(channels, cout, 1, 1, 0)
        for j, c in enumerate(norms, 1):
            setattr(self, f'_norm{j}', nn.BatchNorm2d(c))

    def forward(self, x):
        raise RuntimeError("blocks are executed by the HIP engine through ResNet.forward; they have no standalone forward")


class ResidualBlock(_Block):
    """3x3, 3x3 (residual_block.py:8-99)."""
    KIND = 'basic'


class BottleneckResidualBlock(_Block):
    """1x1, 3x3 (carries the stride), 1x1 (residual_block.py:102-215)."""
    KIND = 'bottleneck'
