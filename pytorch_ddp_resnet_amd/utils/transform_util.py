"""
The reference's input pipeline for a whole batch on the device.

The reference runs its ``data_aug_train`` chain per sample, in Python, on one host thread (data_util.py:218-222 builds a
``DataLoader(num_workers=0)`` over ``tv.transforms.Compose(transforms)``, data_util.py:74): ToTensor -> whitening -> flip ->
padding -> random crop (transform_util.py:36-205; order from config.yaml:6-14).  ``BatchTransform`` takes the same ordered
``data_aug`` mapping and runs the chain as ONE HIP launch (``rn_augment_batch``, csrc/misc.hip) over a uint8 NHWC batch that is
already resident in HBM (CIFAR-10 as uint8 is 150 MB: the whole training set fits in HBM 1,900 times over), writing either the
reference loader's fp32 NCHW batch or the engine's NHWC stem input.

Random draws: one flip bit per sample with probability p (FlipTransform :161-164) and two uniform integers in
[0, H + 2 pad - crop] (RandomCropTransform :201-205), drawn on the device from a ``torch.Generator``; the streams differ from the
reference's host draws by construction (as any two seeds do).

Fitted statistics (``_image_mean`` / ``_image_stddev`` images of shape [C,H,W]) are checkpointed by the reference under the kind
``{transform class name}.lower()`` (data_util.py:79-92); ``state_dict`` / ``load_state_dict`` here use the same keys, so a
reference run's ``standardizewhiteningtransform_0.pth`` loads.  ``fit`` accumulates in float64 on the device, which agrees with the
reference's sample-by-sample float32 recurrence (:58-68, :85-104) to float32 rounding, not bit for bit; load the reference's file
when its exact statistics matter.

ZCAWhitening / RandomScale / Color transforms are not used by any shipped config (and the latter two fail in the reference,
SURVEY.md Q17): they raise NotImplementedError here.
"""
from collections import OrderedDict
from typing import Dict, Optional

import torch

from .. import _lib

_ORDER = ['ToTensorTransform', 'whitening', 'FlipTransform', 'PaddingTransform', 'RandomCropTransform']
_WHITENING = ('ZeroMeanWhiteningTransform', 'StandardizeWhiteningTransform')
_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class BatchTransform(torch.nn.Module):
    def __init__(self, data_shape, data_aug: Dict[str, dict], device='cuda'):
        """data_shape: [H, W, C] of the stored uint8 images (the reference's ``data_shape`` before ToTensor)."""
        super().__init__()
        h, w, c = data_shape
        self.data_shape = (h, w, c)
        stage = -1
        self.whitening, self.p, self.pad_size, self.pad_type, self.crop_size = None, 0.0, 0, 'zero', None
        for name, kw in data_aug.items():
            kw = kw or {}
            slot = 'whitening' if name in _WHITENING else name
            if slot not in _ORDER:
                raise NotImplementedError(f"{name}: not part of the device pipeline (shipped configs use {_ORDER})")
            if _ORDER.index(slot) <= stage:
                raise NotImplementedError(f"{name}: the device pipeline runs the shipped order {_ORDER}")
            stage = _ORDER.index(slot)
            if slot == 'whitening':
                self.whitening = name
            elif name == 'FlipTransform':
                self.p = float(kw['p'])
            elif name == 'PaddingTransform':
                assert kw['pad_type'] in ['zero', 'mirror']
                self.pad_size, self.pad_type = int(kw['pad_size']), kw['pad_type']
            elif name == 'RandomCropTransform':
                self.crop_size = int(kw['crop_size'])
        if 'ToTensorTransform' not in data_aug:
            raise NotImplementedError("the device pipeline starts from uint8 images: ToTensorTransform must come first")
        hp, wp = h + 2 * self.pad_size, w + 2 * self.pad_size
        if self.crop_size is None:
            if hp != wp:
                raise NotImplementedError("without RandomCropTransform the padded image must be square")
            self.crop_size = hp
        # RandomCropTransform :201-203 draws in [0, padded - crop]
        self.t_max, self.l_max = hp - self.crop_size, wp - self.crop_size
        assert self.t_max >= 0 and self.l_max >= 0, "crop larger than the padded image"
        self._image_mean = torch.nn.Parameter(torch.zeros(c, h, w, dtype=torch.float32, device=device), requires_grad=False)
        self._image_stddev = torch.nn.Parameter(torch.ones(c, h, w, dtype=torch.float32, device=device), requires_grad=False)
        self.register_buffer('_fitted', torch.tensor(self.whitening is None))

    @property
    def output_shape(self):
        return [self.data_shape[2], self.crop_size, self.crop_size]

    # ---- fitted statistics ---------------------------------------------------------------------------------------------------
    @property
    def kind_name(self) -> Optional[str]:
        return self.whitening.lower() if self.whitening else None

    def fitted_state_dict(self):
        """the reference's checkpoint of the fitted whitening transform (kind ``self.kind_name``)"""
        sd = OrderedDict(_image_mean=self._image_mean.detach().clone())
        if self.whitening == 'StandardizeWhiteningTransform':
            sd['_image_stddev'] = self._image_stddev.detach().clone()
        sd['_fitted'] = self._fitted.clone()
        return sd

    def load_fitted_state_dict(self, sd):
        want = set(self.fitted_state_dict())
        if set(sd) != want:
            raise RuntimeError(f"fitted transform keys {sorted(sd)} != {sorted(want)}")
        with torch.no_grad():
            self._image_mean.copy_(sd['_image_mean'])
            if '_image_stddev' in sd:
                self._image_stddev.copy_(sd['_image_stddev'])
            self._fitted.copy_(sd['_fitted'])

    @torch.no_grad()
    def fit(self, images_u8, chunk: int = 8192):
        """images_u8: uint8 [N, H, W, C] (host or device)."""
        if self.whitening is None:
            return
        n = images_u8.shape[0]
        dev = self._image_mean.device
        s1 = torch.zeros(self.data_shape, dtype=torch.float64, device=dev)
        s2 = torch.zeros_like(s1)
        for i in range(0, n, chunk):
            x = torch.as_tensor(images_u8[i:i + chunk]).to(dev).to(torch.float64) / 255.
            s1 += x.sum(0)
            s2 += (x * x).sum(0)
        mean = s1 / n
        var = (s2 / n - mean * mean).clamp_(min=0)
        self._image_mean.copy_(mean.permute(2, 0, 1).to(torch.float32))
        if self.whitening == 'StandardizeWhiteningTransform':
            self._image_stddev.copy_(var.sqrt().permute(2, 0, 1).to(torch.float32))
        self._fitted.fill_(True)

    # ---- the batch ----------------------------------------------------------------------------------------------------------
    def draw(self, n: int, generator: Optional[torch.Generator] = None):
        dev = self._image_mean.device
        flip = (torch.rand(n, device=dev, generator=generator) < self.p).to(torch.uint8)
        top = torch.randint(0, self.t_max + 1, (n,), device=dev, generator=generator, dtype=torch.int32)
        left = torch.randint(0, self.l_max + 1, (n,), device=dev, generator=generator, dtype=torch.int32)
        return flip, top, left

    def forward(self, x_u8, flip=None, top=None, left=None, generator=None, nhwc_dtype=None, nhwc_channels=None):
        """x_u8: uint8 [N, H, W, C] on the device.  -> fp32 [N, C, crop, crop] (the reference loader's batch), or, with
        ``nhwc_dtype``, [N, crop, crop, nhwc_channels] in that dtype (the engine's stem input, channels zero-padded)."""
        assert bool(self._fitted), "whitening transform not fitted"
        if not x_u8.is_cuda:
            raise RuntimeError("BatchTransform runs on the device: move the uint8 batch to the GPU first (no CPU fallback)")
        h, w, c = self.data_shape
        if x_u8.dtype != torch.uint8 or tuple(x_u8.shape[1:]) != (h, w, c) or not x_u8.is_contiguous():
            raise ValueError(f"expected a contiguous uint8 [N,{h},{w},{c}] batch, got {x_u8.dtype} {tuple(x_u8.shape)}")
        n = x_u8.shape[0]
        if flip is None:
            flip, top, left = self.draw(n, generator)          # in range by construction: no host read-back on the step's path
        else:
            flip = flip.to(x_u8.device, torch.uint8).contiguous()
            top = top.to(x_u8.device, torch.int32).contiguous()
            left = left.to(x_u8.device, torch.int32).contiguous()
            assert flip.numel() == n and top.numel() == n and left.numel() == n
            assert int(top.min()) >= 0 and int(top.max()) <= self.t_max and int(left.min()) >= 0 and int(left.max()) <= self.l_max
        s = self.crop_size
        std = self._image_stddev.data_ptr() if self.whitening == 'StandardizeWhiteningTransform' else None
        stream = torch.cuda.current_stream().cuda_stream
        if nhwc_dtype is None:
            out = torch.empty(n, c, s, s, dtype=torch.float32, device=x_u8.device)
            args = (out.data_ptr(), None, 0, n, h, w, c, self.pad_size, int(self.pad_type == 'mirror'), s, c)
        else:
            cp = nhwc_channels or c
            out = torch.empty(n, s, s, cp, dtype=nhwc_dtype, device=x_u8.device)
            args = (None, out.data_ptr(), _DTYPES[nhwc_dtype], n, h, w, c, self.pad_size, int(self.pad_type == 'mirror'), s, cp)
        _lib.check(_lib.lib().rn_augment_batch(x_u8.data_ptr(), self._image_mean.data_ptr(), std, flip.data_ptr(), top.data_ptr(),
                                               left.data_ptr(), *args, stream))
        return out
