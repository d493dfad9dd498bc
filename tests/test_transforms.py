"""
Input pipeline, CPU side: the oracle's operator-by-operator restatement (oracle/transforms_ref.py) against
an independent index-map restatement of what the HIP kernel computes (crop -> padding -> flip -> whitening composed backwards),
bit for bit; the host class's handling of the reference's ``data_aug`` mappings; the fitted-statistics checkpoint format.
"""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import transforms_ref as ref
from pytorch_ddp_resnet_amd.utils.transform_util import BatchTransform
from pytorch_ddp_resnet_amd.utils import checkpoint_util as ck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WRN_AUG = {'ToTensorTransform': {}, 'StandardizeWhiteningTransform': {}, 'FlipTransform': {'p': 0.5},
           'PaddingTransform': {'pad_size': 4, 'pad_type': 'mirror'}, 'RandomCropTransform': {'crop_size': 32}}
RN20_AUG = {'ToTensorTransform': {}, 'ZeroMeanWhiteningTransform': {}, 'FlipTransform': {'p': 0.5},
            'PaddingTransform': {'pad_size': 4, 'pad_type': 'zero'}, 'RandomCropTransform': {'crop_size': 32}}
TEST_AUG = {'ToTensorTransform': {}, 'StandardizeWhiteningTransform': {}}


def index_map_pipeline(img, mean, std, flip, top, left, pad, mirror, crop):
    """what rn_augment_batch computes for one sample, in numpy: float32 [C, crop, crop]"""
    H, W, C = img.shape
    out = np.zeros((C, crop, crop), np.float32)
    for oi in range(crop):
        for oj in range(crop):
            hi, wj = top + oi - pad, left + oj - pad
            if mirror:
                hi = -hi if hi < 0 else (2 * (H - 1) - hi if hi >= H else hi)
                wj = -wj if wj < 0 else (2 * (W - 1) - wj if wj >= W else wj)
            elif not (0 <= hi < H and 0 <= wj < W):
                continue
            ws = W - 1 - wj if flip else wj
            v = img[hi, ws].astype(np.float32) / np.float32(255)
            v = v - mean[:, hi, ws]
            if std is not None:
                v = v / std[:, hi, ws]
            out[:, oi, oj] = v
    return out


def make_stats(rng, H, W, C):
    imgs = rng.integers(0, 256, (64, H, W, C), dtype=np.uint8)
    mean = ref.fit_mean(imgs)
    std = ref.fit_stddev(imgs, mean)
    return imgs, mean, std


@pytest.mark.parametrize('aug', [WRN_AUG, RN20_AUG, TEST_AUG], ids=['wrn', 'rn20', 'test'])
def test_oracle_equals_index_map(aug):
    rng = np.random.default_rng(0)
    H = W = 12
    aug = {k: dict(v) for k, v in aug.items()}
    if 'RandomCropTransform' in aug:
        aug['RandomCropTransform']['crop_size'] = 12
    imgs, mean, std = make_stats(rng, H, W, 3)
    pad = aug.get('PaddingTransform', {}).get('pad_size', 0)
    mirror = aug.get('PaddingTransform', {}).get('pad_type') == 'mirror'
    use_std = 'StandardizeWhiteningTransform' in aug
    crop = aug.get('RandomCropTransform', {}).get('crop_size', H)
    for n in range(12):
        f, t, l = bool(rng.integers(2)), int(rng.integers(2 * pad + 1)), int(rng.integers(2 * pad + 1))
        f = f and 'FlipTransform' in aug
        if n == 0:
            t, l = 0, 0
        if n == 1:
            t, l = 2 * pad, 2 * pad
        want = ref.pipeline(imgs[n], aug, mean, std, f, t, l).numpy()
        got = index_map_pipeline(imgs[n], mean.numpy(), std.numpy() if use_std else None, f, t, l, pad, mirror, crop)
        assert want.shape == got.shape
        assert np.array_equal(want, got), (n, f, t, l)


def test_shipped_configs_parse():
    for name, white, ptype in [('wrn-28-10-dropout_cifar10', 'StandardizeWhiteningTransform', 'mirror'),
                               ('resnet-v1-20_cifar10', 'ZeroMeanWhiteningTransform', 'zero')]:
        cfg = yaml.safe_load(open(os.path.join(ROOT, 'models_dir', name, 'config.yaml')))
        tr = BatchTransform([32, 32, 3], cfg['data_aug_train'], device='cpu')
        assert (tr.whitening, tr.p, tr.pad_size, tr.pad_type, tr.crop_size) == (white, 0.5, 4, ptype, 32)
        assert (tr.t_max, tr.l_max) == (8, 8) and tr.output_shape == [3, 32, 32]
        te = BatchTransform([32, 32, 3], cfg['data_aug_test'], device='cpu')
        assert (te.p, te.pad_size, te.crop_size, te.t_max) == (0.0, 0, 32, 0) and te.whitening == white


def test_unsupported_and_misordered_pipelines_raise():
    with pytest.raises(NotImplementedError):
        BatchTransform([32, 32, 3], {'ToTensorTransform': {}, 'ZCAWhiteningTransform': {}}, device='cpu')
    with pytest.raises(NotImplementedError):
        BatchTransform([32, 32, 3], {'ToTensorTransform': {}, 'PaddingTransform': {'pad_size': 4, 'pad_type': 'zero'}, 'FlipTransform': {'p': .5}},
                       device='cpu')
    with pytest.raises(NotImplementedError):
        BatchTransform([32, 32, 3], {'FlipTransform': {'p': .5}}, device='cpu')
    with pytest.raises(AssertionError):   # reference: assert pad_type in ['zero', 'mirror'] (transform_util.py:171)
        BatchTransform([32, 32, 3], {'ToTensorTransform': {}, 'PaddingTransform': {'pad_size': 4, 'pad_type': 'edge'}}, device='cpu')


def test_no_cpu_fallback():
    tr = BatchTransform([8, 8, 3], {'ToTensorTransform': {}}, device='cpu')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        tr(torch.zeros(2, 8, 8, 3, dtype=torch.uint8))


def test_unfitted_whitening_asserts():
    tr = BatchTransform([8, 8, 3], TEST_AUG, device='cpu')
    with pytest.raises(AssertionError):   # reference: assert self._fitted (transform_util.py:71, 107)
        tr(torch.zeros(2, 8, 8, 3, dtype=torch.uint8))


def test_fit_agrees_with_reference_recurrence_and_checkpoints_in_its_format(tmp_path):
    rng = np.random.default_rng(1)
    imgs, mean, std = make_stats(rng, 8, 8, 3)
    tr = BatchTransform([8, 8, 3], TEST_AUG, device='cpu')
    tr.fit(imgs, chunk=24)
    assert bool(tr._fitted)
    np.testing.assert_allclose(tr._image_mean.numpy(), mean.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(tr._image_stddev.numpy(), std.numpy(), rtol=2e-5, atol=0)
    # the reference saves the fitted transform's state_dict under kind = class name lower-cased (data_util.py:79-92)
    sd = tr.fitted_state_dict()
    assert tr.kind_name == 'standardizewhiteningtransform' and list(sd) == ['_image_mean', '_image_stddev', '_fitted']
    assert tuple(sd['_image_mean'].shape) == (3, 8, 8)
    torch.save(sd, tmp_path / ck._format_name(tr.kind_name, 0))
    tr2 = BatchTransform([8, 8, 3], TEST_AUG, device='cpu')
    tr2.load_fitted_state_dict(torch.load(tmp_path / 'standardizewhiteningtransform_0.pth'))
    assert bool(tr2._fitted) and torch.equal(tr2._image_mean, tr._image_mean) and torch.equal(tr2._image_stddev, tr._image_stddev)
    zm = BatchTransform([8, 8, 3], {'ToTensorTransform': {}, 'ZeroMeanWhiteningTransform': {}}, device='cpu')
    assert list(zm.fitted_state_dict()) == ['_image_mean', '_fitted']
    with pytest.raises(RuntimeError):
        zm.load_fitted_state_dict(sd)
