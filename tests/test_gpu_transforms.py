"""
Input pipeline on the device: rn_augment_batch through BatchTransform against the per-sample oracle
(oracle/transforms_ref.py, the reference's own torch operators) -- BIT-EXACT: the chain is a gather plus one IEEE division and one
subtraction / division per element.
"""
import numpy as np
import pytest
import torch

from oracle import transforms_ref as ref
from pytorch_ddp_resnet_amd.utils.transform_util import BatchTransform
from test_transforms import WRN_AUG, RN20_AUG, TEST_AUG

pytestmark = pytest.mark.gpu


def fitted(aug, H=32, W=32, C=3, n=96, seed=0):
    rng = np.random.default_rng(seed)
    imgs = rng.integers(0, 256, (n, H, W, C), dtype=np.uint8)
    tr = BatchTransform([H, W, C], aug)
    tr.fit(imgs)
    return tr, imgs, rng


def oracle_batch(tr, aug, imgs, flip, top, left):
    mean, std = tr._image_mean.cpu(), tr._image_stddev.cpu()
    return torch.stack([ref.pipeline(imgs[i], aug, mean, std, bool(flip[i]), int(top[i]), int(left[i])) for i in range(len(imgs))])


@pytest.mark.parametrize('aug', [WRN_AUG, RN20_AUG, TEST_AUG], ids=['wrn', 'rn20', 'test'])
@pytest.mark.parametrize('n', [1, 96])
def test_batch_equals_oracle_bit_exact(aug, n):
    tr, imgs, rng = fitted(aug)
    imgs = imgs[:n]
    flip = (rng.integers(0, 2, n) * (tr.p > 0)).astype(np.uint8)
    top = rng.integers(0, tr.t_max + 1, n).astype(np.int32)
    left = rng.integers(0, tr.l_max + 1, n).astype(np.int32)
    if n > 4:   # the corners of the draw range
        top[:4], left[:4] = [0, 0, tr.t_max, tr.t_max], [0, tr.l_max, 0, tr.l_max]
    got = tr(torch.from_numpy(imgs).cuda(), torch.from_numpy(flip), torch.from_numpy(top), torch.from_numpy(left))
    want = oracle_batch(tr, aug, imgs, flip, top, left)
    assert got.dtype == torch.float32 and tuple(got.shape) == (n, 3, 32, 32)
    assert torch.equal(got.cpu(), want)


def test_non_square_odd_shapes_and_big_pad():
    aug = {'ToTensorTransform': {}, 'StandardizeWhiteningTransform': {}, 'FlipTransform': {'p': 0.5},
           'PaddingTransform': {'pad_size': 6, 'pad_type': 'mirror'}, 'RandomCropTransform': {'crop_size': 9}}
    tr, imgs, rng = fitted(aug, H=7, W=11, C=5, n=33)
    assert (tr.t_max, tr.l_max) == (10, 14)
    flip = rng.integers(0, 2, 33).astype(np.uint8)
    top = rng.integers(0, tr.t_max + 1, 33).astype(np.int32)
    left = rng.integers(0, tr.l_max + 1, 33).astype(np.int32)
    got = tr(torch.from_numpy(imgs).cuda(), torch.from_numpy(flip), torch.from_numpy(top), torch.from_numpy(left))
    assert torch.equal(got.cpu(), oracle_batch(tr, aug, imgs, flip, top, left))


@pytest.mark.parametrize('dt', [torch.float32, torch.float16, torch.bfloat16])
def test_nhwc_output_is_the_cast_of_the_loader_batch(dt):
    tr, imgs, rng = fitted(WRN_AUG, n=32)
    x = torch.from_numpy(imgs).cuda()
    flip, top, left = tr.draw(32, torch.Generator(device='cuda').manual_seed(3))
    nchw = tr(x, flip, top, left)
    nhwc = tr(x, flip, top, left, nhwc_dtype=dt, nhwc_channels=8)
    assert tuple(nhwc.shape) == (32, 32, 32, 8) and nhwc.dtype == dt
    assert torch.equal(nhwc[..., :3], nchw.permute(0, 2, 3, 1).to(dt))
    assert not nhwc[..., 3:].any()


def test_draw_distributions():
    tr, _, _ = fitted(WRN_AUG, n=8)
    g = torch.Generator(device='cuda').manual_seed(0)
    flip, top, left = tr.draw(200_000, g)
    assert abs(float(flip.float().mean()) - 0.5) < 0.01                  # FlipTransform: Categorical([1-p, p])
    for v in (top, left):                                                  # RandomCropTransform: randint(0, max + 1)
        cnt = torch.bincount(v.long(), minlength=9).float() / v.numel()
        assert cnt.numel() == 9 and float((cnt - 1 / 9).abs().max()) < 0.005
    te, _, _ = fitted(TEST_AUG, n=8)
    flip, top, left = te.draw(1000, g)
    assert not flip.any() and not top.any() and not left.any()


def test_feeds_the_classifier():
    from pytorch_ddp_resnet_amd.architectures.resnet import ResNet
    tr, imgs, _ = fitted(RN20_AUG, n=16)
    x = tr(torch.from_numpy(imgs).cuda(), generator=torch.Generator(device='cuda').manual_seed(1))
    net = ResNet('c3,16,3,1,1 n a r1 ap32,1,0 fc16,10', False, True, 0.0).cuda()
    logits = net(x)
    assert tuple(logits.shape) == (16, 10) and torch.isfinite(logits).all()


def test_bad_arguments_fail_loudly():
    tr, imgs, _ = fitted(TEST_AUG, n=4)
    with pytest.raises(ValueError):
        tr(torch.zeros(4, 3, 32, 32, dtype=torch.uint8, device='cuda'))
    with pytest.raises(ValueError):
        tr(torch.zeros(4, 32, 32, 3, dtype=torch.float32, device='cuda'))
    trn, imgs, _ = fitted(WRN_AUG, n=4)
    z = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(AssertionError):   # a crop offset outside RandomCropTransform's range would read outside the padded image
        trn(torch.from_numpy(imgs).cuda(), z.to(torch.uint8), z + 9, z)
