"""
TEST INFRASTRUCTURE ONLY -- per-sample CPU restatement of the reference's input pipeline.

Follows /root/reference/resnet/utils/transform_util.py, one function per transform's ``forward`` and calling the same torch
operators it calls: ToTensorTransform :36-47, ZeroMeanWhiteningTransform.forward :70-73 / fit :58-68,
StandardizeWhiteningTransform.forward :106-109 / fit :85-104, FlipTransform.forward :161-166, PaddingTransform.forward :182-187,
RandomCropTransform.forward :201-206; the pipeline order is the ``data_aug_train`` / ``data_aug_test`` mapping of
models_dir/*/config.yaml:6-14 (applied left to right by tv.transforms.Compose, data_util.py:74).

PARITY PIN: ``resnet.utils.transform_util`` cannot be imported here (it imports torchvision and PIL at module level; torchvision is
not installed and there is no network -- SURVEY.md 8c), and the reference holds no fixtures for its transforms.  ToTensor is
torchvision's (third party, version unpinned by the reference's requirements): its published algorithm for a uint8 HWC image is
``img.permute(2, 0, 1).contiguous().to(float32).div(255)``.  Everything else below is a torch built-in called exactly as the
reference calls it, so the restatement is pinned operator by operator but NOT by running the reference's module.

The random draws (FlipTransform's Categorical sample, RandomCropTransform's two randint draws) are arguments here: the product
draws them on the device from its own generator, so streams differ from the reference's by construction; parity is on the
transform given the draws, and on the draws' distributions.
"""
import torch


def to_tensor(img_u8_hwc):
    x = torch.as_tensor(img_u8_hwc)
    assert x.dtype == torch.uint8 and x.dim() == 3
    return x.permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def zero_mean_whiten(x, image_mean):
    return x - image_mean


def standardize_whiten(x, image_mean, image_stddev):
    return (x - image_mean) / image_stddev


def flip(x, do_flip):
    return torch.flip(x, dims=(2,)) if do_flip else x


def padding(x, pad_size, pad_type):
    assert pad_type in ('zero', 'mirror')
    pad = (pad_size,) * 4
    if pad_type == 'mirror':
        return torch.nn.functional.pad(x, pad=pad, mode='reflect')
    return torch.nn.functional.pad(x, pad=pad, mode='constant', value=0.)


def crop(x, t_idx, l_idx, crop_size):
    return x[:, t_idx:t_idx + crop_size, l_idx:l_idx + crop_size]


def fit_mean(images_u8_nhwc):
    """the streaming mean of ZeroMean/StandardizeWhiteningTransform.fit, fp32, sample by sample"""
    mean = torch.zeros(to_tensor(images_u8_nhwc[0]).shape, dtype=torch.float32)
    for k, img in enumerate(images_u8_nhwc, 1):
        mean *= (k - 1) / k
        mean += to_tensor(img) / k
    return mean


def fit_stddev(images_u8_nhwc, mean):
    var = torch.zeros_like(mean)
    for k, img in enumerate(images_u8_nhwc, 1):
        var *= (k - 1) / k
        var += torch.square(to_tensor(img) - mean) / k
    return torch.sqrt(var)


def pipeline(img_u8_hwc, data_aug, image_mean=None, image_stddev=None, do_flip=False, t_idx=0, l_idx=0):
    """one sample through an ordered ``data_aug`` mapping (config.yaml:6-14) -> float32 [C, h, w]"""
    x = img_u8_hwc
    for name, kw in data_aug.items():
        if name == 'ToTensorTransform':
            x = to_tensor(x)
        elif name == 'ZeroMeanWhiteningTransform':
            x = zero_mean_whiten(x, image_mean)
        elif name == 'StandardizeWhiteningTransform':
            x = standardize_whiten(x, image_mean, image_stddev)
        elif name == 'FlipTransform':
            x = flip(x, do_flip)
        elif name == 'PaddingTransform':
            x = padding(x, kw['pad_size'], kw['pad_type'])
        elif name == 'RandomCropTransform':
            x = crop(x, t_idx, l_idx, kw['crop_size'])
        else:
            raise NotImplementedError(name)
    return x
