"""
utils/data_util.py on the CPU: the CIFAR archive reader (torchvision's on-disk layout, read without torchvision), the reference's sampler
(data_util.py:155-190), batching with a short last batch (DataLoader default drop_last=False, :218-222) and the fitted-transform checkpoint
protocol (:76-92: load the newest ``{kind}_{step}.pth``, else fit and save at step 1).  The device transform itself is covered by
tests/test_gpu_transforms.py.
"""
import os
import pickle

import numpy as np
import pytest
import torch

from pytorch_ddp_resnet_amd.utils import data_util
from pytorch_ddp_resnet_amd.utils.transform_util import BatchTransform
from test_transforms import WRN_AUG, TEST_AUG


def fake_cifar10(root, n_per_batch=40, n_test=24, seed=0):
    """the layout torchvision leaves under ``root`` for CIFAR10 (python version of the archive)"""
    rng = np.random.default_rng(seed)
    base = os.path.join(root, 'cifar-10-batches-py')
    os.makedirs(base, exist_ok=True)
    out = {}
    for name, n in [(f'data_batch_{i}', n_per_batch) for i in range(1, 6)] + [('test_batch', n_test)]:
        data = rng.integers(0, 256, (n, 3072), dtype=np.uint8)
        labels = rng.integers(0, 10, n).tolist()
        with open(os.path.join(base, name), 'wb') as f:
            pickle.dump({'data': data, 'labels': labels, 'batch_label': name, 'filenames': [f'{name}_{i}.png' for i in range(n)]}, f)
        out[name] = (data, labels)
    return out


def test_cifar10_archive_reader(tmp_path):
    raw = fake_cifar10(str(tmp_path))
    x, y = data_util.load_arrays('CIFAR10', str(tmp_path), True)
    assert x.shape == (200, 32, 32, 3) and x.dtype == np.uint8 and y.shape == (200,) and y.dtype == np.int64
    d3, l3 = raw['data_batch_3']
    # stored planar (R plane, G plane, B plane, row-major): image 7 of the third file is sample 87
    assert np.array_equal(x[87], d3[7].reshape(3, 32, 32).transpose(1, 2, 0)) and y[87] == l3[7]
    xt, yt = data_util.load_arrays('CIFAR10', str(tmp_path), False)
    assert xt.shape == (24, 32, 32, 3) and np.array_equal(yt, np.asarray(raw['test_batch'][1]))
    with pytest.raises(FileNotFoundError, match='nothing is downloaded'):
        data_util.load_arrays('CIFAR100', str(tmp_path), True)
    with pytest.raises(NotImplementedError):
        data_util.load_arrays('ImageNet', str(tmp_path), True)


def test_cifar100_keys(tmp_path):
    base = tmp_path / 'cifar-100-python'
    base.mkdir()
    rng = np.random.default_rng(1)
    for name, n in (('train', 30), ('test', 10)):
        with open(base / name, 'wb') as f:
            pickle.dump({'data': rng.integers(0, 256, (n, 3072), dtype=np.uint8), 'fine_labels': rng.integers(0, 100, n).tolist(),
                         'coarse_labels': rng.integers(0, 20, n).tolist()}, f)
    x, y = data_util.load_arrays('CIFAR100', str(tmp_path), True)
    assert x.shape == (30, 32, 32, 3) and int(y.max()) < 100


def test_sampler_is_the_references(tmp_path):
    """DistributedSampler(seed=0, shuffle=True, drop_last=False): randperm(seed + epoch), padded by wrap-around, strided by rank"""
    ds = list(range(203))
    for world in (1, 2, 4):
        for rank in range(world):
            s = data_util.get_samplers(rank, world, ds, ds)['sampler_train']
            for epoch in (0, 3):
                s.set_epoch(epoch)
                g = torch.Generator().manual_seed(0 + epoch)
                perm = torch.randperm(203, generator=g).tolist()
                total = -(-203 // world) * world
                perm += perm[:total - 203]
                assert list(s) == perm[rank:total:world]


def test_loader_batches_the_sampler_order_with_a_short_last_batch():
    imgs = np.arange(50, dtype=np.uint8).reshape(50, 1, 1, 1).repeat(3, axis=3)
    labels = np.arange(50) % 10
    ds = data_util.DeviceDataset(imgs, labels, lambda x, generator=None: x, 'cpu')           # stub transform: the gathered uint8 batch itself
    samplers = data_util.get_samplers(1, 2, ds, ds)
    dl = data_util.get_dataloaders(ds, ds, **samplers, batch_size=32, world_size=2, num_microbatches=2, rank=1)['dl_train']
    assert dl.batch_size == 8 and len(dl) == 4                                               # 25 indices per rank: 8 + 8 + 8 + 1
    samplers['sampler_train'].set_epoch(2)
    want = list(samplers['sampler_train'])
    got_x, got_y, sizes = [], [], []
    for x, y in dl:
        sizes.append(len(y))
        got_x += x[:, 0, 0, 0].tolist()
        got_y += y.tolist()
    assert sizes == [8, 8, 8, 1] and got_x == want and got_y == [i % 10 for i in want]


def test_fitted_transform_checkpoint_protocol(tmp_path, monkeypatch):
    fake_cifar10(str(tmp_path / 'data'))
    ck = str(tmp_path / 'ckpt')
    d1 = data_util.get_datasets('CIFAR10', str(tmp_path / 'data'), WRN_AUG, TEST_AUG, ck, 'cpu')
    assert os.path.exists(os.path.join(ck, 'standardizewhiteningtransform_1.pth'))           # fitted once, saved at step 1 (data_util.py:87-92)
    sd = torch.load(os.path.join(ck, 'standardizewhiteningtransform_1.pth'))
    assert list(sd) == ['_image_mean', '_image_stddev', '_fitted'] and bool(sd['_fitted'])
    tr, te = d1['dataset_train'].transform, d1['dataset_test'].transform
    assert torch.equal(te._image_mean, tr._image_mean) and torch.equal(te._image_stddev, tr._image_stddev) and bool(te._fitted)
    x = d1['dataset_train'].images.double() / 255
    np.testing.assert_allclose(tr._image_mean.numpy(), x.mean(0).permute(2, 0, 1).numpy(), atol=1e-6)
    # a second run loads the file instead of fitting again
    monkeypatch.setattr(BatchTransform, 'fit', lambda self, *a, **k: (_ for _ in ()).throw(AssertionError('refitted')))
    d2 = data_util.get_datasets('CIFAR10', str(tmp_path / 'data'), WRN_AUG, TEST_AUG, ck, 'cpu')
    assert torch.equal(d2['dataset_train'].transform._image_stddev, tr._image_stddev)
    with pytest.raises(ValueError, match='Fittable test transform'):
        data_util.get_datasets('CIFAR10', str(tmp_path / 'data'), WRN_AUG, {'ToTensorTransform': {}, 'ZeroMeanWhiteningTransform': {}}, ck, 'cpu')


def _two_rank_fit_worker(rank, world, port, data_dir, ck, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fits = []
        orig = BatchTransform.fit
        BatchTransform.fit = lambda self, *a, **k: (fits.append(1), orig(self, *a, **k))[1]
        d = data_util.get_datasets('CIFAR10', data_dir, WRN_AUG, TEST_AUG, ck, 'cpu')
        tr = d['dataset_train'].transform
        torch.save(dict(mean=tr._image_mean.clone(), std=tr._image_stddev.clone(), fits=len(fits)), os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_two_ranks_fit_once_and_agree(tmp_path):
    """world_size 2 (gloo): rank 0 fits and saves the whitening statistics (atomically), rank 1 loads the finished file behind the barrier --
    one fit, one file, identical statistics (every rank fitting and saving the same file at once is the reference's race, data_util.py:76-92)."""
    import socket
    import torch.multiprocessing as mp
    fake_cifar10(str(tmp_path / 'data'))
    ck, out = str(tmp_path / 'ckpt'), str(tmp_path / 'out')
    os.makedirs(out)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    mp.spawn(_two_rank_fit_worker, args=(2, port, str(tmp_path / 'data'), ck, out), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(out, 'rank0.pt')), torch.load(os.path.join(out, 'rank1.pt'))
    assert r0['fits'] == 1 and r1['fits'] == 0
    assert torch.equal(r0['mean'], r1['mean']) and torch.equal(r0['std'], r1['std'])
    assert sorted(os.listdir(ck)) == ['standardizewhiteningtransform_1.pth']            # no temporary file left behind
