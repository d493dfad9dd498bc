"""
Datasets, samplers and loaders of the reference (``resnet/utils/data_util.py``) with the whole dataset RESIDENT on the device.

The reference builds a torchvision dataset, fits / loads the whitening transform, wraps it in ``DistributedSampler(seed=0, shuffle=True,
drop_last=False)`` (:155-190) and a ``DataLoader(batch_size = batch_size // (num_microbatches * world_size), num_workers=0)`` (:193-232)
that runs the transforms per sample on one host thread.  Here:

* ``load_arrays`` reads the dataset files torchvision downloads (the dataset authors' "python version" archives: ``cifar-10-batches-py/
  data_batch_1..5 | test_batch``, ``cifar-100-python/train | test``; pickled dicts, ``data`` uint8 [N, 3072] stored CHW, ``labels`` /
  ``fine_labels``) WITHOUT torchvision, as uint8 [N, 32, 32, 3] -- the array ``torchvision.datasets.CIFAR10.data`` holds (the reference reads
  its ``data[0].shape`` as the initial data shape, :41-45).  Nothing is downloaded (no network): a missing archive is an error that says so.
* the arrays live in HBM (CIFAR-10: 150 MB of uint8); a batch is an index gather + ONE ``rn_augment_batch`` launch (``transform_util.
  BatchTransform``); the sampler IS ``torch.utils.data.DistributedSampler`` with the reference's arguments, so every rank visits the
  same indices in the same order as the reference run (``set_epoch`` is called by the training loop, training.py:88).
* the fitted whitening statistics are checkpointed under the reference's kind and step (``{transform class}.lower()_1.pth``, :79-92),
  and a reference-written file is loaded instead of re-fitting.

ImageNet (variable-size JPEG files: decode + resize) is not covered: ``get_datasets`` raises NotImplementedError for it.
"""
import math
import os
import pickle
from typing import Dict

import numpy as np
import torch

from .checkpoint_util import maybe_load_checkpoint, save_checkpoint
from .transform_util import BatchTransform

_CIFAR = {
    'CIFAR10': dict(base='cifar-10-batches-py', train=[f'data_batch_{i}' for i in range(1, 6)], test=['test_batch'], key='labels'),
    'CIFAR100': dict(base='cifar-100-python', train=['train'], test=['test'], key='fine_labels'),
}


def load_arrays(dataset_cls_name: str, root: str, train: bool):
    """-> (uint8 [N, 32, 32, 3], int64 [N]) from the archives torchvision keeps under ``root``"""
    if dataset_cls_name not in _CIFAR:
        raise NotImplementedError(f"{dataset_cls_name}: only the CIFAR archives are read without torchvision")
    spec = _CIFAR[dataset_cls_name]
    data, labels = [], []
    for name in spec['train' if train else 'test']:
        path = os.path.join(root, spec['base'], name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: place the extracted '{spec['base']}' archive under {root} (nothing is downloaded here)")
        with open(path, 'rb') as f:
            entry = pickle.load(f, encoding='latin1')
        data.append(np.asarray(entry['data'], dtype=np.uint8))
        labels.extend(entry[spec['key']])
    x = np.vstack(data).reshape(-1, 3, 32, 32).transpose((0, 2, 3, 1))          # stored CHW -> HWC, as torchvision does
    return np.ascontiguousarray(x), np.asarray(labels, dtype=np.int64)


class DeviceDataset:
    """uint8 NHWC images + int64 labels on ``device`` and the batch transform that turns a gathered uint8 batch into the classifier's input"""

    def __init__(self, images_u8, labels, transform, device):
        self.images = torch.as_tensor(images_u8).to(device).contiguous()
        self.labels = torch.as_tensor(labels, dtype=torch.int64).to(device)
        self.transform = transform
        self.device = torch.device(device)
        assert self.images.dtype == torch.uint8 and self.images.dim() == 4 and self.images.shape[0] == self.labels.shape[0]

    def __len__(self):
        return int(self.images.shape[0])


class _Fitted:
    """the fitted whitening statistics of a BatchTransform seen as the reference's checkpointable transform"""

    def __init__(self, tr):
        self.tr = tr

    def state_dict(self):
        return self.tr.fitted_state_dict()

    def load_state_dict(self, sd):
        self.tr.load_fitted_state_dict(sd)


def get_datasets(dataset_cls_name: str, data_dir: str, data_aug_train: Dict[str, dict], data_aug_test: Dict[str, dict], checkpoint_dir: str, device,
                 **kwargs) -> Dict[str, DeviceDataset]:
    xtr, ytr = load_arrays(dataset_cls_name, data_dir, True)
    xte, yte = load_arrays(dataset_cls_name, data_dir, False)
    shape = list(xtr.shape[1:])
    t_train = BatchTransform(shape, data_aug_train, device=device)
    ds_train = DeviceDataset(xtr, ytr, t_train, device)
    if t_train.whitening is not None:
        # the reference lets every rank load-or-fit-and-save the same file at once (data_util.py:76-92; its slow host-side fit hides the race).  The
        # fit here takes milliseconds, so rank 0 goes first -- load, else fit and save (atomically) -- and the other ranks load behind a barrier.
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        first = not multi or dist.get_rank() == 0
        if multi and not first:
            dist.barrier()
        step = maybe_load_checkpoint(checkpoint_dir, t_train.kind_name, _Fitted(t_train), 'cpu', None)
        if step == 0:
            t_train.fit(ds_train.images)
            if first:
                save_checkpoint(checkpoint_dir, t_train.kind_name, _Fitted(t_train), 1)
        if multi and first:
            dist.barrier()
    t_test = BatchTransform(shape, data_aug_test, device=device)
    if t_test.whitening is not None:
        if t_test.whitening != t_train.whitening:
            raise ValueError("Fittable test transform not in reusable_transforms.")        # data_util.py:94-98
        t_test.load_fitted_state_dict(t_train.fitted_state_dict())
    return dict(dataset_train=ds_train, dataset_test=DeviceDataset(xte, yte, t_test, device))


def get_samplers(rank: int, world_size: int, dataset_train, dataset_test, **kwargs):
    mk = lambda ds: torch.utils.data.DistributedSampler(dataset=range(len(ds)), num_replicas=world_size, rank=rank, shuffle=True, seed=0, drop_last=False)
    return dict(sampler_train=mk(dataset_train), sampler_test=mk(dataset_test))


class DeviceLoader:
    """iterates ``(x, y)`` batches of one epoch: the sampler's indices in order, ``batch_size`` at a time (the last batch may be short, as
    with DataLoader's default drop_last=False); x = dataset.transform(gathered uint8 batch)"""

    def __init__(self, dataset: DeviceDataset, sampler, batch_size: int, seed: int = 0):
        self.dataset, self.sampler, self.batch_size = dataset, sampler, int(batch_size)
        self.generator = None
        if dataset.device.type == 'cuda':
            self.generator = torch.Generator(device=dataset.device)
            self.generator.manual_seed(seed)

    def __len__(self):
        return math.ceil(len(self.sampler) / self.batch_size)

    def __iter__(self):
        idx = torch.tensor(list(self.sampler), dtype=torch.int64).to(self.dataset.device)
        for i in range(0, idx.numel(), self.batch_size):
            b = idx[i:i + self.batch_size]
            x = self.dataset.transform(self.dataset.images.index_select(0, b), generator=self.generator)
            yield x, self.dataset.labels.index_select(0, b)


def get_dataloaders(dataset_train, dataset_test, sampler_train, sampler_test, batch_size: int, world_size: int, num_microbatches: int, rank: int = 0,
                    **kwargs):
    local_batch_size = batch_size // (num_microbatches * world_size)
    seed = torch.initial_seed() + rank              # the random draws of a rank: its own stream (the reference draws from each process' global RNG)
    return dict(dl_train=DeviceLoader(dataset_train, sampler_train, local_batch_size, seed),
                dl_test=DeviceLoader(dataset_test, sampler_test, local_batch_size, seed + 1))
