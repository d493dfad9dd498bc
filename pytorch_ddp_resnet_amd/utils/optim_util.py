"""Optimizer / scheduler factories by class name (mirrors /root/reference/resnet/utils/optim_util.py:11-30:
``getattr(torch.optim, name)(model.parameters(), **args)``; scheduler name 'None' -> no scheduler).  'SGD' on a HIP-engine
model resolves to the fused flat-buffer implementation of the same rule (utils/fused_sgd.py); RN_TORCH_SGD=1 keeps torch's."""
import importlib
import os


def get_optimizer(optimizer_cls_name, model, optimizer_args):
    if optimizer_cls_name == 'SGD' and os.environ.get('RN_TORCH_SGD', '0') != '1':
        from ..architectures.resnet import ResNet
        p0 = next(model.parameters(), None)
        if isinstance(model, ResNet) and p0 is not None and p0.device.type == 'cuda':
            from .fused_sgd import FusedSGD
            return FusedSGD(model, **optimizer_args)
    cls = getattr(importlib.import_module('torch.optim'), optimizer_cls_name)
    return cls(model.parameters(), **optimizer_args)


def get_scheduler(scheduler_cls_name, optimizer, scheduler_args):
    if scheduler_cls_name == 'None':
        return None
    cls = getattr(importlib.import_module('torch.optim.lr_scheduler'), scheduler_cls_name)
    return cls(optimizer, **scheduler_args)
