"""Optimizer / scheduler factories by class name (mirrors /root/reference/resnet/utils/optim_util.py:11-30:
``getattr(torch.optim, name)(model.parameters(), **args)``; scheduler name 'None' -> no scheduler)."""
import importlib


def get_optimizer(optimizer_cls_name, model, optimizer_args):
    cls = getattr(importlib.import_module('torch.optim'), optimizer_cls_name)
    return cls(model.parameters(), **optimizer_args)


def get_scheduler(scheduler_cls_name, optimizer, scheduler_args):
    if scheduler_cls_name == 'None':
        return None
    cls = getattr(importlib.import_module('torch.optim.lr_scheduler'), scheduler_cls_name)
    return cls(optimizer, **scheduler_args)
