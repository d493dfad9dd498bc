"""
Checkpoint files in the reference's format, so that a run of either code base resumes from the other's files.

Format (from /root/reference/resnet/utils/checkpoint_util.py): one file per checkpointable, ``{kind}_{steps}.pth`` (:16-18),
holding ``torch.save(obj.state_dict())`` (:74-85); on resume the newest step is picked per kind (:21-41), every kind must
agree on it or ``RuntimeError("Checkpoint steps not aligned.")`` (:112-114); only the newest five files of a kind are kept
(:44-49, 85).  The reference's classifier is wrapped in DistributedDataParallel (script.py:64), so its keys carry a
``module.`` prefix (``module._architecture.1.0._conv1.weight``); the product's classifier is the bare ``ResNet`` (the gradient
reduction lives in ddp.GradReducer), so ``ddp_keys(classifier)`` adds / strips that prefix at the file boundary.  Optimizer
files are ``torch.optim.SGD.state_dict()``; ``utils.fused_sgd.FusedSGD`` loads and writes the same structure (parameter
order == ``model.parameters()`` order == the reference's).  Tensors are read by shape: the product keeps convolution weights
channels_last in memory, which ``load_state_dict``'s ``copy_`` and ``state_dict()``'s logical [K,C,R,S] view make invisible.
"""
import os
import re
from typing import Any, Dict, Optional

import torch


def _format_name(kind: str, steps: int, suffix: str = 'pth') -> str:
    return f"{kind}_{steps}.{suffix}"


def _parse_name(filename: str) -> Optional[dict]:
    m = re.fullmatch(r"(\w+)_([0-9]+)\.([a-z]+)", filename)
    return dict(kind=m.group(1), steps=int(m.group(2)), suffix=m.group(3)) if m else None


def _steps_of(base_path: str, kind: str):
    out = set()
    for f in os.listdir(base_path):
        p = _parse_name(f)
        if p is not None and p['kind'] == kind:
            out.add(p['steps'])
    return sorted(out)


def latest_step(base_path: str, kind: str) -> Optional[int]:
    s = _steps_of(base_path, kind)
    return s[-1] if s else None


class ddp_keys:
    """views a module through DistributedDataParallel's key scheme: ``state_dict()`` keys gain ``module.``, ``load_state_dict``
    accepts them with or without it."""

    def __init__(self, module, prefix: str = 'module.'):
        self.module, self.prefix = module, prefix

    def state_dict(self):
        return {self.prefix + k: v for k, v in self.module.state_dict().items()}

    def load_state_dict(self, sd):
        if sd and all(k.startswith(self.prefix) for k in sd):
            sd = {k[len(self.prefix):]: v for k, v in sd.items()}
        return self.module.load_state_dict(sd)


def maybe_load_checkpoint(checkpoint_dir: str, kind_name: str, checkpointable, map_location, steps: Optional[int]) -> int:
    os.makedirs(checkpoint_dir, exist_ok=True)
    steps_ = latest_step(checkpoint_dir, kind_name) if steps is None else steps
    path = os.path.join(checkpoint_dir, _format_name(kind_name, steps_)) if steps_ is not None else None
    if path is None or not os.path.exists(path):
        print(f"Bad {kind_name} checkpoint or none at {checkpoint_dir} with step {steps}.")
        print("Running from scratch.")
        return 0
    checkpointable.load_state_dict(torch.load(path, map_location=map_location))
    print(f"Loaded {kind_name} checkpoint from {checkpoint_dir}, with step {steps_}.")
    return steps_


def save_checkpoint(checkpoint_dir: str, kind_name: str, checkpointable, steps: int, keep: int = 5) -> None:
    os.makedirs(checkpoint_dir, exist_ok=True)
    # written under a name the resume scan does not parse, then renamed: a reader never sees a half-written file (torch.save is not atomic)
    final = os.path.join(checkpoint_dir, _format_name(kind_name, steps))
    tmp = final + '.tmp%d' % os.getpid()
    torch.save(checkpointable.state_dict(), tmp)
    os.replace(tmp, final)
    for old in _steps_of(checkpoint_dir, kind_name)[:-keep]:
        os.remove(os.path.join(checkpoint_dir, _format_name(kind_name, old)))


def maybe_load_checkpoints(checkpoint_dir: str, checkpointables: Dict[str, object], map_location, steps: Optional[int] = None) -> int:
    """-> the step of the loaded checkpoints (0: none found, running from scratch).  Every non-None checkpointable must be at the
    same step."""
    found = [maybe_load_checkpoint(checkpoint_dir, k, c, map_location, steps) for k, c in checkpointables.items() if c is not None]
    if len(set(found)) > 1:
        raise RuntimeError("Checkpoint steps not aligned.")
    return found[0] if found else 0


def save_checkpoints(checkpoint_dir: str, checkpointables: Dict[str, object], steps: int) -> None:
    for k, c in checkpointables.items():
        if c is not None:
            save_checkpoint(checkpoint_dir, k, c, steps)


# ---- checkpoint strategies (/root/reference/resnet/utils/checkpoint_util.py:140-222): WHEN rank 0 writes.  Modules with `_batch_step` / `_epoch_step`
# (and `_lowest_loss`) buffers, so their state_dict is the reference's `checkpoint_strategy_{steps}.pth` ----
class CheckpointStrategy(torch.nn.Module):
    def __init__(self, unit: str):
        assert unit in ('batch', 'epoch')
        super().__init__()
        self._unit = unit
        self.register_buffer('_batch_step', torch.tensor(0))
        self.register_buffer('_epoch_step', torch.tensor(0))

    @property
    def unit(self) -> str:
        return self._unit

    @property
    def batch_step(self) -> int:
        return int(self._batch_step.item())

    @property
    def epoch_step(self) -> int:
        return int(self._epoch_step.item())

    def step(self, unit: str) -> None:
        assert unit in ('batch', 'epoch')
        buf = self._batch_step if unit == 'batch' else self._epoch_step
        buf.add_(1)

    def observe(self, **kwargs) -> bool:
        raise NotImplementedError

    def may_save(self, unit: str) -> bool:
        """can the NEXT observation of `unit` ask for a checkpoint?  (Not in the reference: its loop observes synchronously.  The training loop here reads
        its logging values one microbatch late, so it asks first and closes the step on the host -- before the next optimizer step is enqueued -- whenever
        the answer is yes: the file written for step k then holds the state after exactly k steps, training.py:129-139.)"""
        return self.unit == unit


class FrequencyCheckpointStrategy(CheckpointStrategy):
    """eligible every `frequency`-th observation of its own unit (counted from 0, so the first one is)."""

    def __init__(self, unit: str, frequency: int, **kwargs):
        super().__init__(unit)
        self._frequency = int(frequency)

    def observe(self, unit: str, **kwargs) -> bool:
        cond = getattr(self, f"{unit}_step") % self._frequency == 0
        self.step(unit)
        return bool(cond) if self.unit == unit else False

    def may_save(self, unit: str) -> bool:
        return self.unit == unit and getattr(self, f"{unit}_step") % self._frequency == 0


class PerformanceCheckpointStrategy(CheckpointStrategy):
    """eligible when the observed loss of its own unit is the lowest so far."""

    def __init__(self, unit: str, **kwargs):
        super().__init__(unit)
        self.register_buffer('_lowest_loss', torch.tensor(float('inf')))

    @property
    def lowest_loss(self) -> float:
        return float(self._lowest_loss.item())

    def observe(self, unit: str, loss: float, **kwargs) -> bool:
        cond = loss < self.lowest_loss
        self.step(unit)
        if self.unit != unit:
            return False
        if cond:
            self._lowest_loss.fill_(float(loss))
        return bool(cond)


def get_checkpoint_strategy(checkpoint_strategy_cls_name: str, checkpoint_strategy_args: Optional[Dict[str, Any]]) -> CheckpointStrategy:
    cls = {'FrequencyCheckpointStrategy': FrequencyCheckpointStrategy, 'PerformanceCheckpointStrategy': PerformanceCheckpointStrategy}.get(checkpoint_strategy_cls_name)
    if cls is None:
        raise AttributeError(f"module 'checkpoint_util' has no attribute '{checkpoint_strategy_cls_name}'")
    return cls(**(checkpoint_strategy_args or {}))
