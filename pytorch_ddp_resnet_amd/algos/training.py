"""
The training step around the hot path.  ``train_step`` is this build's counterpart of the reference's step body
(/root/reference/resnet/algos/training.py:92-113): forward -> loss/metrics -> backward (+ gradient mean across ranks)
-> optimizer step once ``num_microbatches`` microbatches have been accumulated (gradients are SUMMED over
microbatches, the loss is not rescaled -- training.py:92-113).  ``training_loop`` keeps the reference's
control flow (epochs over the loader, scheduler step unit, per-epoch evaluation, rank 0 saving
``{checkpoint_strategy, classifier, optimizer, scheduler, scaler}`` whenever the checkpoint strategy says so, at the
reference's two points, training.py:129-139 and :161-171) without its TensorBoard writer.
"""
from collections import Counter
from typing import Optional

import torch

from .metrics import compute_losses_and_metrics, global_means, global_means_async


def requires_loss(scheduler) -> bool:
    """training.py:20-21"""
    return isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau)


def step_scheduler(scheduler, loss) -> None:
    """training.py:24-28: a plateau scheduler is stepped with the loss, every other one without."""
    if requires_loss(scheduler):
        scheduler.step(loss)
    else:
        scheduler.step()


def train_step(classifier, x, y, optimizer=None, reducer=None, world_size=1, microbatch_id=1, num_microbatches=1, accum=None, lazy=False,
               scaler=None):
    """one microbatch.  returns the world-averaged metrics (Counter of floats; with ``lazy`` a metrics.PendingMeans whose
    host copy is still in flight).  ``reducer``: ddp.GradReducer or None; ``accum``: dict used to sum gradients across
    microbatches when num_microbatches > 1; ``scaler``: a torch GradScaler -- the reference's AMP branch (training.py:95-110:
    scaled backward, ``scaler.step`` / ``scaler.update`` at the batch end); the fp16 engine needs it, there is no autocast
    context because the engine's compute dtype is fixed at construction."""
    logits = classifier(x)
    metrics = compute_losses_and_metrics(logits=logits, labels=y)
    if scaler is not None:
        scaler.scale(metrics['loss']).backward()
    else:
        metrics['loss'].backward()
    if reducer is not None:
        reducer.finish()
    if num_microbatches > 1 and accum is not None:
        for k, p in classifier.named_parameters():
            if p.grad is None:
                continue
            if k in accum:
                accum[k] += p.grad
            else:
                accum[k] = p.grad.clone()
            p.grad = None
    out = global_means_async(metrics, world_size) if lazy else global_means(metrics, world_size)
    if optimizer is not None and microbatch_id % num_microbatches == 0:
        if num_microbatches > 1 and accum is not None:
            for k, p in classifier.named_parameters():
                p.grad = accum.pop(k, None)
        if scaler is not None:
            scaler.step(optimizer)
            scaler.update()
        else:
            optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return out


def training_loop(rank, world_size, device, dl_train, dl_test, classifier, optimizer, scheduler=None, scheduler_step_unit='none',
                  num_microbatches=1, global_step=0, max_steps=1, reducer=None, sampler_train=None, log=print, scaler=None,
                  checkpoint_strategy=None, checkpoint_dir=None, **kwargs):
    from .evaluation import evaluation_loop
    from ..utils.checkpoint_util import ddp_keys, save_checkpoints

    def maybe_save(unit, loss, steps):
        """rank 0 only, as the reference (its strategy counters advance on rank 0 alone); the loss is the step's / the epoch's mean"""
        if rank != 0 or checkpoint_strategy is None or checkpoint_dir is None:
            return
        if checkpoint_strategy.observe(unit=unit, loss=loss):
            save_checkpoints(checkpoint_dir, {'checkpoint_strategy': checkpoint_strategy, 'classifier': ddp_keys(classifier), 'optimizer': optimizer,
                                              'scheduler': scheduler, 'scaler': scaler}, steps=steps)

    # a resumed run continues the epoch count (training.py:87-88: sampler_train.set_epoch(checkpoint_strategy.epoch_step)); setup() loads the strategy on every
    # rank, so the start value agrees across ranks, and from here on every rank counts for itself (the reference advances the counter on rank 0 only)
    epoch = checkpoint_strategy.epoch_step if checkpoint_strategy is not None else 0
    saving = rank == 0 and checkpoint_strategy is not None and checkpoint_dir is not None
    while global_step < max_steps:
        if sampler_train is not None:
            sampler_train.set_epoch(epoch)
        classifier.train()
        acc, running = {}, Counter()
        # the logging values of a microbatch are read one microbatch LATE (after the next one has been enqueued), so the
        # device never waits for the host; the printed lines and their order are those of the reference
        pending = []                                       # [(PendingMeans, closes_a_step, global_step)]

        last_loss = [None]

        def resolve(upto):
            nonlocal running
            while len(pending) > upto:
                pm, closes, gs = pending.pop(0)
                running += pm.result()
                if closes:
                    means = {k: v / num_microbatches for k, v in running.items()}
                    last_loss[0] = means.get('loss')
                    if rank == 0:
                        log(f"global step: {gs}... loss: {means.get('loss')}")
                    maybe_save('batch', means.get('loss'), gs + 1)           # training.py:129-139 (in step order; see `saving` below for WHEN)
                    running = Counter()

        for microbatch_id, (x, y) in enumerate(dl_train, 1):
            x, y = x.to(device), y.to(device)
            pm = train_step(classifier, x, y, optimizer, reducer, world_size, microbatch_id, num_microbatches, acc, lazy=True, scaler=scaler)
            closes = microbatch_id % num_microbatches == 0
            pending.append((pm, closes, global_step))
            # A step whose observation may write a checkpoint is closed on the host NOW, before the next train_step enqueues another optimizer step: the
            # files of step k hold the weights, BatchNorm statistics, momentum and scaler state after exactly k steps, as the reference's synchronous loop
            # writes them (training.py:129-139).  Every other step keeps the lagged read (the device never waits for the host).
            resolve(1)                                     # everything but the microbatch just enqueued (the strategy's counters are current after this)
            if closes and saving and checkpoint_strategy.may_save('batch'):
                resolve(0)
            if closes:
                if scheduler is not None and scheduler_step_unit == 'batch':
                    if requires_loss(scheduler):
                        resolve(0)                         # a plateau scheduler needs THIS step's mean loss now (training.py:119-120)
                    step_scheduler(scheduler, last_loss[0])
                global_step += 1
                if global_step >= max_steps:
                    break
        resolve(0)
        if dl_test is not None:
            val = evaluation_loop(world_size, device, dl_test, classifier)
            if scheduler is not None and scheduler_step_unit == 'epoch':
                step_scheduler(scheduler, val.get('loss'))
            if rank == 0:
                log(f"epoch: {epoch}... validation loss: {val.get('loss')}")
            maybe_save('epoch', val.get('loss'), global_step + 1)           # training.py:161-171
        epoch += 1
    return global_step
