"""
Loss and metrics of one microbatch.  Mirrors /root/reference/resnet/algos/metrics.py:10-41: mean cross-entropy,
top-1 / top-5 error, and the world-averaged logging values -- with the three scalar all-reduces + three ``.item()``
host syncs per microbatch (metrics.py:32-36) collapsed into ONE 3-float all-reduce and one host read.
"""
from collections import Counter

import torch
import torch.distributed as dist


def cross_entropy_loss(logits, labels):
    return torch.nn.functional.cross_entropy(logits, labels)


def top_k_err(logits, labels, k):
    topk = torch.topk(logits, k=k, dim=-1).indices
    return 1.0 - torch.eq(topk, labels.unsqueeze(-1)).float().sum(dim=-1).mean(dim=0)


def compute_losses_and_metrics(logits, labels):
    """metrics.py:21-29.  Logits that come straight out of a HIP-engine ``ResNet.forward`` take the fused path (one launch
    forward, one backward: architectures/resnet.py:_LossFn); anything else (CPU tensors, detached logits, eval under
    no_grad) goes through the reference's own torch ops below -- same values (tests/test_gpu_model.py)."""
    from ..architectures.resnet import fused_loss_and_metrics
    fused = fused_loss_and_metrics(logits, labels)
    if fused is not None:
        return {"loss": fused[0], "top1_err": fused[1], "top5_err": fused[2]}
    return {"loss": cross_entropy_loss(logits, labels), "top1_err": top_k_err(logits, labels, 1), "top5_err": top_k_err(logits, labels, 5)}


def global_means(metrics, world_size):
    """for logging only (as the reference): mean over ranks of each metric -> Counter of floats."""
    keys = list(metrics)
    packed = torch.stack([metrics[k].detach().float() for k in keys])
    if world_size > 1 and dist.is_initialized():
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    vals = (packed / world_size).tolist()
    return Counter(dict(zip(keys, vals)))


class PendingMeans:
    """world-averaged logging values of one microbatch whose device->host copy is still in flight.  ``result()`` blocks only
    until THAT copy has landed, so a training loop that resolves step i-1 after it has enqueued step i never idles the GPU
    (the reference reads three scalars with ``.item()`` right after every backward: metrics.py:32-36)."""

    def __init__(self, keys, host, event):
        self.keys, self._host, self._event = keys, host, event

    def result(self):
        if self._event is not None:
            self._event.synchronize()
        vals = self._host.tolist()
        # the fused loss turns a label outside [0, classes) into a NaN loss (torch's cross_entropy device-asserts there); under the fp16 GradScaler a NaN is
        # just a skipped step and a halved scale, so a corrupted label stream would degrade training silently -- make it loud where the value reaches the host
        if 'loss' in self.keys and vals[self.keys.index('loss')] != vals[self.keys.index('loss')]:
            raise FloatingPointError('loss is NaN: a label outside [0, num_classes) (the fused loss marks those rows NaN) or a diverged forward')
        return Counter(dict(zip(self.keys, vals)))


def global_means_async(metrics, world_size):
    """as global_means, but returns a PendingMeans: one 3-float all-reduce, then an asynchronous copy into pinned memory."""
    keys = list(metrics)
    packed = torch.stack([metrics[k].detach().float() for k in keys])
    if world_size > 1 and dist.is_initialized():
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    packed = packed / world_size
    if not packed.is_cuda:
        return PendingMeans(keys, packed, None)
    host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
    host.copy_(packed, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(packed.device))
    return PendingMeans(keys, host, ev)
