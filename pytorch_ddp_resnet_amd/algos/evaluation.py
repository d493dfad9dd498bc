"""Evaluation loop: mirrors /root/reference/resnet/algos/evaluation.py:14-42 (no_grad, ``classifier.eval()`` -- BN uses
running statistics, dropout off --, metrics averaged over batches, then over ranks)."""
from collections import Counter

import torch

from .metrics import compute_losses_and_metrics, global_means


@torch.no_grad()
def evaluation_loop(world_size, device, dl_test, classifier, **kwargs):
    classifier.eval()
    summed, n = Counter(), 0
    for x, y in dl_test:
        x, y = x.to(device), y.to(device)
        summed += Counter({k: v for k, v in compute_losses_and_metrics(logits=classifier(x), labels=y).items()})
        n += 1
    metrics = {k: v / n for k, v in summed.items()}
    return global_means(metrics, world_size)
