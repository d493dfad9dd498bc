"""Throughput of the reference's whole step body (training.py:92-113: forward, loss + top-k metrics, backward, optimizer
step, zero_grad, world-mean of the logging values) through this package's train_step -- next to bench.py's fwd+bwd number.
usage: python tools/train_rate.py [workload] [steps]"""
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS
from pytorch_ddp_resnet_amd import ResNet
from pytorch_ddp_resnet_amd.algos.training import train_step
from pytorch_ddp_resnet_amd.utils.optim_util import get_optimizer

name = sys.argv[1] if len(sys.argv) > 1 else 'wrn-28-10'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = WORKLOADS[name]
torch.manual_seed(0)
m = ResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], cfg['p'], compute_dtype='bf16').cuda().train()
opt = get_optimizer('SGD', m, dict(lr=0.1, momentum=0.9, dampening=0.0, nesterov=True, weight_decay=5e-4))
x = torch.randn(cfg['batch'], 3, cfg['hw'], cfg['hw'], device='cuda')
y = torch.randint(0, cfg['classes'], (cfg['batch'],), device='cuda')
for _ in range(5):
    train_step(m, x, y, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
lazy = os.environ.get('LAZY', '1') == '1'
prev = None
for _ in range(steps):
    cur = train_step(m, x, y, opt, lazy=lazy)
    if lazy and prev is not None:
        out = prev.result()                      # the previous step's values, read after this step was enqueued
    prev = cur
out = prev.result() if lazy else prev
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f'{name}: full step body {dt * 1e3:.3f} ms/step = {cfg["batch"] / dt:.0f} images/s  (optimizer {type(opt).__name__}, last loss {out["loss"]:.4f})')
