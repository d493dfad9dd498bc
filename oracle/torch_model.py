"""
TEST INFRASTRUCTURE ONLY -- torch-CPU restatement of the reference's training step, NCHW fp32.

The reference delegates all arithmetic to stock ATen ops (Conv2d / BatchNorm2d / ReLU / Dropout /
AvgPool2d / MaxPool2d / Linear / CrossEntropyLoss).  This file re-states the *composition* of those ops
(resnet.py:122-166, residual_block.py:67-99,173-215, metrics.py:10-29, training.py:92-113) with
``torch.nn.functional`` calls on a plain parameter dict keyed by the reference's ``state_dict`` names, so that

  * model-level parity cases too large for the numpy restatement (ResNet-v1-20 at batch 128) have a checker
    that runs on the GPU box (where /root/reference does not exist), and
  * ``bench.py`` has the "reference CPU training loop" to time on the host cores (``cpu_baseline.kind = "port"``).

It is pinned against the reference itself by tests/test_oracle_golden.py (fixtures G3/G4 in tests/golden/).
Never imported by the product package.
"""

import torch
import torch.nn.functional as F

from .np_model import parse_spec, block_layout, param_shapes


def init_state(spec, preact, use_proj, seed=0):
    """Reference initialisation (resnet.py:160-163 + torch module defaults): Kaiming-normal on the top-level
    conv only; block convs / Linear keep torch's default kaiming_uniform(a=sqrt(5)) i.e. U(-1/sqrt(fan_in), +);
    BN gamma=1, beta=0, running stats (0, 1).  The random stream is this file's own (portable), not torch's
    module-construction order: parity tests load identical weights on both sides, they never re-draw."""
    g = torch.Generator().manual_seed(seed)
    shapes = dict(param_shapes(spec, preact, use_proj))
    st = {}
    for key, shape in shapes.items():
        prefix, leaf = key.rsplit('.', 1)
        is_bn = (prefix + '.running_mean') in shapes
        if leaf == 'num_batches_tracked':
            st[key] = torch.zeros((), dtype=torch.int64)
        elif is_bn:
            st[key] = torch.ones(shape) if leaf in ('weight', 'running_var') else torch.zeros(shape)
        else:
            wshape = shapes[prefix + '.weight']
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            if leaf == 'weight' and len(shape) == 4 and key.count('.') == 2:      # top-level conv
                st[key] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
            else:
                st[key] = (torch.rand(shape, generator=g) * 2 - 1) / fan_in ** 0.5
    return st


def is_param(key):
    leaf = key.rsplit('.', 1)[1]
    return leaf in ('weight', 'bias')


class TorchResNet:
    """Functional model over a state dict (tensors are used in place: BN running stats are updated like the
    reference's modules do)."""

    def __init__(self, spec, preact, use_proj, dropout_prob=0.0):
        self.spec, self.preact, self.use_proj, self.p = spec, preact, use_proj, dropout_prob
        self.comps = parse_spec(spec)

    def _bn(self, st, pre, x, train):
        if train:
            st[pre + '.num_batches_tracked'] += 1
        return F.batch_norm(x, st[pre + '.running_mean'], st[pre + '.running_var'], st[pre + '.weight'],
                            st[pre + '.bias'], training=train, momentum=0.1, eps=1e-5)

    def _block(self, st, bp, kind, cin, down, x, train):
        convs, norms, cout = block_layout(kind, cin, down, self.preact)
        i = x
        n = len(convs)
        for j, (ci, co, k, s, p) in enumerate(convs, 1):
            w = st[f'{bp}._conv{j}.weight']
            if self.preact:
                x = F.relu(self._bn(st, f'{bp}._norm{j}', x, train))
                x = F.dropout(x, self.p, training=train)
                x = F.conv2d(x, w, None, s, p)
            else:
                x = F.dropout(x, self.p, training=train)
                x = F.conv2d(x, w, None, s, p)
                x = self._bn(st, f'{bp}._norm{j}', x, train)
                if j < n:
                    x = F.relu(x)
        if down:
            i = i[:, :, ::2, ::2]                      # AvgPool2d(k=1, s=2)
            if self.use_proj:
                i = F.conv2d(i, st[f'{bp}._proj.weight'])
            else:
                i = F.pad(i, (0, 0, 0, 0, 0, cin))
        h = i + x
        return h if self.preact else F.relu(h)

    def forward(self, st, x, train=True):
        for idx, comp in enumerate(self.comps):
            pre, kind = f'_architecture.{idx}', comp['kind']
            if kind == 'conv':
                x = F.conv2d(x, st[pre + '.weight'], st[pre + '.bias'], comp['stride'], comp['pad'])
            elif kind == 'norm':
                x = self._bn(st, pre, x, train)
            elif kind == 'act':
                x = F.relu(x)
            elif kind == 'maxpool':
                x = F.max_pool2d(x, comp['k'], comp['stride'], comp['pad'])
            elif kind == 'avgpool':
                x = F.avg_pool2d(x, comp['k'], comp['stride'], comp['pad'])
            elif kind == 'fc':
                x = F.linear(x.flatten(1), st[pre + '.1.weight'], st[pre + '.1.bias'])
            else:
                for b in range(comp['depth']):
                    cin = comp['cin'] if b == 0 else comp['cout']
                    x = self._block(st, f'{pre}.{b}', kind, cin, comp['down'] and b == 0, x, train)
        return x


def losses_and_metrics(logits, labels):
    """metrics.py:10-29"""
    loss = F.cross_entropy(logits, labels)

    def err(k):
        top = torch.topk(logits, k=k, dim=-1).indices
        return 1.0 - torch.eq(top, labels.unsqueeze(-1)).float().sum(dim=-1).mean(dim=0)
    return dict(loss=loss, top1_err=err(1), top5_err=err(min(5, logits.shape[1])))


def train_step(model, st, x, y, train=True):
    """One microbatch of training.py:92-103 without the optimizer: forward, loss/metrics, backward.
    Returns (logits, metrics, grads dict).  Parameters in ``st`` must have requires_grad=True."""
    params = {k: v for k, v in st.items() if is_param(k)}
    for p in params.values():
        p.grad = None
    logits = model.forward(st, x, train)
    metrics = losses_and_metrics(logits, y)
    metrics['loss'].backward()
    return logits.detach(), {k: v.detach() for k, v in metrics.items()}, {k: p.grad for k, p in params.items()}


def make_trainable(st):
    for k, v in st.items():
        if is_param(k):
            v.requires_grad_(True)
    return st
