"""
``ResNet(architecture_spec, preact, use_proj, dropout_prob)`` -- the drop-in boundary of the hot path.

Mirrors /root/reference/resnet/architectures/resnet.py: constructor (:25-32), spec grammar (:122-158, see spec.py),
top-level module order / indices and therefore every ``state_dict`` key (``_architecture.{i}...``), weight init
(:160-163: Kaiming-normal on top-level convs only, torch defaults elsewhere), ``forward(x)`` taking NCHW fp32 and
returning fp32 logits that support autograd (:165-166), ``train()/eval()`` semantics of BatchNorm and Dropout.

What differs is how forward/backward run: the spec is lowered once per input shape to a plan of fused HIP kernels
(engine/lowering.py) executed by librn_hip.so; a single autograd.Function hands the parameter gradients back to
PyTorch so optimizers, DistributedDataParallel hooks and checkpoints keep working unchanged.  There is no CPU or
eager fallback: on a non-GPU device forward raises.
"""
from typing import Dict, Tuple

import torch
from torch import nn

from .residual_block import ResidualBlock, BottleneckResidualBlock, _conv_holder
from .spec import parse_spec

_DTYPES = {'fp32': torch.float32, 'float32': torch.float32, 'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16,
           'fp16': torch.float16, 'float16': torch.float16, 'half': torch.float16}


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, seed, need_grad, *params):
        eng = model._engine(x.shape, model.training, need_grad)
        eng.bind(model._named_tensors(), x=x)
        eng.forward(step_seed=seed, hook_fn=model._hook_fn)
        ctx.eng, ctx.gen, ctx.seed, ctx.model = eng, eng.generation, seed, model
        ctx.keys = model._param_keys
        return eng.t('logits').clone()

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.eng
        if eng.generation != ctx.gen:
            raise RuntimeError("the engine's activation buffers were overwritten by a later forward of the same shape/mode; "
                               "call backward before the next forward")
        if 'dlogits' not in eng.plan.slot_of:
            raise RuntimeError("forward ran without gradient support")
        if dlogits.data_ptr() != eng.t('dlogits').data_ptr():      # the fused loss (_LossFn) has already written it in place
            eng.t('dlogits').copy_(dlogits)
        eng.backward(step_seed=ctx.seed, hook_fn=ctx.model._hook_fn)
        # autograd's AccumulateGrad keeps (steals) the tensors returned here as p.grad.  Handing out views of the engine's
        # flat buffer is zero-copy but means the NEXT backward overwrites p.grad in place -- only safe when the caller
        # consumes the gradients every step (ddp.GradReducer / bench.py set alias_grads).  Default: one flat copy.
        if ctx.model.alias_grads and any(p.grad is not None for p in ctx.model.parameters()):
            raise RuntimeError("alias_grads: parameter .grad must be None before backward (optimizer.zero_grad(set_to_none=True)); "
                               "an existing .grad would alias the buffer the engine has just overwritten")
        flat = eng.flat_grad if ctx.model.alias_grads else eng.flat_grad.clone()
        return (None, None, None, None) + tuple(eng.grad_view(k, flat) for k in ctx.keys)


class _LossFn(torch.autograd.Function):
    """mean cross-entropy + top-1 / top-5 error (reference metrics.py:10-29) on the engine's logits as ONE launch
    (rn_softmax_ce) instead of ATen's log_softmax / nll_loss / topk / eq / mean chains; the backward is one more launch that
    writes dL/dlogits straight into the engine's buffer, multiplied by the upstream gradient of the loss ON THE DEVICE (a
    GradScaler's loss scale under ``scaler.scale(loss).backward()``, training.py:100), so no copy and no host sync."""

    @staticmethod
    def forward(ctx, logits, labels, eng, gen):
        import ctypes as C
        from .. import _lib
        N, O = logits.shape
        out3 = eng.loss_scratch(0)
        st = C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
        _lib.check(eng.L.rn_softmax_ce(C.c_void_p(eng.t('logits').data_ptr()), C.c_void_p(labels.data_ptr()), C.c_void_p(out3.data_ptr()), None,
                                      N, O, 1.0 / N, None, st))
        vals = out3[:3] * (1.0 / N)
        ctx.eng, ctx.gen, ctx.labels, ctx.shape = eng, gen, labels, (N, O)
        loss, t1, t5 = vals[0], vals[1], vals[2]
        ctx.mark_non_differentiable(t1, t5)
        return loss, t1, t5

    @staticmethod
    def backward(ctx, gloss, _g1, _g5):
        import ctypes as C
        from .. import _lib
        eng = ctx.eng
        if eng.generation != ctx.gen:
            raise RuntimeError("the engine's logits were overwritten by a later forward; call backward before the next forward")
        N, O = ctx.shape
        gloss = gloss.detach().float().contiguous()
        st = C.c_void_p(torch.cuda.current_stream(gloss.device).cuda_stream)
        _lib.check(eng.L.rn_softmax_ce(C.c_void_p(eng.t('logits').data_ptr()), C.c_void_p(ctx.labels.data_ptr()), C.c_void_p(eng.loss_scratch(1).data_ptr()),
                                      C.c_void_p(eng.t('dlogits').data_ptr()), N, O, 1.0 / N, C.c_void_p(gloss.data_ptr()), st))
        return eng.t('dlogits'), None, None, None


def fused_loss_and_metrics(logits, labels):
    """-> (loss, top1_err, top5_err) through the engine's fused kernel when ``logits`` is the fresh output of a HIP-engine
    ResNet.forward with gradients enabled; None when it is not (the caller then uses the reference's torch ops)."""
    src = getattr(logits, '_rn_src', None)
    if src is None:
        return None
    eng, gen = src
    if eng.generation != gen or 'dlogits' not in eng.plan.slot_of or not logits.requires_grad:
        return None
    if labels.dtype != torch.int64 or labels.device != logits.device or not labels.is_contiguous() or logits.shape[1] < 5:
        return None
    # the kernel reads labels[n] for every logits row, and the logits from the engine's own buffer (``logits`` is a copy of it): anything else goes
    # to the torch ops (which raise on a shape mismatch); a label outside [0, O) gives a NaN loss in the kernel (torch device-asserts), never an
    # out-of-range read
    if labels.dim() != 1 or logits.dim() != 2 or labels.shape[0] != logits.shape[0] or logits._version != getattr(logits, '_rn_version', -1):
        return None                                # (an in-place edit of the returned logits bumps the version: the engine's copy is then stale)
    return _LossFn.apply(logits, labels, eng, gen)


class ResNet(nn.Module):
    def __init__(self, architecture_spec: str, preact: bool, use_proj: bool, dropout_prob: float,
                 compute_dtype: str = 'fp16', sync_bn: bool = False):
        """:param compute_dtype: 'fp16' (default: the reference's own GPU arithmetic -- fp16 storage and MFMA operands under a
        GradScaler, script.py:63 / training.py:95-110 -- with fp32 accumulation, statistics and master weights; meets the 1e-3 logit
        bound), 'bf16' (no loss scaling needed, 8 significant bits: logits ~4e-3) or 'fp32' (exact-f32 MFMA: the parity mode).
        :param sync_bn: all-reduce BatchNorm statistics over the default process group."""
        super().__init__()
        self._architecture_spec = architecture_spec
        self._preact, self._use_proj, self._dropout_prob = preact, use_proj, dropout_prob
        self._compute_dtype = _DTYPES[compute_dtype]
        self._sync_bn = sync_bn
        self._architecture = self._parse_spec(architecture_spec)
        self._init_weights()
        self._engines: Dict[Tuple, object] = {}
        self._param_keys = [k for k, _ in self.named_parameters()]
        self._step = 0
        self._seed_base = int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFF
        self._hook_fn = None
        self.alias_grads = False        # True: p.grad aliases the engine's flat gradient buffer (no copy; see _EngineFn.backward)

    # ---- construction (module order == reference order) ---------------------------------------------------------
    def _parse_spec(self, spec: str) -> nn.Sequential:
        ms = []
        for comp in parse_spec(spec):
            if comp.kind == 'conv':
                k, s, p = comp.args
                ms.append(_conv_holder(comp.cin, comp.cout, k, s, p, bias=True))
            elif comp.kind == 'maxpool':
                ms.append(nn.MaxPool2d(comp.args[0], comp.args[1], comp.args[2]))
            elif comp.kind == 'avgpool':
                ms.append(nn.AvgPool2d(comp.args[0], comp.args[1], comp.args[2]))
            elif comp.kind in ('basic', 'bottleneck'):
                cls = ResidualBlock if comp.kind == 'basic' else BottleneckResidualBlock
                ms.append(nn.Sequential(*[
                    cls(channels=comp.cin if b == 0 else comp.cout, downsample=comp.down and b == 0, preact=self._preact,
                        use_proj=self._use_proj, dropout_prob=self._dropout_prob) for b in range(comp.depth)]))
            elif comp.kind == 'norm':
                ms.append(nn.BatchNorm2d(comp.cin))
            elif comp.kind == 'act':
                ms.append(nn.ReLU())
            else:
                ms.append(nn.Sequential(nn.Flatten(), nn.Linear(comp.cin, comp.cout)))
        return nn.Sequential(*ms)

    def _init_weights(self):
        for m in self._architecture:
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    # ---- engine plumbing ---------------------------------------------------------------------------------------------
    def _named_tensors(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d

    def _engine(self, xshape, train: bool, need_grad: bool):
        from ..engine.lowering import lower
        from ..engine.executor import Engine
        dev = next(self.parameters()).device
        world = torch.distributed.get_world_size() if (self._sync_bn and torch.distributed.is_initialized()) else 1
        key = (tuple(xshape), train, need_grad, str(dev), self._compute_dtype, world)
        eng = self._engines.get(key)
        if eng is None:
            N, C, H, W = xshape
            plan = lower(self._architecture_spec, self._preact, self._use_proj, self._dropout_prob, N, H, W, train=train,
                         need_grad=need_grad, sync_bn=self._sync_bn, world_size=world, fp32=self._compute_dtype == torch.float32)
            eng = Engine(plan, dev, self._compute_dtype)
            self._engines[key] = eng
        return eng

    def _apply(self, fn, *a, **kw):
        self._engines = {}          # device / dtype moves invalidate the bound buffers
        return super()._apply(fn, *a, **kw)

    def forward(self, x):
        p0 = next(self.parameters())
        if p0.device.type != 'cuda' or x.device != p0.device:
            raise RuntimeError("ResNet.forward runs only on an MI355X ('cuda') device through librn_hip.so; "
                               "there is no CPU / eager fallback (model on %s, input on %s)" % (p0.device, x.device))
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        self._step += 1
        seed = (self._seed_base + self._step) & 0x7FFFFFFFFFFFFFFF
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)   # (grad mode is off inside Function.forward)
        out = _EngineFn.apply(self, x, seed, need_grad, *params)
        eng = self._engine(x.shape, self.training, need_grad)
        out._rn_src = (eng, eng.generation)        # lets algos.metrics.compute_losses_and_metrics take the fused loss path
        out._rn_version = out._version
        return out
