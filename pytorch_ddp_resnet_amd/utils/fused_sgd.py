"""
torch.optim.SGD's update rule as ONE HIP launch over a flat parameter buffer.

The reference builds its optimizer by class name (``getattr(torch.optim, 'SGD')(model.parameters(), **optimizer_args)``,
/root/reference/resnet/utils/optim_util.py:11-18, arguments from config.yaml:22-28: lr, momentum, dampening, nesterov,
weight_decay) and steps it once per batch (training.py:111-113).  For WRN-28-10 that is 80 parameter tensors, i.e. a few
hundred small kernels per step.  Here the parameters of a ``pytorch_ddp_resnet_amd.ResNet`` are re-homed, once, as views of one
flat fp32 buffer laid out exactly like the engine's flat gradient buffer (gradient-production order, 256-byte aligned
slices), and a step is one ``rn_sgd_step`` over (parameters, gradients, momentum) -- same hyper-parameters, same
``param_groups`` (so ``torch.optim.lr_scheduler`` works unchanged), same arithmetic as torch's single-tensor SGD.

Gradients are taken from the engine's flat buffer when ``p.grad`` aliases it (the normal case: the optimizer turns on
``model.alias_grads``, which needs ``zero_grad(set_to_none=True)`` -- the reference's loop does that); gradients that live
elsewhere (micro-batch accumulation) are first gathered into a flat scratch buffer.
"""
import ctypes as C

import torch

from .. import _lib


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, model, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid SGD hyper-parameter")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")        # torch.optim.SGD's check
        self._model = model
        self._named = list(model.named_parameters())
        super().__init__([p for _, p in self._named], dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                                           nesterov=nesterov))
        self._flat = self._mom = self._gtmp = None
        self._first = True
        model.alias_grads = True

    # ---- one-time re-homing of the parameters into the flat buffer ---------------------------------------------
    def _engine_of(self, grad):
        for eng in self._model._engines.values():
            if getattr(eng, 'flat_grad', None) is not None and eng.plan.grad_order:
                return eng
        return None

    def _flatten(self, eng):
        flat = torch.zeros_like(eng.flat_grad)
        for k, p in self._named:
            view = eng.grad_view(k, flat)            # same logical shape / KRSC memory order as the parameter
            if tuple(view.shape) != tuple(p.shape):
                raise _lib.RnError(f"{k}: parameter shape {tuple(p.shape)} does not match the engine's {tuple(view.shape)}")
            view.copy_(p.data)
            p.data = view                             # the engine re-binds on the next forward (data_ptr changed)
        self._flat, self._mom = flat, torch.zeros_like(flat)
        self._layout = eng.grad_offsets
        for k, p in self._named:                      # torch-style per-parameter state (views), e.g. for state_dict()
            self.state[p]['momentum_buffer'] = eng.grad_view(k, self._mom)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self._engine_of(None)
        if eng is None:
            raise _lib.RnError("FusedSGD.step before any forward/backward of the model on the device")
        if self._flat is not None:                    # parameters replaced since (module.to(...), load of new tensors): re-home them
            lo, hi = self._flat.data_ptr(), self._flat.data_ptr() + self._flat.numel() * 4
            if any(not (lo <= p.data_ptr() < hi) for _, p in self._named):
                mom = self._mom
                self._flatten(eng)
                if mom is not None and mom.shape == self._mom.shape and mom.device == self._mom.device:
                    self._mom.copy_(mom)
        if self._flat is None:
            self._flatten(eng)
        k0, p0 = self._named[0]
        grads = None
        if p0.grad is not None:
            for e in self._model._engines.values():  # which engine's flat buffer do the gradients alias?
                if e.plan.grad_order and p0.grad.data_ptr() == e.grad_view(k0).data_ptr():
                    if all(p.grad is not None and p.grad.data_ptr() == e.grad_view(k).data_ptr() for k, p in self._named):
                        grads = e.flat_grad
                    break
        if grads is None:                             # gradients held elsewhere (accumulated micro-batches): gather them
            if self._gtmp is None:
                self._gtmp = torch.zeros_like(self._flat)
            for k, p in self._named:
                v = eng.grad_view(k, self._gtmp)
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
            grads = self._gtmp
        g = self.param_groups[0]
        L = _lib.lib()
        _lib.check(L.rn_sgd_step(C.c_void_p(self._flat.data_ptr()), C.c_void_p(grads.data_ptr()), C.c_void_p(self._mom.data_ptr()),
                                 self._flat.numel(), float(g['lr']), float(g['momentum']), float(g['dampening']), float(g['weight_decay']),
                                 int(bool(g['nesterov'])), int(self._first), 1.0,
                                 C.c_void_p(torch.cuda.current_stream(self._flat.device).cuda_stream)))
        self._first = False
        return loss
