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

Checkpoints: ``state_dict()`` / ``load_state_dict()`` round-trip like torch.optim.SGD's (the reference saves
``optimizer.state_dict()``, checkpoint_util.py:68-85): ``state[p]['momentum_buffer']`` are views of the flat momentum
buffer, a loaded state is copied into it and the first step after a resume continues the momentum instead of re-seeding it.

AMP: ``torch.amp.GradScaler.step(optimizer)`` hands this optimizer the loss scale and the found-inf flag as DEVICE
tensors (``_step_supports_amp_scaling``); the kernel unscales on the fly and skips the whole update when a gradient was
non-finite (training.py:104-110) without a host synchronisation.
"""
import ctypes as C
import weakref

import torch

from .. import _lib


class FusedSGD(torch.optim.Optimizer):
    _step_supports_amp_scaling = True     # GradScaler.step sets .grad_scale / .found_inf (device tensors) around step()

    def __init__(self, model, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid SGD hyper-parameter")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")        # torch.optim.SGD's check
        self._model = model
        self._named = list(model.named_parameters())
        self._flat = self._mom = self._gtmp = None
        self._grad_offsets = weakref.WeakKeyDictionary()      # engine -> byte offset of every parameter's gradient view in its flat buffer
        self._first = True
        self._loaded_mom = None           # momentum buffers of a loaded state_dict waiting for the flat buffer to exist
        super().__init__([p for _, p in self._named], dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                                           nesterov=nesterov))
        model.alias_grads = True

    def add_param_group(self, param_group):
        if getattr(self, 'param_groups', None):
            raise ValueError("FusedSGD steps ONE flat buffer with one set of hyper-parameters: a second param group is not supported "
                             "(use torch.optim.SGD, RN_TORCH_SGD=1, for per-group settings)")
        super().add_param_group(param_group)

    # ---- checkpoint round trip ----------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)          # torch replaces state[p]['momentum_buffer'] by fresh tensors
        if len(self.param_groups) != 1:
            raise ValueError("FusedSGD: the loaded state has more than one param group")
        loaded = {}
        for k, p in self._named:
            mb = self.state.get(p, {}).get('momentum_buffer', None)
            if mb is not None:
                loaded[k] = mb.detach().clone()
        self._loaded_mom = loaded or None
        if self._flat is not None:
            self._adopt_loaded()

    def _adopt_loaded(self):
        """copies a loaded momentum state into the flat buffer and re-points state[p] at the views."""
        if self._loaded_mom is None:
            return
        eng = self._engine_of()
        for k, p in self._named:
            view = eng.grad_view(k, self._mom)
            if k in self._loaded_mom:
                view.copy_(self._loaded_mom[k].to(view.device))
            self.state[p]['momentum_buffer'] = view
        self._first = False                           # the momentum continues: no re-seeding with the first gradient
        self._loaded_mom = None

    # ---- one-time re-homing of the parameters into the flat buffer ---------------------------------------------
    def _engine_of(self):
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

    def flat_grads(self):
        """the engine's flat gradient buffer if every parameter's .grad is a view of it (the product's backward leaves them so), else
        None.  utils.amp.GradScaler inspects / unscales the gradients through it in one launch."""
        k0, p0 = self._named[0]
        if p0.grad is None:
            return None
        for e in self._model._engines.values():       # which engine's flat buffer do the gradients alias?
            if not e.plan.grad_order or p0.grad.data_ptr() != e.grad_view(k0).data_ptr():
                continue
            # every gradient must be the engine's view.  Building 100+ view tensors per call costs ~0.5 ms of host time (the launch
            # thread is close to the critical path at 6.4 ms per step): the byte offsets are computed once per engine, then it is one
            # pointer comparison per parameter
            offs = self._grad_offsets.get(e)
            if offs is None:
                base = e.flat_grad.data_ptr()
                offs = self._grad_offsets[e] = [e.grad_view(k).data_ptr() - base for k, _ in self._named]
            base = e.flat_grad.data_ptr()
            if all(p.grad is not None and p.grad.data_ptr() == base + o for (_, p), o in zip(self._named, offs)):
                return e.flat_grad
            return None
        return None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self._engine_of()
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
        self._adopt_loaded()
        grads = self.flat_grads()
        if grads is None:                             # gradients held elsewhere (accumulated micro-batches): gather them
            missing = [k for k, p in self._named if p.grad is None]
            if missing:
                # torch.optim.SGD skips such parameters entirely (no weight decay, no momentum update); one flat launch
                # cannot skip slices, and every parameter of this model receives a gradient from every backward
                raise _lib.RnError(f"FusedSGD.step: {len(missing)} parameters have no gradient (first: {missing[0]}); "
                                   "run backward first, or use torch.optim.SGD (RN_TORCH_SGD=1) for partial updates")
            if self._gtmp is None:
                self._gtmp = torch.zeros_like(self._flat)
            for k, p in self._named:
                eng.grad_view(k, self._gtmp).copy_(p.grad)
            grads = self._gtmp
        g = self.param_groups[0]
        L = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream(self._flat.device).cuda_stream)
        grad_scale, found_inf = getattr(self, 'grad_scale', None), getattr(self, 'found_inf', None)
        if grad_scale is not None or found_inf is not None:
            if float(g['dampening']) != 0.0 and float(g['momentum']) != 0.0:
                raise _lib.RnError("FusedSGD under a GradScaler needs dampening == 0 (a skipped first step must not consume the "
                                   "momentum seeding; with zero dampening the zero-initialised buffer gives torch's first step exactly)")

            def dev_f32(t):
                if t is None:
                    return None
                t = t.detach().to(device=self._flat.device, dtype=torch.float32).reshape(-1)
                return t if t.is_contiguous() else t.contiguous()
            gs, fi = dev_f32(grad_scale), dev_f32(found_inf)
            _lib.check(L.rn_sgd_step_amp(C.c_void_p(self._flat.data_ptr()), C.c_void_p(grads.data_ptr()), C.c_void_p(self._mom.data_ptr()),
                                         self._flat.numel(), float(g['lr']), float(g['momentum']), float(g['dampening']), float(g['weight_decay']),
                                         int(bool(g['nesterov'])), 0, C.c_void_p(gs.data_ptr()) if gs is not None else None,
                                         C.c_void_p(fi.data_ptr()) if fi is not None else None, stream))
            self._first = False
            return loss
        _lib.check(L.rn_sgd_step(C.c_void_p(self._flat.data_ptr()), C.c_void_p(grads.data_ptr()), C.c_void_p(self._mom.data_ptr()),
                                 self._flat.numel(), float(g['lr']), float(g['momentum']), float(g['dampening']), float(g['weight_decay']),
                                 int(bool(g['nesterov'])), int(self._first), 1.0, stream))
        self._first = False
        return loss
