"""
GradScaler for the engine: ``torch.amp.GradScaler`` (the reference's ``tc.cuda.amp.GradScaler()``, script.py:63 / training.py:95-110) whose
gradient inspection runs over the engine's ONE flat gradient buffer.

``scaler.step(optimizer)`` first looks for non-finite gradients (and ``scaler.unscale_`` also divides them by the scale).  torch does
that by walking every parameter of every group and launching ``_amp_foreach_non_finite_check_and_unscale_`` per device / dtype group --
for WRN-28-10 a Python loop over 100+ tensors and kernels that re-WRITE all 36.5 M gradient elements even for the check-only call
(measured: + 0.6 ms on a 6.9 ms step).  When the optimizer's gradients are views of one flat buffer (``FusedSGD.flat_grads()``) the
same answer is one read-only launch (``rn_amp_check_unscale``).  Everything else -- scale growth / backoff, ``update()``, state_dict,
optimizers it does not know -- is torch's, unchanged.
"""
import ctypes as C

import torch

from .. import _lib


class GradScaler(torch.amp.GradScaler):
    def __init__(self, device='cuda', **kw):
        super().__init__(device, **kw)

    def _unscale_grads_(self, optimizer, inv_scale, found_inf, allow_fp16):
        flat = optimizer.flat_grads() if hasattr(optimizer, 'flat_grads') else None
        if flat is None or not flat.is_cuda:
            return super()._unscale_grads_(optimizer, inv_scale, found_inf, allow_fp16)
        inv = inv_scale.to(device=flat.device, dtype=torch.float32, non_blocking=True)
        fi = found_inf.to(device=flat.device, dtype=torch.float32, non_blocking=True)
        if fi.data_ptr() == found_inf.data_ptr():
            fi = fi.clone()                                # the per-device copy torch's replicator would make
        _lib.check(_lib.lib().rn_amp_check_unscale(C.c_void_p(flat.data_ptr()), flat.numel(), C.c_void_p(inv.data_ptr()), C.c_void_p(fi.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)))
        return {flat.device: fi}
