"""
Engine: owns the device buffers of one lowered plan and runs it through librn_hip.so.

PyTorch is used here for device memory (torch.empty on the caching allocator), the current HIP stream and, in
ddp.py, torch.distributed; all arithmetic of the hot path happens inside the library.
"""
import ctypes as C
import os
from typing import Callable, Dict, Optional

import torch

from .. import _lib
from . import ir
from .lowering import Plan

_TORCH_DT = {'f32': torch.float32, 'i64': torch.int64, 'u8': torch.uint8}
ALIGN = 64          # elements: every gradient starts on a 256-byte boundary inside the flat buffer


def layout_grads(plan: Plan):
    """offsets (in elements) of every parameter gradient inside the flat buffer, in gradient-production order, each
    starting on a 256-byte boundary.  -> (offsets dict, total elements)"""
    shapes = {s.key: s for s in plan.slots if s.role == 'grad'}
    offsets, off = {}, 0
    for key in plan.grad_order:
        offsets[key] = off
        off += (shapes[key].numel + ALIGN - 1) // ALIGN * ALIGN
    return offsets, max(off, 1)


_conv_ws = {}       # device index -> the stream-K workspace of the eight-phase convolution kernels (one per process and device, zero-initialised tickets)


def ensure_conv_workspace(device: torch.device):
    """the convolution kernels' stream-K workspace (csrc/conv_igemm8.hip): allocated once, handed to the library; every engine of the process
    launches its convolutions on one stream at a time, so one workspace serves them all.  RN_NO_STREAMK=1: none (whole tiles only; A/B)."""
    L = _lib.lib()
    if os.environ.get('RN_NO_STREAMK', '0') == '1':
        L.rn_set_conv_workspace(None, 0)
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _conv_ws:
        _conv_ws[idx] = torch.zeros(int(L.rn_conv_workspace_bytes()), dtype=torch.uint8, device=device)
    t = _conv_ws[idx]
    _lib.check(L.rn_set_conv_workspace(C.c_void_p(t.data_ptr()), t.numel()))
    return t


class Engine:
    def __init__(self, plan: Plan, device: torch.device, compute_dtype: torch.dtype):
        if device.type != 'cuda':
            raise _lib.RnError("the HIP engine needs an MI355X device ('cuda:N' under PyTorch-ROCm); there is no CPU path")
        self.L = _lib.lib()
        self.plan, self.device = plan, device
        self.T = compute_dtype
        self.rn_dtype = {torch.float32: ir.RN_F32, torch.bfloat16: ir.RN_BF16, torch.float16: ir.RN_F16}[compute_dtype]
        assert plan.meta['fp32'] == (compute_dtype == torch.float32)
        self.generation = 0
        self._conv_ws = ensure_conv_workspace(device)
        # ---- flat gradient buffer, laid out in the order the backward produces the gradients ----
        self.grad_offsets, total = layout_grads(plan)
        self._grad_slots = {s.key: s for s in plan.slots if s.role == 'grad'}
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        # ---- workspace (shared by all wgrad launches: they are serialised on one stream) ----
        ws_bytes = 0
        for kind, g in plan.meta['ws_need']:
            gs = _lib.geom_struct(g)
            fn = self.L.rn_conv_wgrad_ws_bytes if kind == 'wgrad' else self.L.rn_stem_wgrad_ws_bytes
            ws_bytes = max(ws_bytes, int(fn(C.byref(gs))))
        self.ws_bytes = ws_bytes
        # ---- buffers ----
        self.tensors = [None] * len(plan.slots)
        for i, s in enumerate(plan.slots):
            if s.role in ('param', 'buffer', 'input', 'labels'):
                continue
            if s.role == 'grad':
                o = self.grad_offsets[s.key]
                self.tensors[i] = self.flat_grad[o:o + s.numel].view(s.shape if s.shape else ())
            elif s.role == 'u8':
                self.tensors[i] = torch.zeros(s.shape, dtype=torch.uint8, device=device)
            elif s.role == 'ws':                    # (the side stream's workspace holds two regions: plan.cpp alternates them, the slab sums run on a stream of their own)
                self.tensors[i] = torch.empty(max(ws_bytes, 16) * (2 if s.name == 'workspace_side' else 1), dtype=torch.uint8, device=device)
            else:
                dt = self.T if s.dtype == 'T' else _TORCH_DT[s.dtype]
                self.tensors[i] = torch.zeros(s.shape if s.shape else (), dtype=dt, device=device)
        # ---- native plan ----
        ops = (_lib.RnOp * len(plan.ops))()
        for j, op in enumerate(plan.ops):
            buf, dim, fp = op.packed()
            ops[j].kind, ops[j].flags, ops[j].seed = op.kind, op.flags, op.seed
            for k in range(ir.OP_NBUF):
                ops[j].buf[k] = buf[k]
            for k in range(ir.OP_NDIM):
                ops[j].dim[k] = dim[k]
            for k in range(4):
                ops[j].fp[k] = fp[k]
        self._h = C.c_void_p()
        _lib.check(self.L.rn_plan_create(ops, len(plan.ops), len(plan.slots), self.rn_dtype, C.byref(self._h)))
        for i, sl in enumerate(plan.slots):
            if sl.role == 'ws':                     # the main-stream workspace and the side stream's own
                _lib.check(self.L.rn_plan_set_bytes(self._h, i, max(ws_bytes, 16) * (2 if sl.name == 'workspace_side' else 1)))
            elif sl.role == 'fold':                 # hand-off buffer of a split finalize (zeroed above): the launch checks its size
                _lib.check(self.L.rn_plan_set_bytes(self._h, i, int(sl.shape[0]) * 4))
        # ---- deferred weight-gradient slab sums: on thin networks every weight gradient is a ~10 us kernel followed by a ~5 us launch that
        # only adds its split-K slabs (165 of them per ResNet-v2-164 step); with slab regions of their own the sums of a whole range of ops
        # go out as one batched launch at the end of the range (plan.cpp).  RN_NO_DEFER_REDUCE=1: off (A/B).
        self._reduce_arena = None
        if os.environ.get('RN_NO_DEFER_REDUCE', '0') != '1':
            nbytes = int(self.L.rn_plan_defer_reduce(self._h))
            if nbytes:
                self._reduce_arena = torch.empty(nbytes, dtype=torch.uint8, device=device)
                _lib.check(self.L.rn_plan_set_reduce_arena(self._h, C.c_void_p(self._reduce_arena.data_ptr()), nbytes))
        self._ptrs = (C.c_void_p * len(plan.slots))()
        self._bound = None
        self._hooks = {}
        for h in plan.hooks:
            self._hooks.setdefault(h.at, []).append(h)
        self._hook_points = sorted(self._hooks)
        # hipGraph replay of whole forward / backward ranges: kernel boundaries cost ~1.5 us in a graph vs ~5-10 us of
        # host launch path, which is what bounds thin networks (ResNet-v1-20: ~150 launches, v2-164: ~1500 per step).
        # Only for ranges with nothing host-side in between (no hook consumer) and no per-step kernel argument (the
        # dropout seed is one): captured once per binding, invalidated when any bound pointer changes.
        # ... and none with forked weight-gradient ops: replaying a captured fork/join was measured 25 % SLOWER than eager
        # launches (the graph's internal streams ignore the side stream's low priority)
        self.use_graphs = (not any(op.seed or (op.flags & ir.F_FORK) for op in plan.ops)) and os.environ.get('RN_NO_GRAPHS', '0') != '1'
        # weight-gradient ops on a second stream, overlapped with the data-gradient / BatchNorm chain (plan.cpp)
        _lib.check(self.L.rn_plan_set_overlap(self._h, 0 if os.environ.get('RN_NO_OVERLAP', '0') == '1' else 1))
        self._graphs = {}
        self._profiling = False
        self._loss_scratch = None
        self._xbuf = None

    def loss_scratch(self, i):
        """two 4-float device scratch rows for the fused loss kernel (forward sums / backward by-product)."""
        if self._loss_scratch is None:
            self._loss_scratch = torch.zeros(2, 4, dtype=torch.float32, device=self.device)
        return self._loss_scratch[i]

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self.L.rn_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- binding ------------------------------------------------------------------------------------------
    def bind(self, named: Dict[str, torch.Tensor], x: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None):
        """named: reference-keyed parameters and buffers (conv weights must be KRSC in memory)."""
        sig = []
        for i, s in enumerate(self.plan.slots):
            if s.role in ('param', 'buffer'):
                t = named[s.key]
                if t.device != self.device or (s.dtype == 'f32' and t.dtype != torch.float32):
                    raise _lib.RnError(f"{s.key}: expected a float32 tensor on {self.device}, got {t.dtype} on {t.device}")
                if t.dim() == 4 and not t.permute(0, 2, 3, 1).is_contiguous():
                    raise _lib.RnError(f"{s.key}: convolution weights must be channels_last (KRSC in memory)")
                if t.dim() != 4 and not t.is_contiguous():
                    raise _lib.RnError(f"{s.key}: must be contiguous")
                self.tensors[i] = t
            elif s.role == 'input' and x is not None:
                if tuple(x.shape) != s.shape or x.dtype != torch.float32 or not x.is_contiguous() or x.device != self.device:
                    raise _lib.RnError(f"input must be a contiguous float32 NCHW tensor of shape {s.shape} on {self.device}")
                if self.use_graphs:
                    # a captured range reads fixed addresses: a training loop hands over a NEW tensor every step (training.py:94
                    # x.to(device)), which would invalidate the graphs each time -- the batch is copied into an engine-owned buffer
                    if self._xbuf is None:
                        self._xbuf = torch.empty_like(x)
                    self._xbuf.copy_(x)
                    self.tensors[i] = self._xbuf
                else:
                    self.tensors[i] = x
            elif s.role == 'labels' and labels is not None:
                self.tensors[i] = labels
            t = self.tensors[i]
            sig.append(t.data_ptr() if t is not None else 0)
        if sig != self._bound:
            for i, p in enumerate(sig):
                self._ptrs[i] = p
            _lib.check(self.L.rn_plan_bind(self._h, self._ptrs, len(sig)))
            self._bound = sig
            self._graphs = {}

    # ---- execution ------------------------------------------------------------------------------------------
    MIN_GRAPH_OPS = 8        # a captured range costs ~10-16 us of replay overhead: shorter hook-to-hook ranges run eagerly

    def _cuts(self, first, last, hook_fn):
        """hook points inside [first, last) the consumer acts on -> [(position, [hooks])].  The consumer may declare which
        ones (GradReducer.wanted: bucket ends and SyncBN sums only): 80 gradient hooks per WRN-28-10 backward, 6 buckets."""
        if hook_fn is None:
            return []
        wanted = getattr(getattr(hook_fn, '__self__', None), 'wanted', None)
        cuts = []
        for at in self._hook_points:
            if at < first or at >= last:
                continue
            hooks = [h for h in self._hooks[at] if wanted is None or wanted(self, h)]
            if hooks:
                cuts.append((at, hooks))
        return cuts

    def run(self, first: int, last: int, step_seed: int, hook_fn: Optional[Callable] = None):
        """ops [first, last) in order; `hook_fn(engine, hook)` is called between ops at the plan's hook points.  With hipGraph
        replay enabled every hook-free range is a captured graph: the host-side actions (collective launches) sit BETWEEN graphs,
        so a reducer no longer turns replay off for launch-bound networks (round 1: any hook consumer forced eager launches)."""
        stream = None
        pos = first
        graphs = self.use_graphs and not self._profiling
        for at, hooks in self._cuts(first, last, hook_fn) + [(last, [])]:
            if at > pos:
                if graphs and at - pos >= self.MIN_GRAPH_OPS:
                    self._replay(pos, at, step_seed)
                else:
                    if stream is None:
                        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                    _lib.check(self.L.rn_plan_run(self._h, pos, at, step_seed, stream))
                pos = at
            for h in hooks:
                hook_fn(self, h)
        if stream is None:
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.L.rn_plan_join(self._h, stream))       # forked weight-gradient ops: joined at the end of every range

    def _replay(self, a, b, step_seed):
        key = (a, b)
        g = self._graphs.get(key)
        if g is None:                                      # first call runs eagerly (warm-up), second call captures
            self._graphs[key] = 'warm'
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self.L.rn_plan_run(self._h, a, b, step_seed, stream))
            return
        if g == 'warm':
            g = torch.cuda.CUDAGraph()
            # thread_local: a collective backend's own threads (gloo's device-to-host copies, the RCCL watchdog's event queries) may call
            # synchronising HIP APIs while this thread captures; in the default (global) mode that invalidates the capture -- seen as a
            # timing-dependent hipErrorStreamCaptureInvalidated in the two-rank test.  This thread itself only launches kernels.
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                _lib.check(self.L.rn_plan_run(self._h, a, b, step_seed, stream))
            self._graphs[key] = g
        g.replay()

    def side_wait(self, stream):
        """`stream` (a torch.cuda.Stream) waits for the weight-gradient ops forked so far; the compute stream is not joined."""
        _lib.check(self.L.rn_plan_side_wait(self._h, C.c_void_p(stream.cuda_stream)))

    def join(self):
        """the current stream waits for the weight-gradient ops forked so far (before their results are read)."""
        _lib.check(self.L.rn_plan_join(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def forward(self, step_seed=0, hook_fn=None):
        self.generation += 1
        self.run(0, self.plan.n_fwd, step_seed, hook_fn)

    def backward(self, step_seed=0, hook_fn=None):
        self.run(self.plan.n_fwd, len(self.plan.ops), step_seed, hook_fn)

    def profile(self, enable: bool):
        self._profiling = bool(enable)
        _lib.check(self.L.rn_plan_profile(self._h, int(enable)))

    def profile_read(self):
        """-> list of (op, milliseconds) for the ops launched since profile(True)."""
        n = len(self.plan.ops)
        ms = (C.c_float * n)()
        _lib.check(self.L.rn_plan_profile_read(self._h, ms, n))
        return [(self.plan.ops[i], float(ms[i])) for i in range(n)]

    def t(self, name):
        return self.tensors[self.plan.slot_of[name]]

    def grad_view(self, key, flat=None):
        """the gradient of parameter `key`, shaped like the parameter (conv: [K,C,R,S] view of KRSC storage), as a view
        of `flat` (default: the engine's own flat gradient buffer)."""
        s = self._grad_slots[key]
        o = self.grad_offsets[key]
        g = (self.flat_grad if flat is None else flat)[o:o + s.numel].view(s.shape if s.shape else ())
        return g.permute(0, 3, 1, 2) if g.dim() == 4 else g
