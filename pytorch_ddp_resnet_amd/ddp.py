"""
Data-parallel gradient reduction for the HIP engine: bucketed all-reduce of the flat gradient buffer, overlapped with
the backward that is still producing it, plus the cross-rank sums SyncBN needs.

Replaces what ``torch.nn.parallel.DistributedDataParallel`` does for the reference at
/root/reference/script.py:64-71 (bucketed gradient mean across ranks, fired from autograd hooks inside
``loss.backward()``, training.py:100-102).  The engine's backward is one host-driven op stream, so instead of
autograd hooks the plan carries ``grad_ready`` hook points: when the last gradient of a bucket has been enqueued, an
event is recorded on the compute stream, the communication stream waits on it and issues the collective
(``backend='nccl'`` is RCCL over xGMI on ROCm; ``gloo`` runs the same code on CPU tensors for the tests).  Buckets are
contiguous slices of ``Engine.flat_grad``, which is laid out in gradient-production order, so no copies or flattening
are needed.  Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce is per-link bound
(~1.7 ms for WRN-28-10's 147 MB): few, large buckets keep the links busy without paying per-collective latency.
"""
from typing import List, Optional

import torch
import torch.distributed as dist


class BucketPlan:
    """splits the production-ordered flat gradient buffer into contiguous buckets of at least `cap_bytes`.  The gradients
    produced LAST (stem, first block) get a small bucket of their own (<= `last_cap_bytes`): only that collective cannot
    overlap with the backward, so its size is the exposed tail of every step."""

    def __init__(self, grad_order: List[str], offsets: dict, total_elems: int, cap_bytes: int, first_cap_bytes: Optional[int] = None,
                 last_cap_bytes: int = 0):
        n = len(grad_order)
        starts = [offsets[k] for k in grad_order]
        ends = starts[1:] + [total_elems]
        tail_from = n                                   # index of the first gradient of the tail bucket (n: no tail bucket)
        if last_cap_bytes > 0 and n > 1:
            tail_from = n - 1                           # at least the last gradient, then as many as fit under the cap
            while tail_from > 1 and (total_elems - starts[tail_from - 1]) * 4 <= last_cap_bytes:
                tail_from -= 1
        self.bounds = []          # (last_grad_index, start_elem, end_elem)
        start, cap = 0, (first_cap_bytes or cap_bytes)
        for i in range(n):
            end = ends[i]
            last = i + 1 == n
            cut_for_tail = (i + 1 == tail_from) and end > start
            if (end - start) * 4 >= cap or last or cut_for_tail:
                if end > start:
                    self.bounds.append((i, start, end))
                start, cap = end, cap_bytes
        assert not self.bounds or self.bounds[-1][2] == total_elems


class GradReducer:
    """attach to a ResNet: ``reducer = GradReducer(model, world_size)``; after ``loss.backward()`` call
    ``reducer.finish()`` -- gradients in ``p.grad`` are then the mean over ranks (views of the flat buffer)."""

    def __init__(self, model, world_size: int, bucket_cap_mb: float = 32.0, first_bucket_mb: float = 4.0, last_bucket_mb: float = 1.0, group=None,
                 broadcast_buffers: bool = True, force_hooks: bool = False):
        """:param broadcast_buffers: DDP's default (script.py:64 constructs DistributedDataParallel without arguments, so
        ``broadcast_buffers=True``): before EVERY forward, training or eval, rank 0's BatchNorm running statistics replace
        every other rank's, so all ranks evaluate with rank-0 statistics.  Here the floating-point buffers are
        re-homed once as views of ONE flat tensor and synchronised by one broadcast per forward.  With ``sync_bn`` every rank
        computes its running statistics from the same all-reduced sums with the same arithmetic, i.e. they are identical by
        construction and the broadcast is skipped (tests/test_ddp_gloo.py proves both).  ``num_batches_tracked`` advances by one
        per training forward on every rank alike and is not sent.
        :param force_hooks: act on the hook points even with world_size == 1 (single-GPU rehearsal of the multi-GPU control flow)."""
        self.model, self.world, self.group = model, world_size, group
        self.force = bool(force_hooks)
        self._buf_flat = None
        self._broadcast_buffers = bool(broadcast_buffers) and world_size > 1 and not getattr(model, '_sync_bn', False)
        if self._broadcast_buffers and hasattr(model, 'register_forward_pre_hook'):
            model.register_forward_pre_hook(lambda _m, _inp: self.sync_buffers())
        self.cap, self.first_cap, self.last_cap = int(bucket_cap_mb * 2 ** 20), int(first_bucket_mb * 2 ** 20), int(last_bucket_mb * 2 ** 20)
        import weakref
        self._plans = weakref.WeakKeyDictionary()         # per-engine caches die with the engine (ResNet._apply discards engines)
        self._ends = weakref.WeakKeyDictionary()
        self._eng = None
        self._next = 0
        self._work = []
        self._comm_stream = None
        model._hook_fn = self._on_hook
        model.alias_grads = True          # finish() re-points p.grad at the reduced flat buffer every step

    # ---- C2: rank-0 BatchNorm statistics before every forward ------------------------------------------------------
    def _flatten_buffers(self):
        bufs = [(k, b) for k, b in self.model.named_buffers() if b.is_floating_point()]
        if not bufs:
            return None
        flat = torch.empty(sum(b.numel() for _, b in bufs), dtype=bufs[0][1].dtype, device=bufs[0][1].device)
        o = 0
        for _, b in bufs:
            v = flat[o:o + b.numel()].view(b.shape)
            v.copy_(b.data)
            b.data = v                                   # the module's buffer IS the slice: state_dict / the engine see the same memory
            o += b.numel()
        return flat

    def sync_buffers(self):
        if not self._broadcast_buffers:
            return
        dev = next(self.model.parameters()).device
        if self._buf_flat is None or self._buf_flat.device != dev or any(
                b.is_floating_point() and not (self._buf_flat.data_ptr() <= b.data_ptr() < self._buf_flat.data_ptr() + self._buf_flat.numel() * 4)
                for b in self.model.buffers()):
            self._buf_flat = self._flatten_buffers()     # first use, or the buffers were replaced (.to(), load_state_dict of new tensors)
        if self._buf_flat is not None:
            dist.broadcast(self._buf_flat, src=0, group=self.group)

    def wanted(self, eng, hook) -> bool:
        """hook points this reducer acts on: SyncBN sums, and the gradients that complete a bucket."""
        if self.world <= 1 and not self.force:
            return False
        if hook.action == 'allreduce_f32':
            return True
        if hook.action != 'grad_ready':
            return False
        ends = self._ends.get(eng)
        if ends is None:
            ends = self._ends[eng] = {b[0] for b in self._bplan(eng).bounds}
        return hook.arg in ends

    # ---- hook entry point: called by Engine.run between op ranges -------------------------------------------
    def _on_hook(self, eng, hook):
        if hook.action == 'allreduce_f32':          # SyncBN statistics: needed immediately by the next op
            if self.world > 1:
                dist.all_reduce(eng.tensors[hook.slot], op=dist.ReduceOp.SUM, group=self.group)
            return
        if hook.action != 'grad_ready' or (self.world <= 1 and not self.force):
            return
        if self._eng is not eng or self._next >= len(self._bplan(eng).bounds):
            self._begin(eng)
        bp = self._bplan(eng)
        while self._next < len(bp.bounds) and bp.bounds[self._next][0] <= hook.arg:
            self._launch(eng, bp.bounds[self._next])
            self._next += 1

    def _bplan(self, eng):
        bp = self._plans.get(eng)
        if bp is None:
            bp = BucketPlan(eng.plan.grad_order, eng.grad_offsets, eng.flat_grad.numel(), self.cap, self.first_cap, self.last_cap)
            self._plans[eng] = bp
        return bp

    def _begin(self, eng):
        self._eng, self._next, self._work = eng, 0, []

    def _launch(self, eng, bound):
        _, a, b = bound
        buf = eng.flat_grad[a:b]
        if buf.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=buf.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            self._comm_stream.wait_event(ev)
            if hasattr(eng, 'side_wait'):
                # weight gradients are produced on the engine's side stream: only the COMMUNICATION stream waits for the ones
                # forked so far; the compute stream keeps issuing the data-gradient chain (round 1 joined the compute stream here)
                eng.side_wait(self._comm_stream)
            with torch.cuda.stream(self._comm_stream):
                if self.world > 1 and dist.get_backend(self.group) == 'nccl':
                    # RCCL averages inside the collective (ReduceOp.AVG): no scaling pass over the bucket at all
                    self._work.append(dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
                else:                                            # gloo has no AVG: pre-scale, then SUM (rehearsals of the multi-rank path on one GPU)
                    buf.mul_(1.0 / self.world)
                    if self.world > 1:
                        self._work.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            buf.mul_(1.0 / self.world)
            if self.world > 1:
                self._work.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- after backward ------------------------------------------------------------------------------------------
    def finish(self):
        eng = self._eng
        if (self.world > 1 or self.force) and eng is not None:
            bp = self._bplan(eng)
            while self._next < len(bp.bounds):                  # gradients produced after the last hook point
                self._launch(eng, bp.bounds[self._next])
                self._next += 1
            for w in self._work:
                w.wait()                                         # CUDA: makes the current stream wait; CPU: blocks
            if self._comm_stream is not None:
                torch.cuda.current_stream(eng.flat_grad.device).wait_stream(self._comm_stream)
            self._work = []
            self._next = len(bp.bounds)
        if eng is not None:
            for k, p in self.model.named_parameters():
                p.grad = eng.grad_view(k)


def broadcast_parameters(model, src: int = 0, group=None):
    """what DDP's constructor does at script.py:64 (rank-0 parameters and buffers define the model)."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
