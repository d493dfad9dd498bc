"""Runs a plan through BOTH the HIP engine (through the C ABI) and the numpy interpreter on the same inputs."""
import numpy as np
import torch

from np_interp import NumpyPlan
from pytorch_ddp_resnet_amd.engine import ir
from pytorch_ddp_resnet_amd.engine.executor import Engine
from pytorch_ddp_resnet_amd.engine.lowering import Plan
from pytorch_ddp_resnet_amd.engine.ir import Op, Slot

DEV = torch.device('cuda:0')


def bf16_round(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy().astype(np.float64)


class PlanBuilder:
    def __init__(self):
        self.slots, self.ops, self.ws_need = [], [], []

    def slot(self, name, shape, dtype='T', role=None):
        role = role or ('act' if dtype == 'T' else ('ws' if dtype == 'u8' else 'f32'))
        self.slots.append(Slot(name, role, tuple(shape), dtype))
        return len(self.slots) - 1

    def op(self, kind, **kw):
        self.ops.append(Op(kind, **kw))

    def plan(self, fp32):
        so = {s.name: i for i, s in enumerate(self.slots)}
        return Plan(self.slots, self.ops, len(self.ops), [], [], [], so, True, meta=dict(ws_need=self.ws_need, fp32=fp32))


_last = [None]


def last_engine():
    """the Engine of the latest run_both (a test that launches the same plan again, e.g. to see a hand-off buffer reused)"""
    return _last[0]


def run_both(plan, inputs, fp32, step_seed=0):
    """inputs: {slot name: array}.  returns (hip: {name: float64 array}, ref: {name: float64 array})."""
    T = torch.float32 if fp32 else torch.bfloat16
    eng = Engine(plan, DEV, T)
    _last[0] = eng
    ref = NumpyPlan(plan)
    for name, arr in inputs.items():
        i = plan.slot_of[name]
        s = plan.slots[i]
        arr = np.asarray(arr)
        if s.dtype == 'T' and not fp32:
            arr = bf16_round(arr)
        if s.dtype == 'i64':
            eng.tensors[i].copy_(torch.from_numpy(arr.astype(np.int64)).reshape(s.shape))
            ref.bufs[i] = arr.astype(np.int64).reshape(s.shape)
        else:
            eng.tensors[i].copy_(torch.from_numpy(arr.astype(np.float32)).reshape(s.shape).to(eng.tensors[i].dtype))
            ref.bufs[i] = arr.astype(np.float64).reshape(s.shape)
    eng.bind({})
    eng.run(0, len(plan.ops), step_seed)
    torch.cuda.synchronize()
    ref.run(0, len(plan.ops), step_seed=step_seed)
    hip = {}
    for i, s in enumerate(plan.slots):
        if s.role == 'ws' or eng.tensors[i] is None:
            continue
        hip[s.name] = eng.tensors[i].detach().float().cpu().numpy().astype(np.float64) if s.dtype not in ('i64', 'u8') else eng.tensors[i].cpu().numpy()
    refd = {s.name: ref.bufs[i] for i, s in enumerate(plan.slots) if s.role != 'ws'}
    return hip, refd


def max_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def geom(N, H, W, C, K, k, stride, pad):
    P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return dict(N=N, H=H, W=W, C=C, P=P, Q=Q, K=K, R=k, S=k, stride=stride, pad=pad)
