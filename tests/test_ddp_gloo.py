"""Multi-rank paths on CPU: world_size 2, gloo, 127.0.0.1 ('distributed without a cluster').

(1) GradReducer: bucket partition of the production-ordered flat gradient buffer, hook-driven launches while the
    backward is still 'running', result == mean over ranks, repeated steps.
(2) SyncBN: the plan's allreduce hooks, interpreted with the oracle (np_interp) on 2 ranks x 4 images, reproduce plain
    BN on the concatenated batch of 8 -- logits, gradients (after the DDP mean) and running statistics (G7 semantics)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from filler import fill, fill_state, fill_labels


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


SPEC = dict(spec='c3,16,3,1,1 r1 r1 n a ap16,1,0 fc32,10', preact=True, use_proj=True)


class _FakeEngine:
    """the parts of Engine the reducer touches, on CPU tensors."""

    def __init__(self, plan):
        from pytorch_ddp_resnet_amd.engine.executor import layout_grads
        self.plan = plan
        self.grad_offsets, total = layout_grads(plan)
        self.flat_grad = torch.zeros(total)
        self.tensors = [None] * len(plan.slots)
        self._shape = {s.key: s.shape for s in plan.slots if s.role == 'grad'}

    def grad_view(self, key):
        o, shp = self.grad_offsets[key], self._shape[key]
        n = int(np.prod(shp)) if shp else 1
        g = self.flat_grad[o:o + n].view(shp if shp else ())
        return g.permute(0, 3, 1, 2) if g.dim() == 4 else g


class _FakeModel:
    def __init__(self, eng):
        self._p = {k: torch.nn.Parameter(torch.zeros_like(eng.grad_view(k))) for k in eng.plan.grad_order}
        self._hook_fn = None
        self.alias_grads = False

    def named_parameters(self):
        return self._p.items()


def _reducer_worker(rank, world, port):
    from pytorch_ddp_resnet_amd.engine.lowering import lower
    from pytorch_ddp_resnet_amd.ddp import GradReducer, BucketPlan
    _init(rank, world, port)
    plan = lower(SPEC['spec'], SPEC['preact'], SPEC['use_proj'], 0.0, 4, 32, 32)
    eng = _FakeEngine(plan)
    model = _FakeModel(eng)
    red = GradReducer(model, world, bucket_cap_mb=0.01, first_bucket_mb=0.002)        # tiny caps -> several buckets
    bp = BucketPlan(plan.grad_order, eng.grad_offsets, eng.flat_grad.numel(), red.cap, red.first_cap)
    assert len(bp.bounds) >= 3
    assert bp.bounds[0][1] == 0 and bp.bounds[-1][2] == eng.flat_grad.numel()
    for (_, a0, b0), (_, a1, _) in zip(bp.bounds, bp.bounds[1:]):
        assert b0 == a1 and b0 > a0                                       # contiguous, non-empty, no overlap
    hooks = [h for h in plan.hooks if h.action == 'grad_ready']
    assert [h.arg for h in hooks] == sorted(h.arg for h in hooks)          # gradients become ready in layout order
    assert hooks[-1].arg == len(plan.grad_order) - 1
    for step in range(3):
        ref = torch.zeros_like(eng.flat_grad)
        for r in range(world):
            g = torch.Generator().manual_seed(100 * step + r)
            ref += torch.randn(eng.flat_grad.numel(), generator=g) / world
        mine = torch.randn(eng.flat_grad.numel(), generator=torch.Generator().manual_seed(100 * step + rank))
        done = -1
        for h in hooks:                                                     # the backward "produces" gradients up to h.arg
            last = h.arg
            end = eng.grad_offsets[plan.grad_order[last + 1]] if last + 1 < len(plan.grad_order) else eng.flat_grad.numel()
            start = 0 if done < 0 else (eng.grad_offsets[plan.grad_order[done + 1]] if done + 1 < len(plan.grad_order) else end)
            eng.flat_grad[start:end] = mine[start:end]
            done = last
            model._hook_fn(eng, h)
        red.finish()
        assert torch.allclose(eng.flat_grad, ref, atol=1e-6), (step, (eng.flat_grad - ref).abs().max())
        for k, p in model.named_parameters():
            assert p.grad.data_ptr() == eng.grad_view(k).data_ptr()
    dist.destroy_process_group()


def test_grad_reducer_two_ranks_gloo():
    mp.spawn(_reducer_worker, args=(2, _free_port()), nprocs=2, join=True)


def _syncbn_worker(rank, world, port, out):
    from np_interp import NumpyPlan
    from oracle import np_ops as ops
    from oracle.np_model import param_shapes
    from pytorch_ddp_resnet_amd.engine.lowering import lower
    _init(rank, world, port)
    st = fill_state(param_shapes(SPEC['spec'], SPEC['preact'], SPEC['use_proj']), 3)
    x, y = fill((8, 3, 32, 32), 30), fill_labels(8, 10, 31)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    plan = lower(SPEC['spec'], SPEC['preact'], SPEC['use_proj'], 0.0, 4, 32, 32, sync_bn=True, world_size=world)
    assert any(h.action == 'allreduce_f32' for h in plan.hooks)

    def hook(npl, h):
        if h.action == 'allreduce_f32':
            t = torch.from_numpy(npl.bufs[h.slot])
            dist.all_reduce(t)
    npl = NumpyPlan(plan)
    npl.load_state(st)
    npl['x'] = xs.astype(np.float64)
    npl.forward(hook_fn=hook)
    logits = npl['logits'].copy()
    npl['dlogits'] = ops.cross_entropy_bwd(logits, ys)
    npl.backward(hook_fn=hook)
    grads = npl.grads()
    for k in sorted(grads):                                                # the DDP mean of the gradient all-reduce
        t = torch.from_numpy(np.ascontiguousarray(grads[k]))
        dist.all_reduce(t)
        grads[k] = t.numpy() / world
    if rank == 0:
        np.savez(out, logits=logits, **{'g.' + k: v for k, v in grads.items()}, **{'b.' + k: v for k, v in npl.state().items()})
    dist.destroy_process_group()


def test_syncbn_two_ranks_equals_big_batch(tmp_path):
    from np_interp import NumpyPlan
    from oracle import np_ops as ops
    from oracle.np_model import param_shapes
    from pytorch_ddp_resnet_amd.engine.lowering import lower
    out = str(tmp_path / 'r0.npz')
    mp.spawn(_syncbn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    st = fill_state(param_shapes(SPEC['spec'], SPEC['preact'], SPEC['use_proj']), 3)
    x, y = fill((8, 3, 32, 32), 30), fill_labels(8, 10, 31)
    plan = lower(SPEC['spec'], SPEC['preact'], SPEC['use_proj'], 0.0, 8, 32, 32)          # one rank, batch 8, plain BN
    npl = NumpyPlan(plan)
    npl.load_state(st)
    npl['x'] = x.astype(np.float64)
    npl.forward()
    assert np.abs(got['logits'] - npl['logits'][:4]).max() < 1e-9
    npl['dlogits'] = ops.cross_entropy_bwd(npl['logits'], y)
    npl.backward()
    for k, v in npl.grads().items():
        assert np.abs(got['g.' + k] - v).max() < 1e-9 * max(1.0, np.abs(v).max()), k
    for k, v in npl.state().items():
        assert np.abs(got['b.' + k] - v).max() < 1e-9, k


def _gpu_two_rank_worker(rank, world, port, out, sync_bn):
    """both ranks share cuda:0 (the pool's boxes have one GPU); gloo moves CUDA tensors, the reducer's stream/event
    ordering is the same code path RCCL takes on a real node."""
    from oracle.np_model import param_shapes
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.ddp import GradReducer
    _init(rank, world, port)
    torch.cuda.set_device(0)
    st = fill_state(param_shapes(SPEC['spec'], SPEC['preact'], SPEC['use_proj']), 3)
    x, y = fill((8, 3, 32, 32), 30), fill_labels(8, 10, 31)
    m = ResNet(SPEC['spec'], SPEC['preact'], SPEC['use_proj'], 0.0, compute_dtype='fp32', sync_bn=sync_bn)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    m = m.cuda().train()
    red = GradReducer(m, world, bucket_cap_mb=0.01, first_bucket_mb=0.002)
    xs, ys = torch.from_numpy(x[rank * 4:(rank + 1) * 4]).cuda(), torch.from_numpy(y[rank * 4:(rank + 1) * 4]).cuda()
    for _ in range(2):                                    # second step: buckets and hooks re-arm correctly
        for p in m.parameters():
            p.grad = None
        logits = m(xs)
        torch.nn.functional.cross_entropy(logits, ys).backward()
        red.finish()
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, logits=logits.detach().cpu().numpy(), **{'g.' + k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()})
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('sync_bn', [True, False])
def test_two_ranks_on_gpu_match_single_process(tmp_path, sync_bn):
    from oracle.np_model import param_shapes
    from pytorch_ddp_resnet_amd import ResNet
    out = str(tmp_path / 'r0.npz')
    mp.spawn(_gpu_two_rank_worker, args=(2, _free_port(), out, sync_bn), nprocs=2, join=True)
    got = np.load(out)
    st = fill_state(param_shapes(SPEC['spec'], SPEC['preact'], SPEC['use_proj']), 3)
    x, y = fill((8, 3, 32, 32), 30), fill_labels(8, 10, 31)
    m = ResNet(SPEC['spec'], SPEC['preact'], SPEC['use_proj'], 0.0, compute_dtype='fp32')
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    m = m.cuda().train()
    if sync_bn:      # SyncBN over 2 x 4 images == plain BN on the 8 images (second step: running stats differ, logits/grads do not)
        logits = m(torch.from_numpy(x).cuda())
        torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).cuda()).backward()
        assert np.abs(got['logits'] - logits[:4].detach().cpu().numpy()).max() < 1e-4 * np.abs(got['logits']).max()
        scale = max(float(p.grad.abs().max()) for p in m.parameters())
        for k, p in m.named_parameters():
            assert np.abs(got['g.' + k] - p.grad.cpu().numpy()).max() < 1e-3 * scale, k
    else:            # plain BN: the reduced gradient is the mean of the two per-rank gradients
        gs = []
        for r in range(2):
            for p in m.parameters():
                p.grad = None
            torch.nn.functional.cross_entropy(m(torch.from_numpy(x[r * 4:(r + 1) * 4]).cuda()), torch.from_numpy(y[r * 4:(r + 1) * 4]).cuda()).backward()
            gs.append({k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()})
        scale = max(np.abs(v).max() for v in gs[0].values())
        for k in gs[0]:
            assert np.abs(got['g.' + k] - 0.5 * (gs[0][k] + gs[1][k])).max() < 1e-4 * scale, k


def test_bucket_plan_tail_bucket():
    """the gradients produced last (the only collective that cannot overlap with the backward) get a small bucket."""
    from pytorch_ddp_resnet_amd.ddp import BucketPlan
    sizes = [1000] * 3 + [8_000_000] * 4 + [200_000] * 3 + [4000, 100]
    order = [f'g{i}' for i in range(len(sizes))]
    off, o = {}, 0
    for k, sz in zip(order, sizes):
        off[k] = o
        o += sz
    bp = BucketPlan(order, off, o, 32 << 20, 4 << 20, 1 << 20)
    assert bp.bounds[0][1] == 0 and bp.bounds[-1][2] == o
    for (_, a0, b0), (_, a1, _) in zip(bp.bounds, bp.bounds[1:]):
        assert b0 == a1 and b0 > a0
    last = bp.bounds[-1]
    assert (last[2] - last[1]) * 4 <= (1 << 20) and last[0] == len(order) - 1
    assert (bp.bounds[-2][2] - bp.bounds[-2][1]) * 4 >= (32 << 20)         # the bucket in front of it was not shrunk
    one = BucketPlan(order[:1], {order[0]: 0}, sizes[0], 32 << 20, 4 << 20, 1 << 20)
    assert one.bounds == [(0, 0, sizes[0])]


def _buffer_sync_worker(rank, world, port, out):
    from pytorch_ddp_resnet_amd.ddp import GradReducer
    _init(rank, world, port)
    torch.manual_seed(100 + rank)                                   # ranks start from DIFFERENT statistics
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1), torch.nn.BatchNorm2d(4))
    for mod in m:
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(); mod.running_var.uniform_(0.5, 2.0); mod.num_batches_tracked.fill_(7 + rank)
    keys_before = list(m.state_dict().keys())
    red = GradReducer(m, world)                                      # registers the forward pre-hook (DDP's broadcast_buffers=True)
    x = torch.randn(2, 3, 8, 8)
    m.eval()
    y = m(x)                                                         # eval forward: rank-0 statistics everywhere (script.py:64 semantics)
    st = {k: v.clone() for k, v in m.state_dict().items()}
    flat = red._buf_flat
    assert flat is not None and flat.numel() == 2 * (8 + 4)
    for mod in m:
        if isinstance(mod, torch.nn.BatchNorm2d):                    # the module buffers ARE slices of the flat tensor
            assert flat.data_ptr() <= mod.running_mean.data_ptr() < flat.data_ptr() + flat.numel() * 4
            assert int(mod.num_batches_tracked) == 7 + rank          # integer step counters are not sent
    assert list(m.state_dict().keys()) == keys_before
    m.train()
    m(x)                                                             # training forward: broadcast first, then each rank updates locally
    torch.save(dict(st=st, y=y.detach()), out + f'.{rank}')
    # sync_bn models skip the broadcast: statistics are identical by construction
    m2 = torch.nn.Sequential(torch.nn.BatchNorm2d(4))
    m2._sync_bn = True
    m2[0].running_mean.fill_(float(rank))
    red2 = GradReducer(m2, world)
    m2.eval(); m2(torch.randn(2, 4, 3, 3))
    assert float(m2[0].running_mean[0]) == float(rank) and red2._buf_flat is None
    dist.destroy_process_group()


def test_buffers_follow_rank0_before_every_forward(tmp_path):
    """SURVEY C2 / script.py:64: DistributedDataParallel's default broadcast_buffers=True."""
    out = str(tmp_path / 'bufs')
    mp.spawn(_buffer_sync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
    for k in r0['st']:
        if r0['st'][k].is_floating_point() and ('running' in k):
            assert torch.equal(r0['st'][k], r1['st'][k]), k
    torch.manual_seed(100)                                           # rank 0's own statistics won
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1), torch.nn.BatchNorm2d(4))
    ref[1].running_mean.normal_(); ref[1].running_var.uniform_(0.5, 2.0)
    assert torch.equal(r1['st']['1.running_mean'], ref[1].running_mean)
