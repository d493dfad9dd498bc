"""GPU parity of the whole hot path behind the drop-in boundary (ResNet module -> plan -> librn_hip.so) against the
reference's golden vectors (G4) and the oracle.  Tolerances (relative to the tensor's max-abs):
  fp32 engine: logits 1e-4, loss 1e-5, gradients 1e-3, BN running stats 1e-4  (north-star bound: logits 1e-3)
  bf16 engine: logits 5e-2, identical argmax on the golden batches, gradients 1.5e-1 (bf16 storage of activations
  and gradients through up to 20 layers; reported, not hidden: see DESIGN.md 'Precision')."""
import os

import numpy as np
import pytest
import torch

from filler import fill, fill_state, fill_labels
from oracle.np_model import param_shapes
from oracle import torch_model as tm
from test_oracle_golden import MODELS, model_inputs

pytestmark = pytest.mark.gpu


def build(cfg, st, dtype, p=0.0, **kw):
    from pytorch_ddp_resnet_amd import ResNet
    m = ResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], p, compute_dtype=dtype, **kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    return m.cuda()


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('name', list(MODELS))
def test_models_match_golden(golden, name, dtype):
    cfg = MODELS[name]
    g = golden('g4_' + name)
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    m = build(cfg, st, dtype)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    if dtype != 'fp32' and name in ('v2b_small', 'inet_small'):
        # bottleneck width 4: bf16 moves 16-byte chunks of 8 channels -> rejected loudly, never silently re-routed
        from pytorch_ddp_resnet_amd._lib import RnError
        with pytest.raises(RnError, match='multiples of 8'):
            m(xt)
        return
    # fp16 is the 16-bit engine that has to meet the north-star bound (logits 1e-3); bf16 (8 significant bits) cannot and is
    # bounded at what bf16 storage costs
    tl, tg = {'fp32': (1e-4, 1e-3), 'bf16': (5e-2, 3.5e-1), 'fp16': (1e-3, 1.5e-1)}[dtype]
    S = 256.0 if dtype == 'fp16' else 1.0            # loss scale (fp16 gradients; a GradScaler's job in training)
    m.eval()
    with torch.no_grad():
        le = m(xt)
    assert rel(le, g['eval.logits']) < tl
    m.train()
    logits = m(xt)
    assert logits.dtype == torch.float32 and logits.shape == g['train.logits'].shape
    assert rel(logits, g['train.logits']) < tl
    assert (logits.argmax(1).cpu().numpy() == g['train.logits'].argmax(1)).all()
    loss = torch.nn.functional.cross_entropy(logits, yt)
    assert abs(loss.item() - float(g['train.loss'])) < {'fp32': 1e-5, 'bf16': 3e-2, 'fp16': 2e-3}[dtype] * max(1.0, abs(float(g['train.loss'])))
    (loss * S).backward()
    grads = {k: p.grad / S for k, p in m.named_parameters()}
    pkeys = [k for k, _ in shapes if k.endswith('weight') or k.endswith('bias')]
    norms = np.array([grads[k].double().norm().item() for k in pkeys])
    assert np.abs(norms - g['grad.norms']).max() < min(tg, 0.15) * g['grad.norms'].max()
    gscale = float(g['grad.norms'].max())
    gnorm = float(np.sqrt((g['grad.norms'] ** 2).sum()))
    worst = 0.0
    for k in pkeys:
        if 'grad.' + k in g:
            diff = grads[k].detach().cpu().numpy().astype(np.float64) - g['grad.' + k]
            if dtype == 'fp32':
                assert np.abs(diff).max() < tg * np.abs(g['grad.' + k]).max() + 1e-5 * gscale, k
            else:
                # bf16 activations/gradients: parameters in front of a train-mode BN have gradients that are sums with
                # heavy cancellation (analytically zero for the stem bias), so the bound is L2, relative to the
                # parameter's own norm plus 1 % of the whole gradient's norm
                d2, r2 = float(np.sqrt((diff ** 2).sum())), float(np.sqrt((g['grad.' + k].astype(np.float64) ** 2).sum()))
                worst = max(worst, d2 / (r2 + 0.01 * gnorm))
                assert d2 < tg * r2 + 0.01 * gnorm, (k, d2, r2, gnorm)
    if dtype != 'fp32':
        print(f'{name} {dtype}: logits rel err {rel(logits, g["train.logits"]):.3e}; worst per-parameter gradient L2 error (rel. to own norm + 1% global) {worst:.3f}')
    sd = m.state_dict()
    for k in sd:
        if 'step1.buf.' + k in g:
            assert rel(sd[k], g['step1.buf.' + k]) < {'fp32': 1e-4, 'bf16': 2e-2, 'fp16': 2e-3}[dtype], k


@pytest.mark.parametrize('name', ['rn20', 'wrn_small'])
def test_three_sgd_steps_fp32(golden, name):
    """training.py:92-113 step body with torch.optim.SGD on the engine's gradients: parameters and BN buffers after
    1 and 3 steps (G4 step1/step3)."""
    cfg = MODELS[name]
    g = golden('g4_' + name)
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    m = build(cfg, st, 'fp32')
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, nesterov=nesterov, weight_decay=5e-4 if nesterov else 1e-4)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m.train()
    for step in range(3):
        loss = torch.nn.functional.cross_entropy(m(xt), yt)
        loss.backward()
        opt.step(); opt.zero_grad()
        if step in (0, 2):
            tag = f'step{step + 1}'
            assert abs(loss.item() - float(g[tag + '.loss'])) < 2e-4 * max(1.0, abs(float(g[tag + '.loss'])))
            sums = np.array([p.detach().double().sum().item() for p in m.parameters()])
            assert np.abs(sums - g[tag + '.param_sums']).max() < 1e-3 * max(1.0, np.abs(g[tag + '.param_abs_sums']).max())
            sd = m.state_dict()
            for k in sd:
                if f'{tag}.buf.{k}' in g:
                    assert rel(sd[k], g[f'{tag}.buf.{k}']) < 1e-3, k


COMBOS = [(k, p, j) for k in 'rb' for p in (False, True) for j in (False, True)]


@pytest.mark.parametrize('kind,preact,proj', COMBOS)
@pytest.mark.parametrize('train', [True, False])
def test_all_block_combinations_fp32(kind, preact, proj, train):
    top = 'n a ' if not preact else ''
    tail = 'n a ' if preact else ''
    spec = f'c3,16,3,1,1 {top}{kind}1 {kind}1 {tail}ap4,1,0 fc32,10'
    cfg = dict(spec=spec, preact=preact, use_proj=proj)
    st = fill_state(param_shapes(spec, preact, proj), 7)
    x, y = fill((4, 3, 8, 8), 70), fill_labels(4, 10, 71)
    m = build(cfg, st, 'fp32')
    m.train(train)
    logits = m(torch.from_numpy(x).cuda())
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).cuda()).backward()
    tst = tm.make_trainable({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    lg, _, grads = tm.train_step(tm.TorchResNet(spec, preact, proj), tst, torch.from_numpy(x), torch.from_numpy(y), train=train)
    assert rel(logits, lg) < 1e-4
    scale = max(float(v.abs().max()) for v in grads.values())
    for k, p in m.named_parameters():
        assert (p.grad.cpu() - grads[k]).abs().max().item() < 1e-3 * scale, k
    if train:
        for k, b in m.named_buffers():
            assert rel(b, tst[k]) < 1e-4, k


@pytest.mark.parametrize('dtype', ['fp32', 'fp16'])
@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('spec,preact', __import__('test_lowering').GRAMMAR_CORNERS)
def test_grammar_corners(spec, preact, train, dtype):
    """token sequences outside the shipped configs (resnet.py:122-158 builds any of them): a standalone 'a', a second top-level convolution, an AvgPool2d
    that is not the global pool, 'f' on a map of more than one pixel -- engine vs the torch-CPU port of the reference, logits, every gradient, BN buffers"""
    cfg = dict(spec=spec, preact=preact, use_proj=True)
    st = fill_state(param_shapes(spec, preact, True), 11)
    x, y = fill((4, 3, 8, 8), 110), fill_labels(4, 10, 111)
    m = build(cfg, st, dtype)
    m.train(train)
    logits = m(torch.from_numpy(x).cuda())
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).cuda()).backward()
    tst = tm.make_trainable({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    lg, _, grads = tm.train_step(tm.TorchResNet(spec, preact, True), tst, torch.from_numpy(x), torch.from_numpy(y), train=train)
    tol_l, tol_g = (1e-4, 1e-3) if dtype == 'fp32' else (2e-2, 1e-1)      # fp16: 11-bit storage of every activation and gradient on 4 images
    assert rel(logits, lg) < tol_l
    scale = max(float(v.abs().max()) for v in grads.values())
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == grads[k].shape, k
        assert (p.grad.cpu() - grads[k]).abs().max().item() < tol_g * scale, k
    if train:
        for k, b_ in m.named_buffers():
            assert rel(b_, tst[k]) < (1e-4 if dtype == 'fp32' else 2e-2), k


# every block kind at a size where the weight gradients are of the deferred, batchable kind (64 images of 32 x 32): a queued launch runs later in the
# backward than its op stands, so every way a block hands gradients on (v1 / v2, basic / bottleneck, projection / padded shortcut, the accumulate into
# the block input's gradient) must leave the queued operands alone
BATCH_CASES = [('rn20', None)] + [(f'{k}{int(p)}{int(j)}', (k, p, j)) for k, p, j in COMBOS]


@pytest.mark.parametrize('dtype', ['fp32', 'fp16'])
@pytest.mark.parametrize('case', BATCH_CASES, ids=[c[0] for c in BATCH_CASES])
def test_batched_weight_gradients_are_bit_identical(case, dtype):
    """thin networks queue their weight-gradient launches by tile shape and send each queue out as ONE grid (plan.cpp, rn_conv_wgrad_batch): the
    gradients are those of the single launches, bit for bit (rn_set_variant 1 << 17 turns the queues off), and the batched form really ran."""
    import ctypes as C
    from pytorch_ddp_resnet_amd import _lib
    L = _lib.lib()
    L.rn_set_variant.argtypes = [C.c_int]
    if case[1] is None:
        cfg = MODELS['rn20']                                   # at batch 64 its layers split their pixels 85-227 ways: the deferrable, batchable kind
    else:
        kind, preact, proj = case[1]
        if dtype == 'fp32':
            pytest.skip('block kinds: the 16-bit engine only (same plan, same queues)')
        top, tail = ('n a ' if not preact else ''), ('n a ' if preact else '')
        w = 16 if kind == 'r' else 32                          # bottleneck width = channels / 4: one 16-byte chunk at least
        cfg = dict(spec=f'c3,{w},3,1,1 {top}{kind}2 {kind}2 {tail}ap16,1,0 fc{2 * w},10', preact=preact, use_proj=proj)
    st = fill_state(param_shapes(cfg['spec'], cfg['preact'], cfg['use_proj']), 21)
    xs, ys = torch.from_numpy(fill((64, 3, 32, 32), 210)).cuda(), torch.from_numpy(fill_labels(64, 10, 211)).cuda()

    def grads(variant):
        L.rn_set_variant(variant)
        n0 = int(L.rn_wgrad_batch_launches())
        try:
            m = build(cfg, st, dtype).train()
            out = []
            for _ in range(3):                                 # eager warm-up, capture, replay: the queues live inside the captured ranges too
                for p in m.parameters():
                    p.grad = None
                torch.nn.functional.cross_entropy(m(xs), ys).backward()
                out.append({k: p.grad.clone() for k, p in m.named_parameters()})
            return out, int(L.rn_wgrad_batch_launches()) - n0
        finally:
            L.rn_set_variant(0)
    L.rn_wgrad_batch_launches.restype = C.c_long
    (single, n1), (batched, n2) = grads(1 << 17), grads(0)
    assert n1 == 0 and n2 >= 2                       # (eager warm-up and capture each issue the batched launches; a replayed graph runs no host code)
    for a_, b_ in zip(single, batched):
        for k in a_:
            assert torch.equal(a_[k], b_[k]), k


@pytest.mark.parametrize('preact', [False, True])
def test_dropout_matches_plan_interpreter(preact):
    """p = 0.3: the engine's counter-based masks == the executable spec (np_interp) fed with the same step seed."""
    from np_interp import NumpyPlan
    from oracle import np_ops as ops
    spec = 'c3,8,3,1,1 ' + ('' if preact else 'n a ') + 'r1 r1 ' + ('n a ' if preact else '') + 'ap4,1,0 fc16,10'
    cfg = dict(spec=spec, preact=preact, use_proj=True)
    st = fill_state(param_shapes(spec, preact, True), 9)
    x, y = fill((2, 3, 8, 8), 90), fill_labels(2, 10, 91)
    m = build(cfg, st, 'fp32', p=0.3)
    m.train()
    logits = m(torch.from_numpy(x).cuda())
    eng = next(iter(m._engines.values()))
    seed = (m._seed_base + m._step) & 0x7FFFFFFFFFFFFFFF
    torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).cuda()).backward()
    npl = NumpyPlan(eng.plan)
    npl.load_state(st)
    npl['x'] = x.astype(np.float64)
    npl.forward(step_seed=seed)
    assert rel(logits, npl['logits']) < 1e-4
    npl['dlogits'] = ops.cross_entropy_bwd(npl['logits'], y)
    npl.backward(step_seed=seed)
    ref = npl.grads()
    scale = max(np.abs(v).max() for v in ref.values())
    for k, p in m.named_parameters():
        assert np.abs(p.grad.cpu().numpy() - ref[k]).max() < 1e-3 * scale, k


def test_rn20_batch128_fp32_vs_oracle():
    """BASELINE config 2: ResNet-v1-20, CIFAR shapes, batch 128, reference init; logits within 1e-3 rel (north star),
    identical argmax, loss and every gradient vs the torch-CPU port of the reference step."""
    spec = 'c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10'
    st = tm.init_state(spec, False, False, seed=0)
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(128, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (128,), generator=gen)
    from pytorch_ddp_resnet_amd import ResNet
    m = ResNet(spec, False, False, 0.0, compute_dtype='fp32')
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    m = m.cuda().train()
    logits = m(x.cuda())
    loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    loss.backward()
    tst = tm.make_trainable({k: v.clone() for k, v in st.items()})
    lg, met, grads = tm.train_step(tm.TorchResNet(spec, False, False), tst, x, y)
    r = rel(logits, lg)
    print(f'rn20 b128 fp32: logits rel err {r:.3e}, loss {loss.item():.6f} vs {float(met["loss"]):.6f}')
    assert r < 1e-4
    assert (logits.argmax(1).cpu() == lg.argmax(1)).all()
    assert abs(loss.item() - float(met['loss'])) < 1e-5
    scale = max(float(v.abs().max()) for v in grads.values())
    for k, p in m.named_parameters():
        assert (p.grad.cpu() - grads[k]).abs().max().item() < 1e-3 * scale, k


def test_rn20_batch128_bf16_reported():
    """the bf16 engine on the same batch: argmax agreement and logit error are REPORTED and bounded loosely."""
    spec = 'c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10'
    st = tm.init_state(spec, False, False, seed=0)
    gen = torch.Generator().manual_seed(1234)
    x = torch.randn(128, 3, 32, 32, generator=gen)
    from pytorch_ddp_resnet_amd import ResNet
    m = ResNet(spec, False, False, 0.0, compute_dtype='bf16')
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    m = m.cuda().train()
    with torch.no_grad():
        logits = m(x.cuda())
        lg = tm.TorchResNet(spec, False, False).forward({k: v.clone() for k, v in st.items()}, x, train=True)
    r = rel(logits, lg)
    agree = (logits.argmax(1).cpu() == lg.argmax(1)).float().mean().item()
    print(f'rn20 b128 bf16: logits rel err {r:.3e}, argmax agreement {agree:.3f}')
    assert r < 8e-2 and agree > 0.9


def test_state_dict_roundtrip_and_layout():
    from pytorch_ddp_resnet_amd import ResNet
    m = ResNet('c3,16,3,1,1 n a r1 ap32,1,0 fc16,10', False, False, 0.0).cuda()
    w = m._architecture[3][0]._conv1.weight
    assert w.shape == (16, 16, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = ResNet('c3,16,3,1,1 n a r1 ap32,1,0 fc16,10', False, False, 0.0).cuda()
    m2.load_state_dict(sd)
    x = torch.randn(2, 3, 32, 32, device='cuda')
    m.eval(); m2.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m2(x))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_forked_weight_gradients_are_bit_identical(golden, monkeypatch, dtype):
    """RN_FORK_WGRAD=1: every weight-gradient op runs on the executor's side stream (own workspace, joined at the end of
    the backward).  Same kernels, same order of accumulation -> gradients, logits and BN buffers identical bit for bit to
    the single-stream run, over several steps (a missing dependency would show as stale or torn gradients)."""
    name = 'wrn_small'
    cfg = MODELS[name]
    g = golden('g4_' + name)
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    runs = {}
    for fork in ('0', '1'):
        monkeypatch.setenv('RN_FORK_WGRAD', fork)
        m = build(cfg, st, dtype).train()
        eng_flags = None
        outs = []
        for step in range(3):
            for p_ in m.parameters():
                p_.grad = None
            logits = m(xt + 0.01 * step)
            torch.nn.functional.cross_entropy(logits, yt).backward()
            torch.cuda.synchronize()
            outs.append((logits.detach().clone(), {k: p_.grad.detach().clone() for k, p_ in m.named_parameters()}))
        eng = next(iter(m._engines.values()))
        from pytorch_ddp_resnet_amd.engine import ir
        eng_flags = sum(1 for op in eng.plan.ops if op.flags & ir.F_FORK)
        runs[fork] = (outs, eng_flags, eng.use_graphs)
    assert runs['0'][1] == 0 and runs['1'][1] > 0 and not runs['1'][2]            # forked plan: flagged ops, no graph capture
    for (l0, g0), (l1, g1) in zip(runs['0'][0], runs['1'][0]):
        assert torch.equal(l0, l1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k


def test_graph_replay_survives_a_reducer(golden):
    """a GradReducer's hook points cut the backward into hook-free ranges that are still hipGraph replays (round 1: any hook consumer
    forced eager launches, i.e. every multi-GPU run of the launch-bound nets): same gradients bit for bit, graphs in use, and the
    communication-side actions (pre-scale on the comm stream, bucket events) run between the graphs."""
    from pytorch_ddp_resnet_amd.ddp import GradReducer
    cfg = MODELS['rn20']
    g = golden('g4_rn20')
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    outs = {}
    for tag in ('plain', 'reducer'):
        m = build(cfg, st, 'fp32').train()
        red = GradReducer(m, 1, bucket_cap_mb=0.05, first_bucket_mb=0.01, last_bucket_mb=0.005, force_hooks=True) if tag == 'reducer' else None
        res = []
        for step in range(4):
            for p_ in m.parameters():
                p_.grad = None
            torch.nn.functional.cross_entropy(m(xt + 0.01 * step), yt).backward()
            if red is not None:
                red.finish()
            torch.cuda.synchronize()
            res.append({k: p_.grad.detach().clone() for k, p_ in m.named_parameters()})
        eng = next(e for k, e in m._engines.items() if k[1] and k[2])
        ngraphs = sum(1 for v in eng._graphs.values() if isinstance(v, torch.cuda.CUDAGraph))
        outs[tag] = (res, ngraphs, len(red._bplan(eng).bounds) if red else 0)
    assert outs['plain'][1] == 2                                   # forward + backward
    assert outs['reducer'][2] >= 3 and outs['reducer'][1] >= 3     # several buckets -> several captured backward ranges
    for a, b in zip(outs['plain'][0], outs['reducer'][0]):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_capture_survives_a_thread_that_synchronises(golden):
    """The executor captures plan ranges in `thread_local` error mode (engine/executor.py:_replay).  What failed in round 2 was the DEFAULT (global)
    mode: while the launch thread captured a backward range, a collective backend's own thread (gloo's device-to-host copy / the RCCL watchdog's event
    query) called a synchronising HIP API, which in global mode invalidates every capture in the process -- hipErrorStreamCaptureInvalidated, one run in
    four of the two-rank test.  Here that situation is made on purpose: a helper thread records events on its own stream and issues stream / event
    queries and stream / event synchronisations (the calls those backend threads make) in a tight loop for the whole time the main thread warms up,
    captures and replays the forward and backward graphs.  The capture must succeed, graphs must be in use, and the replayed steps must equal the
    eager ones bit for bit.  (A DEVICE-wide hipDeviceSynchronize from another thread is refused -- hipErrorStreamCaptureUnsupported -- and invalidates the
    capture even in thread_local mode: tried here first; nothing in the product or in the collective backends calls it from a second thread.)"""
    import threading
    cfg = MODELS['rn20']
    g = golden('g4_rn20')
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()

    def steps(m, n):
        res = []
        for step in range(n):
            for p_ in m.parameters():
                p_.grad = None
            torch.nn.functional.cross_entropy(m(xt + 0.01 * step), yt).backward()
            torch.cuda.synchronize()
            res.append({k: p_.grad.detach().clone() for k, p_ in m.named_parameters()})
        return res

    os.environ['RN_NO_GRAPHS'] = '1'
    try:
        eager = steps(build(cfg, st, 'fp32').train(), 4)
    finally:
        del os.environ['RN_NO_GRAPHS']
    stop, calls = threading.Event(), [0]
    side = torch.cuda.Stream()

    def pest():
        ev = torch.cuda.Event()
        while not stop.is_set():
            ev.record(side)
            side.query(); ev.query(); ev.synchronize(); side.synchronize()
            calls[0] += 1

    th = threading.Thread(target=pest, daemon=True)
    th.start()
    try:
        m = build(cfg, st, 'fp32').train()
        got = steps(m, 4)                                   # step 0 eager warm-up, step 1 captures, steps 2-3 replay
    finally:
        stop.set()
        th.join()
    eng = next(e for k, e in m._engines.items() if k[1] and k[2])
    assert calls[0] > 10
    assert sum(1 for v in eng._graphs.values() if isinstance(v, torch.cuda.CUDAGraph)) == 2
    for a, b in zip(eager, got):
        for k in a:
            assert torch.equal(a[k], b[k]), k
