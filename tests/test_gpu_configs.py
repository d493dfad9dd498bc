"""GPU parity of the BASELINE configurations AT SIZE (round-1 review: configs 3, 4, 5 ran in bench.py only).

config 3  WRN-28-10, CIFAR shapes, batch 128            fp32 engine vs the torch-CPU port of the reference step (oracle/torch_model.py)
config 4  ResNet-v2-164, CIFAR-100 shapes, batch 128    fp32, plain BN single process; 2 ranks x 64 with SyncBN == the big batch
config 5  WRN-50-2 (spec B), 224 x 224                  batch 8 fp32 vs the oracle; batch 256 through a size-independent property

Tolerances (relative to the tensor's max-abs):  fp32 engine: logits 1e-4 (north star: 1e-3), loss 1e-5 abs, every gradient 1e-3
of the largest gradient entry; identical argmax.  16-bit engines: measured error is printed and bounded by the figure
DESIGN.md quotes (fp16 is the engine that has to meet the north-star 1e-3; bf16 is bounded loosely and reported).

The replicated-batch property (config 5 at batch 256): a batch made of r copies of b images has the same batch statistics,
the same mean loss and the same parameter gradients as the b images alone (BatchNorm in training mode included), so the full
batch-256 run is checked against the batch-8 oracle exactly -- at a size the CPU oracle could not finish."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_model as tm
from prod_geoms import CONFIGS

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


_ORACLE = {}


def oracle_step(cfg, st, x, y, classes):
    """one training microbatch of the torch-CPU port, in float64 AND in float32 (cached per spec / batch: several tests share one
    run).  -> (logits64, metrics64, grads64, state32 after the step, noise): `noise` is what the reference's own fp32 CPU
    arithmetic differs from fp64 by on this case -- deep nets at random init amplify rounding (measured: WRN-50-2 spec B at batch 8
    has 2.3 % gradient L2 noise between torch fp32 and fp64, ResNet-v2-164 0.4 %), so a fixed 1e-3 bound would test the
    conditioning of the case, not the engine.  The engine is held to the reference's own noise level instead."""
    key = (cfg['spec'], tuple(x.shape), float(x.double().sum()))
    if key not in _ORACLE:
        out = {}
        for dt in (torch.float64, torch.float32):
            tst = tm.make_trainable({k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in st.items()})
            lg, met, grads = tm.train_step(tm.TorchResNet(cfg['spec'], cfg['preact'], cfg['use_proj']), tst, x.to(dt), y)
            out[dt] = (lg.double(), {k: float(v) for k, v in met.items()}, {k: g.detach().double() for k, g in grads.items()},
                       {k: v.detach() for k, v in tst.items()})
        l64, m64, g64, _ = out[torch.float64]
        l32, m32, g32, st32 = out[torch.float32]
        noise = dict(logits=float((l32 - l64).abs().max() / l64.abs().max()), loss=abs(m32['loss'] - m64['loss']),
                     grad={k: float((g32[k] - g64[k]).norm()) for k in g64},
                     grad_total=float(torch.sqrt(sum(((g32[k] - g64[k]) ** 2).sum() for k in g64))))
        _ORACLE[key] = (l64, m64, g64, st32, noise)
    return _ORACLE[key]


def engine_step(cfg, st, x, y, dtype, loss_scale=1.0, **kw):
    from pytorch_ddp_resnet_amd import ResNet
    m = ResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], 0.0, compute_dtype=dtype, **kw)
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    m = m.cuda().train()
    logits = m(x.cuda())
    loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    (loss * loss_scale).backward()
    torch.cuda.synchronize()
    grads = {k: (p.grad.detach().double().cpu() / loss_scale) for k, p in m.named_parameters()}
    return m, logits.detach().double().cpu(), float(loss.detach()), grads


def check_fp32(tag, logits, loss, grads, oracle):
    """fp32 engine vs the fp64 oracle: within 1e-4 on logits (north star: 1e-3), identical argmax, and no further from fp64 than
    4x the reference's own fp32 arithmetic is (plus a floor of 1e-3 of each gradient's norm)."""
    lg, met, g64, _, noise = oracle
    r = rel(logits, lg)
    gtot = float(torch.sqrt(sum((g ** 2).sum() for g in g64.values())))
    gerr = float(torch.sqrt(sum(((grads[k] - g64[k]) ** 2).sum() for k in g64)))
    print(f'{tag} fp32: logits rel err {r:.3e} (torch fp32: {noise["logits"]:.3e}), loss {loss:.6f} vs {met["loss"]:.6f}, '
          f'gradient L2 error {gerr / gtot:.3e} (torch fp32: {noise["grad_total"] / gtot:.3e})')
    assert r < max(1e-4, 4 * noise['logits'])
    assert (logits.argmax(1) == lg.argmax(1)).all()
    assert abs(loss - met['loss']) < max(1e-5, 4 * noise['loss']) * max(1.0, abs(met['loss']))
    assert gerr < 4 * noise['grad_total'] + 1e-3 * gtot
    for k in g64:
        assert float((grads[k] - g64[k]).norm()) < 4 * noise['grad'][k] + 1e-3 * float(g64[k].norm()) + 1e-5 * gtot, k


def inputs(cfg, batch, classes, seed=1234):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, cfg['hw'], cfg['hw'], generator=gen)
    y = torch.randint(0, classes, (batch,), generator=gen)
    return x, y


# ---------------------------------------------------------------------------------------------------- config 3
def test_wrn28_10_batch128_fp32_vs_oracle():
    cfg = CONFIGS['wrn-28-10']
    st = tm.init_state(cfg['spec'], True, True, seed=0)
    x, y = inputs(cfg, 128, 10)
    _, logits, loss, grads = engine_step(cfg, st, x, y, 'fp32')
    check_fp32('wrn-28-10 b128', logits, loss, grads, oracle_step(cfg, st, x, y, 10))


@pytest.mark.parametrize('dtype,bound', [('fp16', 1e-3), ('bf16', 2e-2)])
def test_wrn28_10_batch128_16bit(dtype, bound):
    """the timed engines of bench.py on the headline workload (dropout off: torch's masks cannot be reproduced)."""
    cfg = CONFIGS['wrn-28-10']
    st = tm.init_state(cfg['spec'], True, True, seed=0)
    x, y = inputs(cfg, 128, 10)
    lg = oracle_step(cfg, st, x, y, 10)[0]
    _, logits, loss, grads = engine_step(cfg, st, x, y, dtype, loss_scale=1024.0 if dtype == 'fp16' else 1.0)
    r = rel(logits, lg)
    agree = float((logits.argmax(1) == lg.argmax(1)).float().mean())
    finite = all(bool(torch.isfinite(g).all()) for g in grads.values())
    # the metric is forward + BACKWARD: the timed engine's gradients against the float64 oracle, in total and per parameter (L2, relative; a parameter
    # whose true gradient is ~0 -- the stem bias in front of a BatchNorm -- is measured against 1e-3 of the total norm instead of its own)
    g64 = oracle_step(cfg, st, x, y, 10)[2]
    gtot = float(torch.sqrt(sum((g ** 2).sum() for g in g64.values())))
    gerr = float(torch.sqrt(sum(((grads[k] - g64[k]) ** 2).sum() for k in g64)))
    per = {k: float((grads[k] - g64[k]).norm() / max(float(g64[k].norm()), 1e-3 * gtot)) for k in g64}
    worst = max(per, key=per.get)
    print(f'wrn-28-10 b128 {dtype}: logits rel err {r:.3e}, argmax agreement {agree:.4f}, gradients finite {finite}, '
          f'gradient L2 rel err total {gerr / gtot:.3e}, worst parameter {worst} {per[worst]:.3e}, median {sorted(per.values())[len(per) // 2]:.3e}')
    assert r < bound and finite
    if dtype == 'fp16':
        assert agree == 1.0
        # Measured (round 3, tools/probes/autocast_probe.py on the same weights and batch): this engine 5.0e-4 on the logits and 3.6e-2 on the gradients;
        # PyTorch-ROCm's own autocast(float16) step -- the reference's GPU arithmetic, script.py:63 / training.py:95-110 -- 8.8e-4 and 4.4e-2; the fp32
        # engine 8.9e-4 on the gradients (torch fp32 on the GPU: 1.0e-3).  The net amplifies rounding ~7,500x at random init (fp32: 6e-8 -> 4.5e-4), so
        # fp16 STORAGE of activations and activation gradients (2^-11) lands at a few per cent whoever does it.  Bounds: the reference's own figure.
        ac_logits, ac_grad = _autocast_reference(cfg, st, x, y, g64)
        print(f'    torch autocast(fp16) on the GPU, same case: logits rel err {rel(ac_logits, lg):.3e}, gradient L2 rel err {ac_grad / gtot:.3e}')
        assert gerr / gtot < 6e-2, gerr / gtot
        assert gerr < 1.25 * ac_grad + 1e-3 * gtot, (gerr / gtot, ac_grad / gtot)
        assert per[worst] < 0.5, (worst, per[worst])


def _autocast_reference(cfg, st, x, y, g64):
    """the torch port of the reference step on the GPU under torch.autocast(float16) with a loss scale: what the REFERENCE computes on a GPU."""
    tst = tm.make_trainable({k: v.clone().cuda() for k, v in st.items()})
    net = tm.TorchResNet(cfg['spec'], cfg['preact'], cfg['use_proj'])
    with torch.autocast('cuda', dtype=torch.float16):
        lg = net.forward(tst, x.cuda(), train=True)
        loss = torch.nn.functional.cross_entropy(lg.float(), y.cuda())
    (loss * 1024.0).backward()
    torch.cuda.synchronize()
    gerr = float(torch.sqrt(sum(((tst[k].grad.double().cpu() / 1024.0 - g64[k]) ** 2).sum() for k in g64)))
    return lg.detach().double().cpu(), gerr


# ---------------------------------------------------------------------------------------------------- config 4
def test_v2_164_batch128_fp32_vs_oracle():
    cfg = CONFIGS['v2-164']
    st = tm.init_state(cfg['spec'], True, True, seed=0)
    x, y = inputs(cfg, 128, 100)
    _, logits, loss, grads = engine_step(cfg, st, x, y, 'fp32')
    check_fp32('v2-164 b128', logits, loss, grads, oracle_step(cfg, st, x, y, 100))


def _v2_164_syncbn_worker(rank, world, port, out):
    import torch.distributed as dist
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.ddp import GradReducer
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = CONFIGS['v2-164']
    st = tm.init_state(cfg['spec'], True, True, seed=0)
    x, y = inputs(cfg, 128, 100)
    h = 128 // world
    m = ResNet(cfg['spec'], True, True, 0.0, compute_dtype='fp32', sync_bn=True)
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    m = m.cuda().train()
    red = GradReducer(m, world)
    logits = m(x[rank * h:(rank + 1) * h].cuda())
    torch.nn.functional.cross_entropy(logits, y[rank * h:(rank + 1) * h].cuda()).backward()
    red.finish()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save(dict(logits=logits.detach().cpu(), grads={k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()},
                        bufs={k: b.detach().cpu().clone() for k, b in m.named_buffers()}), out)
    dist.destroy_process_group()


def test_v2_164_syncbn_two_ranks_equal_the_big_batch(tmp_path):
    """BASELINE config 4: 2 ranks x 64 images with SyncBN (163 BN layers, 326 cross-rank sums per step) reproduce the reference's
    plain BN on the 128-image batch: logits of rank 0's half, DDP-mean gradients and the running statistics."""
    import torch.multiprocessing as mp
    from test_ddp_gloo import _free_port
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_v2_164_syncbn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    cfg = CONFIGS['v2-164']
    st = tm.init_state(cfg['spec'], True, True, seed=0)
    x, y = inputs(cfg, 128, 100)
    lg, met, g64, tst, noise = oracle_step(cfg, st, x, y, 100)
    # per-rank loss is the mean over 64 images and the reducer averages over 2 ranks == mean over the 128 images
    assert rel(got['logits'], lg[:64]) < max(1e-4, 4 * noise['logits'])
    gtot = float(torch.sqrt(sum((g ** 2).sum() for g in g64.values())))
    for k in g64:
        assert float((got['grads'][k].double() - g64[k]).norm()) < 4 * noise['grad'][k] + 1e-3 * float(g64[k].norm()) + 1e-5 * gtot, k
    for k, b in got['bufs'].items():
        if b.dtype.is_floating_point:
            assert rel(b, tst[k]) < 1e-4, k


# ---------------------------------------------------------------------------------------------------- config 5
def test_wrn50_2b_batch8_fp32_vs_oracle():
    cfg = CONFIGS['wrn-50-2b']
    st = tm.init_state(cfg['spec'], False, True, seed=0)
    x, y = inputs(cfg, 8, 1000)
    _, logits, loss, grads = engine_step(cfg, st, x, y, 'fp32')
    check_fp32('wrn-50-2b b8', logits, loss, grads, oracle_step(cfg, st, x, y, 1000))


@pytest.mark.parametrize('dtype', ['fp32', 'fp16'])
def test_wrn50_2b_batch256_replicated_batch_property(dtype):
    """full size (batch 256 per GPU, 224 x 224): 32 copies of an 8-image batch == the 8-image oracle (see the module docstring).
    This case is badly conditioned (8 distinct images, 392 samples per channel in the last stage: torch's own fp32 differs from
    fp64 by 2e-5 on the logits, 40x the WRN-28-10 figure), so the 16-bit bound scales with that measured amplification."""
    cfg = CONFIGS['wrn-50-2b']
    st = tm.init_state(cfg['spec'], False, True, seed=0)
    x8, y8 = inputs(cfg, 8, 1000)
    lg, met, g64, _, noise = oracle_step(cfg, st, x8, y8, 1000)
    x, y = x8.repeat(32, 1, 1, 1), y8.repeat(32)
    _, logits, loss, grads = engine_step(cfg, st, x, y, dtype, loss_scale=1024.0 if dtype == 'fp16' else 1.0)
    assert bool(torch.isfinite(logits).all()) and np.isfinite(loss)
    r = rel(logits[:8], lg)
    rep = float((logits.view(32, 8, -1) - logits[:8].unsqueeze(0)).abs().max() / logits.abs().max())
    gn_ref = float(torch.sqrt(sum((g ** 2).sum() for g in g64.values())))
    gn_err = float(torch.sqrt(sum(((grads[k] - g64[k]) ** 2).sum() for k in g64)))
    print(f'wrn-50-2b b256 {dtype}: logits rel err {r:.3e} vs the b8 oracle (torch fp32: {noise["logits"]:.3e}), copies differ by {rep:.2e}, '
          f'loss {loss:.6f} vs {met["loss"]:.6f}, gradient L2 error {gn_err / gn_ref:.3e} (torch fp32: {noise["grad_total"] / gn_ref:.3e})')
    assert all(bool(torch.isfinite(g).all()) for g in grads.values())
    if dtype == 'fp32':
        assert r < max(1e-4, 4 * noise['logits']) and rep < 1e-5 and abs(loss - met['loss']) < max(1e-4, 4 * noise['loss'])
        assert gn_err < 4 * noise['grad_total'] + 1e-3 * gn_ref
        assert (logits[:8].argmax(1) == lg.argmax(1)).all()
    else:
        # fp16 here = what the reference's own autocast(float16) forward gives on this net (2.3e-2 on 32 distinct images, measured in
        # test_wrn50_2b_fp16_on_a_well_conditioned_batch, where the bound is taken from a live autocast run): a fixed ceiling above that figure
        assert r < 4e-2 and rep < 1e-5 and abs(loss - met['loss']) < 1e-2


def test_wrn50_2b_fp16_on_a_well_conditioned_batch():
    """config 5 in the TIMED arithmetic (fp16) on a batch that is not degenerate: 32 DISTINCT images at 224 x 224, forward in train mode, against the
    float64 torch-CPU port -- beside the fp32 engine and PyTorch-ROCm's own autocast(float16) forward (the reference's GPU arithmetic, script.py:63).
    Measured (round 3, tools/probes/autocast_probe50.py): fp32 engine 2.4e-5 (torch fp32: 2.3e-5 CPU, 2.5e-5 GPU); fp16 engine 2.06e-2; torch autocast
    2.29e-2.  The fp16 error is not lost at a stage: against the fp32 engine the block outputs differ by 3.6e-4 after the stem and by a further factor
    of ~1.3 per bottleneck block, up to 8.7e-2 after the sixteenth -- this net (v1 bottlenecks, random init) amplifies ANY rounding ~400x (fp32's 6e-8
    becomes 2.4e-5), so 2^-11 storage lands at 2e-2 whoever does it.  The north star's 1e-3 on this configuration is met by the fp32 engine; the fp16
    engine is held to the reference's own GPU arithmetic, not to a widened constant."""
    cfg = CONFIGS['wrn-50-2b']
    st = tm.init_state(cfg['spec'], False, True, seed=0)
    x, y = inputs(cfg, 32, 1000, seed=4321)
    net = tm.TorchResNet(cfg['spec'], cfg['preact'], cfg['use_proj'])
    with torch.no_grad():
        l64 = net.forward({k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in st.items()}, x.double(), train=True)
        l32 = net.forward({k: v.clone() for k, v in st.items()}, x, train=True).double()
        with torch.autocast('cuda', dtype=torch.float16):
            lac = net.forward({k: v.clone().cuda() for k, v in st.items()}, x.cuda(), train=True).double().cpu()
    torch.cuda.empty_cache()
    from pytorch_ddp_resnet_amd import ResNet
    out = {}
    for dtype in ('fp32', 'fp16'):
        m = ResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], 0.0, compute_dtype=dtype)
        m.load_state_dict({k: v.clone() for k, v in st.items()})
        m = m.cuda().train()
        with torch.no_grad():
            out[dtype] = m(x.cuda()).double().cpu()
        del m
        torch.cuda.empty_cache()
    r32, r16, rt, rac = rel(out['fp32'], l64), rel(out['fp16'], l64), rel(l32, l64), rel(lac, l64)
    agree = float((out['fp16'].argmax(1) == l64.argmax(1)).float().mean())
    print(f'wrn-50-2b b32 (distinct images): logits rel err fp16 engine {r16:.3e}, torch autocast(fp16) {rac:.3e}, fp32 engine {r32:.3e}, torch fp32 {rt:.3e}; '
          f'argmax agreement fp16 {agree:.3f}')
    assert r32 < max(1e-4, 4 * rt) and (out['fp32'].argmax(1) == l64.argmax(1)).all()       # the parity engine: north star with a 40x margin
    assert r16 < 1.25 * rac + 1e-3, (r16, rac)                                                # the timed engine: no worse than the reference's own GPU arithmetic
    assert agree >= 0.9


# ---------------------------------------------------------------------------------------------------- fused loss / AMP
def test_fused_loss_matches_torch_ops_and_carries_the_loss_scale():
    """compute_losses_and_metrics on engine logits takes the one-launch path (metrics.py:10-29 values), its backward applies the
    upstream gradient (a GradScaler's scale) on the device: gradients == scale x the unscaled ones."""
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.algos.metrics import compute_losses_and_metrics, cross_entropy_loss, top_k_err
    cfg = CONFIGS['rn20']
    st = tm.init_state(cfg['spec'], False, False, seed=0)
    x, y = inputs(cfg, 64, 10)
    m = ResNet(cfg['spec'], False, False, 0.0, compute_dtype='fp32')
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    m = m.cuda().train()
    xc, yc = x.cuda(), y.cuda()
    logits = m(xc)
    met = compute_losses_and_metrics(logits, yc)
    assert type(met['loss'].grad_fn).__name__.startswith('_LossFn')
    assert abs(float(met['loss']) - float(cross_entropy_loss(logits.detach(), yc))) < 1e-6
    assert float(met['top1_err']) == pytest.approx(float(top_k_err(logits.detach(), yc, 1)), abs=1e-6)
    assert float(met['top5_err']) == pytest.approx(float(top_k_err(logits.detach(), yc, 5)), abs=1e-6)
    (met['loss'] * 64.0).backward()
    g_fused = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    logits = m(xc)
    torch.nn.functional.cross_entropy(logits, yc).backward()
    scale = max(float(p.grad.abs().max()) for p in m.parameters())
    for k, p in m.named_parameters():
        assert float((g_fused[k] / 64.0 - p.grad).abs().max()) <= 2e-5 * scale, k


@pytest.mark.parametrize('which', ['torch', 'engine'])
def test_fp16_training_with_grad_scaler_matches_fp32_steps(which):
    """(both with torch.amp.GradScaler and with utils.amp.GradScaler, which inspects the flat gradient buffer in one launch)
    the reference's AMP branch (training.py:95-110) on the fp16 engine: scaler.scale(loss).backward(), scaler.step(FusedSGD),
    scaler.update() -- three steps track the fp32 engine + torch.optim.SGD; an injected overflow skips the step on the device."""
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.algos.training import train_step
    from pytorch_ddp_resnet_amd.utils.fused_sgd import FusedSGD
    cfg = CONFIGS['rn20']
    st = tm.init_state(cfg['spec'], False, False, seed=0)
    x, y = inputs(cfg, 64, 10)
    xc, yc = x.cuda(), y.cuda()
    args = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=5e-4)
    ref = ResNet(cfg['spec'], False, False, 0.0, compute_dtype='fp32')
    ref.load_state_dict({k: v.clone() for k, v in st.items()})
    ref = ref.cuda().train()
    opt_ref = torch.optim.SGD(ref.parameters(), **args)
    m = ResNet(cfg['spec'], False, False, 0.0, compute_dtype='fp16')
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    m = m.cuda().train()
    opt = FusedSGD(m, **args)
    from pytorch_ddp_resnet_amd.utils.amp import GradScaler as EngineScaler
    Scaler = torch.amp.GradScaler if which == 'torch' else EngineScaler
    scaler = Scaler('cuda', init_scale=2.0 ** 12)
    for step in range(3):
        train_step(m, xc, yc, optimizer=opt, scaler=scaler)
        torch.nn.functional.cross_entropy(ref(xc), yc).backward()
        opt_ref.step(); opt_ref.zero_grad(set_to_none=True)
    assert scaler.get_scale() == 2.0 ** 12                        # no overflow, no growth yet
    pr = dict(ref.named_parameters())
    num = sum(float(((p.detach() - pr[k].detach()) ** 2).sum()) for k, p in m.named_parameters()) ** 0.5
    den = sum(float(((pr[k].detach() - st[k].cuda()) ** 2).sum()) for k in pr) ** 0.5
    print(f'fp16 + GradScaler: parameter update after 3 steps differs from fp32 by {num / den:.3e} of the update norm')
    assert num < 0.3 * den          # fp16 gradients of this 8-sample-statistics case carry ~10 % error (tests/test_gpu_model.py); the point here is the AMP control flow
    # overflow: a scale that drives the fp16 gradients to inf must skip the update and halve the scale
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    big = Scaler('cuda', init_scale=2.0 ** 40)
    train_step(m, xc, yc, optimizer=opt, scaler=big)
    assert big.get_scale() == 2.0 ** 39
    for k, p in m.named_parameters():
        assert torch.equal(p.detach(), before[k]), k
    # explicit unscale_ (gradient clipping use): the gradients are divided in place once, the step does not divide again
    for p_ in m.parameters():
        p_.grad = None
    logits = m(xc)
    loss = torch.nn.functional.cross_entropy(logits, yc)
    scaler.scale(loss).backward()
    g_scaled = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    scaler.unscale_(opt)
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, g_scaled[k] / scaler.get_scale(), rtol=1e-6, atol=0), k
    w0 = {k: p.detach().clone() for k, p in m.named_parameters()}
    scaler.step(opt)
    scaler.update()
    moved = sum(float((p.detach() - w0[k]).abs().sum()) for k, p in m.named_parameters())
    assert moved > 0 and all(torch.isfinite(p).all() for p in m.parameters())


@pytest.mark.parametrize('dtype', ['fp16', 'fp32'])
def test_learns_a_separable_problem_end_to_end(dtype):
    """200 SGD steps of a small pre-activation WRN (dropout 0.2) on a linearly separable 10-class toy problem, everything on the
    product path (engine forward / backward, fused loss, FusedSGD, GradScaler for fp16): the training loss must fall below 0.25 (from 2.3) and
    the EVAL-mode accuracy (running statistics, no dropout) on fresh samples must exceed 95 %.  Catches what one-step parity
    cannot: statistics that drift, a mask that differs between forward and backward, an update applied to the wrong buffer."""
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.algos.training import train_step
    from pytorch_ddp_resnet_amd.utils.optim_util import get_optimizer
    torch.manual_seed(0)
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(1)
    protos = torch.randn(10, 3, 32, 32, device=dev, generator=g)

    def batch(n):
        y = torch.randint(0, 10, (n,), device=dev, generator=g)
        return protos[y] * 0.7 + torch.randn(n, 3, 32, 32, device=dev, generator=g), y

    m = ResNet('c3,32,3,1,1 r2 r2 n a ap16,1,0 fc64,10', True, True, 0.2, compute_dtype=dtype).to(dev).train()
    opt = get_optimizer('SGD', m, dict(lr=0.05, momentum=0.9, dampening=0.0, nesterov=True, weight_decay=5e-4))
    scaler = torch.amp.GradScaler('cuda') if dtype == 'fp16' else None
    losses = []
    for step in range(200):
        x, y = batch(64)
        out = train_step(m, x, y, opt, None, 1, 1, 1, {}, scaler=scaler)
        if step % 10 == 9:
            losses.append(float(out['loss']))
    assert all(np.isfinite(losses)) and losses[-1] < 0.25 and losses[0] > 1.5, losses
    m.eval()
    with torch.no_grad():
        x, y = batch(512)
        acc = float((m(x).argmax(1) == y).float().mean())
    assert acc > 0.95, (acc, losses)


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the form of the N = 1 command) spawns the two ranks itself -- children of
    torch.distributed.run, started before the parent touches the GPU -- and prints ONE JSON line; rehearsed with both ranks on this box's one GPU
    (RN_BENCH_REHEARSAL=1: collectives through gloo).  A failing child makes the parent exit non-zero."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['RN_BENCH_REHEARSAL'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'rn20', '--no-parity',
                        '--no-cpu-baseline', '--also', ''], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['parallelism'] == 'dp2'
    bad = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--workload', 'rn20', '--dtype', 'nope'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0
