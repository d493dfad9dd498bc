"""GPU parity of the PRODUCTION convolution tiles: the real layer shapes of the BASELINE configurations (tests/prod_geoms.py),
which select the igemm_dma<128xBN> / igemm_ws / igemm_dma<256x32> / wgrad<BKxBC> instantiations that dominate bench.py
(round-1 review: those ran in the bench only).  Forward with residual + fused BatchNorm statistics, data gradient with the
fused BatchNorm-backward sums (mask, gscale, xhat) and every parity class of a stride-2 layer, weight gradient at production
split counts -- through the C ABI (one-op plans), against torch-CPU convolutions (numpy is too slow at these sizes).

The kernel that ran is ASSERTED (rn_kernel_log): a geometry that silently re-routes to another tile fails the test.
Small geometries are additionally run with the wave-specialised schedule forced (rn_set_variant 128) and forbidden (512)
so both schedules see every tile shape.

Tolerances: as tests/test_gpu_kernels.py (fp32 2e-5 of the tensor's max; 16-bit engines: operands pre-rounded on both
sides, one output rounding: bf16 6e-3, fp16 8e-4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from filler import fill
from pytorch_ddp_resnet_amd import _lib
from pytorch_ddp_resnet_amd.engine import ir
from pytorch_ddp_resnet_amd.engine.lowering import conv_stats_rows
from prod_geoms import PROD_GEOMS, IGEMM8_GEOMS, R8_GEOMS, STEM8_GEOMS, geom, resolve

pytestmark = pytest.mark.gpu

DTYPES = ['fp32', 'bf16', 'fp16']
TOL = {'fp32': 2e-5, 'bf16': 6e-3, 'fp16': 8e-4}
TORCH_DT = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}
RN_DT = {'fp32': ir.RN_F32, 'bf16': ir.RN_BF16, 'fp16': ir.RN_F16}


def _round(a, dtype):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(TORCH_DT[dtype]).to(torch.float32)


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def run_conv_case(g, dtype, variant=0, expect_same_names=True, fwd_res=True, dgrad_merge='none', xmask=False, variant2=0):
    """one convolution layer: pack -> forward(+residual, +stats) -> dgrad(+BN-backward sums) -> wgrad, HIP vs torch-CPU.
    fwd_res: the forward adds an identity residual; dgrad_merge: 'none' | 'res' (dx = conv^T(dy) + shortcut gradient) | 'acc' (dx += conv^T(dy)) --
    the operand sets the block backward of the reference lowers to (the eight-phase kernels are specialised per set).  xmask: the mask fed is exactly
    [bn_x * scale + shift > 0] and the op says so (F_MASK_RECOMPUTE -> rn_conv_epilogue.mask_from_x): a kernel may test bn_x instead of reading it."""
    import gpu_harness as h
    from pytorch_ddp_resnet_amd.engine.executor import Engine
    L = _lib.lib()
    fp32 = dtype == 'fp32'
    g = resolve(g, fp32)
    N, Hh, W, C, K, k, s, p = g
    gm = geom(*g)
    P, Q = gm['P'], gm['Q']
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); w = b.slot('w', (K, k, k, C), 'f32')
    wf = b.slot('wf', (K, k * k, C)); wd = b.slot('wd', (C, k * k, K))
    y = b.slot('y', (N, P, Q, K)); res = b.slot('res', (N, P, Q, K)); dy = b.slot('dy', (N, P, Q, K)); dx = b.slot('dx', (N, Hh, W, C))
    st = b.slot('st', (conv_stats_rows(gm), 2, K), 'f32'); dp = b.slot('dp', (conv_stats_rows(gm, True), 2, C), 'f32')
    bx = b.slot('bx', (N, Hh, W, C)); bm = b.slot('bm', (N, Hh, W, C)); coef = b.slot('coef', (4, C), 'f32')
    dw = b.slot('dw', (K, k, k, C), 'f32'); ws = b.slot('workspace', (0,), 'u8')
    b.op(ir.OP_PACK_W, buf=dict(w=w, w_fwd=wf, w_dgrad=wd), dim=dict(K=K, RS=k * k, C=C))
    b.op(ir.OP_CONV_FWD, buf=dict(x=x, w_fwd=wf, y=y, res=res if fwd_res else -1, stats=st),
         dim=dict(gm, res_mode=ir.RES_SAME if fwd_res else 0, res_C=K if fwd_res else 0))
    dres = b.slot('dres', (N, Hh, W, C))
    b.op(ir.OP_CONV_DGRAD, buf=dict(dy=dy, w_dgrad=wd, dx=dx, res=dres if dgrad_merge == 'res' else -1, bn_x=bx, bn_mask=bm, bn_coef=coef, bn_partial=dp),
         dim=dict(gm, res_mode=ir.RES_SAME if dgrad_merge == 'res' else 0, res_C=C if dgrad_merge == 'res' else 0), fp=dict(gscale=1 / 0.7),
         flags=(ir.F_ACCUM if dgrad_merge == 'acc' else 0) | (ir.F_MASK_RECOMPUTE if xmask else 0))
    b.op(ir.OP_CONV_WGRAD, buf=dict(x=x, dy=dy, dw=dw, ws=ws), dim=dict(gm))
    b.ws_need.append(('wgrad', gm))
    plan = b.plan(fp32)
    plan.meta['dtype'] = dtype
    plan.slot_of['ws'] = ws

    sc = (3.0 / (C * k * k)) ** 0.5
    xv = _round(fill((N, Hh, W, C), 2), dtype)
    wv = _round(fill((K, k, k, C), 1, sc), dtype)            # the master is fp32; rounded so both sides multiply the same values
    dyv = _round(fill((N, P, Q, K), 3), dtype)
    resv = _round(fill((N, P, Q, K), 4), dtype)
    bxv = _round(fill((N, Hh, W, C), 5), dtype)
    bmv = _round(fill((N, Hh, W, C), 6), dtype)
    dresv = _round(fill((N, Hh, W, C), 7), dtype)
    cf = torch.from_numpy(np.stack([fill((C,), 71, 0.2, 1.0), fill((C,), 72, 0.1), fill((C,), 73, 0.3), fill((C,), 74, 0.2, 1.0)]))
    if xmask:
        # the sign of the kernel's fp32 fma(x, scale, shift) is the sign of the exact value (no underflow at these magnitudes), which float64 holds exactly;
        # scale of mixed sign so both comparison directions occur
        cf[0] = cf[0] * torch.where(torch.arange(C) % 3 == 0, -1.0, 1.0).to(cf.dtype)
        cf32 = cf.float()
        bmv = ((bxv.double() * cf32[0].double() + cf32[1].double()) > 0).to(bxv.dtype)

    L.rn_set_variant2(variant2)                             # (before the engine: workspace sizes and deferral marks follow the kernel choice)
    try:
        eng = Engine(plan, h.DEV, TORCH_DT[dtype])
    except Exception:
        L.rn_set_variant2(0)
        raise
    feed = dict(x=xv, w=wv, dy=dyv, res=resv, bx=bxv, bm=bmv, coef=cf, dres=dresv)
    if dgrad_merge == 'acc':
        feed['dx'] = dresv                                   # the destination already holds the other branch's gradient
    for name, v in feed.items():
        t = eng.tensors[plan.slot_of[name]]
        t.copy_(v.reshape(t.shape).to(t.dtype))
    eng.bind({})
    L.rn_set_variant(variant)
    L.rn_set_variant2(variant2)
    try:
        L.rn_kernel_log(1)
        eng.run(0, len(plan.ops), 0)
        torch.cuda.synchronize()
        ran = L.rn_kernel_log_read().decode().split(',')
        L.rn_kernel_log(0)
        if expect_same_names:
            want = []
            flags = [1 | (2 if fwd_res else 0), 1 | {'none': 0, 'res': 2, 'acc': 4}[dgrad_merge] | (16 if xmask else 0), 0]
            for ps in range(3):
                want += _lib.conv_kernel_names(ps, RN_DT[dtype], gm, fused_epilogue=flags[ps])
            assert ran == want, (ran, want)
    finally:
        L.rn_set_variant(0)
        L.rn_set_variant2(0)
        L.rn_kernel_log(0)
    out = {n: eng.tensors[plan.slot_of[n]].detach().float().cpu() for n in ('y', 'dx', 'dw', 'st', 'dp')}

    # ---- torch-CPU reference on the same (pre-rounded) operands, fp32 ----
    xn, wn, dyn = _nchw(xv), wv.permute(0, 3, 1, 2).contiguous(), _nchw(dyv)
    y_ref = _nhwc(F.conv2d(xn, wn, None, s, p)) + (resv if fwd_res else 0)
    dx_ref = _nhwc(torch.nn.grad.conv2d_input(xn.shape, wn, dyn, s, p)) + (dresv if dgrad_merge != 'none' else 0)
    dw_ref = torch.nn.grad.conv2d_weight(xn, wn.shape, dyn, s, p).permute(0, 2, 3, 1)
    tol = TOL[dtype]

    def close(a, r, name, t=tol):
        err = float((a - r).abs().max() / r.abs().max().clamp_min(1e-30))
        assert err < t, (name, err, g, dtype, ran)
    close(out['y'], y_ref, 'y')
    close(out['dx'], dx_ref, 'dx')
    close(out['dw'].reshape(dw_ref.shape), dw_ref, 'dw', max(tol, 2e-5) if fp32 else 2e-4)     # dw is fp32 on every engine
    # fused sums are taken over the STORED (rounded) tensors; only the total over partial rows is specified
    ys = out['y'].double().reshape(-1, K)
    s0, s1 = out['st'].double().sum(0)
    assert float((s0 - ys.sum(0)).abs().max()) < 2e-5 * float(ys.abs().sum(0).max()), 'stats sum'
    assert float((s1 - (ys * ys).sum(0)).abs().max()) < 2e-5 * float((ys * ys).sum(0).max()), 'stats sumsq'
    gd = out['dx'].double() * (1 / 0.7) * (bmv.double() > 0)
    xh = (bxv.double() - cf[2].double()) * cf[3].double()
    d0, d1 = out['dp'].double().sum(0)
    scale = float(gd.abs().reshape(-1, C).sum(0).max())
    assert float((d0 - gd.reshape(-1, C).sum(0)).abs().max()) < 3e-5 * scale, 'bn-backward sum g'
    assert float((d1 - (gd * xh).reshape(-1, C).sum(0)).abs().max()) < 3e-5 * scale * float(xh.abs().max()), 'bn-backward sum g*xhat'
    return ran


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('g', PROD_GEOMS)
def test_production_tiles(g, dtype):
    run_conv_case(g, dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('g', [(3, 8, 8, 160, 160, 3, 1, 1), (2, 8, 8, 64, 128, 1, 1, 0), (2, 8, 8, 32, 64, 3, 2, 1), (4, 16, 16, 16, 32, 3, 1, 1), (128, 32, 32, 16, 64, 1, 1, 0),
                               (128, 32, 32, 64, 16, 1, 1, 0), (128, 16, 16, 64, 64, 3, 2, 1)])
def test_mask_from_x_on_the_128_row_kernels(g, dtype):
    """the same promise (rn_conv_epilogue.mask_from_x) reaches the kernels of the CIFAR nets too (the lowering sets it whatever kernel will run): they keep
    reading the mask tensor -- computing it there was built and measured slower -- and the flagged launch must give the same sums:
    every tile family (thin ResNet-v2-164 / ResNet-20 shapes at full batch, a stride-2 layer's parity classes), scales of both signs."""
    ran = run_conv_case(g, dtype, xmask=True, expect_same_names=False)
    assert not any(n.startswith('igemm8') for n in ran), ran


SMALL = [(3, 8, 8, 160, 160, 3, 1, 1), (2, 8, 8, 64, 128, 1, 1, 0), (1, 16, 16, 96, 96, 3, 1, 1), (2, 8, 8, 32, 64, 3, 2, 1), (4, 16, 16, 16, 32, 3, 1, 1)]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('variant', [128, 512])
@pytest.mark.parametrize('g', SMALL)
def test_both_schedules_on_small_tiles(g, variant, dtype):
    """variant 128 forces the wave-specialised kernel, 512 forbids it: every tile shape under both schedules."""
    ran = run_conv_case(g, dtype, variant=variant, expect_same_names=False)
    assert all(('igemm_ws' in n) == (variant == 128) for n in ran if n.startswith('igemm_') and '256x32' not in n), ran


# stride-2 layers: the data gradient is four parity classes (1 / 2 / 2 / 4 taps of a 3 x 3 kernel, 1 / 0 / 0 / 0 of a 1 x 1), launched as ONE grid over a
# table of four argument records (igemm_dma_classes_kernel) or -- rn_set_variant 1 << 18 -- one launch per class.  Column tiles 160 / 128 / 96 / 64 / 32,
# odd map sizes (classes of unequal size, a class with a single row), every operand set of the block backward
STRIDED = [(3, 8, 8, 320, 160, 3, 2, 1), (2, 16, 16, 128, 128, 3, 2, 1), (2, 9, 7, 96, 64, 3, 2, 1), (4, 16, 16, 128, 64, 1, 2, 0), (5, 15, 17, 32, 32, 3, 2, 1),
           (2, 8, 8, 64, 16, 3, 2, 1), (128, 32, 32, 160, 320, 3, 2, 1)]
CLASSES_OFF = 1 << 18


@pytest.mark.parametrize('dtype', ['fp32', 'fp16'])
@pytest.mark.parametrize('merge', ['none', 'res', 'acc'])
@pytest.mark.parametrize('variant', [0, CLASSES_OFF])
@pytest.mark.parametrize('g', STRIDED)
def test_stride2_parity_classes_as_one_grid_and_one_by_one(g, variant, merge, dtype):
    """both launch forms of a stride-2 data gradient against torch-CPU: dx of all four classes, the fused BatchNorm-backward sums (their rows are
    per-class), with a shortcut gradient added / accumulated into an existing gradient; the weight gradient and forward ride along."""
    if g[0] == 128 and (dtype == 'fp32' or merge != 'none'):
        pytest.skip('the production-size case once per 16-bit type')
    ran = run_conv_case(g, dtype, variant=variant, fwd_res=False, dgrad_merge=merge)
    dgrad = [n for n in ran if n.startswith('igemm_')][1:]                     # (the first name is the forward's)
    assert len(dgrad) >= 1 and all(n.startswith(('igemm_dma<128x', 'igemm_ws<128x')) for n in dgrad), ran
    if g[5] == 1:                                                              # a 1 x 1 kernel: ONE class has a tap, the other three are one launch of the tap-less pass
        assert len(dgrad) == 1 and ran.count('dgrad_notap') == 1, ran
    elif variant == 0:
        assert 'dgrad_notap' not in ran, ran
        assert all(n.startswith('igemm_dma<128x') for n in dgrad), ran          # one grid: the two-workgroups-per-CU kernel, one name per class


PATCH_SMALL = [
    (3, 8, 8, 160, 160, 3, 1, 1),        # several images per tile, the last tile partly empty: image-validity of the patch DMAs, row tail
    (5, 16, 16, 128, 128, 3, 1, 1),      # BN = 128 (forward and dgrad)
    (2, 32, 32, 160, 160, 3, 1, 1),      # whole image rows per tile; 160 channels = 5 chunks of 32 / 2.5 chunks of 64
    (1, 16, 16, 320, 320, 3, 1, 1),      # two column tiles
    (9, 8, 8, 128, 128, 3, 1, 1),        # 9 images of 8x8: tile tail, BN = 128
]
PATCH = 16 | (1 << 21)               # rn_set_variant: 16 lifts the minimum-grid rule, 1 << 21 takes the LDS-patch kernel wherever the geometry allows


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('g', PATCH_SMALL)
def test_patch_kernel_on_small_geometries(g, dtype):
    """the LDS-patch 3x3 kernel (16x16x32 MFMA tiles), forward AND data gradient, on shapes that exercise its tails."""
    ran = run_conv_case(g, dtype, variant=PATCH, expect_same_names=False)
    assert sum(n.startswith('igemm_patch<128x') for n in ran) == 2, ran          # forward and dgrad both took it


@pytest.mark.parametrize('g', [(4, 16, 16, 72, 160, 3, 1, 1), (4, 16, 16, 96, 160, 3, 1, 1), (4, 16, 16, 80, 160, 3, 1, 1)])
def test_patch_kernel_channel_tails(g):
    """forward only reaches the patch kernel here (the data gradient has 72 / 96 / 80 output channels): channel counts that end inside
    a chunk (a 32-channel k-step half empty, a chunk half empty)."""
    ran = run_conv_case(g, 'fp16', variant=PATCH, expect_same_names=False)
    assert ran[0].startswith('igemm_patch<128x'), ran


def test_production_set_reaches_every_instantiation():
    """sanity of the list itself: the names it selects include both schedules and the wide tiles."""
    names = set()
    for g in PROD_GEOMS:
        for ps in range(3):
            names.update(_lib.conv_kernel_names(ps, ir.RN_BF16, geom(*resolve(g, False)), True))
    for need in ('igemm_dma<128x160>', 'igemm_ws<128x160>', 'igemm_dma<128x128>', 'igemm_dma<256x32>', 'wgrad<160x160>', 'wgrad<128x128>',
                 'wgrad_reduce', 'wgrad_reduce_wide'):
        assert need in names, (need, sorted(names))


@pytest.mark.parametrize('g', [PROD_GEOMS[0], PROD_GEOMS[3]])
def test_patch_kernel_on_production_shapes(g):
    """WRN-28-10 stage 1 / stage 2 at batch 128 take the patch kernel (grids of 512-1024 workgroups, XCD remap, two column tiles) when the row-patch
    256 x 160 kernel of round 4 is switched off (rn_set_variant2 1): the fallback stays parity-tested at size."""
    ran = run_conv_case(g, 'fp16', expect_same_names=False, variant2=1)
    assert sum(n.startswith('igemm_patch<128x') for n in ran) == 2, ran


def _one_op_engine(kind, g, bufs_shapes, dtype=torch.float16):
    import gpu_harness as h
    from pytorch_ddp_resnet_amd.engine.executor import Engine
    gm = geom(*g)
    b = h.PlanBuilder()
    slots = {name: b.slot(name, shape, dt) for name, (shape, dt) in bufs_shapes.items()}
    b.op(kind, buf={k: v for k, v in slots.items()}, dim=dict(gm, res_mode=0, res_C=0) if kind != ir.OP_CONV_WGRAD else dict(gm))
    if kind == ir.OP_CONV_WGRAD:
        b.ws_need.append(('wgrad', gm))
    plan = b.plan(False)
    if kind == ir.OP_CONV_WGRAD:
        plan.slot_of['ws'] = slots['ws']
    return Engine(plan, h.DEV, dtype), slots


@pytest.mark.parametrize('g', [(4, 16, 16, 64, 160, 3, 1, 1), (2, 32, 32, 160, 160, 3, 1, 1), (3, 8, 8, 96, 128, 3, 1, 1)])
def test_patch_kernel_exact_integers(g):
    """integer-valued operands: every product and partial sum is exact in fp32, so the convolution must equal the reference BIT FOR BIT
    (catches a wrong lane <-> pixel / channel map of the 16x16x32 fragments or a mis-shifted tap, which a tolerance could blur)."""
    N, Hh, W, C, K, k, s_, p = g
    eng, sl = _one_op_engine(ir.OP_CONV_FWD, g, dict(x=((N, Hh, W, C), 'T'), w_fwd=((K, 9, C), 'T'), y=((N, Hh, W, K), 'T')))
    rng = np.random.RandomState(1)
    xv = rng.randint(-2, 3, size=(N, Hh, W, C)).astype(np.float32)
    wv = rng.randint(-1, 2, size=(K, 3, 3, C)).astype(np.float32)
    eng.tensors[sl['x']].copy_(torch.from_numpy(xv).to(torch.float16))
    eng.tensors[sl['w_fwd']].copy_(torch.from_numpy(wv).reshape(K, 9, C).to(torch.float16))
    eng.bind({})
    L = _lib.lib()
    L.rn_set_variant(PATCH)
    try:
        L.rn_kernel_log(1)
        eng.run(0, 1, 0)
        torch.cuda.synchronize()
        assert 'igemm_patch<128x' in L.rn_kernel_log_read().decode()
    finally:
        L.rn_kernel_log(0)
        L.rn_set_variant(0)
    ref = torch.nn.functional.conv2d(_nchw(torch.from_numpy(xv)), torch.from_numpy(wv).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert float(ref.abs().max()) < 2048                     # representable in fp16: the stored output is exact too
    assert torch.equal(eng.tensors[sl['y']].float().cpu(), ref.contiguous())


def test_wgrad_exact_integers():
    """the weight gradient on integer-valued operands equals the reference bit for bit (transposed fragment reads, tap shifts, split-K
    slabs summed in fixed order)."""
    g = (4, 16, 16, 64, 64, 3, 1, 1)
    N, Hh, W, C, K, k, s_, p = g
    eng, sl = _one_op_engine(ir.OP_CONV_WGRAD, g, dict(x=((N, Hh, W, C), 'T'), dy=((N, Hh, W, K), 'T'), dw=((K, 3, 3, C), 'f32'), ws=((0,), 'u8')))
    rng = np.random.RandomState(0)
    xv = rng.randint(-3, 4, size=(N, Hh, W, C)).astype(np.float32)
    dv = rng.randint(-2, 3, size=(N, Hh, W, K)).astype(np.float32)
    eng.tensors[sl['x']].copy_(torch.from_numpy(xv).to(torch.float16)); eng.tensors[sl['dy']].copy_(torch.from_numpy(dv).to(torch.float16))
    eng.bind({})
    eng.run(0, 1, 0)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(_nchw(torch.from_numpy(xv)), (K, C, 3, 3), _nchw(torch.from_numpy(dv)), 1, 1).permute(0, 2, 3, 1)
    assert torch.equal(eng.tensors[sl['dw']].cpu(), ref.contiguous())


IGEMM8 = 1 << 22                     # rn_set_variant: the eight-phase 256-row kernel wherever the geometry allows
IGEMM8_SMALL = [
    (2, 16, 16, 256, 256, 3, 1, 1),      # two row tiles, forward and data gradient, 36 K tiles
    (2, 16, 16, 64, 256, 3, 1, 1),       # one 64-channel chunk per tap (forward only: the data gradient has 64 output channels)
    (3, 14, 14, 256, 512, 1, 1, 0),      # 1x1, two column tiles, row tail (588 = 2 x 256 + 76), 4 K tiles
    (2, 8, 8, 64, 256, 1, 1, 0),         # ONE K tile (prologue only)
    (2, 8, 8, 128, 256, 1, 1, 0),        # two K tiles (no steady-state iteration)
    (2, 8, 8, 192, 256, 1, 1, 0),        # three K tiles (one steady-state iteration)
    (2, 28, 28, 256, 256, 3, 2, 1),      # stride 2: forward on the 14 x 14 grid, data gradient as four parity classes
    (2, 14, 14, 512, 256, 1, 2, 0),      # stride-2 projection shortcut
    (5, 7, 7, 64, 256, 3, 1, 1),         # images of 49 pixels straddle the 256-row tile; padding taps on every side
    (1, 7, 7, 128, 512, 3, 1, 1),        # a single, mostly empty row tile
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('g', IGEMM8_SMALL)
def test_igemm8_on_small_geometries(g, dtype):
    """the eight-phase kernel (conv_igemm8.hip) forced onto small shapes that exercise its prologue / tail modes, row tails, strides and
    parity classes; fused epilogues as in every case of this file."""
    ran = run_conv_case(g, dtype, variant=IGEMM8)
    assert ran[0] == 'igemm8<256x256:res>', ran
    if g[3] % 256 == 0 and not (g[5] == 1 and g[6] == 2):          # (a stride-2 1x1 layer has three parity classes without any tap: plain zero-fill launches)
        assert all(n.startswith('igemm8<256x256:') for n in ran if n.startswith('igemm')), ran


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('merge', ['none', 'res', 'acc'])
@pytest.mark.parametrize('g', [(3, 16, 16, 128, 128, 3, 1, 1), (3, 14, 14, 256, 384, 1, 1, 0), (2, 28, 28, 128, 128, 3, 2, 1)])
def test_igemm8_column_tiles_of_128(g, merge, dtype):
    """output channels that are a multiple of 128 but not of 256 take 256 x 128 tiles (4 x 2 waves of 64 x 64; two waves share a BatchNorm statistics
    row and combine their sums through LDS): forward and data gradient, every operand set, strided parity classes through the general epilogue."""
    ran = run_conv_case(g, dtype, variant=IGEMM8, fwd_res=(merge == 'none'), dgrad_merge=merge)
    assert ran[0].startswith('igemm8<256x128:'), ran
    if g[3] % 256:
        assert all(n.startswith('igemm8<256x128:') for n in ran if n.startswith('igemm')), ran


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('merge', ['none', 'res', 'acc'])
@pytest.mark.parametrize('g', [(2, 16, 16, 256, 256, 3, 1, 1), (3, 14, 14, 256, 512, 1, 1, 0)])
def test_igemm8_epilogue_specialisations(g, merge, dtype):
    """every operand set the eight-phase kernel is specialised for: forward without a residual (plain) and the three data-gradient forms
    (BatchNorm-backward sums alone, + shortcut gradient as residual, + accumulate into dx)."""
    ran = run_conv_case(g, dtype, variant=IGEMM8, fwd_res=False, dgrad_merge=merge)
    assert ran[0] == 'igemm8<256x256:plain>', ran
    assert ran[1] == {'none': 'igemm8<256x256:bnb>', 'res': 'igemm8<256x256:bnb+res>', 'acc': 'igemm8<256x256:bnb+acc>'}[merge], ran


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('g,name', [((2, 16, 16, 256, 256, 3, 1, 1), '256x256:bnb/xmask'), ((3, 14, 14, 512, 256, 1, 1, 0), '256x256:bnb/xmask'),
                                    ((3, 16, 16, 128, 128, 3, 1, 1), '256x128:bnb/xmask'), ((2, 28, 28, 256, 256, 3, 2, 1), '256x256:bnb/xmask/s2'),
                                    ((5, 15, 15, 256, 128, 3, 1, 1), '256x256:bnb/xmask')])
def test_igemm8_mask_from_x(g, name, dtype):
    """the BatchNorm-backward sums with the ReLU mask computed from bn_x (a plain BatchNorm + ReLU in front of the convolution: the lowering promises it
    with F_MASK_RECOMPUTE): the epilogue reads x only; scales of both signs; dense and strided (parity class) destinations, both column tiles, a row tail."""
    ran = run_conv_case(g, dtype, variant=IGEMM8, fwd_res=False, dgrad_merge='none', xmask=True)
    assert all(n == f'igemm8<{name}>' for n in ran[1:-1] if n.startswith('igemm8')) and any(n == f'igemm8<{name}>' for n in ran), ran


@pytest.mark.parametrize('ops,dtype', [((True, 'none'), 'fp16'), ((True, 'none'), 'bf16'), ((False, 'res'), 'fp16'), ((False, 'acc'), 'fp16'), ((False, 'none', True), 'fp16')])
@pytest.mark.parametrize('g', IGEMM8_GEOMS)
def test_igemm8_production_operand_sets(g, ops, dtype):
    """the eight-phase kernels at production grids (>= 160 persistent tiles, the shipped selection rule), with every operand set the full-batch
    WRN-50-2 configuration launches them with: forward +- identity residual; data gradient with the BatchNorm-backward sums alone, with the
    shortcut gradient as residual, accumulating into dx.  The kernel that ran is asserted against the launchers' own choice."""
    xm = len(ops) > 2 and ops[2]
    ran = run_conv_case(g, dtype, fwd_res=ops[0], dgrad_merge=ops[1], xmask=xm)
    bn_f, bn_d = (256 if g[4] % 256 == 0 else 128), (256 if g[3] % 256 == 0 else 128)
    assert ran[0] == f'igemm8<256x{bn_f}:' + ('res>' if ops[0] else 'plain>'), ran
    if g[6] == 1:
        assert ran[1] == f'igemm8<256x{bn_d}:' + {'none': 'bnb/xmask>' if xm else 'bnb>', 'res': 'bnb+res>', 'acc': 'bnb+acc>'}[ops[1]], ran


WGRAD8 = 1 << 30                     # rn_set_variant: the eight-phase weight-gradient kernel at any size (its rule wants >= 8 K tiles per workgroup)
WGRAD8_SMALL = [
    (2, 16, 16, 256, 256, 3, 1, 1),      # 9 taps x one 256 x 256 tile, 8 K tiles each: padding taps, cuts between and inside tiles
    (4, 14, 14, 256, 512, 1, 1, 0),      # dense 1x1, two output-channel tiles, pixel tail (784 = 12 x 64 + 16)
    (2, 28, 28, 256, 256, 3, 2, 1),      # stride 2
    (2, 14, 14, 512, 256, 1, 2, 0),      # stride-2 1x1 (projection shortcut), two input-channel tiles
    (1, 7, 7, 256, 256, 3, 1, 1),        # ONE K tile per tile (49 pixels)
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('g', WGRAD8_SMALL)
def test_wgrad8_on_small_geometries(g, dtype):
    """the eight-phase weight-gradient kernel (conv_wgrad8.hip): transposed fragment reads, per-K-tile pixel decode, pixel splits into slabs."""
    ran = run_conv_case(g, dtype, variant=WGRAD8 | IGEMM8)
    assert 'wgrad8<256x256>' in ran, ran


def test_wgrad8_exact_integers_and_accumulate():
    """integer operands: bit for bit; a second launch with RN_F_ACCUM doubles the gradient exactly."""
    g = (2, 16, 16, 256, 256, 3, 1, 1)
    N, Hh, W, C, K, k, s_, p = g
    eng, sl = _one_op_engine(ir.OP_CONV_WGRAD, g, dict(x=((N, Hh, W, C), 'T'), dy=((N, Hh, W, K), 'T'), dw=((K, 3, 3, C), 'f32'), ws=((0,), 'u8')))
    rng = np.random.RandomState(0)
    xv = rng.randint(-3, 4, size=(N, Hh, W, C)).astype(np.float32)
    dv = rng.randint(-2, 3, size=(N, Hh, W, K)).astype(np.float32)
    eng.tensors[sl['x']].copy_(torch.from_numpy(xv).to(torch.float16)); eng.tensors[sl['dy']].copy_(torch.from_numpy(dv).to(torch.float16))
    eng.bind({})
    L = _lib.lib()
    L.rn_set_variant(WGRAD8)
    try:
        L.rn_kernel_log(1)
        eng.run(0, 1, 0)
        torch.cuda.synchronize()
        assert 'wgrad8<256x256>' in L.rn_kernel_log_read().decode()
    finally:
        L.rn_kernel_log(0)
        L.rn_set_variant(0)
    ref = torch.nn.grad.conv2d_weight(_nchw(torch.from_numpy(xv)), (K, C, 3, 3), _nchw(torch.from_numpy(dv)), 1, 1).permute(0, 2, 3, 1)
    assert torch.equal(eng.tensors[sl['dw']].cpu(), ref.contiguous())


STREAMK_ANY = -(1 << 31)             # rn_set_variant bit 31 (as a C int): stream-K on every grid that is no multiple of the CU count
STREAMK = [
    (29, 16, 16, 256, 256, 3, 1, 1),     # 29 tiles x 36 K tiles over 256 workgroups: every tile cut into ~9 parts (multi-part ticketed sums)
    (37, 32, 32, 256, 512, 1, 1, 0),     # 148 x 2 = 296 tiles: 256 + 40 tiles as stream-K units of a 4-K-tile reduction, cuts inside and between tiles
    (128, 14, 14, 512, 512, 3, 1, 1),    # the production case: 196 tiles of 72 K tiles, 55 units per workgroup
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('g', STREAMK)
def test_igemm8_stream_k(g, dtype):
    """grids that are no multiple of the CU count: the last rounds are cut into K-tile units (stream-K) and cut tiles are summed through the workspace by
    the last-arriving workgroup, in slice order.  Results against the reference, AND bit-identical between two runs (the sum order is fixed)."""
    import gpu_harness as h                               # noqa: F401  (engines set the workspace)
    ran = run_conv_case(g, dtype, variant=IGEMM8 | STREAMK_ANY)
    assert ran[0].startswith('igemm8<256x256:'), ran


def test_igemm8_stream_k_is_reproducible_and_equals_whole_tiles():
    """same convolution three times: stream-K twice (bit-identical: parts are added in slice order whoever arrives last) and with whole tiles only
    (rn_set_variant 1 << 28): equal to fp32 rounding of the differently ordered K sums."""
    g = (29, 16, 16, 256, 256, 3, 1, 1)
    N, Hh, W, C, K, k, s_, p = g
    eng, sl = _one_op_engine(ir.OP_CONV_FWD, g, dict(x=((N, Hh, W, C), 'T'), w_fwd=((K, k * k, C), 'T'), y=((N, Hh, W, K), 'T')))
    rng = np.random.RandomState(3)
    eng.tensors[sl['x']].copy_(torch.from_numpy(rng.randn(N, Hh, W, C).astype(np.float32)).to(torch.float16))
    eng.tensors[sl['w_fwd']].copy_(torch.from_numpy((rng.randn(K, k * k, C) * 0.02).astype(np.float32)).to(torch.float16))
    eng.bind({})
    L = _lib.lib()
    outs = []
    try:
        for v in (IGEMM8 | STREAMK_ANY, IGEMM8 | STREAMK_ANY, IGEMM8 | (1 << 28)):
            L.rn_set_variant(v)
            eng.tensors[sl['y']].zero_()
            eng.run(0, 1, 0)
            torch.cuda.synchronize()
            outs.append(eng.tensors[sl['y']].float().cpu().clone())
    finally:
        L.rn_set_variant(0)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - outs[2]).abs().max()) <= 2e-3 * float(outs[2].abs().max())
    assert not torch.equal(outs[2], torch.zeros_like(outs[2]))


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('g,variant', [(STEM8_GEOMS[0], 0), (STEM8_GEOMS[1], 0), ((3, 40, 56, 8, 256, 7, 2, 3), IGEMM8), ((2, 32, 32, 8, 256, 3, 1, 1), IGEMM8),
                                       ((1, 19, 23, 8, 512, 5, 2, 2), IGEMM8)])
def test_igemm8_stem(g, variant, dtype):
    """the top-level Conv2d(3 -> K, bias=True) (resnet.py:69-75) on the eight-phase kernel: one 16-byte chunk per pixel, a K tile = eight taps, the
    tap walked per lane (tap = 8 g + chunk), taps beyond the last and out-of-image taps zero-filled; bias and BatchNorm statistics in the epilogue.
    The two ImageNet stems at the shipped selection rule, and small 7x7 / 3x3 / 5x5 shapes (odd sizes, one or several K tiles) forced."""
    import gpu_harness as h
    from pytorch_ddp_resnet_amd.engine.executor import Engine
    g = resolve(g, False)
    N, Hh, W, C, K, k, s_, p = g
    gm = geom(*g)
    P, Q = gm['P'], gm['Q']
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); wf = b.slot('wf', (K, k * k, C)); y = b.slot('y', (N, P, Q, K))
    st = b.slot('st', (conv_stats_rows(gm), 2, K), 'f32'); bias = b.slot('bias', (K,), 'f32')
    b.op(ir.OP_CONV_FWD, buf=dict(x=x, w_fwd=wf, y=y, res=-1, stats=st, bias=bias), dim=dict(gm, res_mode=0, res_C=0))
    plan = b.plan(False)
    plan.meta['dtype'] = dtype
    xv = _round(fill((N, Hh, W, C), 2), dtype)
    xv[..., 3:] = 0                                              # the padded channels of the image are zero
    wv = _round(fill((K, k, k, C), 1, (3.0 / (3 * k * k)) ** 0.5), dtype)
    bv = torch.from_numpy(fill((K,), 9, 0.5))
    eng = Engine(plan, h.DEV, TORCH_DT[dtype])
    for name, v in dict(x=xv, wf=wv.reshape(K, k * k, C), bias=bv).items():
        t = eng.tensors[plan.slot_of[name]]
        t.copy_(v.reshape(t.shape).to(t.dtype))
    eng.bind({})
    L = _lib.lib()
    L.rn_set_variant(variant)
    try:
        L.rn_kernel_log(1)
        eng.run(0, 1, 0)
        torch.cuda.synchronize()
        ran = L.rn_kernel_log_read().decode().split(',')
        want = _lib.conv_kernel_names(0, RN_DT[dtype], gm, fused_epilogue=9)
    finally:
        L.rn_kernel_log(0)
        L.rn_set_variant(0)
    assert ran == want == ['igemm8<256x256:stem+bias>'], (ran, want)
    yo = eng.tensors[plan.slot_of['y']].float().cpu()
    ref = _nhwc(F.conv2d(_nchw(xv), wv.permute(0, 3, 1, 2).contiguous(), bv, s_, p))
    err = float((yo - ref).abs().max() / ref.abs().max())
    assert err < TOL[dtype], (err, g, dtype)
    ys = yo.double().reshape(-1, K)
    s0, s1 = eng.tensors[plan.slot_of['st']].double().cpu().sum(0)
    assert float((s0 - ys.sum(0)).abs().max()) < 2e-5 * float(ys.abs().sum(0).max())
    assert float((s1 - (ys * ys).sum(0)).abs().max()) < 2e-5 * float((ys * ys).sum(0).max())


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('shape,variant,fwd_name', [((8, 224, 224, 512), 0, 'igemm8<256x256:bias/rows>'), ((8, 224, 224, 256), 0, 'igemm8<256x256:bias/rows>'),
                                                    ((3, 40, 56, 256), IGEMM8, 'igemm8<256x256:bias/rows>'), ((2, 36, 20, 128), IGEMM8, 'igemm8<256x128:bias/rows>'),
                                                    ((2, 32, 32, 64), 0, None), ((5, 20, 28, 16), 0, None)])
def test_s2d_stem(shape, variant, fwd_name, dtype):
    """the ImageNet stem Conv2d(3 -> K, 7 x 7, stride 2, padding 3, bias) (resnet.py:69-75) as the 16-bit engines lower it: image -> space-to-depth layout
    (rn_img_to_s2d), weights -> [K][4][4][16] (rn_pack_stem_w_s2d), a 4 x 4 / stride-1 / VALID convolution (the eight-phase kernel's row-segment form at the
    shipped rule and forced on small shapes -- a kernel row = one K tile of 128 contiguous bytes; the 128-row kernels otherwise), its weight gradient (im2col
    columns = (tap, chunk)) mapped back to [K][7][7][3] (rn_unpack_stem_dw_s2d).  Against torch's 7 x 7 convolution and its weight gradient on the same
    rounded operands; bias and BatchNorm statistics from the epilogue."""
    import gpu_harness as h
    from pytorch_ddp_resnet_amd.engine.executor import Engine
    N, Hh, W, K = shape
    b = h.PlanBuilder()
    x = b.slot('x', (N, 3, Hh, W), 'f32'); w = b.slot('w', (K, 7, 7, 3), 'f32'); bias = b.slot('bias', (K,), 'f32')
    xs = b.slot('xs', (N, Hh // 2 + 3, W // 2 + 3, 16)); wsd = b.slot('wsd', (K, 16, 16))
    gm = geom(N, Hh // 2 + 3, W // 2 + 3, 16, K, 4, 1, 0)
    P, Q = gm['P'], gm['Q']
    assert (P, Q) == (Hh // 2, W // 2)
    y = b.slot('y', (N, P, Q, K)); st = b.slot('st', (conv_stats_rows(gm), 2, K), 'f32'); dy = b.slot('dy', (N, P, Q, K))
    dws = b.slot('dws', (K, 16, 16), 'f32'); dw = b.slot('dw', (K, 7, 7, 3), 'f32'); ws = b.slot('workspace', (0,), 'u8')
    b.op(ir.OP_IMG_TO_S2D, buf=dict(x=x, out=xs), dim=dict(N=N, C=3, H=Hh, W=W))
    b.op(ir.OP_PACK_STEM_W_S2D, buf=dict(w=w, w_s2d=wsd), dim=dict(K=K, C=3))
    b.op(ir.OP_CONV_FWD, buf=dict(x=xs, w_fwd=wsd, y=y, res=-1, stats=st, bias=bias), dim=dict(gm, res_mode=0, res_C=0))
    b.op(ir.OP_CONV_WGRAD, buf=dict(x=xs, dy=dy, dw=dws, ws=ws), dim=dict(gm))
    b.op(ir.OP_UNPACK_STEM_DW_S2D, buf=dict(dw_s2d=dws, dw=dw), dim=dict(K=K, C=3))
    b.ws_need.append(('wgrad', gm))
    plan = b.plan(False)
    plan.meta['dtype'] = dtype
    plan.slot_of['ws'] = ws
    xv = _round(fill((N, 3, Hh, W), 2), dtype)                    # pre-rounded: both sides multiply the same values
    wv = _round(fill((K, 7, 7, 3), 1, (3.0 / 147) ** 0.5), dtype)
    bv = torch.from_numpy(fill((K,), 9, 0.5))
    dyv = _round(fill((N, P, Q, K), 3), dtype)
    eng = Engine(plan, h.DEV, TORCH_DT[dtype])
    for name, v in dict(x=xv, w=wv, bias=bv, dy=dyv).items():
        t = eng.tensors[plan.slot_of[name]]
        t.copy_(v.reshape(t.shape).to(t.dtype))
    eng.bind({})
    L = _lib.lib()
    L.rn_set_variant(variant)
    try:
        L.rn_kernel_log(1)
        eng.run(0, len(plan.ops), 0)
        torch.cuda.synchronize()
        ran = L.rn_kernel_log_read().decode().split(',')
        want = _lib.conv_kernel_names(0, RN_DT[dtype], gm, fused_epilogue=9) + _lib.conv_kernel_names(2, RN_DT[dtype], gm)
    finally:
        L.rn_kernel_log(0)
        L.rn_set_variant(0)
    assert ran == want, (ran, want)
    if fwd_name:
        assert ran[0] == fwd_name, ran
    assert any(n.startswith('wgrad_im2col<') for n in ran), ran
    wn = wv.permute(0, 3, 1, 2).contiguous()
    yo = eng.tensors[plan.slot_of['y']].float().cpu()
    ref = _nhwc(F.conv2d(xv, wn, bv, 2, 3))
    err = float((yo - ref).abs().max() / ref.abs().max())
    assert err < TOL[dtype], ('y', err, shape, dtype)
    ys = yo.double().reshape(-1, K)
    s0, s1 = eng.tensors[plan.slot_of['st']].double().cpu().sum(0)
    assert float((s0 - ys.sum(0)).abs().max()) < 2e-5 * float(ys.abs().sum(0).max())
    assert float((s1 - (ys * ys).sum(0)).abs().max()) < 2e-5 * float((ys * ys).sum(0).max())
    dw_ref = torch.nn.grad.conv2d_weight(xv, wn.shape, _nchw(dyv), 2, 3).permute(0, 2, 3, 1)
    dwo = eng.tensors[plan.slot_of['dw']].float().cpu()
    errw = float((dwo - dw_ref).abs().max() / dw_ref.abs().max())
    assert errw < 2e-4, ('dw', errw, shape, dtype)
    # the structural zeros of the regrouped filter (tap row / column -1, channel 3 of every image pixel) carry no weight and receive no gradient back
    wsd_o = eng.tensors[plan.slot_of['wsd']].float().cpu().reshape(K, 4, 4, 2, 2, 4)
    assert float(wsd_o[..., 3].abs().max()) == 0 and float(wsd_o[:, 0, :, 0].abs().max()) == 0 and float(wsd_o[:, :, 0, :, 0].abs().max()) == 0


@pytest.mark.parametrize('g', [(2, 16, 16, 128, 256, 3, 1, 1), (3, 14, 14, 64, 256, 1, 1, 0)])
def test_igemm8_exact_integers(g):
    """integer operands: the eight-phase kernel must equal the reference bit for bit (fragment <-> pixel / channel maps, tap walk, stage toggling)."""
    N, Hh, W, C, K, k, s_, p = g
    eng, sl = _one_op_engine(ir.OP_CONV_FWD, g, dict(x=((N, Hh, W, C), 'T'), w_fwd=((K, k * k, C), 'T'), y=((N, Hh, W, K), 'T')))
    rng = np.random.RandomState(2)
    xv = rng.randint(-2, 3, size=(N, Hh, W, C)).astype(np.float32)
    wv = rng.randint(-1, 2, size=(K, k, k, C)).astype(np.float32)
    eng.tensors[sl['x']].copy_(torch.from_numpy(xv).to(torch.float16))
    eng.tensors[sl['w_fwd']].copy_(torch.from_numpy(wv).reshape(K, k * k, C).to(torch.float16))
    eng.bind({})
    L = _lib.lib()
    L.rn_set_variant(IGEMM8)
    try:
        L.rn_kernel_log(1)
        eng.run(0, 1, 0)
        torch.cuda.synchronize()
        assert 'igemm8<256x256:plain>' in L.rn_kernel_log_read().decode()
    finally:
        L.rn_kernel_log(0)
        L.rn_set_variant(0)
    ref = torch.nn.functional.conv2d(_nchw(torch.from_numpy(xv)), torch.from_numpy(wv).permute(0, 3, 1, 2), padding=p).permute(0, 2, 3, 1)
    assert float(ref.abs().max()) < 2048
    assert torch.equal(eng.tensors[sl['y']].float().cpu(), ref.contiguous())


# ---- the row-patch 256 x 160 kernel (conv_igemm8r.hip): 3x3 stride-1 layers with 160 n output channels, the WRN-28-10 family ----
R8_ANY = 2                           # rn_set_variant2: on any grid size
R8_SMALL = [
    (3, 8, 8, 160, 160, 3, 1, 1),        # W = 8: ten-row patches of four images per tile, ONE partly empty tile (192 rows); 15 items, the last group half empty
    (2, 32, 32, 160, 160, 3, 1, 1),      # W = 32: eight image rows per tile, 8 tiles
    (1, 16, 16, 320, 320, 3, 1, 1),      # W = 16: a tile = one image; two column tiles; 30 items = 15 full groups
    (5, 16, 16, 64, 160, 3, 1, 1),       # 64 input channels: 6 items = 3 groups (forward only: the data gradient has 64 output channels)
    (2, 16, 16, 96, 160, 3, 1, 1),       # 96 input channels: 9 items, 5 groups, kernel rows change inside a group
    (9, 8, 8, 160, 320, 3, 1, 1),        # 576 pixels = two whole tiles + a quarter; forward 2 column tiles
    (3, 24, 16, 160, 160, 3, 1, 1),      # H = 24: 16-row tiles straddle images (vertical padding inside a patch)
    (300, 8, 8, 160, 160, 3, 1, 1),      # 75 tiles on 75 workgroups: several per XCD class (the remap with a remainder)
    (70, 16, 16, 160, 320, 3, 1, 1),     # 140 tiles forward (two column tiles), 70 in the data gradient
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('merge', ['none', 'res', 'acc'])
@pytest.mark.parametrize('g', R8_SMALL)
def test_igemm8r_on_small_geometries(g, merge, dtype):
    """the row-patch kernel forced onto small shapes: every map width it takes (8 / 16 / 32), channel counts whose items end inside a group, tiles that
    straddle images, a partly empty tile, every operand set (forward +- residual; data gradient with the BatchNorm-backward sums alone / + shortcut
    gradient / accumulating), fused statistics rows shared by two waves."""
    if g[0] >= 70 and (dtype == 'bf16' or merge != 'none'):
        pytest.skip('the larger grids once')
    ran = run_conv_case(g, dtype, variant2=R8_ANY, fwd_res=(merge == 'none'), dgrad_merge=merge)
    assert ran[0] == 'igemm8r<256x160:' + ('res>' if merge == 'none' else 'plain>'), ran
    if g[3] % 160 == 0:
        assert ran[1] == 'igemm8r<256x160:' + {'none': 'bnb>', 'res': 'bnb+res>', 'acc': 'bnb+acc>'}[merge], ran


def test_igemm8r_more_tiles_than_workgroups():
    """600 tiles on 256 persistent workgroups: two to three tiles per workgroup (the next tile's prologue under the previous tile's epilogue), XCD remap."""
    ran = run_conv_case((300, 16, 16, 160, 320, 3, 1, 1), 'fp16', variant2=R8_ANY)
    assert ran[0] == 'igemm8r<256x160:res>' and ran[1] == 'igemm8r<256x160:bnb>', ran


@pytest.mark.parametrize('g', [(2, 32, 32, 160, 160, 3, 1, 1), (3, 8, 8, 160, 320, 3, 1, 1), (2, 16, 16, 96, 160, 3, 1, 1)])
def test_igemm8r_exact_integers(g):
    """integer-valued operands: the convolution equals the reference BIT FOR BIT (a wrong lane <-> pixel / channel map of the transposed products, of the
    fifth column tile's transpose over pixel tiles, a mis-paired item half or a mis-shifted tap cannot hide behind a tolerance)."""
    N, Hh, W, C, K, k, s_, p = g
    eng, sl = _one_op_engine(ir.OP_CONV_FWD, g, dict(x=((N, Hh, W, C), 'T'), w_fwd=((K, 9, C), 'T'), y=((N, Hh, W, K), 'T')))
    rng = np.random.RandomState(1)
    xv = rng.randint(-2, 3, size=(N, Hh, W, C)).astype(np.float32)
    wv = rng.randint(-1, 2, size=(K, 3, 3, C)).astype(np.float32)
    eng.tensors[sl['x']].copy_(torch.from_numpy(xv).to(torch.float16))
    eng.tensors[sl['w_fwd']].copy_(torch.from_numpy(wv).reshape(K, 9, C).to(torch.float16))
    eng.bind({})
    L = _lib.lib()
    L.rn_set_variant2(R8_ANY)
    try:
        L.rn_kernel_log(1)
        eng.run(0, 1, 0)
        torch.cuda.synchronize()
        assert 'igemm8r<256x160:plain>' in L.rn_kernel_log_read().decode()
    finally:
        L.rn_kernel_log(0)
        L.rn_set_variant2(0)
    ref = torch.nn.functional.conv2d(_nchw(torch.from_numpy(xv)), torch.from_numpy(wv).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    assert float(ref.abs().max()) < 2048                     # representable in fp16: the stored output is exact too
    assert torch.equal(eng.tensors[sl['y']].float().cpu(), ref.contiguous())


@pytest.mark.parametrize('ops,dtype', [((True, 'none'), 'fp16'), ((True, 'none'), 'bf16'), ((False, 'res'), 'fp16'), ((False, 'acc'), 'fp16')])
@pytest.mark.parametrize('g', R8_GEOMS)
def test_igemm8r_production_operand_sets(g, ops, dtype):
    """WRN-28-10's stage-1 / stage-2 layers at batch 128 by the SHIPPED rule (512 / 256 tiles on 256 persistent workgroups), with every operand set the
    block forward / backward launches them with (residual_block.py:67-99, pre-activation + dropout: conv1 plain + statistics, conv2 + identity residual;
    data gradients with the BatchNorm-backward sums alone, + the shortcut gradient, accumulating into dx)."""
    ran = run_conv_case(g, dtype, fwd_res=ops[0], dgrad_merge=ops[1])
    assert ran[0] == 'igemm8r<256x160:' + ('res>' if ops[0] else 'plain>'), ran
    assert ran[1] == 'igemm8r<256x160:' + {'none': 'bnb>', 'res': 'bnb+res>', 'acc': 'bnb+acc>'}[ops[1]], ran


R8_SPLIT_ANY = 262144                # rn_set_variant2: the two-halves form on grids of < 96 tiles too
R8_SPLIT_SMALL = [
    (2, 32, 32, 160, 160, 3, 1, 1),      # 8 groups: 4 + 4
    (1, 16, 16, 320, 320, 3, 1, 1),      # 15 groups: 8 + 7, the second half starts inside a kernel row
    (3, 8, 8, 640, 640, 3, 1, 1),        # the stage it ships on: 30 groups, a partly empty tile, four column tiles
    (2, 16, 16, 480, 160, 3, 1, 1),      # 45 items: 23 groups (12 + 11), the last group half empty
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('merge', ['none', 'res', 'acc'])
@pytest.mark.parametrize('g', R8_SPLIT_SMALL)
def test_igemm8r_split_reduction(g, merge, dtype):
    """the row-patch kernel with every tile's reduction cut into two work items (conv_igemm8r.hip, split form): group ranges that start inside a kernel row,
    the hand-off through the stream-K workspace (write-through parts, ticket, the second arriver adds both parts in slice order), every operand set."""
    if dtype == 'bf16' and merge != 'none':
        pytest.skip('the merged operand sets once (fp16): the suite runs close to a quarter of an hour')
    ran = run_conv_case(g, dtype, variant2=R8_ANY | R8_SPLIT_ANY, fwd_res=(merge == 'none'), dgrad_merge=merge)
    assert ran[0] == 'igemm8r<256x160/2:' + ('res>' if merge == 'none' else 'plain>'), ran
    if g[3] % 160 == 0:
        assert ran[1] == 'igemm8r<256x160/2:' + {'none': 'bnb>', 'res': 'bnb+res>', 'acc': 'bnb+acc>'}[merge], ran


def test_igemm8r_split_is_reproducible_and_ships_on_the_640_stage():
    """WRN-28-10's 640-channel layers at batch 128 (128 tiles) take the split form by the shipped rule; two runs give the same bits (the parts are added in
    slice order, whoever arrives last), and the tickets are left at zero."""
    g = (128, 8, 8, 640, 640, 3, 1, 1)
    N, Hh, W, C, K, k, s_, p = g
    eng, sl = _one_op_engine(ir.OP_CONV_FWD, g, dict(x=((N, Hh, W, C), 'T'), w_fwd=((K, 9, C), 'T'), y=((N, Hh, W, K), 'T')))
    gen = torch.Generator().manual_seed(5)
    eng.tensors[sl['x']].copy_(torch.randn(N, Hh, W, C, generator=gen).to(torch.float16))
    eng.tensors[sl['w_fwd']].copy_((torch.randn(K, 9, C, generator=gen) * 0.02).to(torch.float16))
    eng.bind({})
    L = _lib.lib()
    outs = []
    try:
        L.rn_kernel_log(1)
        for _ in range(3):
            eng.tensors[sl['y']].zero_()
            eng.run(0, 1, 0)
            torch.cuda.synchronize()
            outs.append(eng.tensors[sl['y']].clone())
        assert 'igemm8r<256x160/2:plain>' in L.rn_kernel_log_read().decode()
    finally:
        L.rn_kernel_log(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    x = eng.tensors[sl['x']].float().cpu(); w = eng.tensors[sl['w_fwd']].float().cpu().reshape(K, 3, 3, C)
    ref = torch.nn.functional.conv2d(_nchw(x[:8]), w.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    err = float((outs[0][:8].float().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-3, err


# ---- the 320 x 160 weight-gradient kernel of the 160-channel family (conv_wgrad8r.hip) ----
W8R_ANY = 8                          # rn_set_variant2: at any size (its rule wants >= 2,048 tile x K-tile units)
NO_W9 = 16384                        # rn_set_variant2: never the nine-tap kernel (which otherwise takes the 3x3 stride-1 geometries first)
W8R_SMALL = [
    (2, 16, 16, 160, 160, 3, 1, 1),      # five units (four tap pairs + a single tap: half of the last tile's rows empty), 8 K tiles, padding taps
    (3, 8, 8, 320, 160, 3, 1, 1),        # 320 input channels: a unit = the two 160-channel slices of one tap; 3 K tiles
    (2, 16, 16, 160, 320, 3, 2, 1),      # stride 2, two output-channel tiles
    (4, 8, 8, 320, 640, 1, 2, 0),        # stride-2 1x1 (projection shortcut): ONE unit, four output-channel tiles; one K tile
    (1, 7, 9, 160, 160, 3, 1, 1),        # 63 pixels: a single, partly empty K tile
    (5, 12, 12, 320, 320, 3, 1, 1),      # pixel tail (720 = 11 x 64 + 16), splits that cut the K-tile range unevenly
    (37, 16, 16, 160, 160, 3, 1, 1),     # 148 K tiles x 5 tiles: ~51 splits, more than one item per workgroup for some
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('g', W8R_SMALL)
def test_wgrad8r_on_small_geometries(g, dtype):
    """the 320 x 160 weight-gradient kernel forced onto small shapes: per-pixel LDS rows filled by global_load_lds (zero page for padding taps, pixel tails
    and the pad chunks), whole-row XOR swizzle under the transposed reads, segment pairs across taps / across channel slices, strides, slabs."""
    ran = run_conv_case(g, dtype, variant2=W8R_ANY | R8_ANY | NO_W9)
    assert 'wgrad8r<320x160>' in ran, ran


def test_wgrad8r_exact_integers():
    """integer operands: bit for bit against the reference (lane <-> channel maps of the transposed fragments, tap pairing, slab sums in fixed order)."""
    for g in [(2, 16, 16, 160, 160, 3, 1, 1), (3, 8, 8, 320, 320, 3, 1, 1)]:
        N, Hh, W, C, K, k, s_, p = g
        eng, sl = _one_op_engine(ir.OP_CONV_WGRAD, g, dict(x=((N, Hh, W, C), 'T'), dy=((N, Hh, W, K), 'T'), dw=((K, 3, 3, C), 'f32'), ws=((0,), 'u8')))
        rng = np.random.RandomState(0)
        xv = rng.randint(-3, 4, size=(N, Hh, W, C)).astype(np.float32)
        dv = rng.randint(-2, 3, size=(N, Hh, W, K)).astype(np.float32)
        eng.tensors[sl['x']].copy_(torch.from_numpy(xv).to(torch.float16)); eng.tensors[sl['dy']].copy_(torch.from_numpy(dv).to(torch.float16))
        eng.bind({})
        L = _lib.lib()
        L.rn_set_variant2(W8R_ANY | NO_W9)
        try:
            L.rn_kernel_log(1)
            eng.run(0, 1, 0)
            torch.cuda.synchronize()
            assert 'wgrad8r<320x160>' in L.rn_kernel_log_read().decode()
        finally:
            L.rn_kernel_log(0)
            L.rn_set_variant2(0)
        ref = torch.nn.grad.conv2d_weight(_nchw(torch.from_numpy(xv)), (K, C, 3, 3), _nchw(torch.from_numpy(dv)), s_, 1).permute(0, 2, 3, 1)
        assert torch.equal(eng.tensors[sl['dw']].cpu(), ref.contiguous()), g


@pytest.mark.parametrize('shared_ws', [False, True])
@pytest.mark.parametrize('g,n', [((2, 16, 16, 160, 160, 3, 1, 1), 3), ((3, 8, 8, 320, 320, 3, 1, 1), 2), ((6, 16, 16, 160, 320, 3, 2, 1), 4), ((16, 16, 16, 320, 320, 3, 1, 1), 6)])
def test_wgrad8r_batch_of_layers(g, n, shared_ws):
    """rn_conv_wgrad8r_batch: the weight gradients of n layers of one geometry as ONE launch (a table of layer records, items = (split, tile) per record, slabs
    in every record's own workspace -- or, shared_ws, in ONE workspace that all records name, as the plan executor's side workspace is: every record then
    takes a region of its own -- summed right behind): every layer against torch-CPU, one of them accumulating into an existing gradient; the launch is
    bitwise reproducible."""
    import ctypes as C
    import gpu_harness as h                               # noqa: F401
    L = _lib.lib()
    vp = C.c_void_p

    class Desc(C.Structure):
        _fields_ = [('x', vp), ('dy', vp), ('dw', vp), ('ws', vp), ('ws_bytes', C.c_size_t), ('g', _lib.RnConvGeom), ('flags', C.c_int32)]
    L.rn_conv_wgrad8r_batch.argtypes = [C.POINTER(Desc), C.c_int, C.c_int, C.c_int, vp]
    N, Hh, W, Cc, K, k, st, p = g
    gm = geom(*g)
    gs = _lib.geom_struct(gm)
    P, Q = gm['P'], gm['Q']
    dev = torch.device('cuda', 0)
    rng = np.random.RandomState(5)
    L.rn_set_variant2(W8R_ANY)
    try:
        wsb = int(L.rn_conv_wgrad_ws_bytes(C.byref(gs)))
        xs = [torch.from_numpy(rng.randn(N, Hh, W, Cc).astype(np.float32)).to(torch.float16) for _ in range(n)]
        dys = [torch.from_numpy(rng.randn(N, P, Q, K).astype(np.float32)).to(torch.float16) for _ in range(n)]
        old = torch.from_numpy(rng.randn(K, k, k, Cc).astype(np.float32))
        outs = []
        one_ws = torch.empty(max(n * wsb, 16), dtype=torch.uint8, device=dev)       # (the executor's side workspace is sized for the largest single layer; here: room for all)
        for rep in range(2):
            keep = []
            descs = (Desc * n)()
            dws = []
            for i in range(n):
                xd, dyd = xs[i].to(dev), dys[i].to(dev)
                dw = old.to(dev).clone() if i == 1 else torch.full((K, k, k, Cc), float('nan'), device=dev)
                ws = one_ws if shared_ws else torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
                keep += [xd, dyd, ws]
                dws.append(dw)
                descs[i] = Desc(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ws.data_ptr(), n * wsb if shared_ws else wsb, gs, ir.F_ACCUM if i == 1 else 0)
            L.rn_kernel_log(1)
            _lib.check(L.rn_conv_wgrad8r_batch(descs, n, ir.RN_F16, 256, vp(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            names = L.rn_kernel_log_read().decode()
            assert names.count('wgrad8r<320x160>') + names.count('wgrad9<288x160>') == n, names
            L.rn_kernel_log(0)
            outs.append([d.cpu() for d in dws])
    finally:
        L.rn_kernel_log(0)
        L.rn_set_variant2(0)
    for i in range(n):
        ref = torch.nn.grad.conv2d_weight(_nchw(xs[i].float()), (K, Cc, k, k), _nchw(dys[i].float()), st, p).permute(0, 2, 3, 1)
        if i == 1:
            ref = ref + old
        err = float((outs[0][i] - ref).abs().max() / ref.abs().max())
        assert err < 2e-4, (i, err)
        assert torch.equal(outs[0][i], outs[1][i])


# ---- the nine-tap 288 x 160 weight-gradient kernel (conv_wgrad9.hip): 3x3 stride-1 layers with 160 n output channels ----
W9_SMALL = [
    (2, 16, 16, 160, 160, 3, 1, 1),      # W = 16: four image rows per K tile, five channel slices, 8 K tiles
    (3, 8, 8, 320, 160, 3, 1, 1),        # W = 8: a K tile = one image (every vertical tap leaves it)
    (2, 32, 32, 160, 320, 3, 1, 1),      # W = 32: two image rows per K tile; two output-channel tiles
    (1, 32, 32, 32, 160, 3, 1, 1),       # a single 32-channel slice, one image
    (5, 16, 16, 96, 160, 3, 1, 1),       # 96 input channels: three slices
    (37, 16, 16, 160, 160, 3, 1, 1),     # 148 K tiles x 5 tiles: pixel splits, more than one item per workgroup
    (2, 32, 32, 160, 320, 3, 2, 1),      # stride 2, 16-wide output: a K tile = 4 output rows = a 9-row input patch (20 DMA pieces), two output-channel tiles
    (3, 16, 16, 320, 160, 3, 2, 1),      # stride 2, 8-wide output: a K tile = one image's 8 output rows, 17 input rows of 18 pixels
    (5, 32, 32, 64, 160, 3, 2, 1),       # stride 2, two channel slices, five images
]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('g', W9_SMALL)
def test_wgrad9_on_small_geometries(g, dtype):
    """the nine-tap weight-gradient kernel forced onto small shapes: the staged input patch (pad columns, rows above / below the K tile's image rows, image
    borders as out-of-range DMA offsets), the nine shifted transposed reads of it, the row-bit swizzles of both LDS images, every map width it takes."""
    ran = run_conv_case(g, dtype, variant2=W8R_ANY | R8_ANY)
    assert 'wgrad9<288x160>' in ran, ran


def test_wgrad9_exact_integers():
    """integer operands: bit for bit against the reference (tap <-> wave row tile map, channel halves, shifted patch rows, slab sums in fixed order)."""
    for g in [(2, 16, 16, 160, 160, 3, 1, 1), (3, 8, 8, 320, 320, 3, 1, 1), (2, 32, 32, 64, 160, 3, 1, 1), (2, 32, 32, 96, 160, 3, 2, 1), (3, 16, 16, 32, 320, 3, 2, 1)]:
        N, Hh, W, C, K, k, s_, p = g
        P, Q = Hh // s_, W // s_
        eng, sl = _one_op_engine(ir.OP_CONV_WGRAD, g, dict(x=((N, Hh, W, C), 'T'), dy=((N, P, Q, K), 'T'), dw=((K, 3, 3, C), 'f32'), ws=((0,), 'u8')))
        rng = np.random.RandomState(0)
        xv = rng.randint(-3, 4, size=(N, Hh, W, C)).astype(np.float32)
        dv = rng.randint(-2, 3, size=(N, P, Q, K)).astype(np.float32)
        eng.tensors[sl['x']].copy_(torch.from_numpy(xv).to(torch.float16)); eng.tensors[sl['dy']].copy_(torch.from_numpy(dv).to(torch.float16))
        eng.bind({})
        L = _lib.lib()
        L.rn_set_variant2(W8R_ANY)
        try:
            L.rn_kernel_log(1)
            eng.run(0, 1, 0)
            torch.cuda.synchronize()
            assert 'wgrad9<288x160>' in L.rn_kernel_log_read().decode()
        finally:
            L.rn_kernel_log(0)
            L.rn_set_variant2(0)
        ref = torch.nn.grad.conv2d_weight(_nchw(torch.from_numpy(xv)), (K, C, 3, 3), _nchw(torch.from_numpy(dv)), s_, 1).permute(0, 2, 3, 1)
        assert torch.equal(eng.tensors[sl['dw']].cpu(), ref.contiguous()), g
