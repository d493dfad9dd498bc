"""GPU parity of every HIP kernel against the oracle formulas, through the C ABI (one-op plans).
fp32 path: tolerance 2e-5 relative to the tensor's max (exact-f32 MFMA, different summation order than the oracle's
float64).  bf16 path: inputs are pre-rounded to bf16 on both sides, so the residual error is fp32 accumulation plus one
bf16 rounding of the output (2^-9 relative): tolerance 6e-3."""
import numpy as np
import pytest
import torch

from filler import fill
from pytorch_ddp_resnet_amd.engine import ir

pytestmark = pytest.mark.gpu

TOL = {True: 2e-5, False: 6e-3}
DT = [True, False]      # fp32, bf16


def H():
    import gpu_harness
    return gpu_harness


CONV_GEOMS = [
    # N, H, W, C, K, k, stride, pad
    (2, 8, 8, 16, 32, 3, 1, 1),
    (3, 8, 8, 160, 160, 3, 1, 1),        # BN=160 tile, M tail (192 rows)
    (2, 9, 7, 24, 16, 3, 1, 1),          # odd sizes, K=16 masked columns, C not a multiple of the K tile
    (2, 8, 8, 16, 32, 3, 2, 1),          # stride 2
    (2, 8, 8, 64, 128, 1, 1, 0),         # 1x1
    (2, 8, 8, 32, 64, 1, 2, 0),          # projection shortcut
    (1, 16, 16, 96, 96, 3, 1, 1),        # BN=96 tile
    (2, 7, 7, 128, 256, 3, 1, 1),        # ImageNet-like 7x7 map
    (2, 16, 16, 8, 64, 7, 2, 3),         # 7x7 stride-2 stem on the MFMA route (49 taps, channels padded to one chunk)
    (32, 32, 32, 16, 64, 1, 1, 0),       # thin 1x1 layer on a large map: wgrad split 128 ways, wide slab reduction
    (4, 32, 32, 8, 16, 3, 1, 1),         # 3x3 stem on the MFMA route (im2col wgrad, 72 columns in one 160-wide tile)
]


@pytest.mark.parametrize('fp32', DT)
@pytest.mark.parametrize('g', CONV_GEOMS)
def test_conv_fwd_dgrad_wgrad(g, fp32):
    h = H()
    N, Hh, W, C, K, k, s, p = g
    gm = h.geom(*g)
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); w = b.slot('w', (K, k, k, C), 'f32')
    wf = b.slot('wf', (K, k * k, C)); wd = b.slot('wd', (C, k * k, K))
    y = b.slot('y', (N, gm['P'], gm['Q'], K)); dy = b.slot('dy', (N, gm['P'], gm['Q'], K)); dx = b.slot('dx', (N, Hh, W, C))
    dw = b.slot('dw', (K, k, k, C), 'f32'); ws = b.slot('workspace', (0,), 'u8')
    b.op(ir.OP_PACK_W, buf=dict(w=w, w_fwd=wf, w_dgrad=wd), dim=dict(K=K, RS=k * k, C=C))
    b.op(ir.OP_CONV_FWD, buf=dict(x=x, w_fwd=wf, y=y, res=-1), dim=dict(gm, res_mode=0, res_C=0))
    b.op(ir.OP_CONV_DGRAD, buf=dict(dy=dy, w_dgrad=wd, dx=dx, res=-1), dim=dict(gm, res_mode=0, res_C=0))
    b.op(ir.OP_CONV_WGRAD, buf=dict(x=x, dy=dy, dw=dw, ws=ws), dim=dict(gm))
    b.ws_need.append(('wgrad', gm))
    plan = b.plan(fp32)
    plan.slot_of['ws'] = ws
    wv = fill((K, k, k, C), 1, (3.0 / (C * k * k)) ** 0.5)
    if not fp32:
        wv = h.bf16_round(wv)      # the master is fp32; round so both sides multiply the same values
    hip, ref = h.run_both(plan, dict(x=fill((N, Hh, W, C), 2), w=wv, dy=fill((N, gm['P'], gm['Q'], K), 3)), fp32)
    for name in ('y', 'dx', 'dw'):
        assert h.max_rel(hip[name], ref[name]) < TOL[fp32], name


@pytest.mark.parametrize('fp32', DT)
@pytest.mark.parametrize('mode', [ir.RES_SAME, ir.RES_DOWN2PAD, ir.RES_UP2])
def test_conv_epilogue_residual_and_accum(mode, fp32):
    h = H()
    N, Hh, C, K = 2, 8, 16, 32
    gm = h.geom(N, Hh, Hh, C, K, 3, 1, 1)
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, Hh, C)); wf = b.slot('wf', (K, 9, C)); y = b.slot('y', (N, Hh, Hh, K))
    rshape = {ir.RES_SAME: (N, Hh, Hh, K), ir.RES_DOWN2PAD: (N, 2 * Hh, 2 * Hh, K // 2), ir.RES_UP2: (N, Hh // 2, Hh // 2, 2 * K)}[mode]
    r = b.slot('r', rshape)
    b.op(ir.OP_CONV_FWD, buf=dict(x=x, w_fwd=wf, y=y, res=r), dim=dict(gm, res_mode=mode, res_C=rshape[3]))
    # dgrad of a stride-2 conv, accumulating into a pre-filled dx, with the same residual mode on its own geometry
    g2 = h.geom(N, 2 * Hh, 2 * Hh, K, C, 3, 2, 1)       # conv K->C, stride 2: dy is [N,Hh,Hh,C], dx is [N,2Hh,2Hh,K]
    wd = b.slot('wd', (K, 9, C)); dx = b.slot('dx', (N, 2 * Hh, 2 * Hh, K))
    r2shape = {ir.RES_SAME: (N, 2 * Hh, 2 * Hh, K), ir.RES_DOWN2PAD: (N, 4 * Hh, 4 * Hh, K // 2), ir.RES_UP2: (N, Hh, Hh, 2 * K)}[mode]
    r2 = b.slot('r2', r2shape)
    b.op(ir.OP_CONV_DGRAD, buf=dict(dy=x, w_dgrad=wd, dx=dx, res=r2), dim=dict(g2, res_mode=mode, res_C=r2shape[3]), flags=ir.F_ACCUM)
    plan = b.plan(fp32)
    hip, ref = h.run_both(plan, dict(x=fill((N, Hh, Hh, C), 4), wf=fill((K, 9, C), 5, 0.1), r=fill(rshape, 6), wd=fill((K, 9, C), 7, 0.1),
                                     dx=fill((N, 2 * Hh, 2 * Hh, K), 8), r2=fill(r2shape, 9)), fp32)
    assert h.max_rel(hip['y'], ref['y']) < TOL[fp32]
    assert h.max_rel(hip['dx'], ref['dx']) < TOL[fp32]


@pytest.mark.parametrize('fp32', DT)
@pytest.mark.parametrize('g', [(3, 8, 8, 160, 160, 3, 1, 1), (2, 9, 7, 24, 16, 3, 1, 1), (2, 8, 8, 16, 32, 3, 2, 1), (5, 16, 16, 32, 64, 3, 1, 1)])
def test_conv_fused_epilogues(g, fp32):
    """forward: per-channel (sum, sum^2) of the stored output; dgrad: the BatchNorm-backward sums of the layer that fed
    the conv (mask + gscale + xhat), both reduced inside the conv epilogue.  Only the TOTAL over tile rows is specified."""
    h = H()
    from pytorch_ddp_resnet_amd.engine.lowering import conv_stats_rows
    N, Hh, W, C, K, k, s, p = g
    gm = h.geom(*g)
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); wf = b.slot('wf', (K, k * k, C)); wd = b.slot('wd', (C, k * k, K))
    y = b.slot('y', (N, gm['P'], gm['Q'], K)); dy = b.slot('dy', (N, gm['P'], gm['Q'], K)); dx = b.slot('dx', (N, Hh, W, C))
    res = b.slot('res', (N, gm['P'], gm['Q'], K))
    st = b.slot('st', (conv_stats_rows(gm), 2, K), 'f32'); dp = b.slot('dp', (conv_stats_rows(gm, True), 2, C), 'f32')
    bx = b.slot('bx', (N, Hh, W, C)); bm = b.slot('bm', (N, Hh, W, C)); coef = b.slot('coef', (4, C), 'f32')
    b.op(ir.OP_CONV_FWD, buf=dict(x=x, w_fwd=wf, y=y, res=res, stats=st), dim=dict(gm, res_mode=ir.RES_SAME, res_C=K))
    b.op(ir.OP_CONV_DGRAD, buf=dict(dy=dy, w_dgrad=wd, dx=dx, res=-1, bn_x=bx, bn_mask=bm, bn_coef=coef, bn_partial=dp),
         dim=dict(gm, res_mode=0, res_C=0), fp=dict(gscale=1 / 0.7))
    plan = b.plan(fp32)
    sc = (3.0 / (k * k * C)) ** 0.5
    cf = np.stack([fill((C,), 71, 0.2, 1.0), fill((C,), 72, 0.1), fill((C,), 73, 0.3), fill((C,), 74, 0.2, 1.0)])
    hip, ref = h.run_both(plan, dict(x=fill((N, Hh, W, C), 61), wf=fill((K, k * k, C), 62, sc), wd=fill((C, k * k, K), 63, sc),
                                     res=fill((N, gm['P'], gm['Q'], K), 64), dy=fill((N, gm['P'], gm['Q'], K), 65),
                                     bx=fill((N, Hh, W, C), 66), bm=fill((N, Hh, W, C), 67), coef=cf), fp32)
    tol = TOL[fp32]
    assert h.max_rel(hip['y'], ref['y']) < tol and h.max_rel(hip['dx'], ref['dx']) < tol
    # sums of ~10^3..10^4 terms of mixed sign: compare against the sum of |terms| scale
    ysc = np.abs(ref['y']).reshape(-1, K).sum(0).max()
    assert np.abs(hip['st'].sum(0)[0] - ref['st'].sum(0)[0]).max() < tol * ysc
    assert np.abs(hip['st'].sum(0)[1] - ref['st'].sum(0)[1]).max() < tol * (ref['y'] ** 2).reshape(-1, K).sum(0).max()
    dsc = np.abs(ref['dx']).reshape(-1, C).sum(0).max() / 0.7
    assert np.abs(hip['dp'].sum(0) - ref['dp'].sum(0)).max() < 3 * tol * dsc


def test_wgrad_many_splits_bf16_exact_integers():
    """integer-valued operands make every product and partial sum exact in fp32: the transposed-read fragment
    layout (ds_read_b64_tr_b16 -> 16x16x32 MFMA) must reproduce the oracle bit for bit, asymmetric data."""
    h = H()
    g = (8, 16, 16, 32, 48, 3, 1, 1)
    gm = h.geom(*g)
    N, Hh, W, C, K, k, s, p = g
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); dy = b.slot('dy', (N, Hh, W, K)); dw = b.slot('dw', (K, 3, 3, C), 'f32'); ws = b.slot('workspace', (0,), 'u8')
    b.op(ir.OP_CONV_WGRAD, buf=dict(x=x, dy=dy, dw=dw, ws=ws), dim=dict(gm))
    b.ws_need.append(('wgrad', gm))
    plan = b.plan(False)
    plan.slot_of['ws'] = ws
    rng = np.random.RandomState(0)
    xv = rng.randint(-3, 4, size=(N, Hh, W, C)).astype(np.float64)
    dv = rng.randint(-2, 3, size=(N, Hh, W, K)).astype(np.float64)
    hip, ref = h.run_both(plan, dict(x=xv, dy=dv), False)
    assert np.array_equal(hip['dw'], ref['dw'])


@pytest.mark.parametrize('fp32', DT)
def test_igemm_exact_integers(fp32):
    """A = identity-like weights with asymmetric data: catches a transposed C/D or operand map."""
    h = H()
    g = (2, 8, 8, 32, 32, 3, 1, 1)
    gm = h.geom(*g)
    N, Hh, W, C, K, k, s, p = g
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); wf = b.slot('wf', (K, 9, C)); y = b.slot('y', (N, Hh, W, K))
    b.op(ir.OP_CONV_FWD, buf=dict(x=x, w_fwd=wf, y=y, res=-1), dim=dict(gm, res_mode=0, res_C=0))
    plan = b.plan(fp32)
    rng = np.random.RandomState(1)
    xv = rng.randint(-4, 5, size=(N, Hh, W, C)).astype(np.float64)
    wv = rng.randint(-2, 3, size=(K, 9, C)).astype(np.float64)
    hip, ref = h.run_both(plan, dict(x=xv, wf=wv), fp32)
    if fp32:
        assert np.array_equal(hip['y'], ref['y'])
    else:
        assert h.max_rel(hip['y'], ref['y']) < 4e-3     # |y| up to ~2^8: one bf16 rounding of the output


@pytest.mark.parametrize('fp32', DT)
@pytest.mark.parametrize('C,M_shape', [(16, (2, 8, 8)), (160, (3, 5, 7)), (64, (4, 16, 16))])
def test_bn_forward_backward(C, M_shape, fp32):
    h = H()
    N, Hh, W = M_shape
    M = N * Hh * W
    nblk = 3
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); part = b.slot('part', (nblk, 2, C), 'f32'); coef = b.slot('coef', (4, C), 'f32')
    gamma = b.slot('gamma', (C,), 'f32'); beta = b.slot('beta', (C,), 'f32'); rm = b.slot('rm', (C,), 'f32'); rv = b.slot('rv', (C,), 'f32')
    nbt = b.slot('nbt', (), 'i64'); out = b.slot('out', (N, Hh, W, C)); res = b.slot('res', (N, Hh, W, C))
    dout = b.slot('dout', (N, Hh, W, C)); dpart = b.slot('dpart', (nblk, 2, C), 'f32'); dsum = b.slot('dsum', (2, C), 'f32')
    dg = b.slot('dg', (C,), 'f32'); db = b.slot('db', (C,), 'f32'); dx = b.slot('dx', (N, Hh, W, C)); gout = b.slot('gout', (N, Hh, W, C))
    add = b.slot('add', (N, Hh, W, C))
    b.op(ir.OP_BN_STATS, buf=dict(x=x, partial=part), dim=dict(M=M, C=C, nblk=nblk))
    b.op(ir.OP_BN_FINALIZE, buf=dict(partial=part, gamma=gamma, beta=beta, running_mean=rm, running_var=rv, nbt=nbt, coef=coef),
         dim=dict(nblk=nblk, count=M, C=C), fp=dict(eps=1e-5, momentum=0.1), flags=ir.F_TRAIN)
    b.op(ir.OP_BN_APPLY, buf=dict(x=x, coef=coef, res=res, out=out), dim=dict(N=N, H=Hh, W=W, C=C, res_mode=ir.RES_SAME, res_C=C),
         fp=dict(p=0.3), flags=ir.F_RELU, seed=5)
    fl = ir.F_RELU | ir.F_TRAIN
    b.op(ir.OP_BN_BWD_REDUCE, buf=dict(dout=dout, x=x, mask=out, coef=coef, partial=dpart), dim=dict(M=M, C=C, nblk=nblk), fp=dict(gscale=1 / 0.7), flags=fl)
    b.op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=dpart, dsum=dsum, dgamma=dg, dbeta=db), dim=dict(nblk=nblk, C=C))
    b.op(ir.OP_BN_BWD_APPLY, buf=dict(dout=dout, x=x, mask=out, coef=coef, dsum=dsum, add=add, dx=dx, g_out=gout),
         dim=dict(N=N, H=Hh, W=W, C=C, add_mode=ir.RES_SAME, add_C=C, count=M), fp=dict(gscale=1 / 0.7), flags=fl | ir.F_WRITE_G)
    plan = b.plan(fp32)
    hip, ref = h.run_both(plan, dict(x=fill((N, Hh, W, C), 11, 1.5, 0.4), gamma=fill((C,), 12, 0.25, 1.0), beta=fill((C,), 13, 0.2),
                                     rm=fill((C,), 14, 0.1), rv=fill((C,), 15, 0.25, 1.0), res=fill((N, Hh, W, C), 16), dout=fill((N, Hh, W, C), 17),
                                     add=fill((N, Hh, W, C), 18)), fp32, step_seed=987654321012)
    tol = TOL[fp32]
    for name in ('coef', 'rm', 'rv'):
        assert h.max_rel(hip[name], ref[name]) < 2e-5, name
    assert int(hip['nbt']) == 1
    # the dropout keep pattern must be identical (counter-based hash), up to elements that sit at a rounding edge of relu
    keep_h, keep_r = hip['out'] != 0, ref['out'] != 0
    assert (keep_h != keep_r).mean() < 1e-3
    same = keep_h == keep_r
    assert np.abs(hip['out'] - ref['out'])[same].max() < tol * np.abs(ref['out']).max()
    if same.all():
        for name in ('dsum', 'dg', 'db', 'dx', 'gout'):
            assert h.max_rel(hip[name], ref[name]) < max(tol, 1e-4), name


@pytest.mark.parametrize('nblk,C', [(6272, 128), (1100, 2048), (4200, 520)])
def test_bn_finalize_rows_split_over_workgroups(nblk, C):
    """the finalize kernels of the ImageNet-sized layers (thousands of partial rows: rn_bn_finalize_split / rn_bn_bwd_finalize_split, ticketed hand-off through a
    per-op `fold` buffer) against the plain kernels and the float64 interpreter on the same partial rows; run twice: the finisher leaves the tickets at zero."""
    from pytorch_ddp_resnet_amd.engine.lowering import bn_fold_bytes
    h = H()
    nbytes = bn_fold_bytes(nblk, C)
    assert nbytes > 0, 'this size must take the split form'
    M = nblk * 128
    b = h.PlanBuilder()
    part = b.slot('part', (nblk, 2, C), 'f32'); dpart = b.slot('dpart', (nblk, 2, C), 'f32')
    gamma = b.slot('gamma', (C,), 'f32'); beta = b.slot('beta', (C,), 'f32')
    outs = {}
    for tag in ('s', 'p'):                       # split, plain
        outs[tag] = dict(coef=b.slot('coef' + tag, (4, C), 'f32'), rm=b.slot('rm' + tag, (C,), 'f32'), rv=b.slot('rv' + tag, (C,), 'f32'), nbt=b.slot('nbt' + tag, (), 'i64'),
                         dsum=b.slot('dsum' + tag, (2, C), 'f32'), dg=b.slot('dg' + tag, (C,), 'f32'), db=b.slot('db' + tag, (C,), 'f32'))
        f1 = b.slot('fold1', ((nbytes + 3) // 4,), 'f32', role='fold') if tag == 's' else -1
        f2 = b.slot('fold2', ((nbytes + 3) // 4,), 'f32', role='fold') if tag == 's' else -1
        o = outs[tag]
        b.op(ir.OP_BN_FINALIZE, buf=dict(partial=part, gamma=gamma, beta=beta, running_mean=o['rm'], running_var=o['rv'], nbt=o['nbt'], coef=o['coef'], fold=f1),
             dim=dict(nblk=nblk, count=M, C=C), fp=dict(eps=1e-5, momentum=0.1), flags=ir.F_TRAIN)
        b.op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=dpart, dsum=o['dsum'], dgamma=o['dg'], dbeta=o['db'], fold=f2), dim=dict(nblk=nblk, C=C))
    plan = b.plan(True)
    # partial rows of a 128-pixel slab each: sums ~ 128 * mean, sums of squares ~ 128 * (var + mean^2), so the variance stays positive
    rng = np.random.default_rng(7)
    mean, var = rng.normal(0, 0.5, C), rng.uniform(0.5, 2.0, C)
    pv = np.empty((nblk, 2, C)); pv[:, 0] = 128 * mean + rng.normal(0, 3, (nblk, C)); pv[:, 1] = 128 * (var + mean ** 2) + rng.normal(0, 3, (nblk, C))
    inputs = dict(part=pv, dpart=rng.normal(0, 1, (nblk, 2, C)), gamma=fill((C,), 12, 0.25, 1.0), beta=fill((C,), 13, 0.2))
    for tag in ('s', 'p'):
        inputs['rm' + tag] = fill((C,), 14, 0.1); inputs['rv' + tag] = fill((C,), 15, 0.25, 1.0)
    hip, ref = h.run_both(plan, inputs, True)
    for name in ('coef', 'rm', 'rv', 'dsum', 'dg', 'db'):
        assert h.max_rel(hip[name + 's'], ref[name + 's']) < 2e-5, name           # float64 interpreter
        assert h.max_rel(hip[name + 's'], hip[name + 'p']) < 1e-6, name           # same doubles in another order, rounded to fp32 once
    assert int(hip['nbts']) == 1 and int(hip['nbtp']) == 1
    assert not hip['fold1'][:C // 16].any() and not hip['fold2'][:C // 16].any()    # the tickets are back at zero
    eng = h.last_engine()
    first = {k: eng.tensors[plan.slot_of[k]].clone() for k in ('dsums', 'dgs', 'dbs')}
    eng.run(0, len(plan.ops), 0)
    torch.cuda.synchronize()
    for k, v in first.items():                   # second launch through the same fold buffers: bitwise the same sums
        assert torch.equal(eng.tensors[plan.slot_of[k]], v), k


@pytest.mark.parametrize('fp32', DT)
def test_bn_backward_recomputed_mask(fp32):
    """RN_F_MASK_RECOMPUTE: the ReLU/dropout mask rebuilt from x, the coefficients and the dropout hash gives exactly
    the sums and gradients of the variant that reads the saved output."""
    h = H()
    N, Hh, W, C, nblk = 3, 8, 8, 32, 4
    M = N * Hh * W
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); coef = b.slot('coef', (4, C), 'f32'); out = b.slot('out', (N, Hh, W, C)); dout = b.slot('dout', (N, Hh, W, C))
    outs = {}
    for tag, fl, mk in (('rd', ir.F_RELU | ir.F_TRAIN, out), ('rc', ir.F_RELU | ir.F_TRAIN | ir.F_MASK_RECOMPUTE, -1)):
        part = b.slot('part_' + tag, (nblk, 2, C), 'f32'); dsum = b.slot('dsum_' + tag, (2, C), 'f32')
        dg = b.slot('dg_' + tag, (C,), 'f32'); db = b.slot('db_' + tag, (C,), 'f32'); dx = b.slot('dx_' + tag, (N, Hh, W, C))
        outs[tag] = (part, dsum, dg, db, dx)
    b.op(ir.OP_BN_APPLY, buf=dict(x=x, coef=coef, res=-1, out=out), dim=dict(N=N, H=Hh, W=W, C=C, res_mode=0, res_C=0), fp=dict(p=0.3), flags=ir.F_RELU, seed=9)
    for tag, fl, mk in (('rd', ir.F_RELU | ir.F_TRAIN, out), ('rc', ir.F_RELU | ir.F_TRAIN | ir.F_MASK_RECOMPUTE, -1)):
        part, dsum, dg, db, dx = outs[tag]
        p = 0.3 if mk == -1 else 0.0
        b.op(ir.OP_BN_BWD_REDUCE, buf=dict(dout=dout, x=x, mask=mk, coef=coef, partial=part), dim=dict(M=M, C=C, nblk=nblk), fp=dict(gscale=1 / 0.7, p=p), flags=fl, seed=9)
        b.op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part, dsum=dsum, dgamma=dg, dbeta=db), dim=dict(nblk=nblk, C=C))
        b.op(ir.OP_BN_BWD_APPLY, buf=dict(dout=dout, x=x, mask=mk, coef=coef, dsum=dsum, add=-1, dx=dx, g_out=-1),
             dim=dict(N=N, H=Hh, W=W, C=C, add_mode=0, add_C=0, count=M), fp=dict(gscale=1 / 0.7, p=p), flags=fl, seed=9)
    plan = b.plan(fp32)
    cf = np.stack([fill((C,), 81, 0.3, 1.0), fill((C,), 82, 0.4), fill((C,), 83, 0.3), fill((C,), 84, 0.2, 1.0)])
    hip, ref = h.run_both(plan, dict(x=fill((N, Hh, W, C), 85, 1.5), coef=cf, dout=fill((N, Hh, W, C), 86)), fp32, step_seed=55555555555)
    for name in ('dsum', 'dg', 'db', 'dx'):
        assert np.array_equal(hip[name + '_rd'], hip[name + '_rc']), name           # identical on the GPU, bit for bit
        assert h.max_rel(hip[name + '_rc'], ref[name + '_rc']) < max(TOL[fp32], 1e-4), name


@pytest.mark.parametrize('fp32', DT)
def test_bn_eval_and_residual_modes(fp32):
    h = H()
    N, Hh, C = 2, 8, 32
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, Hh, C)); coef = b.slot('coef', (4, C), 'f32')
    gamma = b.slot('gamma', (C,), 'f32'); beta = b.slot('beta', (C,), 'f32'); rm = b.slot('rm', (C,), 'f32'); rv = b.slot('rv', (C,), 'f32')
    nbt = b.slot('nbt', (), 'i64')
    r_dn = b.slot('r_dn', (N, 2 * Hh, 2 * Hh, C // 2)); r_up = b.slot('r_up', (N, Hh // 2, Hh // 2, 2 * C))
    o1 = b.slot('o1', (N, Hh, Hh, C)); o2 = b.slot('o2', (N, Hh, Hh, C)); dst = b.slot('dst', (N, Hh, Hh, C))
    b.op(ir.OP_BN_FINALIZE, buf=dict(partial=-1, gamma=gamma, beta=beta, running_mean=rm, running_var=rv, nbt=nbt, coef=coef),
         dim=dict(nblk=0, count=1, C=C), fp=dict(eps=1e-5, momentum=0.1))
    b.op(ir.OP_BN_APPLY, buf=dict(x=x, coef=coef, res=r_dn, out=o1), dim=dict(N=N, H=Hh, W=Hh, C=C, res_mode=ir.RES_DOWN2PAD, res_C=C // 2), flags=ir.F_RELU)
    b.op(ir.OP_BN_APPLY, buf=dict(x=x, coef=coef, res=r_up, out=o2), dim=dict(N=N, H=Hh, W=Hh, C=C, res_mode=ir.RES_UP2, res_C=2 * C))
    b.op(ir.OP_ADD_RES, buf=dict(dst=dst, res=r_up), dim=dict(N=N, H=Hh, W=Hh, C=C, res_mode=ir.RES_UP2, res_C=2 * C))
    plan = b.plan(fp32)
    hip, ref = h.run_both(plan, dict(x=fill((N, Hh, Hh, C), 21), gamma=fill((C,), 22, 0.25, 1.0), beta=fill((C,), 23, 0.2), rm=fill((C,), 24, 0.1),
                                     rv=fill((C,), 25, 0.25, 1.0), r_dn=fill((N, 2 * Hh, 2 * Hh, C // 2), 26), r_up=fill((N, Hh // 2, Hh // 2, 2 * C), 27),
                                     dst=fill((N, Hh, Hh, C), 28)), fp32)
    assert int(hip['nbt']) == 0 and h.max_rel(hip['rm'], ref['rm']) == 0
    for name in ('o1', 'o2', 'dst'):
        assert h.max_rel(hip[name], ref[name]) < TOL[fp32], name


@pytest.mark.parametrize('fp32', DT)
@pytest.mark.parametrize('g', [(4, 8, 8, 3, 16, 3, 1, 1), (2, 16, 16, 3, 160, 3, 1, 1), (2, 16, 16, 3, 64, 7, 2, 3), (3, 9, 9, 3, 24, 3, 1, 1)])
def test_stem_fwd_wgrad(g, fp32):
    h = H()
    N, Hh, W, C, K, k, s, p = g
    gm = h.geom(*g)
    b = h.PlanBuilder()
    x = b.slot('x', (N, C, Hh, W), 'f32'); w = b.slot('w', (K, k, k, C), 'f32'); bias = b.slot('bias', (K,), 'f32')
    y = b.slot('y', (N, gm['P'], gm['Q'], K)); dy = b.slot('dy', (N, gm['P'], gm['Q'], K))
    dw = b.slot('dw', (K, k, k, C), 'f32'); db = b.slot('db', (K,), 'f32'); ws = b.slot('workspace', (0,), 'u8')
    b.op(ir.OP_STEM_FWD, buf=dict(x=x, w=w, bias=bias, y=y), dim=dict(gm))
    b.op(ir.OP_STEM_WGRAD, buf=dict(x=x, dy=dy, dw=dw, db=db, ws=ws), dim=dict(gm))
    b.ws_need.append(('stem', gm))
    plan = b.plan(fp32)
    plan.slot_of['ws'] = ws
    hip, ref = h.run_both(plan, dict(x=fill((N, C, Hh, W), 31), w=fill((K, k, k, C), 32, 0.2), bias=fill((K,), 33, 0.1),
                                     dy=fill((N, gm['P'], gm['Q'], K), 34)), fp32)
    assert h.max_rel(hip['y'], ref['y']) < TOL[fp32]
    assert h.max_rel(hip['dw'], ref['dw']) < 2e-5 and h.max_rel(hip['db'], ref['db']) < 2e-5


@pytest.mark.parametrize('fp32', DT)
def test_pools_fc_loss(fp32):
    h = H()
    N, Hh, C, O = 4, 8, 32, 10
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, Hh, C)); mp = b.slot('mp', (N, 4, 4, C)); dmp = b.slot('dmp', (N, 4, 4, C)); dxm = b.slot('dxm', (N, Hh, Hh, C))
    am = b.slot('am', (N, 4, 4, C), 'u8', role='u8')
    w = b.slot('w', (O, C), 'f32'); bias = b.slot('bias', (O,), 'f32'); feat = b.slot('feat', (N, C), 'f32'); logits = b.slot('logits', (N, O), 'f32')
    labels = b.slot('labels', (N,), 'i64'); out3 = b.slot('out3', (4,), 'f32'); dl = b.slot('dl', (N, O), 'f32')
    dxf = b.slot('dxf', (N, 4, 4, C)); dwf = b.slot('dwf', (O, C), 'f32'); dbf = b.slot('dbf', (O,), 'f32')
    dpool = dict(N=N, H=Hh, W=Hh, C=C, k=3, stride=2, pad=1)
    b.op(ir.OP_MAXPOOL_FWD, buf=dict(x=x, y=mp, argmax=am), dim=dpool)
    b.op(ir.OP_MAXPOOL_BWD, buf=dict(dy=dmp, argmax=am, dx=dxm), dim=dpool)
    dfc = dict(N=N, HW=16, C=C, O=O)
    b.op(ir.OP_POOL_FC_FWD, buf=dict(x=mp, w=w, b=bias, feat=feat, logits=logits), dim=dfc)
    b.op(ir.OP_SOFTMAX_CE, buf=dict(logits=logits, labels=labels, out3=out3, dlogits=dl), dim=dict(N=N, O=O), fp=dict(scale=1.0 / N))
    b.op(ir.OP_POOL_FC_BWD, buf=dict(dlogits=dl, feat=feat, w=w, dx=dxf, dw=dwf, db=dbf), dim=dfc)
    plan = b.plan(fp32)
    xv = fill((N, Hh, Hh, C), 41)
    xv[0, 0:3, 0:3, :] = 0.5          # exact ties inside pooling windows: first maximum must win
    hip, ref = h.run_both(plan, dict(x=xv, dmp=fill((N, 4, 4, C), 42), w=fill((O, C), 43, 0.3), bias=fill((O,), 44, 0.1),
                                     labels=np.array([1, 3, 5, 9])), fp32)
    tol = TOL[fp32]
    assert h.max_rel(hip['mp'], ref['mp']) == 0
    assert h.max_rel(hip['dxm'], ref['dxm']) < tol
    for name in ('feat', 'logits', 'dl', 'dwf', 'dbf'):
        assert h.max_rel(hip[name], ref[name]) < 1e-5, name
    assert h.max_rel(hip['dxf'], ref['dxf']) < tol
    assert h.max_rel(hip['out3'][:3], ref['out3'][:3]) < 1e-5


@pytest.mark.parametrize('fp32', DT)
@pytest.mark.parametrize('shape', [(70, 4, 136, 1000), (64, 1, 64, 128), (5, 49, 264, 130), (130, 2, 64, 10), (37, 1, 32, 5), (33, 1, 16, 32), (9, 1, 8, 100),
                                   (3, 1, 8, 1100)])
def test_classifier_head_shapes(fp32, shape):
    """ImageNet-sized heads (O >= 128) take the LDS-tiled fp32 GEMM for logits, dW/db and the broadcast dx, and the loss kernel reads a
    1000-class row a wave at a time: ragged tiles in every dimension (batch, C and O off the 64 x 64 x 16 tiling), accumulate on and off.
    CIFAR heads (O <= 32: several rows of the loss per wave; batch split over the waves of the dW kernel) and a row too long for registers."""
    h = H()
    N, HW, C, O = shape
    for accum in (0, ir.F_ACCUM):
        b = h.PlanBuilder()
        x = b.slot('x', (N, HW, 1, C)); w = b.slot('w', (O, C), 'f32'); bias = b.slot('bias', (O,), 'f32'); feat = b.slot('feat', (N, C), 'f32')
        logits = b.slot('logits', (N, O), 'f32'); labels = b.slot('labels', (N,), 'i64'); out3 = b.slot('out3', (4,), 'f32'); dl = b.slot('dl', (N, O), 'f32')
        dxf = b.slot('dxf', (N, HW, 1, C)); dwf = b.slot('dwf', (O, C), 'f32'); dbf = b.slot('dbf', (O,), 'f32')
        dfc = dict(N=N, HW=HW, C=C, O=O)
        b.op(ir.OP_POOL_FC_FWD, buf=dict(x=x, w=w, b=bias, feat=feat, logits=logits), dim=dfc)
        b.op(ir.OP_SOFTMAX_CE, buf=dict(logits=logits, labels=labels, out3=out3, dlogits=dl), dim=dict(N=N, O=O), fp=dict(scale=1.0 / N))
        b.op(ir.OP_POOL_FC_BWD, buf=dict(dlogits=dl, feat=feat, w=w, dx=dxf, dw=dwf, db=dbf), dim=dfc, flags=accum)
        plan = b.plan(fp32)
        rng = np.random.default_rng(7)
        wv = fill((O, C), 43, 0.3)
        labs = rng.integers(0, O, N)
        xv = fill((N, HW, 1, C), 41)
        hip, ref = h.run_both(plan, dict(x=xv, w=wv, bias=fill((O,), 44, 0.1), labels=labs, dwf=fill((O, C), 45), dbf=fill((O,), 46)), fp32)
        for name in ('feat', 'logits', 'dl', 'dwf', 'dbf'):
            assert h.max_rel(hip[name], ref[name]) < 2e-5, (name, accum)
        assert h.max_rel(hip['dxf'], ref['dxf']) < TOL[fp32], accum
        assert h.max_rel(hip['out3'][:1], ref['out3'][:1]) < 1e-5
        assert (hip['out3'][1:3] == ref['out3'][1:3]).all()


def test_softmax_ce_matches_golden(golden):
    """metrics.py:10-29 pinned by G6 (no-tie rows): loss, top-1/top-5, dlogits."""
    h = H()
    g = golden('g6_metrics')
    N, O = g['logits'].shape
    b = h.PlanBuilder()
    logits = b.slot('logits', (N, O), 'f32'); labels = b.slot('labels', (N,), 'i64'); out3 = b.slot('out3', (4,), 'f32'); dl = b.slot('dl', (N, O), 'f32')
    b.op(ir.OP_SOFTMAX_CE, buf=dict(logits=logits, labels=labels, out3=out3, dlogits=dl), dim=dict(N=N, O=O), fp=dict(scale=1.0 / N))
    hip, _ = h.run_both(b.plan(True), dict(logits=g['logits'], labels=g['labels']), True)
    assert abs(hip['out3'][0] / N - float(g['loss'])) < 1e-5
    assert hip['out3'][1] / N == pytest.approx(float(g['top1_err'])) and hip['out3'][2] / N == pytest.approx(float(g['top5_err']))
    assert h.max_rel(hip['dl'], g['dlogits']) < 1e-5


def test_sgd_step_matches_torch():
    import ctypes as C
    from pytorch_ddp_resnet_amd import _lib
    L = _lib.lib()
    n = 10007
    p = torch.from_numpy(fill((n,), 51)).cuda(); g = torch.from_numpy(fill((n,), 52)).cuda(); buf = torch.zeros(n, device='cuda')
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pt], lr=0.1, momentum=0.9, nesterov=True, weight_decay=5e-4)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in range(3):
        pt.grad = g.clone()
        opt.step()
        _lib.check(L.rn_sgd_step(p.data_ptr(), g.data_ptr(), buf.data_ptr(), n, 0.1, 0.9, 0.0, 5e-4, 1, int(step == 0), 1.0, st))
    torch.cuda.synchronize()
    assert (p - pt.detach()).abs().max().item() < 1e-6


@pytest.mark.parametrize('fp32', DT)
def test_dropout_backward_recomputes_the_mask_from_the_hash(fp32):
    """a kept element whose forward input was exactly 0 still passes its gradient (the mask is the counter hash, not out != 0)."""
    h = H()
    n = 4 * 8 * 8 * 16
    b = h.PlanBuilder()
    x = b.slot('x', (4, 8, 8, 16)); out = b.slot('out', (4, 8, 8, 16)); dout = b.slot('dout', (4, 8, 8, 16)); din = b.slot('din', (4, 8, 8, 16))
    b.op(ir.OP_DROPOUT_FWD, buf=dict(x=x, out=out), dim=dict(n_lo=n, n_hi=0), fp=dict(p=0.3), seed=7)
    b.op(ir.OP_DROPOUT_BWD, buf=dict(dout=dout, out=out, din=din), dim=dict(n_lo=n, n_hi=0), fp=dict(p=0.3), seed=7)
    xv = fill((4, 8, 8, 16), 91)
    xv[:, :4] = 0.0                                   # exact zeros in the forward input
    hip, ref = h.run_both(b.plan(fp32), dict(x=xv, dout=fill((4, 8, 8, 16), 92, 1.0, 2.0)), fp32, step_seed=424242)
    assert h.max_rel(hip['out'], ref['out']) < TOL[fp32] and h.max_rel(hip['din'], ref['din']) < TOL[fp32]
    kept_zero = (hip['din'] != 0) & (np.broadcast_to(xv == 0, hip['din'].shape))
    assert kept_zero.mean() > 0.3                     # ~70 % of the zero-input elements are kept and carry gradient


def test_deferred_slab_sums_equal_the_immediate_reduction_bit_for_bit():
    """rn_conv_wgrad(RN_F_DEFER_REDUCE) + ONE rn_wgrad_reduce_batch over three layers == three rn_conv_wgrad calls with their own
    reduction launches (same per-output summation order), with and without accumulation into dw."""
    import ctypes as C
    from pytorch_ddp_resnet_amd import _lib
    L = _lib.lib()
    vp, i32 = C.c_void_p, C.c_int32

    class Desc(C.Structure):
        _fields_ = [('slabs', vp), ('dw', vp), ('n', C.c_int64), ('splits', i32), ('accumulate', i32)]
    L.rn_conv_wgrad.argtypes = [vp, vp, vp, vp, C.c_size_t, i32, i32, C.POINTER(_lib.RnConvGeom), vp]
    L.rn_conv_wgrad_splits.argtypes = [C.POINTER(_lib.RnConvGeom), i32, i32]
    L.rn_wgrad_reduce_batch.argtypes = [C.POINTER(Desc), i32, vp]
    st = vp(torch.cuda.current_stream().cuda_stream)
    # the last geometry has >= 16,384 output chunks: its immediate reduction is the thread-per-chunk kernel (sixteen accumulators), which
    # must give the bits of the sixteen-lane LDS form the batch launch uses
    geoms = [(64, 32, 32, 16, 16, 3, 1, 1), (64, 16, 16, 32, 32, 3, 1, 1), (64, 32, 32, 64, 16, 1, 1, 0), (128, 32, 32, 160, 160, 3, 1, 1)]
    gen = torch.Generator(device='cuda').manual_seed(3)
    for accum in (0, ir.F_ACCUM):
        descs = (Desc * len(geoms))()
        keep, want, got = [], [], []
        for j, (N, Hh, W, Cc, K, k, s_, p) in enumerate(geoms):
            g = _lib.RnConvGeom(N, Hh, W, Cc, Hh, W, K, k, k, s_, p)
            x = torch.randn(N, Hh, W, Cc, device='cuda', generator=gen).half()
            dy = torch.randn(N, Hh, W, K, device='cuda', generator=gen).half()
            splits = int(L.rn_conv_wgrad_splits(C.byref(g), 2, accum))
            assert splits >= 32                                          # a thin layer: many slabs, the deferrable kind
            n = K * k * k * Cc
            ws = torch.empty(splits * n, device='cuda')
            dw0 = torch.full((n,), 0.5, device='cuda')
            dw1 = dw0.clone()
            _lib.check(L.rn_conv_wgrad(vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw0.data_ptr()), vp(ws.data_ptr()), ws.numel() * 4, accum, 2, C.byref(g), st))
            ws2 = torch.empty(splits * n, device='cuda')
            _lib.check(L.rn_conv_wgrad(vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw1.data_ptr()), vp(ws2.data_ptr()), ws2.numel() * 4, accum | ir.F_DEFER_REDUCE, 2,
                                       C.byref(g), st))
            descs[j] = Desc(ws2.data_ptr(), dw1.data_ptr(), n, splits, 1 if accum else 0)
            keep += [x, dy, ws, ws2]
            want.append(dw0); got.append(dw1)
        torch.cuda.synchronize()
        assert all(torch.equal(b, torch.full_like(b, 0.5)) for b in got)       # nothing summed yet
        _lib.check(L.rn_wgrad_reduce_batch(descs, len(geoms), st))
        torch.cuda.synchronize()
        for a_, b_ in zip(want, got):
            assert torch.equal(a_, b_) and float(a_.abs().max()) > 1.0


@pytest.mark.parametrize('dt', [0, 2])
def test_batched_weight_gradient_launch_writes_the_same_slabs(dt):
    """rn_conv_wgrad_batch: the slab-writing launches of several thin layers of ONE tile shape as one grid == rn_conv_wgrad(RN_F_DEFER_REDUCE) per layer, slab for
    slab and bit for bit (3 x 3 stride 1 and 2, 1 x 1, a ragged pixel count, 17 records = two launches); records of another shape are refused."""
    import ctypes as C
    from pytorch_ddp_resnet_amd import _lib
    L = _lib.lib()
    vp, i32 = C.c_void_p, C.c_int32

    class WDesc(C.Structure):
        _fields_ = [('x', vp), ('dy', vp), ('slabs', vp), ('g', _lib.RnConvGeom), ('flags', i32), ('splits', i32), ('slab_bytes', C.c_uint64)]
    L.rn_conv_wgrad.argtypes = [vp, vp, vp, vp, C.c_size_t, i32, i32, C.POINTER(_lib.RnConvGeom), vp]
    L.rn_conv_wgrad_splits.argtypes = [C.POINTER(_lib.RnConvGeom), i32, i32]
    L.rn_conv_wgrad_batch_key.argtypes = [C.POINTER(_lib.RnConvGeom), i32, i32]
    L.rn_conv_wgrad_batch.argtypes = [C.POINTER(WDesc), i32, i32, vp]
    st = vp(torch.cuda.current_stream().cuda_stream)
    tdt = torch.float32 if dt == 0 else torch.float16
    # (N, H, W, C, K, k, stride, pad): all select the 32 x 32 tile
    geoms = [(32, 32, 32, 16, 16, 3, 1, 1), (32, 32, 32, 32, 32, 3, 2, 1), (32, 16, 16, 32, 16, 1, 1, 0), (33, 19, 17, 16, 32, 3, 1, 1)] * 4 + [(32, 16, 16, 16, 16, 3, 1, 1)]
    gen = torch.Generator(device='cuda').manual_seed(5)
    descs, want, got, keep = (WDesc * len(geoms))(), [], [], []
    key0 = None
    for j, (N, Hh, W, Cc, K, k, s_, p) in enumerate(geoms):
        P, Q = (Hh + 2 * p - k) // s_ + 1, (W + 2 * p - k) // s_ + 1
        g = _lib.RnConvGeom(N, Hh, W, Cc, P, Q, K, k, k, s_, p)
        key = int(L.rn_conv_wgrad_batch_key(C.byref(g), dt, 0))
        assert key > 0 and (key0 is None or key == key0)
        key0 = key
        x = torch.randn(N, Hh, W, Cc, device='cuda', generator=gen).to(tdt)
        dy = torch.randn(N, P, Q, K, device='cuda', generator=gen).to(tdt)
        splits = int(L.rn_conv_wgrad_splits(C.byref(g), dt, 0))
        n = K * k * k * Cc
        ws0, ws1 = torch.full((splits * n,), 7.0, device='cuda'), torch.full((splits * n,), -7.0, device='cuda')
        dw = torch.zeros(n, device='cuda')
        _lib.check(L.rn_conv_wgrad(vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw.data_ptr()), vp(ws0.data_ptr()), ws0.numel() * 4, ir.F_DEFER_REDUCE, dt, C.byref(g), st))
        descs[j] = WDesc(x.data_ptr(), dy.data_ptr(), ws1.data_ptr(), g, 0, splits, ws1.numel() * 4)
        keep += [x, dy, dw]
        want.append(ws0); got.append(ws1)
    nmax = 16
    for lo in range(0, len(geoms), nmax):
        part = (WDesc * min(nmax, len(geoms) - lo))(*[descs[i] for i in range(lo, min(lo + nmax, len(geoms)))])
        _lib.check(L.rn_conv_wgrad_batch(part, len(part), dt, st))
    torch.cuda.synchronize()
    for a_, b_ in zip(want, got):
        assert torch.equal(a_, b_) and float(a_.abs().max()) > 1.0 and not bool((a_ == 7.0).any())
    # a record of another tile shape in the same launch is an argument error, not a wrong gradient
    g_big = _lib.RnConvGeom(32, 16, 16, 64, 16, 16, 64, 3, 3, 1, 1)
    assert int(L.rn_conv_wgrad_batch_key(C.byref(g_big), dt, 0)) not in (0, key0)
    bad = (WDesc * 2)(descs[0], WDesc(descs[0].x, descs[0].dy, descs[0].slabs, g_big, 0, 0, 0))
    assert L.rn_conv_wgrad_batch(bad, 2, dt, st) != 0
    # a record planned with another split count, or a region too small for the launch's splits, is refused (the region would be overrun / mis-summed)
    d0 = descs[0]
    assert L.rn_conv_wgrad_batch((WDesc * 1)(WDesc(d0.x, d0.dy, d0.slabs, d0.g, 0, d0.splits + 1, d0.slab_bytes)), 1, dt, st) != 0
    assert L.rn_conv_wgrad_batch((WDesc * 1)(WDesc(d0.x, d0.dy, d0.slabs, d0.g, 0, d0.splits, d0.slab_bytes - 4)), 1, dt, st) != 0
    # forked launches and the stem's im2col form stay single launches
    assert int(L.rn_conv_wgrad_batch_key(C.byref(descs[0].g), dt, ir.F_FORK)) == 0
    g_stem = _lib.RnConvGeom(32, 32, 32, 4 if dt == 0 else 8, 32, 32, 16, 3, 3, 1, 1)
    assert int(L.rn_conv_wgrad_batch_key(C.byref(g_stem), dt, 0)) == 0


@pytest.mark.parametrize('fp32', DT)
@pytest.mark.parametrize('shape', [(3, 12, 12, 32, 3, 2, 1), (2, 9, 7, 64, 3, 2, 1), (2, 8, 8, 16, 2, 2, 0), (1, 4, 6, 32, 3, 2, 1), (2, 2, 2, 8, 3, 2, 1)])
def test_bn_relu_maxpool_fused(shape, fp32):
    """BN_POOL_FWD / BN_POOL_BWD_REDUCE / BN_POOL_BWD_APPLY (the "n a mp" stem in one pass each way) against the interpreter's
    composition of bn-apply, relu, maxpool and their backward, AND against the unfused op chain on the device (same pooled output and
    argmax bit for bit: the fused forward rounds the normalised value to the compute dtype before comparing, as the stored tensor would be)."""
    h = H()
    N, Hh, W, C, k, st, pd = shape
    P, Q = (Hh + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
    M = N * Hh * W
    b = h.PlanBuilder()
    x = b.slot('x', (N, Hh, W, C)); coef = b.slot('coef', (4, C), 'f32')
    y = b.slot('y', (N, P, Q, C)); am = b.slot('am', (N, P, Q, C), 'u8', role='u8')
    a2 = b.slot('a2', (N, Hh, W, C)); y2 = b.slot('y2', (N, P, Q, C)); am2 = b.slot('am2', (N, P, Q, C), 'u8', role='u8')
    dy = b.slot('dy', (N, P, Q, C)); nblk = min(N * Hh, 5)
    part = b.slot('part', (nblk, 2, C), 'f32'); dsum = b.slot('dsum', (2, C), 'f32'); dg = b.slot('dg', (C,), 'f32'); db = b.slot('db', (C,), 'f32')
    dx = b.slot('dx', (N, Hh, W, C))
    rows = min(N * Hh, 7); sums = b.slot('sums', (rows, 2, C), 'f32'); cs = b.slot('cs', (2, C), 'f32'); csq = b.slot('csq', (C,), 'f32'); cdb = b.slot('cdb', (C,), 'f32')
    da2 = b.slot('da2', (N, Hh, W, C)); part2 = b.slot('part2', (nblk, 2, C), 'f32'); dsum2 = b.slot('dsum2', (2, C), 'f32')
    dg2 = b.slot('dg2', (C,), 'f32'); db2 = b.slot('db2', (C,), 'f32'); dx2 = b.slot('dx2', (N, Hh, W, C))
    dpool = dict(N=N, H=Hh, W=W, C=C, k=k, stride=st, pad=pd)
    fl = ir.F_RELU | ir.F_TRAIN
    b.op(ir.OP_BN_POOL_FWD, buf=dict(x=x, coef=coef, y=y, argmax=am), dim=dpool, flags=ir.F_RELU)
    b.op(ir.OP_BN_POOL_BWD_REDUCE, buf=dict(dy=dy, argmax=am, x=x, coef=coef, partial=part), dim=dict(dpool, nblk=nblk), flags=fl)
    b.op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part, dsum=dsum, dgamma=dg, dbeta=db), dim=dict(nblk=nblk, C=C))
    # the production form: the forward keeps each window's winning input element, the sums are taken over the windows (pooled resolution)
    ys = b.slot('ys', (N, P, Q, C)); ams = b.slot('ams', (N, P, Q, C), 'u8', role='u8'); xsel = b.slot('xsel', (N, P, Q, C))
    parts = b.slot('parts', (nblk, 2, C), 'f32'); dsums = b.slot('dsums', (2, C), 'f32'); dgs = b.slot('dgs', (C,), 'f32'); dbs = b.slot('dbs', (C,), 'f32')
    b.op(ir.OP_BN_POOL_FWD, buf=dict(x=x, coef=coef, y=ys, argmax=ams, xsel=xsel), dim=dpool, flags=ir.F_RELU)
    b.op(ir.OP_BN_POOL_BWD_REDUCE, buf=dict(dy=dy, argmax=ams, x=x, coef=coef, partial=parts, xsel=xsel), dim=dict(dpool, nblk=nblk, npix=N * P * Q), flags=fl)
    b.op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=parts, dsum=dsums, dgamma=dgs, dbeta=dbs), dim=dict(nblk=nblk, C=C))
    # the apply pass also leaves the per-channel sums of the gradient it stores (the bias gradient of a biased producer, summed by the finalize kernel)
    b.op(ir.OP_BN_POOL_BWD_APPLY, buf=dict(dy=dy, argmax=am, x=x, coef=coef, dsum=dsum, dx=dx, sums=sums), dim=dict(dpool, count=M, rows=rows), flags=fl)
    b.op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=sums, dsum=cs, dgamma=csq, dbeta=cdb), dim=dict(nblk=rows, C=C))
    # the unfused chain on the same inputs
    b.op(ir.OP_BN_APPLY, buf=dict(x=x, coef=coef, res=-1, out=a2), dim=dict(N=N, H=Hh, W=W, C=C, res_mode=0, res_C=0), fp=dict(p=0.0), flags=ir.F_RELU)
    b.op(ir.OP_MAXPOOL_FWD, buf=dict(x=a2, y=y2, argmax=am2), dim=dpool)
    b.op(ir.OP_MAXPOOL_BWD, buf=dict(dy=dy, argmax=am2, dx=da2), dim=dpool)
    b.op(ir.OP_BN_BWD_REDUCE, buf=dict(dout=da2, x=x, mask=-1, coef=coef, partial=part2), dim=dict(M=M, C=C, nblk=nblk), fp=dict(gscale=1.0, p=0.0),
         flags=fl | ir.F_MASK_RECOMPUTE)
    b.op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part2, dsum=dsum2, dgamma=dg2, dbeta=db2), dim=dict(nblk=nblk, C=C))
    b.op(ir.OP_BN_BWD_APPLY, buf=dict(dout=da2, x=x, mask=-1, coef=coef, dsum=dsum2, add=-1, dx=dx2, g_out=-1),
         dim=dict(N=N, H=Hh, W=W, C=C, add_mode=0, add_C=0, count=M), fp=dict(gscale=1.0, p=0.0), flags=fl | ir.F_MASK_RECOMPUTE)
    plan = b.plan(fp32)
    xv = fill((N, Hh, W, C), 51)
    cv = np.stack([np.abs(fill((C,), 52)) + 0.5, fill((C,), 53, 0.3), fill((C,), 54, 0.2), np.abs(fill((C,), 55)) + 0.7])
    hip, ref = h.run_both(plan, dict(x=xv, coef=cv, dy=fill((N, P, Q, C), 56)), fp32)
    tol = TOL[fp32]
    assert h.max_rel(hip['y'], ref['y']) < tol
    assert np.array_equal(hip['y'], hip['y2']) and np.array_equal(hip['am'], hip['am2'])          # fused == unfused, bit for bit
    assert np.array_equal(hip['ys'], hip['y']) and np.array_equal(hip['ams'], hip['am'])
    for a_, b_ in (('dsums', 'dsum'), ('dgs', 'dg'), ('dbs', 'db')):            # windows added unrounded vs each pixel's sum rounded first
        assert h.max_rel(hip[a_], hip[b_]) < (1e-5 if fp32 else 5e-3), a_
    for a_, b_ in (('dsum', 'dsum2'), ('dg', 'dg2'), ('db', 'db2'), ('dx', 'dx2')):
        assert h.max_rel(hip[a_], hip[b_]) < 1e-5 if fp32 else h.max_rel(hip[a_], hip[b_]) < 2e-2, a_
    colsum = hip['dx'].reshape(-1, C).astype(np.float64).sum(0)                # of the STORED gradient; its true value is 0 (BatchNorm removes the mean): absolute bound
    scale = np.abs(hip['dx']).reshape(-1, C).astype(np.float64).sum(0).max()
    assert np.abs(hip['cdb'] - colsum).max() <= 1e-5 * scale + 1e-6
    if fp32:       # in 16 bits a window's winner can differ from the float64 interpreter's after rounding (near-ties): the device-side unfused chain above is the reference there
        assert h.max_rel(hip['dsum'], ref['dsum']) < tol and h.max_rel(hip['dx'], ref['dx']) < tol
        assert h.max_rel(hip['xsel'], ref['xsel']) < tol and h.max_rel(hip['dsums'], ref['dsums']) < tol
