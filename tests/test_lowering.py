"""CPU proof of the lowering: the plan IR, interpreted with the oracle's formulas (tests/np_interp.py), reproduces
the reference's golden vectors (G4) and the oracle on every (kind, preact, downsample, use_proj) block combination."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from filler import fill, fill_state, fill_labels
from np_interp import NumpyPlan, keep_mask
from oracle import np_ops as ops
from oracle.np_model import NumpyResNet, param_shapes
from oracle import torch_model as tm
from pytorch_ddp_resnet_amd.engine import ir
from pytorch_ddp_resnet_amd.engine.lowering import lower
from test_oracle_golden import MODELS, model_inputs


def run_plan(cfg, st, x, y, train=True, p=0.0, step_seed=0, **kw):
    plan = lower(cfg['spec'], cfg['preact'], cfg['use_proj'], p, x.shape[0], x.shape[2], x.shape[3], train=train, **kw)
    npl = NumpyPlan(plan)
    npl.load_state(st)
    npl['x'] = x.astype(np.float64)
    npl.forward(step_seed=step_seed)
    logits = npl['logits'].copy()
    npl['dlogits'] = ops.cross_entropy_bwd(logits, y)
    npl.backward(step_seed=step_seed)
    return plan, npl, logits


@pytest.mark.parametrize('name', [n for n in MODELS if n != 'rn20'])
def test_plan_matches_golden(golden, name):
    cfg = MODELS[name]
    g = golden('g4_' + name)
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    plan, npl, logits = run_plan(cfg, st, x, y, train=True)
    assert rel_err(logits, g['train.logits']) < 2e-5
    grads = npl.grads()
    pkeys = [k for k, _ in shapes if k.endswith('weight') or k.endswith('bias')]
    assert sorted(pkeys) == sorted(grads) == sorted(plan.grad_order) == sorted(plan.param_keys)
    norms = np.array([np.sqrt((grads[k] ** 2).sum()) for k in pkeys])
    assert np.abs(norms - g['grad.norms']).max() < 1e-4 * g['grad.norms'].max()
    gscale = float(g['grad.norms'].max())
    for k in pkeys:
        if 'grad.' + k in g:
            assert np.abs(grads[k] - g['grad.' + k]).max() < 1e-4 * np.abs(g['grad.' + k]).max() + 1e-6 * gscale, k
    for k, v in npl.state().items():
        if 'step1.buf.' + k in g:
            assert rel_err(v, g['step1.buf.' + k]) < 2e-5, k
    # eval-mode plan
    plan_e = lower(cfg['spec'], cfg['preact'], cfg['use_proj'], 0.0, x.shape[0], 32, 32, train=False, need_grad=False)
    npe = NumpyPlan(plan_e)
    npe.load_state(st)
    npe['x'] = x.astype(np.float64)
    npe.forward()
    assert rel_err(npe['logits'], g['eval.logits']) < 2e-5
    assert len(plan_e.ops) == plan_e.n_fwd


@pytest.mark.parametrize('fp32', [True, False])
def test_imagenet_stem_lowerings_match_golden(golden, fp32):
    """the 7 x 7 / stride-2 stem has two lowerings: channels padded to one chunk (fp32 engine) and the space-to-depth form (16-bit engines: a 4 x 4 VALID
    convolution over [N][H/2+3][W/2+3][16], weights regrouped to [K][4][4][16], weight gradient mapped back).  Both, interpreted in float64, must reproduce the
    reference's golden logits and gradients (the stem weight gradient included)."""
    cfg = MODELS['inet_small']
    g = golden('g4_inet_small')
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    plan, npl, logits = run_plan(cfg, st, x, y, train=True, fp32=fp32)
    kinds = {o.kind for o in plan.ops}
    assert (ir.OP_IMG_TO_S2D in kinds) == (not fp32) and (ir.OP_IMG_TO_NHWC in kinds) == fp32
    assert rel_err(logits, g['train.logits']) < 2e-5
    grads = npl.grads()
    pkeys = [k for k, _ in shapes if k.endswith('weight') or k.endswith('bias')]
    norms = np.array([np.sqrt((grads[k] ** 2).sum()) for k in pkeys])
    assert np.abs(norms - g['grad.norms']).max() < 1e-4 * g['grad.norms'].max()
    k0 = '_architecture.0.weight'
    assert grads[k0].shape == (16, 3, 7, 7)                 # reference layout [K][C][R][S]
    if 'grad.' + k0 in g:
        assert np.abs(grads[k0] - g['grad.' + k0]).max() < 1e-4 * np.abs(g['grad.' + k0]).max() + 1e-6 * float(g['grad.norms'].max())


COMBOS = [(k, p, j) for k in 'rb' for p in (False, True) for j in (False, True)]


@pytest.mark.parametrize('kind,preact,proj', COMBOS)
@pytest.mark.parametrize('train', [True, False])
def test_plan_all_block_combinations(kind, preact, proj, train):
    """two stacks: block 0 keeps the shape, block 1 downsamples -> all 16 (kind, preact, down, proj) cases."""
    top = 'n a ' if not preact else ''
    tail = 'n a ' if preact else ''
    spec = f'c3,16,3,1,1 {top}{kind}1 {kind}1 {tail}ap4,1,0 fc32,10'
    cfg = dict(spec=spec, preact=preact, use_proj=proj)
    shapes = param_shapes(spec, preact, proj)
    st = fill_state(shapes, 7)
    x, y = fill((4, 3, 8, 8), 70), fill_labels(4, 10, 71)
    plan, npl, logits = run_plan(cfg, st, x, y, train=train)
    tst = tm.make_trainable({k: torch.from_numpy(v.astype(np.float64) if v.dtype.kind == 'f' else v.copy()) for k, v in st.items()})
    model = tm.TorchResNet(spec, preact, proj)
    lg, m, grads = tm.train_step(model, tst, torch.from_numpy(x).double(), torch.from_numpy(y), train=train)
    assert rel_err(logits, lg.numpy()) < 1e-9
    mine = npl.grads()
    scale = max(float(v.abs().max()) for v in grads.values())
    for k, v in grads.items():
        assert np.abs(mine[k] - v.numpy()).max() < 1e-9 * scale + 1e-12, k
    if train:
        for k, v in npl.state().items():
            assert rel_err(v, tst[k].detach().numpy()) < 1e-9, k


@pytest.mark.parametrize('preact', [False, True])
def test_plan_dropout_sites(preact):
    """p > 0: the plan's counter-based masks, fed to the oracle network as explicit masks, give the same result."""
    spec = 'c3,8,3,1,1 ' + ('' if preact else 'n a ') + 'r1 r1 ' + ('n a ' if preact else '') + 'ap4,1,0 fc16,10'
    cfg = dict(spec=spec, preact=preact, use_proj=True)
    shapes = param_shapes(spec, preact, True)
    st = fill_state(shapes, 9)
    x, y = fill((2, 3, 8, 8), 90), fill_labels(2, 10, 91)
    p, seed = 0.3, 12345678901
    plan, npl, logits = run_plan(cfg, st, x, y, train=True, p=p, step_seed=seed)
    masks = {}
    for op in plan.ops[:plan.n_fwd]:
        if op.seed:
            out = plan.slots[op.buf['out']]
            m = keep_mask(out.numel, p, op.seed, seed).reshape(out.shape).astype(np.float64)
            if op.kind == ir.OP_DROPOUT_FWD:
                masks[op.note + '._dropout1'] = m
            else:
                bp, a = op.note.rsplit('.a', 1)
                masks[f'{bp}._dropout{int(a) + (0 if preact else 1)}'] = m
    assert len(masks) == 4
    net = NumpyResNet(spec, preact, True, dropout_prob=p)
    lg, cache, _ = net.forward(st, x, train=True, dropout_masks=masks)
    assert rel_err(logits, lg) < 1e-10
    grads = net.backward(cache, ops.cross_entropy_bwd(lg, y))
    mine = npl.grads()
    for k, v in grads.items():
        assert np.abs(mine[k] - v).max() < 1e-9 * max(np.abs(v).max(), 1e-3), k
    frac = np.mean([m.mean() for m in masks.values()])
    assert abs(frac - 0.7) < 0.05


def test_plan_with_fused_dgrad_reduction(golden):
    """the optional BN-backward reduction inside the dgrad epilogue gives the same gradients (kept off by default)."""
    name = 'wrn_small'
    cfg = MODELS[name]
    g = golden('g4_' + name)
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    plan, npl, logits = run_plan(cfg, st, x, y, train=True, fuse_dgrad=True)
    assert sum(op.kind == ir.OP_BN_BWD_REDUCE for op in plan.ops) < 3
    assert any(op.kind == ir.OP_CONV_DGRAD and op.buf.get('bn_partial', -1) >= 0 for op in plan.ops)
    grads = npl.grads()
    pkeys = [k for k, _ in shapes if k.endswith('weight') or k.endswith('bias')]
    norms = np.array([np.sqrt((grads[k] ** 2).sum()) for k in pkeys])
    assert np.abs(norms - g['grad.norms']).max() < 1e-4 * g['grad.norms'].max()


def test_plan_structure_wrn():
    plan = lower('c3,160,3,1,1 r4 r4 r4 n a ap8,1,0 fc640,10', True, True, 0.3, 128, 32, 32, fp32=False)
    kinds = [op.kind for op in plan.ops]
    assert kinds.count(ir.OP_CONV_FWD) == 27 and kinds.count(ir.OP_STEM_FWD) == 0     # 27 convs, stem on the MFMA route
    assert kinds.count(ir.OP_BN_FINALIZE) == 25 and kinds.count(ir.OP_CONV_WGRAD) == 27
    assert len(plan.grad_order) == len(plan.param_keys) == 80                       # 80 param tensors
    n = sum(s.numel for s in plan.slots if s.role == 'param')
    assert n == 36688330
    for op in plan.ops:
        op.packed()


# token sequences the reference's grammar accepts (resnet.py:122-158) that none of the shipped configs uses: an 'a' that follows no 'n', a second
# top-level convolution, an AvgPool2d that is not the global pool in front of 'f' (with and without padding), 'f' on an unpooled / partly pooled map
# (Flatten() orders the features (c, h, w); the engine's maps are (h, w, c))
GRAMMAR_CORNERS = [
    ('c3,16,3,1,1 a r1 ap8,1,0 fc16,10', False),
    ('c3,16,3,1,1 n a c16,32,3,2,1 n a r1 ap4,1,0 fc32,10', False),
    ('c3,16,3,1,1 r1 n a c16,16,1,1,0 a ap8,1,0 fc16,10', True),
    ('c3,16,3,1,1 n a r1 ap2,2,0 fc256,10', False),
    ('c3,16,3,1,1 r1 n a fc1024,10', True),
    ('c3,16,3,1,1 n a ap3,2,1 r1 ap4,1,0 fc16,10', False),
    ('c3,16,3,1,1 n a mp3,2,1 a ap2,1,0 fc144,10', False),
]


@pytest.mark.parametrize('spec,preact', GRAMMAR_CORNERS)
@pytest.mark.parametrize('train', [True, False])
def test_plan_grammar_corners(spec, preact, train):
    cfg = dict(spec=spec, preact=preact, use_proj=True)
    st = fill_state(param_shapes(spec, preact, True), 11)
    x, y = fill((4, 3, 8, 8), 110), fill_labels(4, 10, 111)
    plan, npl, logits = run_plan(cfg, st, x, y, train=train)
    for op in plan.ops:
        op.packed()
    tst = tm.make_trainable({k: torch.from_numpy(v.astype(np.float64) if v.dtype.kind == 'f' else v.copy()) for k, v in st.items()})
    lg, m, grads = tm.train_step(tm.TorchResNet(spec, preact, True), tst, torch.from_numpy(x).double(), torch.from_numpy(y), train=train)
    assert rel_err(logits, lg.numpy()) < 1e-9
    mine = npl.grads()
    assert set(mine) == set(grads)
    scale = max(float(v.abs().max()) for v in grads.values())
    for k, v in grads.items():
        assert np.abs(mine[k].reshape(v.shape) - v.numpy()).max() < 1e-9 * scale + 1e-12, k
    if train:
        for k, v in npl.state().items():
            assert rel_err(v, tst[k].detach().numpy()) < 1e-9, k


def test_unsupported_patterns_fail_loudly():
    with pytest.raises(NotImplementedError):
        lower('c3,16,3,1,1 n a ap8,1,0 fc16,32 fc32,10', False, False, 0.0, 2, 8, 8)        # a classifier that is not the last component
    with pytest.raises(ValueError):
        lower('c3,16,3,1,1 n a r1 fc16,10', False, False, 0.0, 2, 8, 8)                     # 'f' narrower than the flattened map
    with pytest.raises(ValueError):
        lower('c3,16,3,1,1 n a c8,16,3,1,1 ap8,1,0 fc16,10', False, False, 0.0, 2, 8, 8)      # a mid-network 'c' whose input width is not the map's
    with pytest.raises(NotImplementedError):
        lower('c3,16,3,1,1 n a c16,12,3,1,1 ap8,1,0 fc12,10', False, False, 0.0, 2, 8, 8, fp32=False)     # ... or whose channel counts are no multiple of the 16-byte chunk (8 in 16 bits)
    with pytest.raises(ValueError):
        lower('c3,16,3,1,1 n a ap9,1,0 fc16,10', False, False, 0.0, 2, 8, 8)                  # AvgPool2d larger than the (padded) map: torch raises too
    with pytest.raises(ValueError):
        lower('c3,16,3,1,1 x1 ap8,1,0 fc16,10', False, False, 0.0, 2, 8, 8)
    with pytest.raises(AttributeError):
        lower('c3,16 r1 ap8,1,0 fc16,10', False, False, 0.0, 2, 8, 8)
