"""Pins the oracle (oracle/np_ops.py, np_model.py, torch_model.py) against the golden vectors captured from the
reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from filler import fill, fill_state, fill_labels
from oracle import np_ops as ops
from oracle.np_model import NumpyResNet, param_shapes, loss_and_metrics, sgd_step
from oracle import torch_model as tm

TOL = 2e-5      # golden vectors are fp32 results of ATen kernels; oracle is fp64 formulas


def to_nhwc(x):
    return np.transpose(x, (0, 2, 3, 1)).astype(np.float64)


CONV_CASES = ['c3x3s1', 'c3x3s2', 'c1x1s1', 'c3x3s1_odd', 'stem3', 'stem7']


@pytest.mark.parametrize('name', CONV_CASES)
def test_g1_conv(golden, name):
    g = golden('g1_conv')
    N, C, K, H, k, s, p, bias, i = g[name + '.meta']
    w = fill((K, C, k, k), 100 + i, (3.0 / (C * k * k)) ** 0.5).astype(np.float64)
    b = fill((K,), 150 + i, 0.1).astype(np.float64) if bias else None
    x = to_nhwc(fill((N, C, H, H), 200 + i))
    y = ops.conv2d_fwd(x, w, s, p, b)
    assert rel_err(y, g[name + '.y']) < TOL
    dy = to_nhwc(fill((N, K, y.shape[1], y.shape[2]), 300 + i))
    assert rel_err(ops.conv2d_dgrad(dy, w, s, p, H, H), g[name + '.dx']) < TOL
    assert rel_err(ops.conv2d_wgrad(x, dy, k, k, s, p), g[name + '.dw']) < TOL
    if bias:
        assert rel_err(ops.bias_grad(dy), g[name + '.db']) < TOL


def test_g2_bn(golden):
    g = golden('g2_bn')
    N, C, H = g['meta']
    gamma, beta = fill((C,), 1, 0.25, 1.0).astype(np.float64), fill((C,), 2, 0.2).astype(np.float64)
    rm, rv = fill((C,), 3, 0.1).astype(np.float64), fill((C,), 4, 0.25, 1.0).astype(np.float64)
    x = to_nhwc(fill((N, C, H, H), 5, 2.0, 0.3))
    dy = to_nhwc(fill((N, C, H, H), 6))
    y, (mean, invstd), (nrm, nrv) = ops.bn_train_fwd(x, gamma, beta, rm, rv)
    assert rel_err(y, g['train_y']) < TOL
    assert rel_err(nrm, g['running_mean']) < TOL and rel_err(nrv, g['running_var']) < TOL
    dx, dg, db = ops.bn_train_bwd(dy, x, gamma, mean, invstd)
    assert rel_err(dx, g['train_dx']) < TOL and rel_err(dg, g['dgamma']) < TOL and rel_err(db, g['dbeta']) < TOL
    assert int(g['nbt']) == 1
    # eval mode uses the UPDATED running stats (the golden ran train first)
    y2 = ops.bn_eval_fwd(x, gamma, beta, nrm, nrv)
    assert rel_err(y2, g['eval_y']) < TOL
    dx2, _, _ = ops.bn_eval_bwd(dy, x, gamma, nrm, nrv)
    assert rel_err(dx2, g['eval_dx']) < TOL


BLOCKS = [(k, p, d, j) for k in ('basic', 'bottleneck') for p in (0, 1) for d in (0, 1) for j in (0, 1)]


@pytest.mark.parametrize('kind,preact,down,proj', BLOCKS)
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_g3_blocks(golden, kind, preact, down, proj, mode):
    """single blocks: driven through NumpyResNet's block routines directly."""
    g = golden('g3_blocks')
    name = f'{kind}_p{preact}_d{down}_j{proj}'
    N, C, H, idx = g[name + '.meta']
    from oracle.np_model import block_layout
    convs, norms, cout = block_layout(kind, C, bool(down), bool(preact))
    # reference module state_dict order: _conv1.., [_proj], _norm1.. (see residual_block.py ctor order)
    shapes = [(f'_conv{j}.weight', (co, ci, k, k)) for j, (ci, co, k, s, p) in enumerate(convs, 1)]
    if down and proj:
        shapes.append(('_proj.weight', (cout, C, 1, 1)))
    for j, c in enumerate(norms, 1):
        shapes += [(f'_norm{j}.weight', (c,)), (f'_norm{j}.bias', (c,)), (f'_norm{j}.running_mean', (c,)),
                   (f'_norm{j}.running_var', (c,)), (f'_norm{j}.num_batches_tracked', ())]
    st = fill_state(shapes, 10 + idx)
    st = {'B.' + k: (v.astype(np.float64) if v.dtype.kind == 'f' else v) for k, v in st.items()}
    if mode == 'eval':   # the golden ran the train pass first on the same module: eval sees the updated buffers
        for k in list(st):
            if 'running_' in k:
                st[k] = g[f'{name}.train.buf.{k[2:]}'].astype(np.float64)
    net = NumpyResNet('a', bool(preact), bool(proj))
    x = to_nhwc(fill((N, C, H, H), 500 + idx))
    cache, new_state = [], {}
    y = net._block_fwd(st, 'B', kind, C, bool(down), x, mode == 'train', {}, cache, new_state)
    assert rel_err(y, g[f'{name}.{mode}.y']) < TOL
    dy = to_nhwc(fill((N, cout, y.shape[1], y.shape[2]), 600 + idx))
    grads = {}
    net._st = st
    dx = net._block_bwd(st, dy, cache, grads)
    assert rel_err(dx, g[f'{name}.{mode}.dx']) < TOL
    for k, v in grads.items():
        assert rel_err(v, g[f'{name}.{mode}.grad.{k[2:]}']) < 5e-5, k
    if mode == 'train':
        for k, v in new_state.items():
            assert rel_err(v, g[f'{name}.train.buf.{k[2:]}']) < TOL, k


MODELS = {
    'rn20':      dict(spec='c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10', preact=False, use_proj=False),
    'wrn_small': dict(spec='c3,16,3,1,1 r1 r1 r1 n a ap8,1,0 fc64,10', preact=True, use_proj=True),
    'v2b_small': dict(spec='c3,16,3,1,1 b2 b2 b2 n a ap8,1,0 fc64,100', preact=True, use_proj=True),
    'v2pad_small': dict(spec='c3,8,3,1,1 r1 r1 n a ap16,1,0 fc16,10', preact=True, use_proj=False),
    'inet_small': dict(spec='c3,16,7,2,3 n a mp3,2,1 b1 b1 ap4,1,0 fc32,10', preact=False, use_proj=True),
}


def model_inputs(g, cfg):
    n, hw, classes, sseed, xseed, yseed, nesterov = [int(v) for v in g['meta']]
    shapes = param_shapes(cfg['spec'], cfg['preact'], cfg['use_proj'])
    st = fill_state(shapes, sseed)
    x = fill((n, 3, hw, hw), xseed)
    y = fill_labels(n, classes, yseed)
    return shapes, st, x, y, bool(nesterov)


@pytest.mark.parametrize('name', list(MODELS))
def test_g4_models_numpy(golden, name):
    if name == 'rn20':
        pytest.skip('numpy restatement of the full rn20 takes minutes; covered by the torch port below')
    cfg = MODELS[name]
    g = golden('g4_' + name)
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    net = NumpyResNet(cfg['spec'], cfg['preact'], cfg['use_proj'])
    logits, _, _ = net.forward(st, x, train=False)
    assert rel_err(logits, g['eval.logits']) < TOL
    logits, cache, new_state = net.forward(st, x, train=True)
    assert rel_err(logits, g['train.logits']) < TOL
    m = loss_and_metrics(logits, y)
    assert abs(m['loss'] - g['train.loss']) < 1e-5 * abs(g['train.loss'])
    assert m['top1_err'] == pytest.approx(float(g['train.top1_err'])) and m['top5_err'] == pytest.approx(float(g['train.top5_err']))
    grads = net.backward(cache, ops.cross_entropy_bwd(logits, y))
    pkeys = [k for k, _ in shapes if k.endswith('weight') or k.endswith('bias')]
    norms = np.array([np.sqrt((grads[k] ** 2).sum()) for k in pkeys])
    assert np.abs(norms - g['grad.norms']).max() < 1e-4 * g['grad.norms'].max()
    gscale = float(g['grad.norms'].max())
    for k in pkeys:
        if 'grad.' + k in g:   # bias grads in front of a train-mode BN are analytically zero: absolute floor
            assert np.abs(grads[k] - g['grad.' + k]).max() < 1e-4 * np.abs(g['grad.' + k]).max() + 1e-6 * gscale, k
    for k, v in new_state.items():
        if 'step1.buf.' + k in g:
            assert rel_err(v, g['step1.buf.' + k]) < TOL, k
    # SGD step 1 (optim_util.py:11-18 + config.yaml:22-28)
    params = {k: np.asarray(st[k], dtype=np.float64) for k in pkeys}
    sgd_step(params, grads, {}, 0.1, 0.9, 5e-4 if nesterov else 1e-4, nesterov)
    sums = np.array([params[k].sum() for k in pkeys])
    assert np.abs(sums - g['step1.param_sums']).max() < 1e-4 * max(1.0, np.abs(g['step1.param_abs_sums']).max())


@pytest.mark.parametrize('name', list(MODELS))
def test_g4_models_torch_port(golden, name):
    """3 SGD steps with the torch port == the reference (logits, loss, grads, BN buffers, params)."""
    cfg = MODELS[name]
    g = golden('g4_' + name)
    shapes, st, x, y, nesterov = model_inputs(g, cfg)
    st = tm.make_trainable({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    model = tm.TorchResNet(cfg['spec'], cfg['preact'], cfg['use_proj'])
    with torch.no_grad():
        assert rel_err(model.forward(st, torch.from_numpy(x), train=False).numpy(), g['eval.logits']) < TOL
    pkeys = [k for k in st if tm.is_param(k)]
    opt = torch.optim.SGD([st[k] for k in pkeys], lr=0.1, momentum=0.9, nesterov=nesterov,
                          weight_decay=5e-4 if nesterov else 1e-4)
    for step in range(3):
        logits, m, grads = tm.train_step(model, st, torch.from_numpy(x), torch.from_numpy(y))
        if step == 0:
            assert rel_err(logits.numpy(), g['train.logits']) < TOL
            assert abs(float(m['loss']) - float(g['train.loss'])) < 1e-5
            assert float(m['top1_err']) == pytest.approx(float(g['train.top1_err']))
            assert float(m['top5_err']) == pytest.approx(float(g['train.top5_err']))
            norms = np.array([grads[k].double().norm().item() for k in pkeys])
            assert np.abs(norms - g['grad.norms']).max() < 1e-4 * g['grad.norms'].max()
        opt.step()
        if step in (0, 2):
            tag = f'step{step + 1}'
            sums = np.array([st[k].detach().double().sum().item() for k in pkeys])
            assert np.abs(sums - g[tag + '.param_sums']).max() < 2e-4 * max(1.0, np.abs(g[tag + '.param_abs_sums']).max())
            for k in st:
                if f'{tag}.buf.{k}' in g:
                    assert rel_err(st[k].detach().numpy(), g[f'{tag}.buf.{k}']) < 1e-4, k


def test_g5_grammar(golden):
    g = golden('g5_grammar')
    specs = {
        'rn20': ('c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10', False, False),
        'wrn2810': ('c3,160,3,1,1 r4 r4 r4 n a ap8,1,0 fc640,10', True, True),
        'v2_164': ('c3,64,3,1,1 b18 b18 b18 n a ap8,1,0 fc256,100', True, True),
        'wrn50a': ('c3,256,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc2048,1000', False, True),
        'wrn50b': ('c3,512,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc4096,1000', False, True),
    }
    for name, (spec, preact, proj) in specs.items():
        shapes = param_shapes(spec, preact, proj)
        assert [k for k, _ in shapes] == list(g[name + '.keys'])
        assert [','.join(map(str, s)) for _, s in shapes] == list(g[name + '.shapes'])
        n = sum(int(np.prod(s)) for k, s in shapes if k.endswith('.weight') or k.endswith('.bias'))
        assert n == int(g[name + '.nparams'])


def test_g6_metrics(golden):
    g = golden('g6_metrics')
    lg, lb = g['logits'].astype(np.float64), g['labels']
    assert abs(ops.cross_entropy_fwd(lg, lb) - g['loss']) < 1e-6
    assert ops.topk_err(lg, lb, 1) == pytest.approx(float(g['top1_err']))
    assert ops.topk_err(lg, lb, 5) == pytest.approx(float(g['top5_err']))
    assert rel_err(ops.cross_entropy_bwd(lg, lb), g['dlogits']) < 1e-5
    # exact ties: torch.topk's order is implementation-defined, so the reference value must lie between the
    # "label wins all ties" and "label loses all ties" counts; the oracle's lowest-index-wins rule lies there too
    lg, lb = g['tied_logits'].astype(np.float64), g['tied_labels']
    assert abs(ops.cross_entropy_fwd(lg, lb) - g['tied_loss']) < 1e-6
    lab = lg[np.arange(len(lb)), lb][:, None]
    for k, key in ((1, 'tied_top1_err'), (5, 'tied_top5_err')):
        lo = 1.0 - ((lg > lab).sum(1) < k).mean()
        hi = 1.0 - (((lg >= lab).sum(1) - 1) < k).mean()
        assert lo - 1e-9 <= float(g[key]) <= hi + 1e-9
        assert lo - 1e-9 <= ops.topk_err(lg, lb, k) <= hi + 1e-9


def test_g7_syncbn_oracle(golden):
    """statistics combined from two half-batches (what the SyncBN all-reduce carries) == plain BN on the full batch."""
    g = golden('g7_syncbn')
    N, C, H = g['meta']
    gamma, beta = fill((C,), 11, 0.25, 1.0).astype(np.float64), fill((C,), 12, 0.2).astype(np.float64)
    x = to_nhwc(fill((N, C, H, H), 13, 1.5, -0.2))
    halves = [x[:4], x[4:]]
    s = sum(h.reshape(-1, C).sum(0) for h in halves); ss = sum((h.reshape(-1, C) ** 2).sum(0) for h in halves)
    m = sum(h.shape[0] * H * H for h in halves)
    mean = s / m; var = ss / m - mean ** 2
    y = (x - mean) / np.sqrt(var + 1e-5) * gamma + beta
    assert rel_err(y, g['y']) < TOL
    assert rel_err(0.1 * mean, g['running_mean']) < TOL
    assert rel_err(0.9 + 0.1 * var * m / (m - 1), g['running_var']) < TOL


def test_g8_eval_loop(golden):
    """evaluation.py:14-42: eval-mode forward over two batches, metrics averaged over batches."""
    g = golden('g8_eval')
    cfg = MODELS['wrn_small']
    shapes = param_shapes(cfg['spec'], cfg['preact'], cfg['use_proj'])
    st = {k: torch.from_numpy(v) for k, v in fill_state(shapes, 41).items()}
    model = tm.TorchResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], dropout_prob=0.3)
    tot = {}
    with torch.no_grad():
        for b in range(2):
            x = torch.from_numpy(fill((4, 3, 32, 32), 1000 + b)); y = torch.from_numpy(fill_labels(4, 10, 1100 + b))
            for k, v in tm.losses_and_metrics(model.forward(st, x, train=False), y).items():
                tot[k] = tot.get(k, 0.0) + float(v) / 2
    for k in ('loss', 'top1_err', 'top5_err'):
        assert tot[k] == pytest.approx(float(g[k]), rel=1e-5)
