"""Host-side mirrors of the reference interface around the hot path: config surface, metrics, factories (CPU), and the
script.py entrypoint + step harness end to end on one GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ('backend world_size master_addr master_port dataset_cls_name data_aug_train data_aug_test architecture_spec preact use_proj '
        'dropout_prob max_steps batch_size num_microbatches optimizer_cls_name optimizer_args scheduler_cls_name scheduler_step_unit '
        'scheduler_args checkpoint_strategy_cls_name checkpoint_strategy_args').split()       # the surface to keep


def test_config_surface_and_q18_fix():
    from pytorch_ddp_resnet_amd.utils.config_util import ConfigParser
    for run in os.listdir(os.path.join(ROOT, 'models_dir')):
        c = ConfigParser(defaults={'mode': 'train', 'data_dir': 'd', 'checkpoint_dir': 'c', 'log_dir': 'l'})
        c.read(os.path.join(ROOT, 'models_dir', run, 'config.yaml'))
        for k in KEYS:
            c.get(k); c[k]
        assert isinstance(c.get('master_port'), str) and c.get('mode') == 'train'
        with pytest.raises(KeyError):
            c.get('missing_key')

        def f(architecture_spec, preact, **kw):
            return architecture_spec, preact, len(kw)
        spec, preact, n = f(**c)                     # the reference's ConfigParser passes NO kwargs here
        assert spec == c['architecture_spec'] and n >= len(KEYS)
        assert dict(c.items())['batch_size'] == c.get('batch_size')
    wrn = ConfigParser(None); wrn.read(os.path.join(ROOT, 'models_dir', 'wrn-28-10-dropout_cifar10', 'config.yaml'))
    assert (wrn.get('architecture_spec'), wrn.get('preact'), wrn.get('use_proj'), wrn.get('dropout_prob')) == \
        ('c3,160,3,1,1 r4 r4 r4 n a ap8,1,0 fc640,10', True, True, 0.3)
    rn = ConfigParser(None); rn.read(os.path.join(ROOT, 'models_dir', 'resnet-v1-20_cifar10', 'config.yaml'))
    assert (rn.get('architecture_spec'), rn.get('backend'), rn.get('world_size')) == ('c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10', 'gloo', 2)


def test_metrics_match_golden(golden):
    from pytorch_ddp_resnet_amd.algos.metrics import compute_losses_and_metrics, global_means
    g = golden('g6_metrics')
    m = compute_losses_and_metrics(torch.from_numpy(g['logits']), torch.from_numpy(g['labels']))
    assert abs(float(m['loss']) - float(g['loss'])) < 1e-6
    assert float(m['top1_err']) == pytest.approx(float(g['top1_err'])) and float(m['top5_err']) == pytest.approx(float(g['top5_err']))
    gm = global_means(m, 1)
    assert gm['loss'] == pytest.approx(float(g['loss']), rel=1e-6)


def test_factories():
    from pytorch_ddp_resnet_amd.utils.optim_util import get_optimizer, get_scheduler
    lin = torch.nn.Linear(2, 2)
    opt = get_optimizer('SGD', lin, dict(lr=0.1, momentum=0.9, dampening=0.0, nesterov=True, weight_decay=5e-4))
    assert isinstance(opt, torch.optim.SGD)
    assert get_scheduler('None', opt, {}) is None
    assert isinstance(get_scheduler('MultiStepLR', opt, dict(milestones=[1, 2], gamma=0.2)), torch.optim.lr_scheduler.MultiStepLR)


@pytest.mark.gpu
def test_train_step_microbatches_sum_gradients():
    """training.py:92-113: two microbatches accumulate SUMMED gradients before one optimizer step."""
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.algos.training import train_step
    torch.manual_seed(0)
    spec = 'c3,16,3,1,1 n a r1 ap32,1,0 fc16,10'
    m = ResNet(spec, False, False, 0.0, compute_dtype='fp32').cuda().train()
    m2 = ResNet(spec, False, False, 0.0, compute_dtype='fp32').cuda().train()
    m2.load_state_dict(m.state_dict())
    xs = [torch.randn(4, 3, 32, 32, device='cuda') for _ in range(2)]
    ys = [torch.randint(0, 10, (4,), device='cuda') for _ in range(2)]
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    acc = {}
    for i in (1, 2):
        train_step(m, xs[i - 1], ys[i - 1], opt, None, 1, i, 2, acc)
    opt2 = torch.optim.SGD(m2.parameters(), lr=0.1)
    for i in (0, 1):                                       # plain autograd accumulation into .grad
        torch.nn.functional.cross_entropy(m2(xs[i]), ys[i]).backward()
    opt2.step()
    for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p, q, atol=1e-6), k


@pytest.mark.gpu
def test_script_entrypoint_single_gpu(tmp_path, capsys):
    """script.py train + eval on one GPU (RCCL process group of size 1, synthetic batches)."""
    import yaml
    import script
    run = tmp_path / 'tiny'
    run.mkdir()
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'models_dir', 'resnet-v1-20_cifar10', 'config.yaml')))
    cfg.update(world_size=1, master_addr='127.0.0.1', master_port='29517', max_steps=3, batch_size=16)
    yaml.safe_dump(cfg, open(run / 'config.yaml', 'w'))
    for mode in ('train', 'eval'):
        args = script.create_argparser().parse_args(['--mode', mode, '--models_dir', str(tmp_path), '--run_name', 'tiny', '--data_dir', 'synthetic'])
        config = script.get_config(args)
        (script.train if mode == 'train' else script.evaluate)(0, config)
    out = capsys.readouterr().out
    assert 'global step: 2' in out and 'Test metrics' in out


@pytest.mark.gpu
def test_script_saves_and_resumes(tmp_path, capsys):
    """script.py train: rank 0 writes {checkpoint_strategy, classifier, optimizer, scheduler, scaler}_{steps}.pth at the reference's points
    (/root/reference/resnet/algos/training.py:129-139,161-171); a second run RESUMES at that step (script.py:84-94) instead of starting over."""
    import yaml
    import script
    run = tmp_path / 'tiny'
    run.mkdir()
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'models_dir', 'resnet-v1-20_cifar10', 'config.yaml')))
    cfg.update(world_size=1, master_addr='127.0.0.1', master_port='29519', max_steps=2, batch_size=16,
               checkpoint_strategy_cls_name='FrequencyCheckpointStrategy', checkpoint_strategy_args=dict(unit='batch', frequency=1))
    yaml.safe_dump(cfg, open(run / 'config.yaml', 'w'))
    argv = ['--mode', 'train', '--models_dir', str(tmp_path), '--run_name', 'tiny', '--data_dir', 'synthetic']
    script.train(0, script.get_config(script.create_argparser().parse_args(argv)))
    files = sorted(os.listdir(run / 'checkpoints'))
    for kind in ('checkpoint_strategy', 'classifier', 'optimizer', 'scheduler', 'scaler'):
        assert f'{kind}_2.pth' in files, files                # saved after step index 1 as steps = 2
    sd = torch.load(run / 'checkpoints' / 'classifier_2.pth')
    assert all(k.startswith('module.') for k in sd)            # DistributedDataParallel's key scheme, as the reference writes it
    capsys.readouterr()
    cfg['max_steps'] = 4
    yaml.safe_dump(cfg, open(run / 'config.yaml', 'w'))
    script.train(0, script.get_config(script.create_argparser().parse_args(argv)))
    out = capsys.readouterr().out
    assert 'Loaded classifier checkpoint' in out and 'with step 2' in out
    assert 'global step: 2...' in out and 'global step: 3...' in out and 'global step: 0...' not in out      # resumed, not restarted
    assert 'classifier_4.pth' in os.listdir(run / 'checkpoints')


@pytest.mark.gpu
def test_script_eval_reads_a_reference_checkpoint(tmp_path, capsys):
    """script.py --mode eval over a checkpoint directory WRITTEN BY THE REFERENCE (tests/golden/ckpt): the classifier is restored (no
    'Running from scratch' for it) and the logits of the golden evaluation batch are the reference's (g9_checkpoint.npz: eval_logits)."""
    import shutil
    import numpy as np
    import yaml
    import script
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from filler import fill
    run = tmp_path / 'ck'
    (run / 'checkpoints').mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, 'tests', 'golden', 'ckpt', 'classifier_3.pth'), run / 'checkpoints' / 'classifier_3.pth')
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'models_dir', 'resnet-v1-20_cifar10', 'config.yaml')))
    cfg.update(world_size=1, master_addr='127.0.0.1', master_port='29521', batch_size=16, compute_dtype='fp32',
               architecture_spec='c3,16,3,1,1 r1 r1 r1 n a ap8,1,0 fc64,10', preact=True, use_proj=True, dropout_prob=0.0)
    yaml.safe_dump(cfg, open(run / 'config.yaml', 'w'))
    args = script.create_argparser().parse_args(['--mode', 'eval', '--models_dir', str(tmp_path), '--run_name', 'ck', '--data_dir', 'synthetic'])
    config = script.get_config(args)
    system = script.setup(0, config)
    try:
        assert system['global_step'] == 3
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g9_checkpoint.npz'))
        m = system['classifier'].eval()
        with torch.no_grad():
            lg = m(torch.from_numpy(fill((4, 3, 32, 32), 701)).cuda())
        assert np.abs(lg.cpu().numpy() - g['eval_logits']).max() < 1e-4 * np.abs(g['eval_logits']).max()
    finally:
        torch.distributed.destroy_process_group()
    assert 'Loaded classifier checkpoint' in capsys.readouterr().out


@pytest.mark.gpu
@pytest.mark.parametrize('nesterov', [False, True])
def test_fused_sgd_matches_torch_sgd(nesterov):
    """utils/fused_sgd.py: one rn_sgd_step over the flat parameter buffer == torch.optim.SGD (optim_util.py:11-18 of the
    reference) over five steps with momentum, weight decay, a MultiStepLR schedule; parameters stay usable by the engine
    (they are re-homed as views of the flat buffer) and the per-parameter momentum state is exposed like torch's."""
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.utils.optim_util import get_optimizer, get_scheduler
    from pytorch_ddp_resnet_amd.utils.fused_sgd import FusedSGD
    torch.manual_seed(0)
    spec = 'c3,16,3,1,1 n a r1 r1 ap16,1,0 fc32,10'
    m = ResNet(spec, False, True, 0.0, compute_dtype='fp32').cuda().train()
    m2 = ResNet(spec, False, True, 0.0, compute_dtype='fp32').cuda().train()
    m2.load_state_dict(m.state_dict())
    args = dict(lr=0.1, momentum=0.9, dampening=0.0, nesterov=nesterov, weight_decay=5e-4)
    opt = get_optimizer('SGD', m, args)
    assert isinstance(opt, FusedSGD)
    opt2 = torch.optim.SGD(m2.parameters(), **args)
    sch, sch2 = get_scheduler('MultiStepLR', opt, dict(milestones=[2, 4], gamma=0.1)), torch.optim.lr_scheduler.MultiStepLR(opt2, milestones=[2, 4], gamma=0.1)
    for step in range(5):
        x, y = torch.randn(8, 3, 32, 32, device='cuda'), torch.randint(0, 10, (8,), device='cuda')
        for mm, oo, ss in ((m, opt, sch), (m2, opt2, sch2)):
            oo.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(mm(x), y).backward()
            oo.step()
            ss.step()
        for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), (step, k, float((p - q).abs().max()))
    assert abs(opt.param_groups[0]['lr'] - opt2.param_groups[0]['lr']) < 1e-12
    base = opt._flat.data_ptr()
    for k, p in m.named_parameters():                      # parameters are views of ONE flat buffer, momentum state exposed per parameter
        assert base <= p.data_ptr() < base + opt._flat.numel() * 4
        assert torch.allclose(opt.state[p]['momentum_buffer'], opt2.state[dict(m2.named_parameters())[k]]['momentum_buffer'], rtol=1e-5, atol=1e-6), k


@pytest.mark.gpu
def test_fused_sgd_with_accumulated_microbatches():
    """gradients that do not alias the engine's flat buffer (summed micro-batches, training.py:92-113) are gathered first."""
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.algos.training import train_step
    from pytorch_ddp_resnet_amd.utils.fused_sgd import FusedSGD
    torch.manual_seed(0)
    spec = 'c3,16,3,1,1 n a r1 ap32,1,0 fc16,10'
    m = ResNet(spec, False, False, 0.0, compute_dtype='fp32').cuda().train()
    m2 = ResNet(spec, False, False, 0.0, compute_dtype='fp32').cuda().train()
    m2.load_state_dict(m.state_dict())
    xs = [torch.randn(4, 3, 32, 32, device='cuda') for _ in range(4)]
    ys = [torch.randint(0, 10, (4,), device='cuda') for _ in range(4)]
    opt, opt2 = FusedSGD(m, lr=0.1, momentum=0.9), torch.optim.SGD(m2.parameters(), lr=0.1, momentum=0.9)
    for rnd in range(2):
        acc = {}
        for i in (1, 2):
            train_step(m, xs[2 * rnd + i - 1], ys[2 * rnd + i - 1], opt, None, 1, i, 2, acc)
        opt2.zero_grad(set_to_none=True)
        for i in (0, 1):
            torch.nn.functional.cross_entropy(m2(xs[2 * rnd + i]), ys[2 * rnd + i]).backward()
        opt2.step()
        for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), (rnd, k)


def test_training_loop_lagged_logging_is_complete_and_ordered():
    """the logging values are read one microbatch late (metrics.global_means_async): every step is still logged once, in
    order, with the values a blocking read gives (any module works on CPU here: the loop is host logic)."""
    from pytorch_ddp_resnet_amd.algos.training import training_loop, train_step
    from pytorch_ddp_resnet_amd.algos.metrics import global_means, global_means_async, compute_losses_and_metrics
    torch.manual_seed(0)
    xs = [(torch.randn(6, 12), torch.randint(0, 10, (6,))) for _ in range(3)]

    def make():
        torch.manual_seed(1)
        m = torch.nn.Linear(12, 10)
        return m, torch.optim.SGD(m.parameters(), lr=0.1)
    m, opt = make()
    lines = []
    steps = training_loop(0, 1, torch.device('cpu'), xs, None, m, opt, num_microbatches=1, max_steps=7, log=lines.append)
    assert steps == 7 and [ln.split('...')[0] for ln in lines] == [f'global step: {i}' for i in range(7)]
    m2, opt2 = make()
    ref = []
    while len(ref) < 7:
        for x, y in xs:
            ref.append(train_step(m2, x, y, opt2)['loss'])
            if len(ref) == 7:
                break
    got = [float(ln.split('loss: ')[1]) for ln in lines]
    assert all(abs(a - b) < 1e-6 for a, b in zip(got, ref)), (got, ref)
    met = compute_losses_and_metrics(torch.randn(5, 10), torch.randint(0, 10, (5,)))
    assert global_means_async(met, 1).result() == global_means(met, 1)


@pytest.mark.parametrize('strategy', ['frequency', 'performance'])
def test_batch_checkpoints_hold_the_state_after_exactly_k_steps(tmp_path, strategy):
    """training.py:129-139: the files written for step k hold the weights and the optimizer state after exactly k steps -- although the loop reads its logging
    values one microbatch late -- and saving at k, then resuming, continues the uninterrupted run bit for bit (any module works on CPU: host logic)."""
    from pytorch_ddp_resnet_amd.algos.training import training_loop, train_step
    from pytorch_ddp_resnet_amd.utils.checkpoint_util import FrequencyCheckpointStrategy, PerformanceCheckpointStrategy, maybe_load_checkpoints, ddp_keys
    torch.manual_seed(0)
    xs = [(torch.randn(6, 12), torch.randint(0, 10, (6,))) for _ in range(4)]

    def make():
        torch.manual_seed(1)
        m = torch.nn.Linear(12, 10)
        return m, torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)

    def strat():
        return FrequencyCheckpointStrategy('batch', 2) if strategy == 'frequency' else PerformanceCheckpointStrategy('batch')
    m, opt = make()
    d = str(tmp_path / 'ck')
    import os
    os.makedirs(d)
    training_loop(0, 1, torch.device('cpu'), xs, None, m, opt, num_microbatches=1, max_steps=7, log=lambda s: None, checkpoint_strategy=strat(), checkpoint_dir=d)
    # the reference states: after k steps, for every k
    m2, opt2 = make()
    states, k = {0: None}, 0
    while k < 7:
        for x, y in xs:
            train_step(m2, x, y, opt2)
            k += 1
            states[k] = ({n: v.clone() for n, v in m2.state_dict().items()}, {i: s['momentum_buffer'].clone() for i, s in opt2.state_dict()['state'].items()})
            if k == 7:
                break
    saved = sorted(int(f.split('_')[1].split('.')[0]) for f in os.listdir(d) if f.startswith('classifier_'))
    assert saved and (strategy != 'frequency' or saved == [1, 3, 5, 7]), saved          # observations 0, 2, 4, 6 -> files of steps 1, 3, 5, 7
    for ksaved in saved:
        m3, opt3 = make()
        x0, y0 = xs[0]
        train_step(m3, x0, y0, opt3)                       # (a momentum buffer to load into)
        assert maybe_load_checkpoints(d, {'classifier': ddp_keys(m3), 'optimizer': opt3}, 'cpu', steps=ksaved) == ksaved
        want_w, want_mom = states[ksaved]
        for n, v in m3.state_dict().items():
            assert torch.equal(v, want_w[n]), (ksaved, n)
        for i, st in opt3.state_dict()['state'].items():
            assert torch.equal(st['momentum_buffer'], want_mom[i]), (ksaved, i)


def test_resumed_run_continues_the_epoch_count(tmp_path):
    """training.py:87-88: the epoch handed to the sampler continues from the checkpoint strategy's epoch counter after a resume."""
    from pytorch_ddp_resnet_amd.algos.training import training_loop
    from pytorch_ddp_resnet_amd.utils.checkpoint_util import FrequencyCheckpointStrategy

    class Sampler:
        def __init__(self):
            self.epochs = []

        def set_epoch(self, e):
            self.epochs.append(e)
    torch.manual_seed(0)
    xs = [(torch.randn(6, 12), torch.randint(0, 10, (6,))) for _ in range(2)]
    m = torch.nn.Linear(12, 10)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    cs = FrequencyCheckpointStrategy('epoch', 1)
    cs.step('epoch'); cs.step('epoch'); cs.step('epoch')            # as loaded from a checkpoint written after three epochs
    sm = Sampler()
    training_loop(0, 1, torch.device('cpu'), xs, None, m, opt, max_steps=4, log=lambda s: None, sampler_train=sm, checkpoint_strategy=cs, checkpoint_dir=None)
    assert sm.epochs == [3, 4], sm.epochs


@pytest.mark.parametrize('name,preact,proj', [('rn20', False, False), ('wrn2810', True, True), ('v2_164', True, True), ('wrn50a', False, True), ('wrn50b', False, True)])
def test_state_dict_keys_and_shapes_match_the_reference(golden, name, preact, proj):
    """G5: ordered state_dict keys, shapes, parameter order and parameter count of the reference module for the five BASELINE specs."""
    import torch
    from pytorch_ddp_resnet_amd import ResNet
    g = golden('g5_grammar')
    specs = {'rn20': 'c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10', 'wrn2810': 'c3,160,3,1,1 r4 r4 r4 n a ap8,1,0 fc640,10',
             'v2_164': 'c3,64,3,1,1 b18 b18 b18 n a ap8,1,0 fc256,100', 'wrn50a': 'c3,256,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc2048,1000',
             'wrn50b': 'c3,512,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc4096,1000'}
    m = ResNet(specs[name], preact, proj, 0.0)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[name + '.keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g[name + '.shapes']]
    assert [k for k, _ in m.named_parameters()] == [str(k) for k in g[name + '.param_keys']]
    assert sum(p.numel() for p in m.parameters()) == int(g[name + '.nparams'])


def test_init_statistics_match_the_reference(golden):
    """resnet.py:160-163: Kaiming-normal on the top-level conv only (std sqrt(2/27) for the 3x3 stem), torch's default
    kaiming_uniform(a=sqrt(5)) -> U(+-1/sqrt(fan_in)) on block convs (max-abs just under 1/sqrt(144) for 16-channel 3x3)."""
    import torch
    from pytorch_ddp_resnet_amd import ResNet
    g = golden('g5_grammar')
    torch.manual_seed(0)
    m = ResNet('c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10', False, False, 0.0)
    sd = m.state_dict()
    std, amax = float(sd['_architecture.0.weight'].std()), float(sd['_architecture.3.0._conv1.weight'].abs().max())
    assert abs(std - (2.0 / 27) ** 0.5) < 0.03 and abs(std - float(g['rn20.init_stem_std'])) < 0.05       # 432 samples: ~3.5 % sampling error
    assert amax <= 1.0 / 12 + 1e-7 and abs(amax - float(g['rn20.init_block_absmax'])) < 2e-3
    assert float(sd['_architecture.0.bias'].abs().max()) <= 1.0 / 27 ** 0.5 + 1e-7                       # Conv2d default bias init
    bn = sd['_architecture.1.weight']
    assert bool((bn == 1).all()) and bool((sd['_architecture.1.running_var'] == 1).all()) and int(sd['_architecture.1.num_batches_tracked']) == 0


def test_dropout_mask_spec_keep_rate_and_decorrelation():
    """the counter-hash dropout mask (spec: tests/np_interp.py keep_mask == csrc/common.h rn_keep): keep rate 1-p, and the masks of
    two sites, or of two steps, are independent (their agreement rate is p^2 + (1-p)^2)."""
    from np_interp import keep_mask
    n, p = 1 << 18, 0.3
    a = keep_mask(n, p, site=3, step_seed=123456789)
    assert abs(a.mean() - (1 - p)) < 4e-3
    for other in (keep_mask(n, p, site=4, step_seed=123456789), keep_mask(n, p, site=3, step_seed=123456790),
                  keep_mask(n, p, site=3, step_seed=123456789 + (1 << 32))):
        assert abs(other.mean() - (1 - p)) < 4e-3
        assert abs((a == other).mean() - (p * p + (1 - p) ** 2)) < 6e-3
        assert abs(np.corrcoef(a.astype(float), other.astype(float))[0, 1]) < 0.01
    # neighbouring elements are independent too (lag-1 autocorrelation)
    assert abs(np.corrcoef(a[:-1].astype(float), a[1:].astype(float))[0, 1]) < 0.01
