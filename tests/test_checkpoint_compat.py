"""Checkpoint compatibility with the reference (/root/reference/resnet/utils/checkpoint_util.py:16-18,52-85).

tests/golden/ckpt/{classifier,optimizer}_3.pth were WRITTEN BY THE REFERENCE's own code (make_golden.py g9: DDP-wrapped
classifier, torch.optim.SGD, three steps, resnet.utils.checkpoint_util.save_checkpoints): `module.`-prefixed keys,
KCRS-contiguous weights, momentum buffers.  The product must resume from them: same eval logits, same fourth step."""
import os
import shutil

import numpy as np
import pytest
import torch

from filler import fill, fill_labels
from test_oracle_golden import MODELS

CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ckpt')


def _model(dtype='fp32'):
    from pytorch_ddp_resnet_amd import ResNet
    cfg = MODELS['wrn_small']
    return ResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], 0.0, compute_dtype=dtype)


def test_reference_checkpoint_loads_by_the_reference_rules(tmp_path, golden):
    from pytorch_ddp_resnet_amd.utils import checkpoint_util as cu
    g = golden('g9_checkpoint')
    d = str(tmp_path / 'ck')
    shutil.copytree(CKPT, d)
    m = _model()
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=5e-4)
    assert cu.latest_step(d, 'classifier') == 3
    step = cu.maybe_load_checkpoints(d, {'classifier': cu.ddp_keys(m), 'optimizer': opt, 'scheduler': None}, 'cpu')
    assert step == 3
    # keys written by the reference == keys the product writes (with DDP's prefix), same order
    assert list(cu.ddp_keys(m).state_dict().keys()) == [str(k) for k in g['keys']]
    w = m.state_dict()['_architecture.1.0._conv1.weight']
    assert w.shape == (16, 16, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()      # still channels_last in memory after the load
    assert len(opt.state) == len(list(m.parameters()))
    # write in the same format; the reference's naming / keep-5 / alignment rules
    for s in range(4, 11):
        cu.save_checkpoints(d, {'classifier': cu.ddp_keys(m), 'optimizer': opt, 'scheduler': None}, steps=s)
    assert sorted(f for f in os.listdir(d) if f.startswith('classifier_')) == [f'classifier_{s}.pth' for s in (10, 6, 7, 8, 9)]
    back = torch.load(os.path.join(d, 'classifier_10.pth'))
    ref = torch.load(os.path.join(CKPT, 'classifier_3.pth'))
    assert list(back.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(back[k], ref[k]) and back[k].shape == ref[k].shape, k
    os.remove(os.path.join(d, 'optimizer_10.pth'))
    with pytest.raises(RuntimeError, match='not aligned'):
        cu.maybe_load_checkpoints(d, {'classifier': cu.ddp_keys(m), 'optimizer': opt}, 'cpu')
    empty = str(tmp_path / 'none')
    assert cu.maybe_load_checkpoints(empty, {'classifier': cu.ddp_keys(m), 'optimizer': opt}, 'cpu') == 0


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [True, False])
def test_resume_from_reference_checkpoint_on_the_engine(golden, fused):
    """evaluate with the reference-trained weights, then take the FOURTH training step: loss, parameters and momentum == reference."""
    from pytorch_ddp_resnet_amd.utils import checkpoint_util as cu
    from pytorch_ddp_resnet_amd.utils.fused_sgd import FusedSGD
    from pytorch_ddp_resnet_amd.algos.metrics import compute_losses_and_metrics
    g = golden('g9_checkpoint')
    m = _model('fp32').cuda()
    args = dict(lr=0.1, momentum=0.9, dampening=0.0, nesterov=True, weight_decay=5e-4)
    opt = FusedSGD(m, **args) if fused else torch.optim.SGD(m.parameters(), **args)
    assert cu.maybe_load_checkpoints(CKPT, {'classifier': cu.ddp_keys(m), 'optimizer': opt}, 'cuda') == 3
    x, y = torch.from_numpy(fill((4, 3, 32, 32), 701)).cuda(), torch.from_numpy(fill_labels(4, 10, 801)).cuda()
    m.eval()
    with torch.no_grad():
        lg = m(x)
        met = compute_losses_and_metrics(lg, y)
    assert np.abs(lg.cpu().numpy() - g['eval_logits']).max() < 1e-4 * np.abs(g['eval_logits']).max()
    assert abs(float(met['loss']) - float(g['eval_loss'])) < 1e-5 and float(met['top1_err']) == pytest.approx(float(g['eval_top1']))
    m.train()
    loss = compute_losses_and_metrics(m(x), y)['loss']
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    assert abs(float(loss) - float(g['step4.loss'])) < 1e-4 * max(1.0, abs(float(g['step4.loss'])))
    sums = np.array([p.detach().double().sum().item() for p in m.parameters()])
    assert np.abs(sums - g['step4.param_sums']).max() < 1e-3 * max(1.0, np.abs(g['step4.param_abs_sums']).max())
    moms = np.array([opt.state[p]['momentum_buffer'].double().sum().item() for p in m.parameters()])
    assert np.abs(moms - g['step4.mom_sums']).max() < 1e-3 * max(1.0, np.abs(g['step4.mom_sums']).max())
    # and the optimizer state round-trips through state_dict() (the reference checkpoints optimizer.state_dict())
    sd = opt.state_dict()
    assert len(sd['state']) == len(list(m.parameters())) and all('momentum_buffer' in v for v in sd['state'].values())
