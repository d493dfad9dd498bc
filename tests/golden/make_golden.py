"""
Generates the golden vectors in this directory by IMPORTING THE REFERENCE (read-only at /root/reference) on CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Only data is written (inputs are the closed-form filler of filler.py; outputs come from the reference's own
modules: resnet.architectures.{resnet,residual_block}, resnet.algos.{metrics,evaluation}, and the torch.nn ops
those files call).  /root/reference does not exist on the GPU box: tests read the .npz files, never this script.
Fixture groups follow SURVEY.md section 8(c): G1 conv kernels, G2 BatchNorm, G3 blocks, G4 models (+SGD steps),
G5 grammar/state_dict keys, G6 metrics, G7 SyncBN oracle, G8 evaluation loop, G9 a reference-written checkpoint.
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True

from filler import fill, fill_state, fill_labels                      # noqa: E402
from resnet.architectures.resnet import ResNet                        # noqa: E402
from resnet.architectures.residual_block import ResidualBlock, BottleneckResidualBlock  # noqa: E402
from resnet.algos.metrics import compute_losses_and_metrics           # noqa: E402
from resnet.algos.evaluation import evaluation_loop                   # noqa: E402

torch.set_num_threads(4)
T = torch.from_numpy

SPECS = {
    'rn20':    dict(spec='c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10', preact=False, use_proj=False),
    'wrn2810': dict(spec='c3,160,3,1,1 r4 r4 r4 n a ap8,1,0 fc640,10', preact=True, use_proj=True),
    'v2_164':  dict(spec='c3,64,3,1,1 b18 b18 b18 n a ap8,1,0 fc256,100', preact=True, use_proj=True),
    'wrn50a':  dict(spec='c3,256,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc2048,1000', preact=False, use_proj=True),
    'wrn50b':  dict(spec='c3,512,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc4096,1000', preact=False, use_proj=True),
}

# small models exercised end to end (G4)
MODELS = {
    'rn20':      dict(spec=SPECS['rn20']['spec'], preact=False, use_proj=False, n=8, hw=32, classes=10),
    'wrn_small': dict(spec='c3,16,3,1,1 r1 r1 r1 n a ap8,1,0 fc64,10', preact=True, use_proj=True, n=4, hw=32, classes=10),
    'v2b_small': dict(spec='c3,16,3,1,1 b2 b2 b2 n a ap8,1,0 fc64,100', preact=True, use_proj=True, n=4, hw=32, classes=100),
    'v2pad_small': dict(spec='c3,8,3,1,1 r1 r1 n a ap16,1,0 fc16,10', preact=True, use_proj=False, n=4, hw=32, classes=10),
    'inet_small': dict(spec='c3,16,7,2,3 n a mp3,2,1 b1 b1 ap4,1,0 fc32,10', preact=False, use_proj=True, n=4, hw=32, classes=10),
}


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f'{name}.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays')


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().numpy()


def load_filled(module, seed):
    shapes = [(k, tuple(v.shape)) for k, v in module.state_dict().items()]
    st = fill_state(shapes, seed)
    module.load_state_dict({k: T(v) for k, v in st.items()})
    return shapes


# ------------------------------------------------------------------------------------------- G1
def g1_conv():
    out = {}
    cases = [  # name, N, C, K, H, k, stride, pad, bias
        ('c3x3s1', 2, 8, 16, 8, 3, 1, 1, False), ('c3x3s2', 2, 8, 16, 8, 3, 2, 1, False),
        ('c1x1s1', 2, 16, 4, 7, 1, 1, 0, False), ('c3x3s1_odd', 2, 4, 8, 7, 3, 1, 1, False),
        ('stem3', 2, 3, 16, 8, 3, 1, 1, True), ('stem7', 2, 3, 8, 16, 7, 2, 3, True),
    ]
    for i, (name, N, C, K, H, k, s, p, bias) in enumerate(cases):
        conv = torch.nn.Conv2d(C, K, (k, k), (s, s), (p, p), bias=bias)     # as resnet.py:69-75 / residual_block.py:34-57
        conv.weight.data = T(fill((K, C, k, k), 100 + i, (3.0 / (C * k * k)) ** 0.5))
        if bias:
            conv.bias.data = T(fill((K,), 150 + i, 0.1))
        x = T(fill((N, C, H, H), 200 + i)).requires_grad_(True)
        y = conv(x)
        dy = T(fill(tuple(y.shape), 300 + i))
        y.backward(dy)
        out[name + '.y'] = nhwc(y); out[name + '.dx'] = nhwc(x.grad)
        out[name + '.dw'] = conv.weight.grad.numpy()
        if bias:
            out[name + '.db'] = conv.bias.grad.numpy()
        out[name + '.meta'] = np.array([N, C, K, H, k, s, p, int(bias), i])
    save('g1_conv', **out)


# ------------------------------------------------------------------------------------------- G2
def g2_bn():
    out = {}
    N, C, H = 4, 8, 6
    bn = torch.nn.BatchNorm2d(C)                                           # as resnet.py:111-112
    bn.weight.data = T(fill((C,), 1, 0.25, 1.0)); bn.bias.data = T(fill((C,), 2, 0.2))
    bn.running_mean.data = T(fill((C,), 3, 0.1)); bn.running_var.data = T(fill((C,), 4, 0.25, 1.0))
    x = T(fill((N, C, H, H), 5, 2.0, 0.3)).requires_grad_(True)
    bn.train()
    y = bn(x)
    dy = T(fill(tuple(y.shape), 6))
    y.backward(dy)
    out.update(train_y=nhwc(y), train_dx=nhwc(x.grad), dgamma=bn.weight.grad.numpy().copy(), dbeta=bn.bias.grad.numpy().copy(),
               running_mean=bn.running_mean.numpy().copy(), running_var=bn.running_var.numpy().copy(),
               nbt=bn.num_batches_tracked.numpy().copy())
    bn.eval()
    x2 = T(fill((N, C, H, H), 5, 2.0, 0.3)).requires_grad_(True)
    y2 = bn(x2)
    y2.backward(dy)
    out.update(eval_y=nhwc(y2), eval_dx=nhwc(x2.grad), meta=np.array([N, C, H]))
    save('g2_bn', **out)


# ------------------------------------------------------------------------------------------- G3
def g3_blocks():
    out = {}
    N, C, H = 2, 8, 8
    idx = 0
    for kind, cls in (('basic', ResidualBlock), ('bottleneck', BottleneckResidualBlock)):
        for preact in (False, True):
            for down in (False, True):
                for proj in (False, True):
                    name = f'{kind}_p{int(preact)}_d{int(down)}_j{int(proj)}'
                    blk = cls(channels=C, downsample=down, preact=preact, use_proj=proj, dropout_prob=0.0)
                    load_filled(blk, 10 + idx)
                    for mode in ('train', 'eval'):
                        blk.train(mode == 'train')
                        for p in blk.parameters():
                            p.grad = None
                        x = T(fill((N, C, H, H), 500 + idx)).requires_grad_(True)
                        y = blk(x)
                        dy = T(fill(tuple(y.shape), 600 + idx))
                        y.backward(dy)
                        out[f'{name}.{mode}.y'] = nhwc(y)
                        out[f'{name}.{mode}.dx'] = nhwc(x.grad)
                        for k, p in blk.named_parameters():
                            out[f'{name}.{mode}.grad.{k}'] = p.grad.numpy().copy()
                        if mode == 'train':
                            for k, b in blk.named_buffers():
                                out[f'{name}.train.buf.{k}'] = b.numpy().copy()
                    out[f'{name}.meta'] = np.array([N, C, H, idx])
                    idx += 1
    save('g3_blocks', **out)


# ------------------------------------------------------------------------------------------- G4
def g4_models():
    for mi, (name, cfg) in enumerate(MODELS.items()):
        out = {}
        net = ResNet(architecture_spec=cfg['spec'], preact=cfg['preact'], use_proj=cfg['use_proj'], dropout_prob=0.0)
        load_filled(net, 40 + mi)
        n, hw, classes = cfg['n'], cfg['hw'], cfg['classes']
        x = T(fill((n, 3, hw, hw), 700 + mi))
        y = T(fill_labels(n, classes, 800 + mi))
        # eval-mode forward first (running stats untouched)
        net.eval()
        with torch.no_grad():
            out['eval.logits'] = net(x).numpy().copy()
        # three SGD steps (config.yaml:22-28 of the rn20 run: lr .1, momentum .9, wd 1e-4, nesterov False;
        # the wrn run uses nesterov True, wd 5e-4) -- training.py:92-113 step body
        nesterov = cfg['preact']
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, dampening=0.0, nesterov=nesterov,
                              weight_decay=5e-4 if nesterov else 1e-4)
        net.train()
        for step in range(3):
            logits = net(x)
            m = compute_losses_and_metrics(logits, y) if classes >= 5 else None
            loss = m['loss']
            loss.backward()
            if step == 0:
                out['train.logits'] = logits.detach().numpy().copy()
                out['train.loss'] = loss.detach().numpy().copy()
                out['train.top1_err'] = m['top1_err'].numpy().copy(); out['train.top5_err'] = m['top5_err'].numpy().copy()
                names = [k for k, _ in net.named_parameters()]
                out['grad.norms'] = np.array([p.grad.double().norm().item() for p in net.parameters()])
                out['grad.sums'] = np.array([p.grad.double().sum().item() for p in net.parameters()])
                keep = names if name != 'rn20' else [k for k in names if ('.0.' in k and '_architecture.3' in k) or k.count('.') == 2
                                                      or '_architecture.5.0' in k or '.1.' in k[-12:]]
                for k, p in net.named_parameters():
                    if k in keep:
                        out['grad.' + k] = p.grad.numpy().copy()
            opt.step(); opt.zero_grad()
            if step in (0, 2):
                tag = f'step{step + 1}'
                out[tag + '.loss'] = loss.detach().numpy().copy()
                out[tag + '.param_sums'] = np.array([p.detach().double().sum().item() for p in net.parameters()])
                out[tag + '.param_abs_sums'] = np.array([p.detach().double().abs().sum().item() for p in net.parameters()])
                for k, b in net.named_buffers():
                    if name != 'rn20' or k.count('.') == 2 or '_architecture.3.0' in k or '_architecture.5.2' in k:
                        out[f'{tag}.buf.{k}'] = b.numpy().copy()
        out['meta'] = np.array([n, hw, classes, 40 + mi, 700 + mi, 800 + mi, int(nesterov)])
        save('g4_' + name, **out)


# ------------------------------------------------------------------------------------------- G5
def g5_grammar():
    out = {}
    for name, cfg in SPECS.items():
        net = ResNet(architecture_spec=cfg['spec'], preact=cfg['preact'], use_proj=cfg['use_proj'], dropout_prob=0.0)
        sd = net.state_dict()
        out[name + '.keys'] = np.array(list(sd.keys()))
        out[name + '.shapes'] = np.array([','.join(map(str, v.shape)) for v in sd.values()])
        out[name + '.nparams'] = np.array(sum(p.numel() for p in net.parameters()))
        out[name + '.param_keys'] = np.array([k for k, _ in net.named_parameters()])
        if name in ('rn20',):
            # init statistics of the reference initialiser (resnet.py:160-163): stem std, block-conv max-abs
            torch.manual_seed(0)
            net = ResNet(architecture_spec=cfg['spec'], preact=cfg['preact'], use_proj=cfg['use_proj'], dropout_prob=0.0)
            out[name + '.init_stem_std'] = np.array(net.state_dict()['_architecture.0.weight'].std().item())
            out[name + '.init_block_absmax'] = np.array(net.state_dict()['_architecture.3.0._conv1.weight'].abs().max().item())
    save('g5_grammar', **out)


# ------------------------------------------------------------------------------------------- G6
def g6_metrics():
    """no-tie logits pin loss / top-1 / top-5 / dlogits exactly; a second set with exact ties records what
    torch.topk (implementation-defined tie order) returned, for a bounds check only."""
    out = {}
    logits = fill((16, 10), 900, 3.0)
    labels = fill_labels(16, 10, 901)
    m = compute_losses_and_metrics(T(logits), T(labels))
    out.update(logits=logits, labels=labels, loss=m['loss'].numpy(), top1_err=m['top1_err'].numpy(), top5_err=m['top5_err'].numpy())
    lg = T(logits).requires_grad_(True)
    compute_losses_and_metrics(lg, T(labels))['loss'].backward()
    out['dlogits'] = lg.grad.numpy()
    tied = logits.copy()
    tied[3, :] = 0.5                                      # all tied
    tied[5, 2] = tied[5, 7] = tied[5].max() + 1.0         # two-way tie at the top
    tlabels = labels.copy(); tlabels[5] = 7
    mt = compute_losses_and_metrics(T(tied), T(tlabels))
    out.update(tied_logits=tied, tied_labels=tlabels, tied_loss=mt['loss'].numpy(), tied_top1_err=mt['top1_err'].numpy(),
               tied_top5_err=mt['top5_err'].numpy())
    save('g6_metrics', **out)


# ------------------------------------------------------------------------------------------- G7
def g7_syncbn():
    """plain BN of the reference on the concatenated batch (2 ranks x 4 images): what SyncBN must reproduce."""
    C, H = 8, 4
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data = T(fill((C,), 11, 0.25, 1.0)); bn.bias.data = T(fill((C,), 12, 0.2))
    x = T(fill((8, C, H, H), 13, 1.5, -0.2)).requires_grad_(True)
    bn.train()
    y = bn(x)
    dy = T(fill(tuple(y.shape), 14))
    y.backward(dy)
    save('g7_syncbn', y=nhwc(y), dx=nhwc(x.grad), dgamma=bn.weight.grad.numpy(), dbeta=bn.bias.grad.numpy(),
         running_mean=bn.running_mean.numpy(), running_var=bn.running_var.numpy(), meta=np.array([8, C, H]))


# ------------------------------------------------------------------------------------------- G8
def g8_eval_loop():
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = '29431'
    torch.distributed.init_process_group('gloo', world_size=1, rank=0)
    cfg = MODELS['wrn_small']
    net = ResNet(architecture_spec=cfg['spec'], preact=cfg['preact'], use_proj=cfg['use_proj'], dropout_prob=0.3)
    load_filled(net, 41)
    batches = [(T(fill((4, 3, 32, 32), 1000 + b)), T(fill_labels(4, 10, 1100 + b))) for b in range(2)]
    m = evaluation_loop(world_size=1, device='cpu', dl_test=batches, classifier=net)
    torch.distributed.destroy_process_group()
    save('g8_eval', loss=np.array(m['loss']), top1_err=np.array(m['top1_err']), top5_err=np.array(m['top5_err']))


# ------------------------------------------------------------------------------------------- G9
def g9_checkpoint():
    """a checkpoint written by the REFERENCE's own code path: DistributedDataParallel-wrapped classifier (script.py:64, gloo,
    world_size 1) + torch.optim.SGD, three training steps (training.py:92-113), then resnet.utils.checkpoint_util.save_checkpoints
    -> tests/golden/ckpt/{classifier,optimizer}_3.pth (the files themselves are the fixture: `module.`-prefixed keys, KCRS-contiguous
    weights, SGD momentum buffers).  Expected values after resuming: eval-mode logits/metrics and the fourth training step."""
    import shutil
    from resnet.utils.checkpoint_util import save_checkpoints
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = '29432'
    torch.distributed.init_process_group('gloo', world_size=1, rank=0)
    cfg = MODELS['wrn_small']
    net = ResNet(architecture_spec=cfg['spec'], preact=cfg['preact'], use_proj=cfg['use_proj'], dropout_prob=0.0)
    load_filled(net, 41)
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    opt = torch.optim.SGD(ddp.parameters(), lr=0.1, momentum=0.9, dampening=0.0, nesterov=True, weight_decay=5e-4)
    x, y = T(fill((4, 3, 32, 32), 701)), T(fill_labels(4, 10, 801))
    ddp.train()
    for step in range(3):
        loss = compute_losses_and_metrics(ddp(x), y)['loss']
        loss.backward()
        opt.step(); opt.zero_grad()
    ck = os.path.join(HERE, 'ckpt')
    shutil.rmtree(ck, ignore_errors=True)
    save_checkpoints(checkpoint_dir=ck, checkpointables={'classifier': ddp, 'optimizer': opt, 'scheduler': None}, steps=3)
    out = {}
    ddp.eval()
    with torch.no_grad():
        lg = ddp(x)
        m = compute_losses_and_metrics(lg, y)
    out.update(eval_logits=lg.numpy().copy(), eval_loss=m['loss'].numpy().copy(), eval_top1=m['top1_err'].numpy().copy())
    ddp.train()
    loss = compute_losses_and_metrics(ddp(x), y)['loss']
    loss.backward()
    opt.step(); opt.zero_grad()
    out['step4.loss'] = loss.detach().numpy().copy()
    out['step4.param_sums'] = np.array([p.detach().double().sum().item() for p in net.parameters()])
    out['step4.param_abs_sums'] = np.array([p.detach().double().abs().sum().item() for p in net.parameters()])
    out['step4.mom_sums'] = np.array([opt.state[p]['momentum_buffer'].double().sum().item() for p in ddp.parameters()])
    out['keys'] = np.array(list(ddp.state_dict().keys()))
    torch.distributed.destroy_process_group()
    save('g9_checkpoint', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9']
    fns = dict(g1=g1_conv, g2=g2_bn, g3=g3_blocks, g4=g4_models, g5=g5_grammar, g6=g6_metrics, g7=g7_syncbn, g8=g8_eval_loop, g9=g9_checkpoint)
    for w in which:
        fns[w]()
