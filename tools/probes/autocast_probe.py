import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import torch_model as tm
from prod_geoms import CONFIGS
cfg = CONFIGS['wrn-28-10']
st = tm.init_state(cfg['spec'], True, True, seed=0)
gen = torch.Generator().manual_seed(1234)
x = torch.randn(128, 3, 32, 32, generator=gen); y = torch.randint(0, 10, (128,), generator=gen)
t0 = time.time()
res = {}
for name, dt, dev, ac in (('f64cpu', torch.float64, 'cpu', False), ('f32gpu', torch.float32, 'cuda', False), ('ac16gpu', torch.float32, 'cuda', True)):
    tst = tm.make_trainable({k: (v.clone().to(dt).to(dev) if v.is_floating_point() else v.clone().to(dev)) for k, v in st.items()})
    net = tm.TorchResNet(cfg['spec'], True, True)
    t1 = time.time()
    if ac:
        with torch.autocast('cuda', dtype=torch.float16):
            lg = net.forward(tst, x.to(dev), train=True)
            loss = torch.nn.functional.cross_entropy(lg.float(), y.to(dev))
        (loss * 1024.0).backward()
        grads = {k: v.grad.double().cpu() / 1024.0 for k, v in tst.items() if v.requires_grad}
    else:
        lg = net.forward(tst, x.to(dt).to(dev), train=True)
        loss = torch.nn.functional.cross_entropy(lg, y.to(dev))
        loss.backward()
        grads = {k: v.grad.double().cpu() for k, v in tst.items() if v.requires_grad}
    if dev == 'cuda': torch.cuda.synchronize()
    res[name] = (lg.detach().double().cpu(), grads)
    print(name, 'took %.1f s' % (time.time() - t1), flush=True)
l64, g64 = res['f64cpu']
gtot = float(torch.sqrt(sum((g ** 2).sum() for g in g64.values())))
for name in ('f32gpu', 'ac16gpu'):
    lg, g = res[name]
    gerr = float(torch.sqrt(sum(((g[k] - g64[k]) ** 2).sum() for k in g64)))
    print(name, 'logits rel %.3e' % float((lg - l64).abs().max() / l64.abs().max()), 'grad L2 rel %.3e' % (gerr / gtot))
