"""forward of WRN-50-2 spec B on 32 distinct 224 x 224 images, train-mode BN: float64 CPU (truth), PyTorch-ROCm fp32 and autocast(float16) on the GPU
(= the reference's own GPU arithmetic), and this engine in fp32 / fp16 -- plus where the fp16 engine's error enters (per-stage activation error vs the
fp32 engine).  usage: python tools/probes/autocast_probe50.py"""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import torch_model as tm
from prod_geoms import CONFIGS
from pytorch_ddp_resnet_amd import ResNet
cfg = CONFIGS['wrn-50-2b']
st = tm.init_state(cfg['spec'], False, True, seed=0)
gen = torch.Generator().manual_seed(4321)
x = torch.randn(32, 3, 224, 224, generator=gen)
net = tm.TorchResNet(cfg['spec'], False, True)
with torch.no_grad():
    t = time.time(); l64 = net.forward({k: (v.double() if v.is_floating_point() else v) for k, v in st.items()}, x.double(), train=True); print('f64 cpu %.0f s' % (time.time() - t), flush=True)
    stg = {k: v.cuda() for k, v in st.items()}
    l32g = net.forward({k: v.clone() for k, v in stg.items()}, x.cuda(), train=True).double().cpu()
    with torch.autocast('cuda', dtype=torch.float16):
        lac = net.forward({k: v.clone() for k, v in stg.items()}, x.cuda(), train=True).double().cpu()
rel = lambda a: float((a - l64).abs().max() / l64.abs().max())
print('torch fp32 GPU %.3e   torch autocast(fp16) GPU %.3e   argmax agree %.3f' % (rel(l32g), rel(lac), float((lac.argmax(1) == l64.argmax(1)).float().mean())))
outs = {}
for dt in ('fp32', 'fp16'):
    m = ResNet(cfg['spec'], False, True, 0.0, compute_dtype=dt); m.load_state_dict({k: v.clone() for k, v in st.items()}); m = m.cuda().train()
    with torch.no_grad():
        lg = m(x.cuda()).double().cpu()
    eng = next(e for k, e in m._engines.items())
    acts = {s.name: eng.tensors[i].detach().float().cpu() for i, s in enumerate(eng.plan.slots) if s.role == 'act' and eng.tensors[i] is not None and eng.tensors[i].dim() == 4}
    outs[dt] = (lg, acts)
    print(dt, 'engine %.3e  argmax agree %.3f' % (rel(lg), float((lg.argmax(1) == l64.argmax(1)).float().mean())), flush=True)
    del m, eng; torch.cuda.empty_cache()
a32, a16 = outs['fp32'][1], outs['fp16'][1]
for name in a32:
    if name in a16 and a32[name].shape == a16[name].shape and ('.out' in name or name.endswith('.h') or 'stem' in name or name.count('.') <= 2):
        d = float((a16[name] - a32[name]).norm() / a32[name].norm().clamp_min(1e-30))
        print(f'  {name:40s} rel L2 diff fp16 vs fp32 engine {d:.3e}')
