"""
bench.py -- images/sec of one training microbatch of the hot path (forward + loss + backward + gradient all-reduce;
optimizer step excluded) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (default = the configuration BASELINE.json's metric is quoted on): WRN-28-10, CIFAR-10 shapes, batch 128 per
GPU, dropout 0.3 as in /root/reference/models_dir/wrn-28-10-dropout_cifar10/config.yaml:15-18; synthetic N(0,1) NCHW
fp32 batch resident in HBM (one fixed batch per rank, seed 1234+rank), random-init weights under manual_seed(0).
Prints ONE JSON line (rank 0).  `roofline` is the MFMA roofline of the dominant kernel (the implicit-GEMM convolution
launches: forward + dgrad + wgrad), measured with per-launch HIP events on the launch stream in an instrumented pass
of the same steps; `cpu_baseline` times the torch-CPU port of the reference step (oracle/torch_model.py) on the host.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: spec, preact, use_proj, dropout, classes, HxW, per-GPU batch, cpu-baseline batch
    'rn20':      dict(spec='c3,16,3,1,1 n a r3 r3 r3 ap8,1,0 fc64,10', preact=False, use_proj=False, p=0.0, classes=10, hw=32, batch=128, cpu_batch=128),
    'wrn-28-10': dict(spec='c3,160,3,1,1 r4 r4 r4 n a ap8,1,0 fc640,10', preact=True, use_proj=True, p=0.3, classes=10, hw=32, batch=128, cpu_batch=16),
    'v2-164':    dict(spec='c3,64,3,1,1 b18 b18 b18 n a ap8,1,0 fc256,100', preact=True, use_proj=True, p=0.0, classes=100, hw=32, batch=128, cpu_batch=16),
    'wrn-50-2a': dict(spec='c3,256,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc2048,1000', preact=False, use_proj=True, p=0.0, classes=1000, hw=224, batch=256, cpu_batch=4),
    'wrn-50-2b': dict(spec='c3,512,7,2,3 n a mp3,2,1 b3 b4 b6 b3 ap7,1,0 fc4096,1000', preact=False, use_proj=True, p=0.0, classes=1000, hw=224, batch=256, cpu_batch=2),
}

PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp32': 157.3}      # MI355X dense MFMA peaks (guides/MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def conv_flops(op):
    """ALGORITHMIC flops of a convolution launch (2 per MAC of the reference's layer): the stem runs on the MFMA path with its 3 image
    channels zero-padded to one 16-byte chunk (C = 4 / 8 in the op) -- the padding is not work."""
    d = op.dim
    if d.get('alg_macs'):                                # the space-to-depth stem: the lowering records the reference layer's MACs per output (7 * 7 * 3)
        return 2.0 * d['N'] * d['P'] * d['Q'] * d['K'] * d['alg_macs']
    c = 3 if d['C'] <= 8 else d['C']
    return 2.0 * d['N'] * d['P'] * d['Q'] * d['K'] * d['R'] * d['S'] * c


def conv_bytes(op, ir, elem):
    """algorithmic HBM bytes of one convolution launch: every operand read once, every result written once (weights included;
    fused-epilogue operands -- residual, accumulate target, BatchNorm-backward x and mask -- counted where the op has them)."""
    d, b = op.dim, op.buf
    x_, y_, w_ = d['N'] * d['H'] * d['W'] * d['C'], d['N'] * d['P'] * d['Q'] * d['K'], d['K'] * d['R'] * d['S'] * d['C']
    if op.kind == ir.OP_CONV_FWD:
        return elem * (x_ + y_ + w_ + (y_ if b.get('res', -1) >= 0 else 0))
    if op.kind == ir.OP_CONV_DGRAD:
        # (a mask the lowering marks as computable from bn_x -- F_MASK_RECOMPUTE -- is no operand: the eight-phase kernels do not read it)
        extra = sum(x_ for k in ('res', 'bn_x', 'bn_mask') if b.get(k, -1) >= 0 and not (k == 'bn_mask' and op.flags & ir.F_MASK_RECOMPUTE)) + (x_ if op.flags & ir.F_ACCUM else 0)
        return elem * (y_ + x_ + w_ + extra)
    return elem * (x_ + y_) + 4 * w_                     # wgrad: fp32 dw


def cpu_baseline(cfg, steps=3):
    """the reference's CPU training loop (torch-CPU port, fp32 NCHW, same region: fwd + loss/metrics + bwd) on a
    bounded sample of the same workload: reduced batch, a few steps."""
    from oracle import torch_model as tm
    torch.manual_seed(0)
    st = tm.make_trainable(tm.init_state(cfg['spec'], cfg['preact'], cfg['use_proj'], seed=0))
    model = tm.TorchResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], cfg['p'])
    gen = torch.Generator().manual_seed(1234)
    b = cfg['cpu_batch']
    x = torch.randn(b, 3, cfg['hw'], cfg['hw'], generator=gen)
    y = torch.randint(0, cfg['classes'], (b,), generator=gen)
    # the reference's best CPU figure, not torch's default thread count: at batch 16 all 128 hardware threads of the GPU box oversubscribe the
    # layers (round 3: 5.8 images/s on 128 threads, 10.8 on the survey's 8): one step per candidate count, then `steps` timed steps at the best
    ncpu = os.cpu_count() or 1
    cands = sorted({t for t in (8, 16, 32, 64, 128, torch.get_num_threads()) if t <= ncpu}) or [torch.get_num_threads()]
    default_threads = torch.get_num_threads()
    tm.train_step(model, st, x, y)                     # warm-up
    sweep = {}
    for t in cands:
        torch.set_num_threads(t)
        tm.train_step(model, st, x, y)
        t0 = time.perf_counter()
        tm.train_step(model, st, x, y)
        sweep[t] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for _ in range(steps):
        tm.train_step(model, st, x, y)
    dt = (time.perf_counter() - t0) / steps
    torch.set_num_threads(default_threads)
    return dict(value=b / dt, unit='images/sec', cores=best, kind='port',
                sample=f'{steps} steps of batch {b} (fwd+loss+bwd, fp32 NCHW torch-CPU port of the reference step), {dt * 1e3:.0f} ms/step on {best} threads '
                       f'(best of a one-step sweep: ' + ', '.join(f'{t}: {v * 1e3:.0f} ms' for t, v in sweep.items()) + ')',
                cpu=_cpu_model(), torch=torch.__version__)


def parity(cfg, dtype, dev):
    """logits / loss of the TIMED engine (same dtype, same kernels) against the torch-CPU port of the reference on a reduced batch
    of the same workload, same weights, BatchNorm in training mode, dropout off (torch's CPU masks cannot be reproduced on the
    device).  The north star asks: logits within 1e-3 relative, identical argmax."""
    from oracle import torch_model as tm
    from pytorch_ddp_resnet_amd import ResNet
    b = cfg['cpu_batch']
    st = tm.init_state(cfg['spec'], cfg['preact'], cfg['use_proj'], seed=0)
    gen = torch.Generator().manual_seed(4321)
    x = torch.randn(b, 3, cfg['hw'], cfg['hw'], generator=gen)
    y = torch.randint(0, cfg['classes'], (b,), generator=gen)
    ref_st = tm.make_trainable({k: v.clone() for k, v in st.items()})
    lg, met, ref_grads = tm.train_step(tm.TorchResNet(cfg['spec'], cfg['preact'], cfg['use_proj']), ref_st, x, y)      # forward + loss + backward, fp32 CPU
    lg, ref_loss = lg.detach(), float(met['loss'])
    m = ResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], 0.0, compute_dtype=dtype)
    m.load_state_dict({k: v.clone() for k, v in st.items()})
    m = m.to(dev).train()
    scale = 1024.0 if dtype == 'fp16' else 1.0
    out = m(x.to(dev))
    loss_t = torch.nn.functional.cross_entropy(out, y.to(dev))
    (loss_t * scale).backward()
    torch.cuda.synchronize()
    logits, loss = out.detach().float().cpu(), float(loss_t.detach())
    gn = sum(float(((p.grad.detach().cpu().double() / scale - ref_grads[k].double()) ** 2).sum()) for k, p in m.named_parameters()) ** 0.5
    rn = sum(float((g.double() ** 2).sum()) for g in ref_grads.values()) ** 0.5
    return dict(logits_rel_err=float((logits - lg).abs().max() / lg.abs().max()), loss_abs_err=abs(loss - ref_loss),
                argmax_equal=bool((logits.argmax(1) == lg.argmax(1)).all()), grad_l2_rel_err=gn / rn, batch=b,
                against='torch-CPU fp32 port of the reference step (oracle/torch_model.py: forward, loss, backward), train-mode BN, dropout 0',
                note='fp16: PyTorch-ROCm autocast(float16) on the same net differs from fp64 by 8.8e-4 (logits) / 4.4e-2 (gradients) at batch 128 '
                     '(tests/test_gpu_configs.py::test_wrn28_10_batch128_16bit)' if dtype == 'fp16' else None)


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def self_launch(n):
    """one rank per GPU through torch.distributed.run (rendezvous on 127.0.0.1, a free port), same arguments; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='wrn-28-10', choices=list(WORKLOADS))
    ap.add_argument('--dtype', default='fp16', choices=['fp16', 'bf16', 'fp32'],
                    help='engine arithmetic: fp16 = the reference\'s own GPU arithmetic (autocast + GradScaler, script.py:63), fp32 = exact-f32 MFMA parity mode')
    ap.add_argument('--also', default='fp32', help='comma-separated dtypes measured in short side runs and reported under "other_dtypes" (N=1 only; "" = none)')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: the workload\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--breakdown', action='store_true', help='print the per-op-kind time table to stderr')
    ap.add_argument('--sync-bn', action='store_true')
    ap.add_argument('--optimizer', default='none', choices=['none', 'fused', 'torch'],
                    help='also run the reference\'s SGD update (config.yaml:22-28) inside the timed step; the headline metric is fwd+bwd: none')
    ap.add_argument('--dropout', type=float, default=None, help='override the workload\'s dropout probability (diagnostics)')
    ap.add_argument('--per-op', type=int, default=0, help='with --breakdown: list the N slowest single ops with their geometry')
    ap.add_argument('--host-input', action='store_true',
                    help='diagnostic (never the headline): the batch starts in pinned host memory and crosses PCIe inside every timed step, as with the reference\'s DataLoader + x.to(device) (training.py:93-94)')
    args = ap.parse_args()

    cfg = dict(WORKLOADS[args.workload])
    if args.batch:
        cfg['batch'] = args.batch
    if args.dropout is not None:
        cfg['p'] = args.dropout
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` started as the N=1 command is: launch the N ranks ourselves, as CHILD processes of torch.distributed.run, BEFORE this
        # process has touched the GPU (nothing above initialises HIP), relay their output and exit with their code -- never re-exec a process
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    # rehearsal hook for a one-GPU box: RN_BENCH_REHEARSAL=1 puts every rank on cuda:0 and moves the collectives through
    # gloo, so the multi-rank control flow of this file can be exercised without a second GPU (never set by the driver)
    rehearsal = os.environ.get('RN_BENCH_REHEARSAL', '0') == '1'
    dev = torch.device('cuda', 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        if rehearsal:
            torch.distributed.init_process_group('gloo')
        else:
            torch.distributed.init_process_group('nccl', device_id=dev)    # RCCL over xGMI

    res = measure(args, cfg, args.dtype, dev, world, rank, args.steps, args.warmup, main_run=True)

    if rank == 0:
        out = {
            'metric': 'images/sec fwd+bwd (whole node)', 'value': round(res['value'], 1), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(res['ms'], 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f"{args.workload} CIFAR-{cfg['classes']} 32x32" if cfg['hw'] == 32 else f"{args.workload} 224x224", 'architecture_spec': cfg['spec'],
                       'preact': cfg['preact'], 'use_proj': cfg['use_proj'], 'dropout_prob': cfg['p'], 'batch_per_gpu': cfg['batch'],
                       'global_batch': cfg['batch'] * world, 'parallelism': f'dp{world}',
                       'timed_region': 'fwd + CE loss/top-k + bwd + grad all-reduce' + (' (no optimizer step)' if args.optimizer == 'none' else f' + SGD step ({args.optimizer})'),
                       'sync_bn': bool(args.sync_bn), 'loss_scale': res['loss_scale'], 'grads_finite': res['grads_finite'], **({'host_input': True} if args.host_input else {})},
            'step_ms_spread': res['spread'],
            'roofline': res['roof'],
        }
    if world == 1:
        # the same workload on the other engines, short runs: the fp32 engine (exact-f32 MFMA) is the parity engine, so the record
        # always carries a figure at the reference's own CPU precision next to the headline
        others = {}
        for dt_ in [d for d in args.also.split(',') if d and d != args.dtype]:
            r = measure(args, cfg, dt_, dev, world, rank, max(3, min(args.steps, 8)), 2, main_run=False)
            others[dt_] = {'value': round(r['value'], 1), 'unit': 'images/sec', 'ms_per_step': round(r['ms'], 3), 'roofline': r['roof']}
            if not args.no_parity:
                others[dt_]['parity'] = parity(cfg, dt_, dev)
        if others:
            out['other_dtypes'] = others
        if not args.no_parity:
            out['parity'] = parity(cfg, args.dtype, dev)
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def measure(args, cfg, dtype, dev, world, rank, steps, warmup, main_run):
    """W untimed + K timed steps of the workload on the `dtype` engine, then the instrumented pass for the roofline record."""
    from pytorch_ddp_resnet_amd import ResNet
    from pytorch_ddp_resnet_amd.engine import ir
    from pytorch_ddp_resnet_amd.ddp import GradReducer
    from pytorch_ddp_resnet_amd.algos.metrics import compute_losses_and_metrics

    torch.manual_seed(0)                                                       # identical replicas on every rank
    model = ResNet(cfg['spec'], cfg['preact'], cfg['use_proj'], cfg['p'], compute_dtype=dtype, sync_bn=args.sync_bn).to(dev).train()
    gen = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(cfg['batch'], 3, cfg['hw'], cfg['hw'], generator=gen).to(dev)
    y = torch.randint(0, cfg['classes'], (cfg['batch'],), generator=gen).to(dev)
    reducer = GradReducer(model, world) if world > 1 else None
    model.alias_grads = True            # gradients are consumed (dropped) every step: no defensive copy of the flat buffer

    params = list(model.parameters())
    # fp16: the loss is scaled on the device as torch's GradScaler does (training.py:100 scaler.scale(loss).backward()); the scale
    # is found during warm-up by GradScaler's own backoff rule (halve while any gradient is non-finite) and then held fixed
    loss_scale = torch.tensor(65536.0 if dtype == 'fp16' else 1.0, device=dev)
    opt = None
    if args.optimizer != 'none' and main_run:
        sgd_args = dict(lr=0.1, momentum=0.9, dampening=0.0, nesterov=True, weight_decay=5e-4)        # the WRN-28-10 run's optimizer_args
        if args.optimizer == 'fused':
            from pytorch_ddp_resnet_amd.utils.fused_sgd import FusedSGD
            opt = FusedSGD(model, **sgd_args)
        else:
            opt = torch.optim.SGD(model.parameters(), **sgd_args)

    # with an optimizer in the step the fp16 engine runs under a real GradScaler (scaled backward, unscale + skip-on-inf inside the
    # optimizer step, dynamic scale), i.e. training.py:100-110: applying the SCALED gradients would blow the weights up to inf within
    # a few steps -- and a network full of inf / nan runs 6 % "faster" (the chip clocks higher on trivial data)
    scaler = None
    if opt is not None and dtype == 'fp16':
        from pytorch_ddp_resnet_amd.utils.amp import GradScaler
        scaler = torch.amp.GradScaler('cuda') if os.environ.get('RN_TORCH_SCALER', '0') == '1' else GradScaler('cuda')
    x_host = y_host = None
    if getattr(args, 'host_input', False):
        x_host, y_host = x.cpu().pin_memory(), y.cpu().pin_memory()

    def step():
        nonlocal x, y
        for p_ in params:
            p_.grad = None                                   # optimizer.zero_grad() of training.py:113 (set_to_none)
        if x_host is not None:
            x, y = x_host.to(dev, non_blocking=True), y_host.to(dev, non_blocking=True)
        logits = model(x)
        loss = compute_losses_and_metrics(logits, y)['loss']          # CE + top-1/top-5 as one launch (metrics.py:10-29)
        if scaler is not None:
            scaler.scale(loss).backward()
        elif dtype == 'fp16':
            (loss * loss_scale).backward()
        else:
            loss.backward()
        if reducer is not None:
            reducer.finish()
        if scaler is not None:
            scaler.step(opt)
            scaler.update()
        elif opt is not None:
            opt.step()
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(max(warmup, 1)):
        step()
        if dtype == 'fp16' and scaler is None and i < 8:     # loss-scale backoff, outside the timed region
            eng0 = next(e for k, e in model._engines.items() if k[1] and k[2])
            bad = torch.logical_not(torch.isfinite(eng0.flat_grad).all()).float()
            if world > 1:
                torch.distributed.all_reduce(bad)
            if bad.item() > 0:
                loss_scale.mul_(0.5)
    fence()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()                                  # per-step spread only; nothing synchronises inside the region
    fence()
    dt = time.perf_counter() - t0
    eng_t = next(e for k, e in model._engines.items() if k[1] and k[2])
    grads_finite = bool(torch.isfinite(eng_t.flat_grad).all())          # after the timed region: the steps that were timed carried finite gradients
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    ms = dt / steps * 1e3
    value = world * cfg['batch'] * steps / dt
    spread = dict(median=round(per_step[len(per_step) // 2], 3), p10=round(per_step[int(0.1 * (len(per_step) - 1))], 3),
                  p90=round(per_step[int(round(0.9 * (len(per_step) - 1)))], 3), source='HIP events between steps on the launch stream (rank 0)')

    # ---- instrumented pass: per-launch HIP events on the launch stream (same steps).  EVERY rank runs it -- a step contains
    # the gradient collectives, which all ranks must enter -- and rank 0 reports its own measurements ----
    roof = None
    eng = next(e for k, e in model._engines.items() if k[1] and k[2])
    nprof = max(3, min(10, steps))
    agg, conv_t, conv_f, nconv = {}, 0.0, 0.0, 0
    per_op = {}
    roof_t, hbm_bound = 0.0, 0                           # per-launch rooflines: max(FLOPs / MFMA peak, bytes / HBM peak)
    elem = 4 if dtype == 'fp32' else 2
    from pytorch_ddp_resnet_amd import _lib
    for _ in range(nprof):
        eng.profile(True)
        step()
        torch.cuda.synchronize()
        for op, t_ms in eng.profile_read():
            name = ir.OP_NAMES[op.kind]
            agg[name] = agg.get(name, 0.0) + t_ms / nprof
            per_op[id(op)] = (op, per_op.get(id(op), (op, 0.0))[1] + t_ms / nprof)
            if op.kind in (ir.OP_CONV_FWD, ir.OP_CONV_DGRAD, ir.OP_CONV_WGRAD):
                conv_t += t_ms
                conv_f += conv_flops(op)
                nconv += 1
                t_m, t_h = conv_flops(op) / (PEAK_TFLOPS[dtype] * 1e12), conv_bytes(op, ir, elem) / (HBM_PEAK_GBS * 1e9)
                roof_t += max(t_m, t_h) * 1e3
                hbm_bound += 1 if t_h > t_m else 0
    eng.profile(False)
    # which convolution kernels the plan launches: the log is per thread and autograd runs the backward on a thread of its own, so the plan's two op
    # ranges are run once more from THIS thread (eagerly, no reducer hooks: every rank does the same, the gradients of this pass are not used)
    _lib.lib().rn_kernel_log(1)
    graphs, eng.use_graphs = eng.use_graphs, False
    eng.forward(step_seed=0)
    eng.backward(step_seed=0)
    torch.cuda.synchronize()
    eng.use_graphs = graphs
    kernels = sorted(set(n for n in _lib.lib().rn_kernel_log_read().decode().split(',') if n and 'reduce' not in n))
    _lib.lib().rn_kernel_log(0)
    if rank == 0:
        achieved = conv_f / (conv_t * 1e-3) / 1e12 if conv_t > 0 else 0.0
        peak = PEAK_TFLOPS[dtype]
        # HBM-side bytes per conv launch: from the committed PMC passes of this command (FETCH_SIZE x2 + WRITE_SIZE, separate
        # rocprofv3 --pmc runs, tools/summarize_profiles.py); a PMC pass cannot run inside the timed process
        traffic, traffic_src = None, None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'latest_pmc_summary.json')
        if os.path.exists(pmc):
            ks = json.load(open(pmc))
            if ks.get('workload', 'wrn-28-10') == args.workload and ks.get('dtype', 'bf16') == dtype:
                tot_b = tot_n = 0.0
                for name, e in ks['kernels'].items():
                    if name.startswith(('igemm_', 'wgrad_kernel')) and 'hbm_read_MB_per_launch' in e:
                        n_l = e['launches_in_trace']
                        tot_b += n_l * (e['hbm_read_MB_per_launch'] + e.get('hbm_write_MB_per_launch', 0.0)) * 1e6
                        tot_n += n_l
                if tot_n:
                    traffic, traffic_src = int(tot_b / tot_n), f"profiles/{ks['tag']}_pmc_summary.json"
        roof = dict(bound='mfma', achieved=round(achieved, 1), peak=peak, unit='TFLOP/s', frac=round(achieved / peak, 4), traffic=traffic,
                    traffic_unit='bytes per conv launch (HBM side)', traffic_source=traffic_src,
                    kernel='implicit-GEMM convolution launches (forward, dgrad, wgrad): ' + ' '.join(kernels),
                    launches_per_step=nconv // nprof, avg_launch_ms=round(conv_t / max(nconv, 1), 4),
                    algorithmic_gflop_per_launch=round(conv_f / max(nconv, 1) / 1e9, 2),
                    conv_ms_per_step=round(conv_t / nprof, 3), all_ops_ms_per_step=round(sum(agg.values()), 3),
                    # the same launches against their OWN rooflines: several are bound by their output / fused-operand bytes, not by
                    # the matrix pipe (1x1 layers with a short reduction, the 7x7 stem), which a pure MFMA fraction under-reports
                    layer_roofline_ms_per_step=round(roof_t / nprof, 3), frac_of_layer_rooflines=round(roof_t / conv_t, 4) if conv_t > 0 else None,
                    hbm_bound_launches_per_step=hbm_bound // nprof)
        if args.breakdown and main_run:
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
                print(f'  {k:22s} {v:9.3f} ms/step', file=sys.stderr)
            for op, t in sorted(per_op.values(), key=lambda ot: -ot[1])[:args.per_op]:
                d = op.dim
                geo = (f"N{d['N']} {d['H']}x{d['W']} C{d['C']}->K{d['K']} k{d['R']} s{d['stride']}" if 'R' in d else
                       ' '.join(f'{k}{v}' for k, v in d.items() if k in ('N', 'H', 'W', 'C', 'M')))
                extra = f" {conv_flops(op) / (t * 1e-3) / 1e12:7.1f} TFLOP/s" if 'R' in d and t > 0 else ''
                if 'R' in d and t > 0:                  # the launch's own roofline: max(FLOPs / MFMA peak, algorithmic bytes / HBM peak), and the time over it
                    bound = max(conv_flops(op) / (PEAK_TFLOPS[dtype] * 1e12), conv_bytes(op, ir, elem) / (HBM_PEAK_GBS * 1e9)) * 1e6
                    extra += f"  bound {bound:6.1f} us (+{t * 1e3 - bound:6.1f})"
                print(f'    {ir.OP_NAMES[op.kind]:18s} {t * 1e3:9.1f} us  {geo}{extra}  [{op.note}]', file=sys.stderr)
    if world > 1:
        torch.distributed.barrier()
    ls = float(scaler.get_scale()) if scaler is not None else float(loss_scale.item())
    del model, x, y
    torch.cuda.empty_cache()
    return dict(ms=ms, value=value, spread=spread, roof=roof, loss_scale=ls, grads_finite=grads_finite)


if __name__ == '__main__':
    main()
