"""Micro-benchmark of the convolution launchers on the WRN-28-10 shapes (A/B inside one process).
usage: python tools/conv_bench.py [fwd|dgrad|wgrad] [iters]   env RN_CONV_VARIANT selects kernel variants ("v" or "v/v2": rn_set_variant / rn_set_variant2
words; run interleaved, RN_CONV_ROUNDS rounds, median reported: single runs of one kernel differ by up to 10 % on one box), RN_CONV_DTYPE fp16|bf16,
RN_CONV_EP=1: the MODEL's operand sets instead of plain stores (forward: identity residual + fused BatchNorm statistics; data gradient: the
BatchNorm-backward sums over x and the mask tensor), RN_CONV_COLD=1: a 512 MiB fill between launches (operands out of L2 / Infinity Cache)."""
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_ddp_resnet_amd import _lib

L = _lib.lib()
from pytorch_ddp_resnet_amd.engine.executor import ensure_conv_workspace
_WS = ensure_conv_workspace(torch.device('cuda', 0))          # stream-K / wgrad8 workspaces, as an Engine sets them
vp = C.c_void_p
L.rn_conv_fwd.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(_lib.RnConvGeom), vp, vp]
L.rn_conv_dgrad.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_lib.RnConvGeom), vp, vp]
L.rn_conv_wgrad.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(_lib.RnConvGeom), vp]
L.rn_set_variant.argtypes = [C.c_int]
L.rn_set_variant2.argtypes = [C.c_int]


class RnConvEpilogue(C.Structure):          # include/rn_hip.h rn_conv_epilogue
    _fields_ = [('partial', vp), ('bn_x', vp), ('bn_mask', vp), ('bn_coef', vp), ('gscale', C.c_float), ('bias', vp), ('mask_from_x', C.c_int)]


def setv(v):
    L.rn_set_variant(v[0])
    L.rn_set_variant2(v[1])

SHAPES = [(128, 32, 32, 160, 160, 3), (128, 16, 16, 320, 320, 3), (128, 8, 8, 640, 640, 3)]
if os.environ.get('RN_CONV_SHAPES'):          # "N,H,W,C,K,k;..."
    SHAPES = [tuple(int(v) for v in sh.split(',')) for sh in os.environ['RN_CONV_SHAPES'].split(';')]


def run(which, iters, variants, dtype=torch.bfloat16):
    st = vp(torch.cuda.current_stream().cuda_stream)
    for (N, H, W, Cc, K, ks) in SHAPES:
        g = _lib.RnConvGeom(N, H, W, Cc, H, W, K, ks, ks, 1, ks // 2)
        x = torch.randn(N, H, W, Cc, device='cuda').to(dtype)
        w = (torch.randn(K, ks * ks, Cc, device='cuda') * 0.05).to(dtype)
        wd = (torch.randn(Cc, ks * ks, K, device='cuda') * 0.05).to(dtype)
        y = torch.empty(N, H, W, K, device='cuda', dtype=dtype)
        dy = torch.randn(N, H, W, K, device='cuda').to(dtype)
        dx = torch.empty(N, H, W, Cc, device='cuda', dtype=dtype)
        dw = torch.empty(K, ks * ks, Cc, device='cuda')
        wsb = 16                                            # the workspace fits every variant's selection
        for v in [(0, 0)] + list(variants):
            setv(v)
            wsb = max(wsb, int(L.rn_conv_wgrad_ws_bytes(C.byref(g))))
        setv((0, 0))
        use_ep = os.environ.get('RN_CONV_EP') == '1'
        cold = torch.empty(512 << 20, dtype=torch.uint8, device='cuda') if os.environ.get('RN_CONV_COLD') == '1' else None
        rows = (N * H * W + 127) // 128
        st_f = torch.zeros(rows, 2, K, device='cuda'); st_d = torch.zeros(rows, 2, Cc, device='cuda')
        res = torch.randn(N, H, W, K, device='cuda').to(dtype)
        bx = torch.randn(N, H, W, Cc, device='cuda').to(dtype); bm = (torch.rand(N, H, W, Cc, device='cuda') > 0.5).to(dtype)
        coef = torch.rand(4, Cc, device='cuda') + 0.5
        ep_f = RnConvEpilogue(st_f.data_ptr(), None, None, None, 1.0, None, 0)
        ep_d = RnConvEpilogue(st_d.data_ptr(), bx.data_ptr(), bm.data_ptr(), coef.data_ptr(), 1.0 / 0.7, None, 0)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device='cuda')
        rn = {torch.bfloat16: 1, torch.float16: 2, torch.float32: 0}[dtype]
        flops = 2.0 * N * H * W * K * ks * ks * Cc
        ref = {}
        rounds = int(os.environ.get('RN_CONV_ROUNDS', '5'))
        times = {v: [] for v in variants}
        errs = {}
        for rd in range(rounds):                          # variants interleaved: box drift and clock state hit all of them alike
            for v in variants:
                setv(v)

                def call():
                    if which == 'fwd':
                        if use_ep:
                            _lib.check(L.rn_conv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), res.data_ptr(), 1, K, rn, C.byref(g), C.byref(ep_f), st))
                        else:
                            _lib.check(L.rn_conv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, 0, 0, rn, C.byref(g), None, st))
                    elif which == 'dgrad':
                        if use_ep:
                            _lib.check(L.rn_conv_dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, 0, 0, 0, rn, C.byref(g), C.byref(ep_d), st))
                        else:
                            _lib.check(L.rn_conv_dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, 0, 0, 0, rn, C.byref(g), None, st))
                    else:
                        _lib.check(L.rn_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, 0, rn, C.byref(g), st))
                for _ in range(3):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if cold is None:
                    e0.record()
                    for _ in range(iters):
                        call()
                    e1.record()
                    torch.cuda.synchronize()
                    times[v].append(e0.elapsed_time(e1) / iters * 1e3)
                else:                                     # per-launch event pairs around the convolution only
                    tot = 0.0
                    for _ in range(iters):
                        cold.fill_(1)
                        e0.record()
                        call()
                        e1.record()
                        torch.cuda.synchronize()
                        tot += e0.elapsed_time(e1)
                    times[v].append(tot / iters * 1e3)
                if rd == 0:
                    out = (y if which == 'fwd' else dx if which == 'dgrad' else dw).float().clone()
                    if not ref:
                        ref['o'] = out
                    errs[v] = float((out - ref['o']).abs().max()) / (float(ref['o'].abs().max()) + 1e-30)
        for v in variants:
            t = sorted(times[v])
            us = t[len(t) // 2]
            print(f'{which} {N}x{H}x{W} C{Cc}->K{K} k{ks} variant {v}: median {us:7.1f} us (min {t[0]:.1f} max {t[-1]:.1f})  {flops / us / 1e6:7.1f} TFLOP/s  vs first variant: {errs[v]:.2e}', flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    variants = [tuple(int(q) for q in (v.split('/') + ['0'])[:2]) for v in os.environ.get('RN_CONV_VARIANT', '0,1').split(',')]
    run(which, iters, variants, dtype={'bf16': torch.bfloat16, 'fp16': torch.float16}[os.environ.get('RN_CONV_DTYPE', 'fp16')])
