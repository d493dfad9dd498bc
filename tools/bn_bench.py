"""Micro-benchmark of the BatchNorm apply passes on the WRN-28-10 shapes, rotating over enough buffer sets to defeat the
256 MB Infinity Cache (A/B of kernel variants inside one process).
usage: python tools/bn_bench.py [iters]   env RN_BN_VARIANT selects library variants (rn_set_variant)."""
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_ddp_resnet_amd import _lib

L = _lib.lib()
vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
L.rn_bn_apply.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, C.c_uint32, C.c_uint64, vp]
L.rn_bn_bwd_apply.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, C.c_double, f32, C.c_uint32, C.c_uint64, vp]
L.rn_set_variant.argtypes = [i32]
SHAPES = [(128, 32, 32, 160), (128, 16, 16, 320), (128, 8, 8, 640), (128, 32, 32, 16)]
F_RELU, F_TRAIN, F_RECOMP = 1, 2, 128


def main(iters, variants):
    from pytorch_ddp_resnet_amd.engine import ir
    relu, train, recomp = ir.F_RELU, ir.F_TRAIN, ir.F_MASK_RECOMPUTE
    st = vp(torch.cuda.current_stream().cuda_stream)
    dt = torch.bfloat16
    for (N, H, W, Cc) in SHAPES:
        M = N * H * W
        nset = max(2, int(600e6 / (M * Cc * 2 * 4)) + 1)
        sets = [[torch.randn(M, Cc, device='cuda').to(dt) for _ in range(4)] for _ in range(nset)]
        coef = torch.randn(4, Cc, device='cuda').abs() + 0.5
        dsum = torch.randn(2, Cc, device='cuda')
        for name in ('apply', 'bwd_apply', 'bwd_apply+add', 'bwd_apply+mask'):
            for v in variants:
                L.rn_set_variant(v)

                def call(i):
                    x, d, a, o = sets[i % nset]
                    if name == 'apply':
                        _lib.check(L.rn_bn_apply(x.data_ptr(), coef.data_ptr(), None, o.data_ptr(), 1, N, H, W, Cc, 0, 0, relu, 0.0, 0, 0, st))
                    else:
                        fl = relu | train | (0 if name.endswith('mask') else recomp)
                        _lib.check(L.rn_bn_bwd_apply(d.data_ptr(), x.data_ptr(), a.data_ptr() if name.endswith('mask') else None, coef.data_ptr(), dsum.data_ptr(),
                                                     a.data_ptr() if name.endswith('add') else None, o.data_ptr(), None, 1, N, H, W, Cc, 1 if name.endswith('add') else 0, 0,
                                                     fl, 1.0, float(M), 0.0, 0, 0, st))
                for i in range(3):
                    call(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(iters):
                    call(i)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / iters * 1e3
                nt = {'apply': 2, 'bwd_apply': 3, 'bwd_apply+add': 4, 'bwd_apply+mask': 4}[name]
                print(f'{name:15s} M{M} C{Cc} variant {v}: {us:7.1f} us  {nt * M * Cc * 2 / us / 1e6:6.2f} TB/s', flush=True)


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, [int(v) for v in os.environ.get('RN_BN_VARIANT', '0').split(',')])
