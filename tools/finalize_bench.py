"""Launch time of the BatchNorm-backward finalize kernel, plain vs rows-split (rn_bn_bwd_finalize / rn_bn_bwd_finalize_split), over the
(partial rows, channels) sizes the configurations produce.  One process, HIP events on the launch stream, cold-ish partials (a 1 GB sweep between
timed batches would be closer to the model; here the buffers of all sizes together exceed the L2s).  Usage: python tools/finalize_bench.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, '.')
from pytorch_ddp_resnet_amd import _lib  # noqa: E402

L = _lib.lib()
L.rn_bn_fold_bytes.argtypes = [C.c_int, C.c_int]
L.rn_bn_fold_bytes.restype = C.c_size_t
vp = C.c_void_p
L.rn_bn_bwd_finalize.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]
L.rn_bn_bwd_finalize_split.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp]
dev = torch.device('cuda', 0)
st = torch.cuda.current_stream().cuda_stream
sizes = [(1024, 160), (256, 320), (1024, 16), (1568, 256), (1568, 1024), (392, 2048), (392, 512), (6272, 128), (6272, 512), (25088, 512), (8192, 512), (2048, 512), (98, 4096)]
bufs = {}
for nblk, ch in sizes:
    p = torch.randn(nblk, 2, ch, device=dev)
    nb = int(L.rn_bn_fold_bytes(nblk, ch))
    bufs[(nblk, ch)] = (p, torch.zeros(2, ch, device=dev), torch.zeros(ch, device=dev), torch.zeros(ch, device=dev), torch.zeros(max(nb, 4) // 4, device=dev), nb)


def run(nblk, ch, split):
    p, ds, dg, db, fold, nb = bufs[(nblk, ch)]
    if split:
        _lib.check(L.rn_bn_bwd_finalize_split(p.data_ptr(), nblk, ds.data_ptr(), dg.data_ptr(), db.data_ptr(), ch, 0, fold.data_ptr(), nb, st))
    else:
        _lib.check(L.rn_bn_bwd_finalize(p.data_ptr(), nblk, ds.data_ptr(), dg.data_ptr(), db.data_ptr(), ch, 0, st))


for nblk, ch in sizes:
    nb = bufs[(nblk, ch)][5]
    out = []
    for split in (False, True):
        if split and not nb:
            out.append(None)
            continue
        run(nblk, ch, split)
        ref = bufs[(nblk, ch)][1].clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        tot = 0.0
        for _ in range(reps):                      # another size's partials between the launches: this size's rows are not L2-warm
            for o_n, o_c in sizes[:4]:
                if (o_n, o_c) != (nblk, ch):
                    run(o_n, o_c, False)
            e0.record(); run(nblk, ch, split); e1.record(); e1.synchronize()
            tot += e0.elapsed_time(e1)
        out.append((tot / reps * 1e3, ref))
    mb = nblk * 2 * ch * 4 / 1e6
    a = out[0][0]
    if out[1] is None:
        print(f'rows {nblk:6d} C {ch:5d} {mb:7.1f} MB  plain {a:7.1f} us   (not split)')
    else:
        same = torch.allclose(out[0][1], out[1][1], rtol=1e-6, atol=1e-4)
        print(f'rows {nblk:6d} C {ch:5d} {mb:7.1f} MB  plain {a:7.1f} us   split {out[1][0]:7.1f} us   equal {same}')
