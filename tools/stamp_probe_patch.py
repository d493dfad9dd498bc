"""In-kernel s_memrealtime stamps of the LDS-patch 3x3 kernel on one shape (diagnostic): where a tile's life goes."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_ddp_resnet_amd import _lib
L = _lib.lib()
vp = C.c_void_p
L.rn_conv_fwd.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(_lib.RnConvGeom), vp, vp]
L.rn_set_stamp_buffer.argtypes = [vp]
L.rn_set_variant.argtypes = [C.c_int]
L.rn_set_variant(int(os.environ.get('RN_VARIANT', '0')))
N, H, W, Cc, K, ks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '128,32,32,160,160,3').split(',')]
g = _lib.RnConvGeom(N, H, W, Cc, H, W, K, ks, ks, 1, ks // 2)
dt = torch.float16
x = torch.randn(N, H, W, Cc, device='cuda').to(dt); w = (torch.randn(K, ks * ks, Cc, device='cuda') * 0.05).to(dt)
y = torch.empty(N, H, W, K, device='cuda', dtype=dt)
st = vp(torch.cuda.current_stream().cuda_stream)
grid = ((N * H * W + 127) // 128) * ((K + 159) // 160 if K % 160 == 0 else (K + 127) // 128)
buf = torch.zeros(grid * 16 + 64, dtype=torch.int64, device='cuda')
for i in range(5):
    if i == 4:
        L.rn_set_stamp_buffer(buf.data_ptr())
    _lib.check(L.rn_conv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, 0, 0, 2, C.byref(g), None, st))
torch.cuda.synchronize()
L.rn_set_stamp_buffer(None)
s = buf[:grid * 16].view(grid, 16).cpu().double()
t0 = s[:, 0].min()
print('workgroups', grid, ' kernel span %.1f us' % ((s[:, 6].max() - t0) / 100))
seq = [(0, 'entry'), (7, 'setup done'), (8, 'patch + 2 weight tiles issued'), (1, 'patch + tile 0 landed'), (11, 'chunk 0 done (9 steps)'), (12, 'patch 1 landed'),
       (13, 'chunk 1 done'), (14, 'patch 2 landed'), (2, 'K loop done'), (3, 'h0 parked'), (4, 'h0 stored'), (5, 'h1 parked'), (6, 'h1 stored')]
for r in (s[s[:, 0] < s[:, 0].median()], s[s[:, 0] >= s[:, 0].median()]):
    print('--- group of', len(r), 'workgroups: start %.1f us' % ((r[:, 0].mean() - t0) / 100))
    prev = 0
    for slot, name in seq[1:]:
        if (r[:, slot] == 0).all():
            continue
        d = (r[:, slot] - r[:, prev]) / 100
        print(f'  -> {name:<32s}: mean {d.mean():6.2f} us  min {d.min():6.2f}  max {d.max():6.2f}')
        prev = slot
    print('  total %.2f us' % ((r[:, 6] - r[:, 0]).mean() / 100))
