"""In-kernel s_memtime stamps of the implicit-GEMM forward kernel on one WRN-28-10 shape (diagnostic)."""
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
dt = torch.bfloat16
x = torch.randn(N, H, W, Cc, device='cuda').to(dt); w = (torch.randn(K, ks * ks, Cc, device='cuda') * 0.05).to(dt)
y = torch.empty(N, H, W, K, device='cuda', dtype=dt)
st = vp(torch.cuda.current_stream().cuda_stream)
grid = ((N * H * W + 127) // 128) * ((K + 159) // 160 if K % 160 == 0 else (K + 127) // 128)
buf = torch.zeros(grid * 16 + 64, dtype=torch.int64, device='cuda')
for i in range(5):
    if i == 4:
        L.rn_set_stamp_buffer(buf.data_ptr())
    _lib.check(L.rn_conv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, 0, 0, 1, C.byref(g), None, st))
torch.cuda.synchronize()
L.rn_set_stamp_buffer(None)
s = buf[:grid * 16].view(grid, 16).cpu().double()
t0 = s[:, 0].min()
print('entry stamps: min %.0f  p1 %.0f  median %.0f  max %.0f (ticks rel. to min)' % (0, s[:, 0].kthvalue(10).values - t0, s[:, 0].median() - t0, s[:, 0].max() - t0))
names = ['entry', 'loop start', 'loop end', 'h0 parked', 'h0 stored', 'h1 parked', 'h1 stored']
print('workgroups', grid, ' (100 MHz ticks -> us = ticks/100)' )
for r in (s[s[:, 0] < s[:, 0].median()], s[s[:, 0] >= s[:, 0].median()]):
    print('--- group of', len(r), 'workgroups: start %.1f us' % ((r[:, 0].mean() - t0) / 100))
    print('  entry -> descriptors %.2f us -> tap tables %.2f us -> rows decoded %.2f us' % (((r[:, 10] - r[:, 0]) / 100).mean(), ((r[:, 9] - r[:, 10]) / 100).mean(), ((r[:, 7] - r[:, 9]) / 100).mean()))
    print('  entry -> setup done %.2f us -> DMA issued %.2f us -> tile 0 landed %.2f us' % (((r[:, 7] - r[:, 0]) / 100).mean(), ((r[:, 8] - r[:, 7]) / 100).mean(), ((r[:, 1] - r[:, 8]) / 100).mean()))
    for i in range(1, 7):
        d = (r[:, i] - r[:, i - 1]) / 100
        print(f'  {names[i - 1]:>10s} -> {names[i]:<10s}: mean {d.mean():7.2f} us  min {d.min():7.2f}  max {d.max():7.2f}')
    print('  total %.2f us' % ((r[:, 6] - r[:, 0]).mean() / 100))
print('kernel span %.1f us' % ((s[:, 6].max() - t0) / 100))
