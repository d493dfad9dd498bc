"""In-kernel stamps of the row-patch convolution kernel (conv_igemm8r.hip) on one shape: where a workgroup's first two tiles spend their time (diagnostic).
usage: python tools/stamp_probe8r.py N,H,W,C,K [probe]     probe: 0 shipped, 1 no LDS-DMA in the K loop, 2 no MFMA, 3 no fragment reads (timing only)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_ddp_resnet_amd import _lib
L = _lib.lib()
vp = C.c_void_p
L.rn_conv_fwd.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(_lib.RnConvGeom), vp, vp]
L.rn_set_stamp_buffer.argtypes = [vp]
N, H, W, Cc, K = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '128,32,32,160,160').split(',')]
probe = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L.rn_set_variant2(2 | (probe << 4))
g = _lib.RnConvGeom(N, H, W, Cc, H, W, K, 3, 3, 1, 1)
dt = torch.float16
x = torch.randn(N, H, W, Cc, device='cuda').to(dt); w = (torch.randn(K, 9, Cc, device='cuda') * 0.05).to(dt)
y = torch.empty(N, H, W, K, device='cuda', dtype=dt)
st = vp(torch.cuda.current_stream().cuda_stream)
grid = min(256, (N * H * W + 255) // 256 * (K // 160))
buf = torch.zeros(grid * 16 + 64, dtype=torch.int64, device='cuda')
for i in range(5):
    if i == 4:
        L.rn_set_stamp_buffer(buf.data_ptr())
    _lib.check(L.rn_conv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, 0, 0, 2, C.byref(g), None, st))
torch.cuda.synchronize()
L.rn_set_stamp_buffer(None)
s = buf[:grid * 16].view(grid, 16).cpu().double()
names = ['loop top', 'roles computed', 'prologue issued', 'epilogue done', 'first K tile landed', 'K loop done']
print(f'shape {N}x{H}x{W} C{Cc}->K{K} probe {probe}: {grid} workgroups (100 MHz ticks -> us)')
for t in range(2):
    if float(s[:, 6 * t + 5].min()) == 0:
        continue
    print(f' tile {t} of a workgroup')
    for i in range(1, 6):
        d = (s[:, 6 * t + i] - s[:, 6 * t + i - 1]) / 100
        print(f'  {names[i - 1]:>20s} -> {names[i]:<20s}: mean {d.mean():7.2f} us  min {d.min():7.2f}  max {d.max():7.2f}')
    print('  tile total %.2f us' % ((s[:, 6 * t + 5] - s[:, 6 * t]).mean() / 100))
t0 = s[:, 0].min()
print(' first stamp spread over workgroups: %.2f us; last K loop end - first stamp: %.2f us' % ((s[:, 0].max() - t0) / 100, (s[:, :12].max() - t0) / 100))
