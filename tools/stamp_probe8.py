"""In-kernel stamps of the eight-phase convolution kernel (conv_igemm8.hip) on one shape: where a workgroup's SECOND tile spends its time (diagnostic).
usage: python tools/stamp_probe8.py N,H,W,C,K,k"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_ddp_resnet_amd import _lib
L = _lib.lib()
vp = C.c_void_p
L.rn_conv_fwd.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(_lib.RnConvGeom), vp, vp]
L.rn_set_stamp_buffer.argtypes = [vp]
L.rn_set_variant.argtypes = [C.c_int]
L.rn_set_variant(1 << 22)
N, H, W, Cc, K, ks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '256,28,28,256,1024,1').split(',')]
g = _lib.RnConvGeom(N, H, W, Cc, H, W, K, ks, ks, 1, ks // 2)
dt = torch.float16
x = torch.randn(N, H, W, Cc, device='cuda').to(dt); w = (torch.randn(K, ks * ks, Cc, device='cuda') * 0.05).to(dt)
y = torch.empty(N, H, W, K, device='cuda', dtype=dt)
st = vp(torch.cuda.current_stream().cuda_stream)
grid = 256
buf = torch.zeros(grid * 16 + 64, dtype=torch.int64, device='cuda')
for i in range(5):
    if i == 4:
        L.rn_set_stamp_buffer(buf.data_ptr())
    _lib.check(L.rn_conv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, 0, 0, 2, C.byref(g), None, st))
torch.cuda.synchronize()
L.rn_set_stamp_buffer(None)
s = buf[:grid * 16].view(grid, 16).cpu().double()
names = ['loop top', 'roles computed', 'prologue issued', 'epilogue done', 'K tile 0 landed', 'K loop done']
print(f'shape {N}x{H}x{W} C{Cc}->K{K} k{ks}: second tile of {grid} workgroups (100 MHz ticks -> us)')
for i in range(1, 6):
    d = (s[:, i] - s[:, i - 1]) / 100
    print(f'  {names[i - 1]:>16s} -> {names[i]:<16s}: mean {d.mean():7.2f} us  min {d.min():7.2f}  max {d.max():7.2f}')
print('  total %.2f us' % ((s[:, 5] - s[:, 0]).mean() / 100))
