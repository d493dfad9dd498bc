"""times rn_augment_batch (BatchTransform) on the GPU and the per-sample oracle chain on one host core (the reference's loader runs
its transforms that way: DataLoader(num_workers=0), data_util.py:218-222).  Prints one JSON line."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from oracle import transforms_ref as ref                      # noqa: E402  (CPU leg of a measurement, never the product path)
from pytorch_ddp_resnet_amd.utils.transform_util import BatchTransform   # noqa: E402

AUG = {'ToTensorTransform': {}, 'StandardizeWhiteningTransform': {}, 'FlipTransform': {'p': 0.5},
       'PaddingTransform': {'pad_size': 4, 'pad_type': 'mirror'}, 'RandomCropTransform': {'crop_size': 32}}
rng = np.random.default_rng(0)
imgs = rng.integers(0, 256, (50000, 32, 32, 3), dtype=np.uint8)
tr = BatchTransform([32, 32, 3], AUG)
tr.fit(imgs)
dev = torch.from_numpy(imgs).cuda()
out = {}
g = torch.Generator(device='cuda').manual_seed(0)
for n in (128, 4096, 50000):
    x = dev[:n]
    for _ in range(3):
        tr(x, generator=g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        tr(x, generator=g, nhwc_dtype=torch.float16, nhwc_channels=4) if '--nhwc' in sys.argv else tr(x, generator=g)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byts = n * (32 * 32 * 3 + 32 * 32 * 3 * 4)
    out[f'gpu_n{n}'] = dict(ms=round(ms, 4), img_per_s=round(n / ms * 1e3), GBps=round(byts / ms / 1e6, 1))
mean, std = tr._image_mean.cpu(), tr._image_stddev.cpu()
f, t, l = (a.cpu().numpy() for a in tr.draw(2000, g))
t0 = time.perf_counter()
for i in range(2000):
    ref.pipeline(imgs[i], AUG, mean, std, bool(f[i]), int(t[i]), int(l[i]))
out['cpu_per_sample_img_per_s'] = round(2000 / (time.perf_counter() - t0))
print(json.dumps(out))
