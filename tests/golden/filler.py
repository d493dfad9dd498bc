"""
Closed-form, platform-portable pseudo-random filler used by the golden-vector generator AND by the tests,
so that no weight/input blobs need to be committed: only the reference's OUTPUTS are stored in the .npz files.

fill(shape, seed) -> float32 array in [-1, 1): v_i = 2*frac(sin(i*12.9898 + seed*78.233) * 43758.5453) - 1,
evaluated in float64 (libm differences of 1 ulp move v by ~1e-11, far below float32 resolution).
"""

import numpy as np


def fill(shape, seed, scale=1.0, offset=0.0):
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * 12.9898 + float(seed) * 78.233) * 43758.5453
    v = 2.0 * (v - np.floor(v)) - 1.0
    return (v * scale + offset).astype(np.float32).reshape(shape)


def fill_state(shapes, seed=0):
    """shapes: ordered [(key, shape)] in reference state_dict order -> dict key -> float32 / int64 array.
    Conv / linear weights ~ U(-1,1)*sqrt(3/fan_in) (unit-variance-preserving), biases small, BN affine and
    running statistics deliberately non-trivial."""
    table = dict(shapes)
    st = {}
    for j, (key, shape) in enumerate(shapes):
        prefix, leaf = key.rsplit('.', 1)
        s = seed * 1000 + j
        is_bn = (prefix + '.running_mean') in table
        if leaf == 'num_batches_tracked':
            st[key] = np.zeros((), dtype=np.int64)
        elif is_bn and leaf == 'weight':
            st[key] = fill(shape, s, 0.25, 1.0)
        elif is_bn and leaf == 'bias':
            st[key] = fill(shape, s, 0.2)
        elif leaf == 'running_mean':
            st[key] = fill(shape, s, 0.1)
        elif leaf == 'running_var':
            st[key] = fill(shape, s, 0.25, 1.0)
        elif leaf == 'weight':
            fan_in = int(np.prod(shape[1:]))
            st[key] = fill(shape, s, (3.0 / fan_in) ** 0.5)
        else:
            st[key] = fill(shape, s, 0.1)
    return st


def fill_labels(n, classes, seed):
    return (np.floor((fill((n,), seed) * 0.5 + 0.5) * classes).astype(np.int64)) % classes
