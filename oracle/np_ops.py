"""
TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the arithmetic on the hot path.

Nothing under ``pytorch_ddp_resnet_amd/`` may import this package: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as the checker.

The reference (lucaslingle/pytorch_ddp_resnet) has no arithmetic of its own: every op on its hot path
is a stock ``torch.nn`` module whose kernel lives in PyTorch ATen (pinned ``torch==1.10.1``,
/root/reference/setup.py:11), which is not vendored under /root/reference.  The formulas below are
therefore a restatement of the *published* semantics of those ATen ops, anchored on the reference's
call sites (cited per function), and pinned by the golden vectors in ``tests/golden/`` which were produced
by importing the reference itself (``tests/golden/make_golden.py``).

Layout: activations are NHWC ``[N, H, W, C]`` (the engine's HBM layout); convolution weights are given in
the reference's KCRS ``[K, C, R, S]`` order.  Everything is computed in the dtype of the inputs
(pass float64 arrays for a high-precision check, float32 to mimic the reference's fp32).
"""

import numpy as np


# ----------------------------------------------------------------------------------------------
# Convolution (reference call sites: resnet.py:69-75 stem Conv2d with bias;
# residual_block.py:34-47,51-57,129-159 block convs, bias=False)
# ----------------------------------------------------------------------------------------------

def _im2col(x, R, S, stride, pad):
    """x: [N,H,W,C] -> cols [N,P,Q,R,S,C] (zero padding), P=(H+2p-R)//s+1."""
    N, H, W, C = x.shape
    P = (H + 2 * pad - R) // stride + 1
    Q = (W + 2 * pad - S) // stride + 1
    xp = np.zeros((N, H + 2 * pad, W + 2 * pad, C), dtype=x.dtype)
    xp[:, pad:pad + H, pad:pad + W, :] = x
    cols = np.empty((N, P, Q, R, S, C), dtype=x.dtype)
    for r in range(R):
        for s in range(S):
            cols[:, :, :, r, s, :] = xp[:, r:r + stride * P:stride, s:s + stride * Q:stride, :]
    return cols, P, Q


def conv2d_fwd(x, w_kcrs, stride, pad, bias=None):
    """y[n,p,q,k] = sum_{c,r,s} x[n, p*stride+r-pad, q*stride+s-pad, c] * w[k,c,r,s] (+ bias[k])."""
    K, C, R, S = w_kcrs.shape
    cols, P, Q = _im2col(x, R, S, stride, pad)
    wm = np.transpose(w_kcrs, (2, 3, 1, 0)).reshape(R * S * C, K)  # [(r,s,c), k]
    y = cols.reshape(-1, R * S * C) @ wm
    y = y.reshape(x.shape[0], P, Q, K)
    if bias is not None:
        y = y + bias.reshape(1, 1, 1, K)
    return y


def conv2d_wgrad(x, dy, R, S, stride, pad):
    """dw[k,c,r,s] = sum_{n,p,q} dy[n,p,q,k] * x[n, p*stride+r-pad, q*stride+s-pad, c]."""
    cols, P, Q = _im2col(x, R, S, stride, pad)
    C = x.shape[3]
    K = dy.shape[3]
    dwm = cols.reshape(-1, R * S * C).T @ dy.reshape(-1, K)          # [(r,s,c), k]
    return np.transpose(dwm.reshape(R, S, C, K), (3, 2, 0, 1)).copy()  # -> KCRS


def conv2d_dgrad(dy, w_kcrs, stride, pad, H, W):
    """dx[n,h,w,c] = sum_{k,r,s : h=p*stride+r-pad, w=q*stride+s-pad} dy[n,p,q,k] * w[k,c,r,s]."""
    K, C, R, S = w_kcrs.shape
    N, P, Q, _ = dy.shape
    wm = np.transpose(w_kcrs, (2, 3, 1, 0)).reshape(R * S * C, K)
    dcols = (dy.reshape(-1, K) @ wm.T).reshape(N, P, Q, R, S, C)
    dxp = np.zeros((N, H + 2 * pad, W + 2 * pad, C), dtype=dy.dtype)
    for r in range(R):
        for s in range(S):
            dxp[:, r:r + stride * P:stride, s:s + stride * Q:stride, :] += dcols[:, :, :, r, s, :]
    return dxp[:, pad:pad + H, pad:pad + W, :].copy()


def bias_grad(dy):
    return dy.reshape(-1, dy.shape[-1]).sum(axis=0)


# ----------------------------------------------------------------------------------------------
# BatchNorm2d (resnet.py:111-112; residual_block.py:58-61,160-165): eps=1e-5, momentum=0.1,
# affine, track_running_stats.  Train: biased batch variance for normalisation, UNBIASED variance
# into running_var.  Eval: running statistics.
# ----------------------------------------------------------------------------------------------

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def bn_batch_stats(x):
    """per-channel (mean, biased var, count) over N*H*W."""
    xm = x.reshape(-1, x.shape[-1])
    m = xm.shape[0]
    mean = xm.sum(axis=0) / m
    var = ((xm - mean) ** 2).sum(axis=0) / m
    return mean, var, m


def bn_train_fwd(x, gamma, beta, running_mean, running_var, eps=BN_EPS, momentum=BN_MOMENTUM):
    """returns y, (mean, invstd), (new_running_mean, new_running_var)."""
    mean, var, m = bn_batch_stats(x)
    invstd = 1.0 / np.sqrt(var + eps)
    y = (x - mean) * invstd * gamma + beta
    unbiased = var * (m / max(m - 1, 1))
    new_rm = (1 - momentum) * running_mean + momentum * mean
    new_rv = (1 - momentum) * running_var + momentum * unbiased
    return y, (mean, invstd), (new_rm, new_rv)


def bn_eval_fwd(x, gamma, beta, running_mean, running_var, eps=BN_EPS):
    return (x - running_mean) / np.sqrt(running_var + eps) * gamma + beta


def bn_train_bwd(dy, x, gamma, mean, invstd):
    """dx, dgamma, dbeta for train-mode BN (batch statistics take part in the gradient)."""
    dym = dy.reshape(-1, dy.shape[-1])
    xhat = ((x - mean) * invstd).reshape(-1, x.shape[-1])
    m = dym.shape[0]
    dbeta = dym.sum(axis=0)
    dgamma = (dym * xhat).sum(axis=0)
    dx = (gamma * invstd) * (dym - dbeta / m - xhat * (dgamma / m))
    return dx.reshape(x.shape), dgamma, dbeta


def bn_eval_bwd(dy, x, gamma, running_mean, running_var, eps=BN_EPS):
    invstd = 1.0 / np.sqrt(running_var + eps)
    xhat = (x - running_mean) * invstd
    dym = dy.reshape(-1, dy.shape[-1])
    return dy * (gamma * invstd), (dym * xhat.reshape(dym.shape)).sum(axis=0), dym.sum(axis=0)


# ----------------------------------------------------------------------------------------------
# ReLU / Dropout (residual_block.py:62-65,166-171).  Dropout takes an explicit keep-mask: torch's CPU
# generator stream is not reproducible elsewhere, so parity runs use p=0 / eval or feed the mask.
# ----------------------------------------------------------------------------------------------

def relu_fwd(x):
    return np.maximum(x, 0)


def relu_bwd(dy, y):
    return dy * (y > 0)


def dropout_fwd(x, keep_mask, p):
    return x * keep_mask / (1.0 - p) if p > 0 else x


def dropout_bwd(dy, keep_mask, p):
    return dy * keep_mask / (1.0 - p) if p > 0 else dy


# ----------------------------------------------------------------------------------------------
# Shortcut (residual_block.py:48-57,89-94,150-159,205-210): AvgPool2d(k=1,s=2) == x[:, ::2, ::2];
# then either the 1x1 projection or zero-padding of C_in extra channels at the END.
# ----------------------------------------------------------------------------------------------

def subsample2(x):
    return x[:, ::2, ::2, :].copy()


def subsample2_bwd(dy, H, W):
    dx = np.zeros((dy.shape[0], H, W, dy.shape[3]), dtype=dy.dtype)
    dx[:, ::2, ::2, :] = dy
    return dx


def pad_channels(x, extra):
    return np.concatenate([x, np.zeros(x.shape[:3] + (extra,), dtype=x.dtype)], axis=3)


def pad_channels_bwd(dy, c_in):
    return dy[..., :c_in].copy()


# ----------------------------------------------------------------------------------------------
# Pools (resnet.py:77-87)
# ----------------------------------------------------------------------------------------------

def avgpool_fwd(x, k, stride, pad):
    """AvgPool2d(count_include_pad=True, the torch default)."""
    N, H, W, C = x.shape
    cols, P, Q = _im2col(x, k, k, stride, pad)
    return cols.reshape(N, P, Q, k * k, C).sum(axis=3) / (k * k)


def avgpool_bwd(dy, k, stride, pad, H, W):
    N, P, Q, C = dy.shape
    dxp = np.zeros((N, H + 2 * pad, W + 2 * pad, C), dtype=dy.dtype)
    for r in range(k):
        for s in range(k):
            dxp[:, r:r + stride * P:stride, s:s + stride * Q:stride, :] += dy / (k * k)
    return dxp[:, pad:pad + H, pad:pad + W, :].copy()


def maxpool_fwd(x, k, stride, pad):
    """MaxPool2d with -inf padding. returns y and argmax (flat index into the k*k window, first max wins)."""
    N, H, W, C = x.shape
    P = (H + 2 * pad - k) // stride + 1
    Q = (W + 2 * pad - k) // stride + 1
    xp = np.full((N, H + 2 * pad, W + 2 * pad, C), -np.inf, dtype=x.dtype)
    xp[:, pad:pad + H, pad:pad + W, :] = x
    win = np.empty((N, P, Q, k * k, C), dtype=x.dtype)
    for r in range(k):
        for s in range(k):
            win[:, :, :, r * k + s, :] = xp[:, r:r + stride * P:stride, s:s + stride * Q:stride, :]
    arg = win.argmax(axis=3)
    return win.max(axis=3), arg


def maxpool_bwd(dy, arg, k, stride, pad, H, W):
    N, P, Q, C = dy.shape
    dxp = np.zeros((N, H + 2 * pad, W + 2 * pad, C), dtype=dy.dtype)
    for r in range(k):
        for s in range(k):
            sel = (arg == r * k + s)
            dxp[:, r:r + stride * P:stride, s:s + stride * Q:stride, :] += dy * sel
    return dxp[:, pad:pad + H, pad:pad + W, :].copy()


# ----------------------------------------------------------------------------------------------
# Flatten + Linear (resnet.py:117-120).  Flatten of NCHW [N,C,1,1] == the channel vector.
# ----------------------------------------------------------------------------------------------

def linear_fwd(f, w, b):
    return f @ w.T + b


def linear_bwd(dlogits, f, w):
    return dlogits @ w, dlogits.T @ f, dlogits.sum(axis=0)  # df, dw, db


# ----------------------------------------------------------------------------------------------
# Loss + metrics (metrics.py:10-29): mean cross entropy; top-k error via topk indices.
# ----------------------------------------------------------------------------------------------

def cross_entropy_fwd(logits, labels):
    z = logits - logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1))
    nll = lse - z[np.arange(len(labels)), labels]
    return nll.mean()


def cross_entropy_bwd(logits, labels):
    z = logits - logits.max(axis=1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(axis=1, keepdims=True)
    p[np.arange(len(labels)), labels] -= 1.0
    return p / len(labels)


def topk_err(logits, labels, k):
    """1 - mean(label in top-k).  torch.topk tie order is implementation-defined; the rank rule used here
    (an entry outranks the label iff it is strictly greater, or equal with a lower index) is the one the
    golden G6 vectors pin."""
    lab = logits[np.arange(len(labels)), labels][:, None]
    idx = np.arange(logits.shape[1])[None, :]
    ahead = (logits > lab) | ((logits == lab) & (idx < labels[:, None]))
    return 1.0 - (ahead.sum(axis=1) < k).mean()
