"""
Executable specification of the plan IR: runs a lowered Plan op by op on numpy arrays using the oracle's formulas.

Used (a) on CPU to prove the lowering (data flow, fusion flags, gradient merges) against the golden vectors before any
kernel exists, and (b) on the GPU box as the per-op checker for the HIP kernels (tests run one-op plans through both).
Test infrastructure: imports oracle/, never imported by the product.
"""
import numpy as np

from oracle import np_ops as ops
from pytorch_ddp_resnet_amd.engine import ir


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def keep_mask(n, p, site, step_seed):
    """must match csrc/common.h: rn_keep().  One 32-bit hash per PAIR of elements; element idx keeps iff its 16-bit half (low half: even idx, high half: odd idx)
    >= floor(p * 2^32) >> 16."""
    key = lowbias32((site * 0x9E3779B9) ^ (step_seed & 0xFFFFFFFF) ^ ((step_seed >> 32) * 0x85EBCA6B))
    idx = np.arange(n, dtype=np.uint64)
    h = lowbias32((idx >> np.uint64(1)) ^ key)
    half = np.where((idx & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    thr = np.uint64(min(int(p * 4294967296.0), 0xFFFFFFFF)) >> np.uint64(16)
    return half >= thr


def pool_winner(x, arg, k, stride, pad):
    """x[N,H,W,C] gathered at each pooling window's argmax (window-local flat index r*k+s) -> [N,P,Q,C]"""
    N, H, W, C = x.shape
    P, Q = arg.shape[1], arg.shape[2]
    xp = np.zeros((N, H + 2 * pad, W + 2 * pad, C), dtype=x.dtype)
    xp[:, pad:pad + H, pad:pad + W, :] = x
    out = np.zeros((N, P, Q, C), dtype=x.dtype)
    for r in range(k):
        for s in range(k):
            out += xp[:, r:r + stride * P:stride, s:s + stride * Q:stride, :] * (arg == r * k + s)
    return out


def res_read(res, mode, N, H, W, C):
    """the residual / merge operand as seen from a destination of shape [N,H,W,C]."""
    if mode == ir.RES_SAME:
        return res
    out = np.zeros((N, H, W, C), dtype=res.dtype)
    if mode == ir.RES_DOWN2PAD:
        cr = res.shape[3]
        out[..., :cr] = res[:, ::2, ::2, :][:, :H, :W, :]
    elif mode == ir.RES_UP2:
        out[:, ::2, ::2, :] = res[..., :C]
    return out


class NumpyPlan:
    def __init__(self, plan, dtype=np.float64):
        self.plan = plan
        self.dtype = dtype
        self.bufs = [None] * len(plan.slots)
        for i, s in enumerate(plan.slots):
            if s.role == 'ws':
                continue
            if s.role == 'u8':
                self.bufs[i] = np.zeros(s.shape, dtype=np.uint8)
                continue
            dt = {'T': dtype, 'f32': dtype, 'i64': np.int64, 'u8': np.uint8}[s.dtype]
            self.bufs[i] = np.zeros(s.shape, dtype=dt)

    # -- binding helpers -----------------------------------------------------------------------------------
    def load_state(self, state):
        """state: reference-keyed dict (conv weights KCRS)."""
        for i, s in enumerate(self.plan.slots):
            if s.role in ('param', 'buffer'):
                v = np.asarray(state[s.key])
                if v.ndim == 4:
                    v = np.transpose(v, (0, 2, 3, 1))
                self.bufs[i] = v.astype(self.dtype if v.dtype.kind == 'f' else v.dtype).reshape(s.shape).copy()

    def grads(self):
        """-> dict key -> gradient in reference layout (KCRS)."""
        out = {}
        for i, s in enumerate(self.plan.slots):
            if s.role == 'grad':
                v = self.bufs[i]
                out[s.key] = np.transpose(v, (0, 3, 1, 2)) if v.ndim == 4 else v
        return out

    def state(self):
        return {s.key: self.bufs[i] for i, s in enumerate(self.plan.slots) if s.role == 'buffer'}

    def __getitem__(self, name):
        return self.bufs[self.plan.slot_of[name]]

    def __setitem__(self, name, v):
        self.bufs[self.plan.slot_of[name]] = np.asarray(v)

    # -- execution -----------------------------------------------------------------------------------------
    def run(self, first, last, step_seed=0, hook_fn=None):
        hooks = {}
        for h in self.plan.hooks:
            hooks.setdefault(h.at, []).append(h)
        for i in range(first, last):
            for h in hooks.get(i, []):
                if hook_fn:
                    hook_fn(self, h)
            self.exec_op(self.plan.ops[i], step_seed)

    def forward(self, **kw):
        self.run(0, self.plan.n_fwd, **kw)

    def backward(self, **kw):
        self.run(self.plan.n_fwd, len(self.plan.ops), **kw)

    def exec_op(self, op, step_seed):
        B = lambda n: (self.bufs[op.buf[n]] if op.buf.get(n, -1) >= 0 else None)   # noqa: E731
        d, k = op.dim, op.kind

        def put(n, v):
            self.bufs[op.buf[n]] = np.array(v, dtype=self.bufs[op.buf[n]].dtype).reshape(self.bufs[op.buf[n]].shape)   # copy: slots never alias

        def w_kcrs(w, K, R, S, C):
            return np.transpose(w.reshape(K, R, S, C), (0, 3, 1, 2))

        if k == ir.OP_STEM_FWD:
            x = np.transpose(B('x'), (0, 2, 3, 1)).astype(self.dtype)
            put('y', ops.conv2d_fwd(x, w_kcrs(B('w'), d['K'], d['R'], d['S'], d['C']), d['stride'], d['pad'], B('bias')))
        elif k == ir.OP_PACK_W:
            w = B('w').reshape(d['K'], d['RS'], d['C'])
            if op.buf.get('w_fwd', -1) >= 0:
                put('w_fwd', w)
            if op.buf.get('w_dgrad', -1) >= 0:
                put('w_dgrad', np.transpose(w, (2, 1, 0)))
        elif k == ir.OP_CONV_FWD:
            y = ops.conv2d_fwd(B('x'), w_kcrs(B('w_fwd'), d['K'], d['R'], d['S'], d['C']), d['stride'], d['pad'],
                               B('bias') if op.buf.get('bias', -1) >= 0 else None)
            if d.get('res_mode', 0):
                y = y + res_read(B('res'), d['res_mode'], d['N'], d['P'], d['Q'], d['K'])
            put('y', y)
            if op.buf.get('stats', -1) >= 0:      # fused epilogue: per-tile partial sums (only their total is specified)
                part = np.zeros_like(B('stats'))
                ym = y.reshape(-1, d['K'])
                part[0, 0], part[0, 1] = ym.sum(0), (ym * ym).sum(0)
                put('stats', part)
        elif k == ir.OP_BN_STATS:
            x = B('x').reshape(d['M'], d['C'])
            nblk = d['nblk']
            rows = (d['M'] + nblk - 1) // nblk
            part = np.zeros((nblk, 2, d['C']), dtype=self.dtype)
            for b in range(nblk):
                xb = x[b * rows:(b + 1) * rows]
                part[b, 0], part[b, 1] = xb.sum(0), (xb * xb).sum(0)
            put('partial', part)
        elif k == ir.OP_BN_FINALIZE:
            C = d['C']
            gamma, beta = B('gamma'), B('beta')
            if op.flags & ir.F_TRAIN:
                part = B('partial').reshape(-1, 2, C)[:d['nblk']]
                m = d['count']
                mean = part[:, 0].sum(0) / m
                var = np.maximum(part[:, 1].sum(0) / m - mean * mean, 0.0)
                self.bufs[op.buf['running_mean']] = (1 - op.fp['momentum']) * B('running_mean') + op.fp['momentum'] * mean
                self.bufs[op.buf['running_var']] = (1 - op.fp['momentum']) * B('running_var') + op.fp['momentum'] * var * (m / max(m - 1, 1))
                self.bufs[op.buf['nbt']] = B('nbt') + 1
            else:
                mean, var = B('running_mean'), B('running_var')
            invstd = 1.0 / np.sqrt(var + op.fp['eps'])
            scale = gamma * invstd
            put('coef', np.stack([scale, beta - mean * scale, mean, invstd]))
        elif k == ir.OP_BN_APPLY:
            coef = B('coef')
            y = B('x') * coef[0] + coef[1]
            if d.get('res_mode', 0):
                y = y + res_read(B('res'), d['res_mode'], d['N'], d['H'], d['W'], d['C'])
            if op.flags & ir.F_RELU:
                y = np.maximum(y, 0)
            p = op.fp.get('p', 0.0)
            if p > 0:
                y = y * keep_mask(y.size, p, op.seed, step_seed).reshape(y.shape) / (1.0 - p)
            put('out', y)
        elif k == ir.OP_DROPOUT_FWD:
            x, p = B('x'), op.fp['p']
            put('out', x * keep_mask(x.size, p, op.seed, step_seed).reshape(x.shape) / (1.0 - p))
        elif k == ir.OP_DROPOUT_BWD:
            p, g = op.fp['p'], B('dout')
            put('din', g * keep_mask(g.size, p, op.seed, step_seed).reshape(g.shape) / (1.0 - p))      # the hash, not (out != 0): exact zeros keep their gradient
        elif k == ir.OP_ADD_RES:
            put('dst', B('dst') + res_read(B('res'), d['res_mode'], d['N'], d['H'], d['W'], d['C']))
        elif k == ir.OP_MAXPOOL_FWD:
            y, arg = ops.maxpool_fwd(B('x'), d['k'], d['stride'], d['pad'])
            put('y', y)
            if op.buf.get('argmax', -1) >= 0:
                # window-local flat index r*k+s of the FIRST maximum among the in-bounds taps (padding is -inf)
                put('argmax', arg.astype(np.uint8))
        elif k == ir.OP_BN_POOL_FWD:                  # maxpool([relu](bn(x))) without storing the normalised activation
            coef = B('coef')
            a = B('x') * coef[0] + coef[1]
            if op.flags & ir.F_RELU:
                a = np.maximum(a, 0)
            y, arg = ops.maxpool_fwd(a, d['k'], d['stride'], d['pad'])
            put('y', y)
            if op.buf.get('argmax', -1) >= 0:
                put('argmax', arg.astype(np.uint8))
            if op.buf.get('xsel', -1) >= 0:                # the input element under each window's argmax
                put('xsel', pool_winner(B('x'), arg, d['k'], d['stride'], d['pad']))
        elif k == ir.OP_BN_POOL_BWD_REDUCE and op.buf.get('xsel', -1) >= 0:
            coef, C, nblk = B('coef'), d['C'], d['nblk']   # sums over the WINDOWS: gradient of a window goes to its winner, masked by the winner's sign
            g, xs = B('dy').reshape(-1, C), B('xsel').reshape(-1, C)
            if op.flags & ir.F_RELU:
                g = g * ((xs * coef[0] + coef[1]) > 0)
            part = np.zeros((nblk, 2, C), dtype=self.dtype)
            part[0, 0] = g.sum(0)
            part[0, 1] = (g * ((xs - coef[2]) * coef[3])).sum(0)
            put('partial', part)
        elif k in (ir.OP_BN_POOL_BWD_REDUCE, ir.OP_BN_POOL_BWD_APPLY):
            coef = B('coef')
            g = ops.maxpool_bwd(B('dy'), B('argmax').astype(np.int64), d['k'], d['stride'], d['pad'], d['H'], d['W'])
            if op.flags & ir.F_RELU:
                g = g * ((B('x') * coef[0] + coef[1]) > 0)
            xhat = (B('x') - coef[2]) * coef[3]
            C = d['C']
            if k == ir.OP_BN_POOL_BWD_REDUCE:
                nblk, rows_total = d['nblk'], d['N'] * d['H']
                gm, xm = g.reshape(rows_total, -1, C), xhat.reshape(rows_total, -1, C)
                part = np.zeros((nblk, 2, C), dtype=self.dtype)
                for b in range(nblk):                     # workgroup b walks image rows b, b + nblk, ...
                    part[b, 0] = gm[b::nblk].sum((0, 1))
                    part[b, 1] = (gm[b::nblk] * xm[b::nblk]).sum((0, 1))
                put('partial', part)
            else:
                dsum = B('dsum')
                dx = coef[0] * (g - dsum[0] / d['count'] - xhat * (dsum[1] / d['count'])) if op.flags & ir.F_TRAIN else coef[0] * g
                put('dx', dx)
                if op.buf.get('sums', -1) >= 0:           # per-channel (sum dx, 0): which workgroup row holds what is the kernel's business, the total is not
                    part = np.zeros((d['rows'], 2, C), dtype=self.dtype)
                    part[0, 0] = dx.reshape(-1, C).sum(0)             # second row: zeros (no consumer of a sum of squares here)
                    put('sums', part)
        elif k == ir.OP_MAXPOOL_BWD:
            put('dx', ops.maxpool_bwd(B('dy'), B('argmax').astype(np.int64), d['k'], d['stride'], d['pad'], d['H'], d['W']))
        elif k == ir.OP_POOL_FC_FWD:
            feat = B('x').reshape(d['N'], d['HW'], d['C']).mean(1)
            put('feat', feat)
            put('logits', ops.linear_fwd(feat, B('w'), B('b')))
        elif k == ir.OP_POOL_FC_BWD:
            df, dw, db = ops.linear_bwd(B('dlogits'), B('feat'), B('w'))
            if op.flags & ir.F_ACCUM:
                dw = dw + B('dw').reshape(dw.shape); db = db + B('db').reshape(db.shape)
            put('dw', dw); put('db', db)
            if not (op.flags & ir.F_NO_DX):
                dx = np.repeat(df[:, None, :] / d['HW'], d['HW'], axis=1)
                put('dx', dx)
        elif k in (ir.OP_BN_BWD_REDUCE, ir.OP_BN_BWD_APPLY):
            coef = B('coef')
            g = B('dout') * op.fp.get('gscale', 1.0)
            if op.flags & ir.F_MASK_RECOMPUTE:      # mask = [x*scale+shift > 0] & dropout keep hash (not read from memory)
                m = (B('x') * coef[0] + coef[1]) > 0
                if op.fp.get('p', 0.0) > 0:
                    m = m & keep_mask(m.size, op.fp['p'], op.seed, step_seed).reshape(m.shape)
                g = g * m
            elif op.flags & ir.F_RELU:
                g = g * (B('mask') > 0)
            xhat = (B('x') - coef[2]) * coef[3]
            C = d['C']
            if k == ir.OP_BN_BWD_REDUCE:
                nblk, M = d['nblk'], d['M']
                rows = (M + nblk - 1) // nblk
                gm, xm = g.reshape(M, C), xhat.reshape(M, C)
                part = np.zeros((nblk, 2, C), dtype=self.dtype)
                for b in range(nblk):
                    part[b, 0] = gm[b * rows:(b + 1) * rows].sum(0)
                    part[b, 1] = (gm[b * rows:(b + 1) * rows] * xm[b * rows:(b + 1) * rows]).sum(0)
                put('partial', part)
            else:
                dsum = B('dsum')
                if op.flags & ir.F_TRAIN:
                    m = d['count']
                    dx = coef[0] * (g - dsum[0] / m - xhat * (dsum[1] / m))
                else:
                    dx = coef[0] * g
                if d.get('add_mode', 0):
                    dx = dx + res_read(B('add'), d['add_mode'], d['N'], d['H'], d['W'], C)
                put('dx', dx)
                if op.flags & ir.F_WRITE_G:
                    put('g_out', g)
        elif k == ir.OP_BN_BWD_FINALIZE:
            part = B('partial').reshape(-1, 2, d['C'])[:d['nblk']]
            s = part.sum(0)
            put('dsum', s)
            put('dbeta', s[0]); put('dgamma', s[1])
        elif k == ir.OP_CONV_DGRAD:
            dx = ops.conv2d_dgrad(B('dy'), np.transpose(B('w_dgrad').reshape(d['C'], d['R'], d['S'], d['K']), (3, 0, 1, 2)),
                                  d['stride'], d['pad'], d['H'], d['W'])
            if d.get('res_mode', 0):
                dx = dx + res_read(B('res'), d['res_mode'], d['N'], d['H'], d['W'], d['C'])
            if op.flags & ir.F_ACCUM:
                dx = dx + B('dx')
            put('dx', dx)
            if op.buf.get('bn_partial', -1) >= 0:   # fused BatchNorm-backward reduction of the layer that fed this conv
                coef = B('bn_coef')
                g = dx * op.fp.get('gscale', 1.0)
                if op.buf.get('bn_mask', -1) >= 0:
                    g = g * (B('bn_mask') > 0)
                xhat = (B('bn_x') - coef[2]) * coef[3]
                part = np.zeros_like(B('bn_partial'))
                part[0, 0], part[0, 1] = g.reshape(-1, d['C']).sum(0), (g * xhat).reshape(-1, d['C']).sum(0)
                put('bn_partial', part)
        elif k == ir.OP_CONV_WGRAD:
            dw = ops.conv2d_wgrad(B('x'), B('dy'), d['R'], d['S'], d['stride'], d['pad'])   # KCRS
            put('dw', np.transpose(dw, (0, 2, 3, 1)))
        elif k == ir.OP_STEM_WGRAD:
            x = np.transpose(B('x'), (0, 2, 3, 1)).astype(self.dtype)
            dw = ops.conv2d_wgrad(x, B('dy'), d['R'], d['S'], d['stride'], d['pad'])
            put('dw', np.transpose(dw, (0, 2, 3, 1)))
            put('db', ops.bias_grad(B('dy')))
        elif k == ir.OP_SOFTMAX_CE:
            lg, lb = B('logits').astype(np.float64), B('labels')
            n = len(lb)
            out = np.zeros(4)
            out[0] = ops.cross_entropy_fwd(lg, lb) * n
            out[1] = ops.topk_err(lg, lb, 1) * n
            out[2] = ops.topk_err(lg, lb, min(5, lg.shape[1])) * n
            put('out3', out)
            if op.buf.get('dlogits', -1) >= 0:
                put('dlogits', ops.cross_entropy_bwd(lg, lb) * n * op.fp['scale'])
        elif k == ir.OP_IMG_TO_NHWC:
            out = np.zeros(self.bufs[op.buf['out']].shape, dtype=self.dtype)
            out[..., :d['C']] = np.transpose(B('x'), (0, 2, 3, 1))
            put('out', out)
        elif k == ir.OP_PACK_STEM_W:
            wp = np.zeros((d['K'], d['RS'], d['CP']), dtype=self.dtype)
            wp[..., :d['C']] = B('w').reshape(d['K'], d['RS'], d['C'])
            put('w_padded', wp)
        elif k == ir.OP_UNPACK_STEM_DW:
            v = B('dw_padded').reshape(d['K'], d['RS'], d['CP'])[..., :d['C']]
            put('dw', v + B('dw').reshape(v.shape) if op.flags & ir.F_ACCUM else v)
        elif k == ir.OP_IMG_TO_S2D:                    # [N,C,H,W] -> [N,H/2+3,W/2+3,16]: s2d pixel (i, j) channel (dy, dx, c of 4) = x[c, 2i+dy, 2j+dx]; 2 zero pixels before, 1 after
            x = B('x')
            N, C, H, W = x.shape
            out = np.zeros((N, H // 2 + 3, W // 2 + 3, 16), dtype=self.dtype)
            for dy in range(2):
                for dx in range(2):
                    out[:, 2:2 + H // 2, 2:2 + W // 2, (dy * 2 + dx) * 4:(dy * 2 + dx) * 4 + C] = np.transpose(x[:, :, dy::2, dx::2], (0, 2, 3, 1))
            put('out', out)
        elif k in (ir.OP_PACK_STEM_W_S2D, ir.OP_UNPACK_STEM_DW_S2D):
            K, C = d['K'], d['C']                      # filter tap (r', s') channel (dy, dx, c) <-> 7x7 weight (2r'+dy-1, 2s'+dx-1, c)
            idx = np.full((7, 7, C), -1, dtype=np.int64)
            for r in range(7):
                for t in range(7):
                    for c in range(C):
                        idx[r, t, c] = ((r + 1) // 2) * 64 + ((t + 1) // 2) * 16 + (((r + 1) % 2) * 2 + (t + 1) % 2) * 4 + c
            if k == ir.OP_PACK_STEM_W_S2D:
                wp = np.zeros((K, 256), dtype=self.dtype)
                wp[:, idx.reshape(-1)] = B('w').reshape(K, -1)
                put('w_s2d', wp.reshape(K, 16, 16))
            else:
                v = B('dw_s2d').reshape(K, 256)[:, idx.reshape(-1)].reshape(K, 7, 7, C)
                put('dw', v + B('dw').reshape(v.shape) if op.flags & ir.F_ACCUM else v)
        elif k == ir.OP_RELU_FWD:
            put('y', np.maximum(B('x'), 0))
        elif k == ir.OP_RELU_BWD:
            put('dx', B('dy') * (B('y') > 0))
        elif k in (ir.OP_AVGPOOL_FWD, ir.OP_AVGPOOL_BWD):          # AvgPool2d(k, s, p), zero padding counted in the divisor (torch's default)
            kk, st, pd, H, W = d['k'], d['stride'], d['pad'], d['H'], d['W']
            P, Q = (H + 2 * pd - kk) // st + 1, (W + 2 * pd - kk) // st + 1
            if k == ir.OP_AVGPOOL_FWD:
                xp = np.pad(B('x'), ((0, 0), (pd, pd), (pd, pd), (0, 0)))
                y = np.zeros((d['N'], P, Q, d['C']), dtype=xp.dtype)
                for r in range(kk):
                    for s_ in range(kk):
                        y += xp[:, r:r + st * (P - 1) + 1:st, s_:s_ + st * (Q - 1) + 1:st, :]
                put('y', y / (kk * kk))
            else:
                dxp = np.zeros((d['N'], H + 2 * pd, W + 2 * pd, d['C']), dtype=B('dy').dtype)
                for r in range(kk):
                    for s_ in range(kk):
                        dxp[:, r:r + st * (P - 1) + 1:st, s_:s_ + st * (Q - 1) + 1:st, :] += B('dy') / (kk * kk)
                put('dx', dxp[:, pd:pd + H, pd:pd + W, :])
        elif k == ir.OP_PERMUTE_F32:
            put('out', np.transpose(B('in').reshape(d['A'], d['B'], d['C']), (0, 2, 1)))
        elif k == ir.OP_ZERO:
            self.bufs[op.buf['dst']][...] = 0
        else:
            raise NotImplementedError(ir.OP_NAMES[k])
