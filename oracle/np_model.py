"""
TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference network (forward AND hand-written backward).

Follows /root/reference/resnet/architectures/resnet.py:122-158 (spec grammar / module order),
residual_block.py:67-99 (basic block), :173-215 (bottleneck block), metrics.py:10-29 (loss / top-k).
Parameters are addressed by the reference's own ``state_dict`` keys (``_architecture.{i}...``), weights KCRS.
No autograd: the backward below is the algorithm the HIP engine implements, written out.
"""

import re
import numpy as np

from . import np_ops as ops


# ---------------------------------------------------------------------------------------------
# spec grammar (resnet.py:16-22, 122-158): prefix dispatch in the order c, mp, ap, r, b, n, a, f
# ---------------------------------------------------------------------------------------------

def _ints(tok, n):
    m = re.match(r"([a-z]+)" + ",".join([r"([0-9]+)"] * n), tok)
    return [int(v) for v in m.groups()[1:]]


def parse_spec(spec):
    """-> list of dicts, one per token, with the running channel count resolved."""
    toks = spec.split()
    out, ch = [], None
    for n, t in enumerate(toks):
        if t.startswith('c'):
            i, o, k, s, p = _ints(t, 5)
            out.append(dict(kind='conv', cin=i, cout=o, k=k, stride=s, pad=p)); ch = o
        elif t.startswith('mp'):
            k, s, p = _ints(t, 3); out.append(dict(kind='maxpool', k=k, stride=s, pad=p))
        elif t.startswith('ap'):
            k, s, p = _ints(t, 3); out.append(dict(kind='avgpool', k=k, stride=s, pad=p))
        elif t.startswith('r') or t.startswith('b'):
            down = toks[n - 1].startswith(t[0])          # resnet.py:135,142 (n-1 == -1 wraps, as there)
            cin, cout = ch, (2 * ch if down else ch)
            out.append(dict(kind='basic' if t[0] == 'r' else 'bottleneck', depth=_ints(t, 1)[0],
                            cin=cin, cout=cout, down=down)); ch = cout
        elif t.startswith('n'):
            out.append(dict(kind='norm', c=ch))
        elif t.startswith('a'):
            out.append(dict(kind='act'))
        elif t.startswith('f'):
            i, o = _ints(t, 2); out.append(dict(kind='fc', cin=i, cout=o))
        else:
            raise ValueError("Unknown component in architecture spec.")
    return out


def block_layout(kind, cin, down, preact):
    """conv/norm shapes of one block (residual_block.py:26-65 / :120-171)."""
    cout = 2 * cin if down else cin
    if kind == 'basic':
        convs = [(cin, cout, 3, 2 if down else 1, 1), (cout, cout, 3, 1, 1)]
        norms = [cin if preact else cout, cout]
    else:
        cb = cin // 2 if down else cin // 4
        convs = [(cin, cb, 1, 1, 0), (cb, cb, 3, 2 if down else 1, 1), (cb, cout, 1, 1, 0)]
        norms = [cin if preact else cb, cb, cb if preact else cout]
    return convs, norms, cout


def param_shapes(spec, preact, use_proj):
    """ordered (key, shape) list of state_dict() -- the G5 grammar fixture pins this."""
    keys = []

    def bn(prefix, c):
        keys.extend([(prefix + '.weight', (c,)), (prefix + '.bias', (c,)), (prefix + '.running_mean', (c,)),
                     (prefix + '.running_var', (c,)), (prefix + '.num_batches_tracked', ())])
    for idx, comp in enumerate(parse_spec(spec)):
        pre = f'_architecture.{idx}'
        if comp['kind'] == 'conv':
            keys.append((pre + '.weight', (comp['cout'], comp['cin'], comp['k'], comp['k'])))
            keys.append((pre + '.bias', (comp['cout'],)))
        elif comp['kind'] == 'norm':
            bn(pre, comp['c'])
        elif comp['kind'] == 'fc':
            keys.append((pre + '.1.weight', (comp['cout'], comp['cin'])))
            keys.append((pre + '.1.bias', (comp['cout'],)))
        elif comp['kind'] in ('basic', 'bottleneck'):
            for b in range(comp['depth']):
                cin = comp['cin'] if b == 0 else comp['cout']
                down = comp['down'] and b == 0
                convs, norms, cout = block_layout(comp['kind'], cin, down, preact)
                bp = f'{pre}.{b}'
                for j, (ci, co, k, s, p) in enumerate(convs, 1):
                    keys.append((f'{bp}._conv{j}.weight', (co, ci, k, k)))
                if down and use_proj:
                    keys.append((f'{bp}._proj.weight', (cout, cin, 1, 1)))
                for j, c in enumerate(norms, 1):
                    bn(f'{bp}._norm{j}', c)
    return keys


# ---------------------------------------------------------------------------------------------
# forward / backward
# ---------------------------------------------------------------------------------------------

class NumpyResNet:
    """state: dict key -> np.ndarray (reference key names).  train flag selects BN batch/running stats.
    dropout_masks: optional dict 'key-of-dropout-site' -> keep mask (NHWC); p>0 without masks is an error."""

    def __init__(self, spec, preact, use_proj, dropout_prob=0.0, dtype=np.float64):
        self.spec, self.preact, self.use_proj, self.p = spec, preact, use_proj, dropout_prob
        self.comps = parse_spec(spec)
        self.dtype = dtype

    # -- helpers -------------------------------------------------------------------------------
    def _bn_fwd(self, st, pre, x, train, cache, new_state):
        g, b = st[pre + '.weight'], st[pre + '.bias']
        rm, rv = st[pre + '.running_mean'], st[pre + '.running_var']
        if train:
            y, (mean, invstd), (nrm, nrv) = ops.bn_train_fwd(x, g, b, rm, rv)
            new_state[pre + '.running_mean'], new_state[pre + '.running_var'] = nrm, nrv
            new_state[pre + '.num_batches_tracked'] = st[pre + '.num_batches_tracked'] + 1
            cache.append(('bn', pre, x, mean, invstd, True))
        else:
            y = ops.bn_eval_fwd(x, g, b, rm, rv)
            cache.append(('bn', pre, x, None, None, False))
        return y

    def _bn_bwd(self, st, rec, dy, grads):
        _, pre, x, mean, invstd, train = rec
        g = st[pre + '.weight']
        if train:
            dx, dg, db = ops.bn_train_bwd(dy, x, g, mean, invstd)
        else:
            dx, dg, db = ops.bn_eval_bwd(dy, x, g, st[pre + '.running_mean'], st[pre + '.running_var'])
        grads[pre + '.weight'] = grads.get(pre + '.weight', 0) + dg
        grads[pre + '.bias'] = grads.get(pre + '.bias', 0) + db
        return dx

    def _drop(self, x, site, train, masks, cache):
        if self.p > 0 and train:
            m = masks[site]
            cache.append(('drop', m))
            return ops.dropout_fwd(x, m, self.p)
        cache.append(('drop', None))
        return x

    def _conv(self, st, key, x, stride, pad, cache, bias_key=None):
        w = st[key]
        cache.append(('conv', key, bias_key, x, stride, pad))
        return ops.conv2d_fwd(x, w, stride, pad, st[bias_key] if bias_key else None)

    # -- block forward (residual_block.py:67-99 / :173-215) --------------------------------------
    def _block_fwd(self, st, bp, kind, cin, down, x, train, masks, cache, new_state):
        convs, norms, cout = block_layout(kind, cin, down, self.preact)
        i = x
        nconv = len(convs)
        for j, (ci, co, k, s, p) in enumerate(convs, 1):
            if self.preact:
                x = self._bn_fwd(st, f'{bp}._norm{j}', x, train, cache, new_state)
                x = ops.relu_fwd(x); cache.append(('relu', x))
                x = self._drop(x, f'{bp}._dropout{j}', train, masks, cache)
                x = self._conv(st, f'{bp}._conv{j}.weight', x, s, p, cache)
            else:
                x = self._drop(x, f'{bp}._dropout{j}', train, masks, cache)
                x = self._conv(st, f'{bp}._conv{j}.weight', x, s, p, cache)
                x = self._bn_fwd(st, f'{bp}._norm{j}', x, train, cache, new_state)
                if j < nconv:
                    x = ops.relu_fwd(x); cache.append(('relu', x))
        # shortcut, always from the raw block input (residual_block.py:68,89-94)
        sc_cache = []
        if down:
            H, W = i.shape[1], i.shape[2]
            i2 = ops.subsample2(i)
            if self.use_proj:
                sc = ops.conv2d_fwd(i2, st[f'{bp}._proj.weight'], 1, 0)
                sc_cache = ('proj', i2, H, W)
            else:
                sc = ops.pad_channels(i2, cin)
                sc_cache = ('pad', cin, H, W)
        else:
            sc, sc_cache = i, ('id',)
        h = sc + x
        if not self.preact:
            h = ops.relu_fwd(h)
        cache.append(('block_end', bp, sc_cache, h if not self.preact else None, nconv))
        return h

    def _block_bwd(self, st, dh, cache, grads):
        _, bp, sc_cache, hpost, nconv = cache.pop()
        if hpost is not None:
            dh = ops.relu_bwd(dh, hpost)
        dx = dh
        for j in range(nconv, 0, -1):
            if self.preact:
                dx = self._conv_bwd(st, cache.pop(), dx, grads)
                dx = self._drop_bwd(cache.pop(), dx)
                dx = ops.relu_bwd(dx, cache.pop()[1])
                dx = self._bn_bwd(st, cache.pop(), dx, grads)
            else:
                if j < nconv:
                    dx = ops.relu_bwd(dx, cache.pop()[1])
                dx = self._bn_bwd(st, cache.pop(), dx, grads)
                dx = self._conv_bwd(st, cache.pop(), dx, grads)
                dx = self._drop_bwd(cache.pop(), dx)
        if sc_cache[0] == 'id':
            di = dh
        elif sc_cache[0] == 'pad':
            di = ops.subsample2_bwd(ops.pad_channels_bwd(dh, sc_cache[1]), sc_cache[2], sc_cache[3])
        else:
            _, i2, H, W = sc_cache
            key = f'{bp}._proj.weight'
            grads[key] = grads.get(key, 0) + ops.conv2d_wgrad(i2, dh, 1, 1, 1, 0)
            di = ops.subsample2_bwd(ops.conv2d_dgrad(dh, st[key], 1, 0, i2.shape[1], i2.shape[2]), H, W)
        return dx + di

    def _conv_bwd(self, st, rec, dy, grads, need_dx=True):
        _, key, bias_key, x, stride, pad = rec
        w = st[key]
        grads[key] = grads.get(key, 0) + ops.conv2d_wgrad(x, dy, w.shape[2], w.shape[3], stride, pad)
        if bias_key:
            grads[bias_key] = grads.get(bias_key, 0) + ops.bias_grad(dy)
        if not need_dx:
            return None
        return ops.conv2d_dgrad(dy, w, stride, pad, x.shape[1], x.shape[2])

    def _drop_bwd(self, rec, dy):
        return ops.dropout_bwd(dy, rec[1], self.p) if rec[1] is not None else dy

    # -- whole net ---------------------------------------------------------------------------------
    def forward(self, state, x_nchw, train=True, dropout_masks=None):
        """returns logits [N,classes], cache (for backward), new_state (BN running stats after this forward)."""
        st = {k: np.asarray(v, dtype=self.dtype) if np.asarray(v).dtype.kind == 'f' else np.asarray(v)
              for k, v in state.items()}
        x = np.transpose(np.asarray(x_nchw, dtype=self.dtype), (0, 2, 3, 1))
        cache, new_state = [], {}
        for idx, comp in enumerate(self.comps):
            pre, kind = f'_architecture.{idx}', comp['kind']
            if kind == 'conv':
                x = self._conv(st, pre + '.weight', x, comp['stride'], comp['pad'], cache, pre + '.bias')
            elif kind == 'norm':
                x = self._bn_fwd(st, pre, x, train, cache, new_state)
            elif kind == 'act':
                x = ops.relu_fwd(x); cache.append(('relu', x))
            elif kind == 'maxpool':
                H, W = x.shape[1], x.shape[2]
                x, arg = ops.maxpool_fwd(x, comp['k'], comp['stride'], comp['pad'])
                cache.append(('maxpool', arg, comp, H, W))
            elif kind == 'avgpool':
                cache.append(('avgpool', comp, x.shape[1], x.shape[2]))
                x = ops.avgpool_fwd(x, comp['k'], comp['stride'], comp['pad'])
            elif kind == 'fc':
                # Flatten of NCHW: [N,C,H,W] -> [N, C*H*W]; with H=W=1 this is the channel vector
                f = np.transpose(x, (0, 3, 1, 2)).reshape(x.shape[0], -1)
                cache.append(('fc', pre, f, x.shape))
                x = ops.linear_fwd(f, st[pre + '.1.weight'], st[pre + '.1.bias'])
            else:
                for b in range(comp['depth']):
                    cin = comp['cin'] if b == 0 else comp['cout']
                    cache.append(('block_begin', comp['kind']))
                    x = self._block_fwd(st, f'{pre}.{b}', comp['kind'], cin, comp['down'] and b == 0, x,
                                        train, dropout_masks or {}, cache, new_state)
        self._st = st
        return x, cache, new_state

    def backward(self, cache, dlogits):
        """returns grads dict (reference keys; conv weights KCRS) -- consumes ``cache``."""
        st, grads = self._st, {}
        cache = list(cache)
        d = np.asarray(dlogits, dtype=self.dtype)
        first_conv = True
        while cache:
            rec = cache[-1]
            tag = rec[0]
            if tag == 'fc':
                _, pre, f, shp = cache.pop()
                df, dw, db = ops.linear_bwd(d, f, st[pre + '.1.weight'])
                grads[pre + '.1.weight'], grads[pre + '.1.bias'] = dw, db
                N, H, W, C = shp
                d = np.transpose(df.reshape(N, C, H, W), (0, 2, 3, 1))
            elif tag == 'avgpool':
                _, comp, H, W = cache.pop()
                d = ops.avgpool_bwd(d, comp['k'], comp['stride'], comp['pad'], H, W)
            elif tag == 'maxpool':
                _, arg, comp, H, W = cache.pop()
                d = ops.maxpool_bwd(d, arg, comp['k'], comp['stride'], comp['pad'], H, W)
            elif tag == 'relu':
                d = ops.relu_bwd(d, cache.pop()[1])
            elif tag == 'bn':
                d = self._bn_bwd(st, cache.pop(), d, grads)
            elif tag == 'conv':
                # a top-level conv whose input is the image has no dgrad consumer
                is_stem = len(cache) == 1
                d = self._conv_bwd(st, cache.pop(), d, grads, need_dx=not is_stem)
            elif tag == 'block_end':
                d = self._block_bwd(st, d, cache, grads)
                assert cache.pop()[0] == 'block_begin'
            else:
                raise RuntimeError(tag)
        return grads


def loss_and_metrics(logits, labels):
    """metrics.py:21-29"""
    return dict(loss=ops.cross_entropy_fwd(logits, labels), top1_err=ops.topk_err(logits, labels, 1),
                top5_err=ops.topk_err(logits, labels, 5))


def sgd_step(params, grads, bufs, lr, momentum, weight_decay, nesterov, dampening=0.0):
    """torch.optim.SGD update rule (optim_util.py:11-18 instantiates it; args config.yaml:22-28)."""
    for k in params:
        g = grads[k] + weight_decay * params[k]
        if momentum:
            if k not in bufs:
                bufs[k] = g.copy()
            else:
                bufs[k] = momentum * bufs[k] + (1 - dampening) * g
            g = g + momentum * bufs[k] if nesterov else bufs[k]
        params[k] = params[k] - lr * g
