"""
Lowers a parsed ``architecture_spec`` to the plan IR: the list of fused HIP launches that make up one forward and one
backward of the network, plus the table of device buffers they read and write.

What is mirrored (semantics, not code): the module order of ``ResNet._parse_spec`` (/root/reference/resnet/
architectures/resnet.py:122-158) and the data flow of ``ResidualBlock.forward`` (residual_block.py:67-99) and
``BottleneckResidualBlock.forward`` (:173-215): v1 ``[drop] conv bn relu``, v2 ``bn relu [drop] conv``, shortcut from
the raw block input (identity / subsample+zero-pad / subsample+1x1 projection), post-add ReLU only in v1.

How it runs here: every BN+ReLU(+dropout) is one elementwise pass (BN_APPLY) fed by statistics that the producing
convolution's epilogue already reduced (a statistics pass of its own only where no convolution produces the tensor), and
the two reduction sums of its backward come out of the consuming convolution's dgrad epilogue; the residual
add rides in the epilogue of the block's last convolution (v2) or of the last BN_APPLY (v1); the backward is written
out explicitly (no autograd inside the engine) and merges gradient forks in kernel epilogues instead of separate adds.
Weight gradients are produced in the order recorded in ``Plan.grad_order`` so the flat gradient buffer can be
all-reduced bucket by bucket while the backward is still running.
"""
from dataclasses import dataclass, field
import os
from typing import Dict, List, Optional, Tuple

from ..architectures.spec import Component, block_convs, parse_spec
from . import ir
from .ir import Op, Slot, Hook

BN_EPS, BN_MOMENTUM = 1e-5, 0.1        # torch.nn.BatchNorm2d defaults, which the reference keeps (resnet.py:111-112)


@dataclass
class T:
    """an NHWC activation tensor living in slot `s`."""
    s: int
    N: int
    H: int
    W: int
    C: int

    @property
    def M(self):
        return self.N * self.H * self.W


@dataclass
class Plan:
    slots: List[Slot]
    ops: List[Op]
    n_fwd: int                       # ops[:n_fwd] forward, ops[n_fwd:] backward
    hooks: List[Hook]
    grad_order: List[str]            # parameter keys in the order their gradients are produced by the backward
    param_keys: List[str]            # parameter keys in forward (state_dict) order
    slot_of: Dict[str, int]          # well-known slots: 'x', 'logits', 'dlogits', 'labels', 'loss3', 'ws'
    train: bool
    meta: dict = field(default_factory=dict)


def conv_stats_rows(g: dict, dgrad: bool = False) -> int:
    """rows of the [rows][2][C] partial-sum buffer a fused conv epilogue writes (= rn_conv_stats_rows in the library)."""
    R = ir.CONV_STATS_ROWS
    if not dgrad:
        return (g['N'] * g['P'] * g['Q'] + R - 1) // R
    st, rows = g['stride'], 0
    for pa in range(st):
        for pb in range(st):
            pc, qc = (g['H'] - pa + st - 1) // st, (g['W'] - pb + st - 1) // st
            if pc > 0 and qc > 0:
                rows += (g['N'] * pc * qc + R - 1) // R
    return rows


def bn_fold_bytes(nblk: int, C: int) -> int:
    """bytes of the hand-off buffer of a finalize launch whose partial rows are split over workgroups; 0 = not split (= rn_bn_fold_bytes in the
    library, csrc/bn.hip fold_splits: checked against it by tests/test_host_surface.py, and by the launch itself)."""
    ncb = (C + 15) // 16
    nbytes = nblk * 2 * C * 4
    if nblk < 512 or not (nbytes >= (16 << 20) or (ncb <= 8 and nbytes >= (4 << 20))):
        return 0
    S = min(nblk // 256, max(1, 1024 // ncb), 64)
    if S < 2:
        return 0
    return ((ncb * 4 + 63) // 64) * 64 + ncb * S * 2 * 16 * 8


def bn_partials(M: int, C: int) -> int:
    """number of row-slabs a statistics pass is split into (one workgroup each)."""
    rows_per = min(256, max(32, M // 512))        # >= 256 workgroups once M >= 8192
    return max(1, min(1024, (M + rows_per - 1) // rows_per))


class Lowering:
    def __init__(self, spec: str, preact: bool, use_proj: bool, dropout_prob: float, N: int, H: int, W: int,
                 train: bool = True, need_grad: bool = True, sync_bn: bool = False, world_size: int = 1,
                 fp32: bool = True, with_loss: bool = False, fuse_dgrad: bool = True):
        self.comps = parse_spec(spec)
        self.preact, self.use_proj, self.p = preact, use_proj, float(dropout_prob)
        self.N, self.H, self.W = N, H, W
        self.train, self.need_grad = train, need_grad
        self.sync = sync_bn and world_size > 1 and train
        self.world = world_size
        self.fp32 = fp32
        self.with_loss = with_loss
        self.slots: List[Slot] = []
        self.fwd: List[Op] = []
        self.bwd: List[Op] = []
        self.fwd_hooks: List[Hook] = []
        self.bwd_hooks: List[Hook] = []
        self._back = []                 # stack of backward emitters
        self.grad_order: List[str] = []
        self.param_keys: List[str] = []
        self._named: Dict[str, int] = {}
        self._site = 0
        self._packed: Dict[str, Tuple[int, int]] = {}
        self._ws_need = []              # geometries of ops that share the workspace slot
        self.fwd_packs: List[Op] = []   # weight-pack ops, emitted in front of the forward
        self._stats_of: Dict[int, Tuple[int, int]] = {}    # tensor slot -> (partial slot, rows) written by its producer's epilogue
        self._wants_colsum = set()                         # outputs of biased convolutions: their gradient's per-channel sums are the bias gradient
        self._colsum_of: Dict[int, Tuple[int, int]] = {}   # gradient slot -> (partial slot, rows) left by the pass that wrote it
        self._tail_bn: Dict[int, dict] = {}                # output slot of a BN(+add)+ReLU -> its record: the consumer's dgrad reduces its backward sums
        self._dpart_of: Dict[int, Tuple[int, int]] = {}    # gradient tensor slot -> BN-backward partial sums from the dgrad epilogue
        self._site_of: Dict[int, Tuple[float, int, bool]] = {}   # BN_APPLY output slot -> (dropout p, site, mask recomputable)
        self.fuse = True                # BN batch statistics in the producing conv's epilogue (free: no extra operand reads)
        # BN-backward sums reduced in the dgrad epilogue (its operands are read as 16-byte chunks by the LDS-staged epilogue):
        # removes the bn_bwd_reduce pass (-1.15 ms) for +0.36 ms of dgrad per WRN-28-10 step (measured)
        self.fuse_dgrad = fuse_dgrad and os.environ.get('RN_NO_DGRAD_FUSION', '0') != '1'      # (A/B: the BatchNorm-backward sums as a pass of their own)
        # a top-level "n [a] mp" as ONE pass forward and two gather passes backward (RN_NO_POOL_FUSION=1: the separate ops, A/B)
        self.fuse_pool = os.environ.get('RN_NO_POOL_FUSION', '0') != '1'

    # ---- slots ------------------------------------------------------------------------------------------
    def slot(self, name, role, shape, dtype, key=None):
        self.slots.append(Slot(name, role, tuple(int(d) for d in shape), dtype, key))
        return len(self.slots) - 1

    def act(self, name, N, H, W, C):
        return T(self.slot(name, 'act', (N, H, W, C), 'T'), N, H, W, C)

    def f32(self, name, shape):
        return self.slot(name, 'f32', shape, 'f32')

    def fold(self, name, nblk, C):
        """the hand-off buffer of ONE finalize op over many partial rows (rows split over workgroups, csrc/bn.hip), or -1: the plain kernel"""
        nbytes = 0 if os.environ.get('RN_NO_FOLD', '0') == '1' else bn_fold_bytes(nblk, C)
        return self.slot(name, 'fold', ((nbytes + 3) // 4,), 'f32') if nbytes else -1

    def param(self, key, shape):
        if key not in self._named:
            self._named[key] = self.slot(key, 'param', shape, 'f32', key)
            self.param_keys.append(key)
        return self._named[key]

    def buffer(self, key, shape, dtype='f32'):
        if key not in self._named:
            self._named[key] = self.slot(key, 'buffer', shape, dtype, key)
        return self._named[key]

    def grad(self, key, shape):
        g = 'grad:' + key
        if g not in self._named:
            self._named[g] = self.slot(g, 'grad', shape, 'f32', key)
        if key not in self.grad_order:
            self.grad_order.append(key)
        return self._named[g]

    def ws(self):
        if 'ws' not in self._named:
            self._named['ws'] = self.slot('workspace', 'ws', (0,), 'u8')
        return self._named['ws']

    def ws_side(self):
        if 'ws_side' not in self._named:
            self._named['ws_side'] = self.slot('workspace_side', 'ws', (0,), 'u8')
        return self._named['ws_side']

    def fork_wgrad(self, g) -> bool:
        """weight-gradient ops that run on the executor's side stream, beside the data-gradient / BatchNorm chain (which is
        the critical path of the backward and leaves matrix-pipe / HBM idle time in every kernel's prologue, epilogue and
        tail round).  Measured (eager launches): WRN-28-10 +6 %, WRN-50-2 +2.5...4.7 %.  Thin CIFAR nets are launch-bound
        and run as hipGraphs, where a captured fork/join replays 25 % slower than no fork at all, so only GEMMs of at
        least 5 GFLOP are forked and a plan with forked ops is not graph-captured (executor.py)."""
        mode = os.environ.get('RN_FORK_WGRAD', 'auto')
        if mode in ('0', '1'):
            return mode == '1'
        return 2.0 * g['N'] * g['P'] * g['Q'] * g['K'] * g['R'] * g['S'] * g['C'] >= 5e9

    # ---- small emitters ---------------------------------------------------------------------------------
    def geom(self, x: T, K, k, stride, pad):
        P = (x.H + 2 * pad - k) // stride + 1
        Q = (x.W + 2 * pad - k) // stride + 1
        return dict(N=x.N, H=x.H, W=x.W, C=x.C, P=P, Q=Q, K=K, R=k, S=k, stride=stride, pad=pad)

    def packed_weights(self, key, K, k, C, need_dgrad):
        """per-forward repack of one conv weight into the compute dtype (KRSC for fwd/wgrad, CRSK for dgrad).
        fp32 plans read the master parameter directly as the forward operand.  (Packing the transposed copy by an op of its
        own on the side stream during the forward was measured 3 % slower: twice the reads, and it competes with the
        forward's kernels.)"""
        w = self.param(key, (K, k, k, C))
        if key in self._packed:
            return self._packed[key]
        need_dgrad = need_dgrad and self.need_grad
        wf = w if self.fp32 else self.slot(key + ':fwd', 'act', (K, k * k, C), 'T')
        wd = self.slot(key + ':dgrad', 'act', (C, k * k, K), 'T') if need_dgrad else -1
        if (not self.fp32) or need_dgrad:
            flags = (ir.F_SKIP_FWD_PACK if self.fp32 else 0) | (ir.F_NEED_DGRAD_PACK if need_dgrad else 0)
            # all weight packs sit at the head of the forward (they depend on parameters only): the executor turns the run
            # into ONE launch (rn_pack_weights_batch)
            self.fwd_packs.append(Op(ir.OP_PACK_W, buf=dict(w=w, w_fwd=-1 if self.fp32 else wf, w_dgrad=wd),
                                     dim=dict(K=K, RS=k * k, C=C), flags=flags, note=key))
        self._packed[key] = (wf, wd)
        return wf, wd

    def conv_fwd(self, x: T, key, K, k, stride, pad, name, res: Optional[T] = None, res_mode=ir.RES_NONE, need_dgrad=True, stats=True, bias=-1):
        """stats: the epilogue also reduces the per-channel (sum, sum^2) of the stored output -- the batch statistics of
        the BatchNorm that reads it next (train mode) -- so that layer needs no statistics pass."""
        g = self.geom(x, K, k, stride, pad)
        wf, wd = self.packed_weights(key, K, k, x.C, need_dgrad)
        y = self.act(name, x.N, g['P'], g['Q'], K)
        part = -1
        if stats and self.train and self.fuse:
            rows = conv_stats_rows(g)
            part = self.f32(name + ':stats', (rows, 2, K))
            self._stats_of[y.s] = (part, rows)
        self.fwd.append(Op(ir.OP_CONV_FWD, buf=dict(x=x.s, w_fwd=wf, y=y.s, res=res.s if res else -1, stats=part, bias=bias),
                           dim=dict(g, res_mode=res_mode, res_C=res.C if res else 0), note=key))
        return y, g, wd

    def conv_bwd(self, ops: List[Op], x: T, dy: T, key, g, wd, dx_name, need_dx=True, res: Optional[T] = None,
                 res_mode=ir.RES_NONE, accum_into: Optional[T] = None, fuse_bn: Optional[dict] = None):
        """emits wgrad (+ dgrad).  returns dx tensor (or None).  fuse_bn = dict(x, mask, coef, p) of the BN+ReLU(+dropout)
        that produced this conv's input: the dgrad epilogue then also reduces that layer's two backward sums."""
        dw = self.grad(key, (g['K'], g['R'], g['S'], g['C']))
        fork = self.fork_wgrad(g)
        ops.append(Op(ir.OP_CONV_WGRAD, buf=dict(x=x.s, dy=dy.s, dw=dw, ws=self.ws_side() if fork else self.ws()), dim=dict(g),
                      flags=ir.F_FORK if fork else 0, note=key))
        self._ws_need.append(('wgrad', dict(g)))
        self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
        if not need_dx:
            return None
        dx = accum_into or self.act(dx_name, x.N, x.H, x.W, x.C)
        buf = dict(dy=dy.s, w_dgrad=wd, dx=dx.s, res=res.s if res else -1)
        fp, flags = {}, (ir.F_ACCUM if accum_into else 0)
        # (measured, round 4: the two stride-2 data gradients WITHOUT the fused sums -- 91 -> 68 and 79 -> 65 us -- plus a pass of their own for the sums: 5.95 vs 5.92 ms per step; fused stays)
        if fuse_bn is not None and self.fuse_dgrad:        # with accum_into: this must be the LAST accumulation into dx
            rows = conv_stats_rows(g, dgrad=True)
            part = self.f32(dx_name + ':dpartial', (rows, 2, x.C))
            mask = fuse_bn['mask']
            buf.update(bn_x=fuse_bn['x'].s, bn_mask=mask.s if mask is not None else -1, bn_coef=fuse_bn['coef'], bn_partial=part)
            fp['gscale'] = 1.0 / (1.0 - fuse_bn['p']) if fuse_bn['p'] > 0 else 1.0
            # a plain BatchNorm + ReLU (nothing added before the ReLU, no dropout -- bn_apply's own record, the one bn_bwd recomputes its mask by): the mask is
            # [x * scale + shift > 0] of the x the sums read anyway --
            # the kernel may leave the mask tensor unread (rn_conv_epilogue.mask_from_x; a third of a data-gradient epilogue's reads)
            if mask is not None and self._site_of.get(mask.s, (0, 0, False))[2] and fuse_bn['p'] == 0 and os.environ.get('RN_NO_XMASK', '0') != '1':
                flags |= ir.F_MASK_RECOMPUTE
            self._dpart_of[dx.s] = (part, rows)
        ops.append(Op(ir.OP_CONV_DGRAD, buf=buf, dim=dict(g, res_mode=res_mode, res_C=res.C if res else 0), fp=fp, flags=flags, note=key))
        return dx

    def bn_coef(self, x: T, pre: str):
        """statistics (+ SyncBN hook) + finalize -> coef slot [4][C]."""
        C = x.C
        gamma, beta = self.param(pre + '.weight', (C,)), self.param(pre + '.bias', (C,))
        rm, rv = self.buffer(pre + '.running_mean', (C,)), self.buffer(pre + '.running_var', (C,))
        nbt = self.buffer(pre + '.num_batches_tracked', (), 'i64')
        coef = self.f32(pre + ':coef', (4, C))
        if self.train:
            if x.s in self._stats_of:                      # statistics already reduced by the producing conv's epilogue
                part, nblk = self._stats_of[x.s]
            else:
                nblk = bn_partials(x.M, C)
                part = self.f32(pre + ':partial', (nblk, 2, C))
                self.fwd.append(Op(ir.OP_BN_STATS, buf=dict(x=x.s, partial=part), dim=dict(M=x.M, C=C, nblk=nblk), note=pre))
            count = x.M
            if self.sync:
                # SyncBN: the [rows][2][C] partials are first summed locally to one [2][C] row (the backward-finalize kernel
                # is exactly that reduction), THAT row is all-reduced (1.3 KB instead of 1.3 MB per layer), then every rank
                # finalizes over the global row count
                local = self.f32(pre + ':local', (2, C))
                sg, sb = self.f32(pre + ':local_g', (C,)), self.f32(pre + ':local_b', (C,))      # by-products, unused
                self.fwd.append(Op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part, dsum=local, dgamma=sg, dbeta=sb, fold=self.fold(pre + ':lfold', nblk, C)),
                                   dim=dict(nblk=nblk, C=C), note=pre))
                self.fwd_hooks.append(Hook(len(self.fwd), 'allreduce_f32', slot=local))
                part, nblk = local, 1
                count = x.M * self.world
            self.fwd.append(Op(ir.OP_BN_FINALIZE, buf=dict(partial=part, gamma=gamma, beta=beta, running_mean=rm,
                                                           running_var=rv, nbt=nbt, coef=coef, fold=self.fold(pre + ':fold', nblk, C)),
                               dim=dict(nblk=nblk, count=count, C=C), fp=dict(eps=BN_EPS, momentum=BN_MOMENTUM),
                               flags=ir.F_TRAIN, note=pre))
        else:
            self.fwd.append(Op(ir.OP_BN_FINALIZE, buf=dict(partial=-1, gamma=gamma, beta=beta, running_mean=rm,
                                                           running_var=rv, nbt=nbt, coef=coef),
                               dim=dict(nblk=0, count=x.M, C=C), fp=dict(eps=BN_EPS, momentum=BN_MOMENTUM), note=pre))
        return coef

    def bn_apply(self, x: T, coef, name, relu, drop=False, res: Optional[T] = None, res_mode=ir.RES_NONE):
        out = self.act(name, x.N, x.H, x.W, x.C)
        p = self.p if (drop and self.train) else 0.0
        site = 0
        if p > 0:
            self._site += 1
            site = self._site
        self.fwd.append(Op(ir.OP_BN_APPLY, buf=dict(x=x.s, coef=coef, res=res.s if res else -1, out=out.s),
                           dim=dict(N=x.N, H=x.H, W=x.W, C=x.C, res_mode=res_mode, res_C=res.C if res else 0),
                           fp=dict(p=p), flags=ir.F_RELU if relu else 0, seed=site, note=name))
        # recomputing the mask in the backward pays only without dropout (fma + compare); re-evaluating the dropout hash
        # per element made bn_bwd_reduce 1.7x slower than reading the saved output (measured, WRN-28-10 p=0.3)
        self._site_of[out.s] = (p, site, res is None and p == 0.0)
        return out, p

    def bn_bwd(self, ops, dout: T, x: T, mask: Optional[T], coef, pre, dx_name, p=0.0, add: Optional[T] = None,
               add_mode=ir.RES_NONE, write_g_name=None):
        """backward of out = [drop][relu](bn(x) [+ res]).  returns (dx, g or None)."""
        C = x.C
        gscale = 1.0 / (1.0 - p) if p > 0 else 1.0
        dsum = self.f32(pre + ':dsum', (2, C))
        fl = (ir.F_RELU if mask is not None else 0) | (ir.F_TRAIN if self.train else 0)
        # the ReLU/dropout mask of out = drop(relu(x*scale+shift)) is a function of x, the coefficients and the dropout
        # hash: recompute it instead of reading the saved output (one tensor read less in each backward pass)
        site, mslot = 0, (mask.s if mask else -1)
        if mask is not None and self._site_of.get(mask.s, (0, 0, False))[2]:
            p_fwd, site, _ = self._site_of[mask.s]
            assert abs(p_fwd - p) < 1e-12
            fl |= ir.F_MASK_RECOMPUTE
            mslot = -1
        apply_fl, apply_mslot, apply_site = fl, mslot, site
        if dout.s in self._dpart_of:                       # the two sums were reduced by the dgrad that produced dout
            part, nblk = self._dpart_of[dout.s]
        else:
            nblk = bn_partials(x.M, C)
            part = self.f32(pre + ':dpartial', (nblk, 2, C))
            ops.append(Op(ir.OP_BN_BWD_REDUCE, buf=dict(dout=dout.s, x=x.s, mask=mslot, coef=coef, partial=part),
                          dim=dict(M=x.M, C=C, nblk=nblk), fp=dict(gscale=gscale, p=p if mslot < 0 else 0.0), flags=fl, seed=site, note=pre))
        dg, db = self.grad(pre + '.weight', (C,)), self.grad(pre + '.bias', (C,))
        count = x.M
        ops.append(Op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part, dsum=dsum, dgamma=dg, dbeta=db, fold=self.fold(pre + ':dfold', nblk, C)), dim=dict(nblk=nblk, C=C), note=pre))
        if self.sync:
            # dgamma/dbeta stay local sums (the gradient all-reduce averages them); dsum must be global
            self.bwd_hooks.append(Hook(len(ops), 'allreduce_f32', slot=dsum))
            count = x.M * self.world
        self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
        dx = self.act(dx_name, x.N, x.H, x.W, C)
        g = self.act(write_g_name, x.N, x.H, x.W, C) if write_g_name else None
        ops.append(Op(ir.OP_BN_BWD_APPLY,
                      buf=dict(dout=dout.s, x=x.s, mask=apply_mslot, coef=coef, dsum=dsum,
                               add=add.s if add else -1, dx=dx.s, g_out=g.s if g else -1),
                      dim=dict(N=x.N, H=x.H, W=x.W, C=C, add_mode=add_mode, add_C=add.C if add else 0, count=count),
                      fp=dict(gscale=gscale, p=p if apply_mslot < 0 else 0.0), flags=apply_fl | (ir.F_WRITE_G if g else 0), seed=apply_site, note=pre))
        return dx, g

    # ---- residual blocks --------------------------------------------------------------------------------
    def block(self, bp: str, kind: str, cin: int, down: bool, i: T) -> T:
        convs, norms, cout = block_convs(kind, cin, down, self.preact)
        n = len(convs)
        proj = down and self.use_proj
        recs = []
        sc_g = sc_wd = None
        if self.preact:
            x = i
            for j, (ci, co, k, s, pd) in enumerate(convs, 1):
                coef = self.bn_coef(x, f'{bp}._norm{j}')
                a, p = self.bn_apply(x, coef, f'{bp}.a{j}', relu=True, drop=True)
                res, mode = None, ir.RES_NONE
                if j == n:
                    if not down:
                        res, mode = i, ir.RES_SAME
                    elif proj:
                        res, sc_g, sc_wd = self.conv_fwd(i, f'{bp}._proj.weight', cout, 1, 2, 0, f'{bp}.sc', stats=False)
                        mode = ir.RES_SAME
                    else:
                        res, mode = i, ir.RES_DOWN2PAD
                y, g, wd = self.conv_fwd(a, f'{bp}._conv{j}.weight', co, k, s, pd, f'{bp}.y{j}', res, mode)
                recs.append(dict(x=x, a=a, coef=coef, p=p, g=g, wd=wd, j=j))
                x = y
            h = x

            def backward(dh: T, ops):
                gcur = dh
                for r in reversed(recs):
                    j = r['j']
                    da = self.conv_bwd(ops, r['a'], gcur, f'{bp}._conv{j}.weight', r['g'], r['wd'], f'{bp}.da{j}',
                                       fuse_bn=dict(x=r['x'], mask=r['a'], coef=r['coef'], p=r['p']))
                    add, mode = None, ir.RES_NONE
                    if j == 1 and not proj:
                        add, mode = dh, (ir.RES_SAME if not down else ir.RES_UP2)
                    gcur, _ = self.bn_bwd(ops, da, r['x'], r['a'], r['coef'], f'{bp}._norm{j}', f'{bp}.dx{j}', r['p'], add, mode)
                if proj:
                    self.conv_bwd(ops, i, dh, f'{bp}._proj.weight', sc_g, sc_wd, None, accum_into=gcur)
                return gcur
        else:
            x = i
            p_in = 0.0
            if self.p > 0 and self.train:
                self._site += 1
                x0 = self.act(f'{bp}.x0', i.N, i.H, i.W, i.C)
                self.fwd.append(Op(ir.OP_DROPOUT_FWD, buf=dict(x=i.s, out=x0.s), dim=dict(n_lo=i.M * i.C & 0x7fffffff, n_hi=(i.M * i.C) >> 31),
                                   fp=dict(p=self.p), seed=self._site, note=bp))
                x, p_in, site_in = x0, self.p, self._site
            h = None
            for j, (ci, co, k, s, pd) in enumerate(convs, 1):
                y, g, wd = self.conv_fwd(x, f'{bp}._conv{j}.weight', co, k, s, pd, f'{bp}.y{j}')
                coef = self.bn_coef(y, f'{bp}._norm{j}')
                rec = dict(x=x, y=y, coef=coef, g=g, wd=wd, j=j)
                if j < n:
                    a, p = self.bn_apply(y, coef, f'{bp}.a{j}', relu=True, drop=True)
                    rec.update(out=a, p=p)
                    x = a
                else:
                    if not down:
                        res, mode = i, ir.RES_SAME
                    elif proj:
                        res, sc_g, sc_wd = self.conv_fwd(i, f'{bp}._proj.weight', cout, 1, 2, 0, f'{bp}.sc', stats=False)
                        mode = ir.RES_SAME
                    else:
                        res, mode = i, ir.RES_DOWN2PAD
                    h, _ = self.bn_apply(y, coef, f'{bp}.h', relu=True, drop=False, res=res, res_mode=mode)
                    rec.update(out=h, p=0.0)
                    self._tail_bn[h.s] = dict(x=y, mask=h, coef=coef, p=0.0)
                recs.append(rec)

            def backward(dh: T, ops):
                gcur, gm = None, None
                for r in reversed(recs):
                    j = r['j']
                    dout = dh if j == n else gcur
                    dy, g_ = self.bn_bwd(ops, dout, r['y'], r['out'], r['coef'], f'{bp}._norm{j}', f'{bp}.dy{j}', r['p'],
                                         write_g_name=f'{bp}.gm' if j == n else None)
                    if j == n:
                        gm = g_
                    if j > 1:
                        prev = recs[j - 2]          # the BN+ReLU(+dropout) whose output is this conv's input
                        gcur = self.conv_bwd(ops, r['x'], dy, f'{bp}._conv{j}.weight', r['g'], r['wd'], f'{bp}.da{j - 1}',
                                             fuse_bn=dict(x=prev['y'], mask=prev['out'], coef=prev['coef'], p=prev['p']))
                    else:
                        plain = p_in == 0.0
                        res, mode = None, ir.RES_NONE
                        if not proj:
                            res, mode = gm, (ir.RES_SAME if not down else ir.RES_UP2)
                        # the BN (+add+ReLU) that produced this block's input: its backward sums are reduced by whichever
                        # dgrad completes di (conv1's, or the projection's accumulate when there is one)
                        tail = self._tail_bn.get(i.s) if plain else None
                        if plain:
                            gcur = self.conv_bwd(ops, r['x'], dy, f'{bp}._conv{j}.weight', r['g'], r['wd'], f'{bp}.di', res=res, res_mode=mode,
                                                 fuse_bn=None if proj else tail)
                        else:   # dropout1 sits between the block input and conv1: its mask applies before the merge
                            t = self.conv_bwd(ops, r['x'], dy, f'{bp}._conv{j}.weight', r['g'], r['wd'], f'{bp}.dx0')
                            gcur = self.act(f'{bp}.di', i.N, i.H, i.W, i.C)
                            ops.append(Op(ir.OP_DROPOUT_BWD, buf=dict(dout=t.s, out=r['x'].s, din=gcur.s),
                                          dim=dict(n_lo=i.M * i.C & 0x7fffffff, n_hi=(i.M * i.C) >> 31), fp=dict(p=p_in), seed=site_in, note=bp))
                            if res is not None:
                                ops.append(Op(ir.OP_ADD_RES, buf=dict(dst=gcur.s, res=res.s),
                                              dim=dict(N=i.N, H=i.H, W=i.W, C=i.C, res_mode=mode, res_C=res.C), note=bp))
                        if proj:
                            self.conv_bwd(ops, i, gm, f'{bp}._proj.weight', sc_g, sc_wd, f'{bp}.di', accum_into=gcur, fuse_bn=tail)
                return gcur
        self._back.append(backward)
        return h

    # ---- whole network ----------------------------------------------------------------------------------
    def lower(self) -> Plan:
        comps = self.comps
        cur: Optional[T] = None
        named = {}
        idx = 0
        while idx < len(comps):
            c = comps[idx]
            pre = f'_architecture.{idx}'
            if c.kind == 'conv' and cur is not None:
                # a top-level convolution deeper in the network (resnet.py:126-131 builds Conv2d(i, o, k, s, p) with bias wherever a 'c' token stands):
                # the block convolutions' kernels with the bias in the epilogue; the bias gradient is the per-channel sum of dy
                k, s, pd = c.args
                ce = 4 if self.fp32 else 8
                if c.cin != cur.C:
                    raise ValueError(f"{pre}: convolution expects {c.cin} input channels, the map has {cur.C}")
                if c.cin % ce or c.cout % ce:
                    raise NotImplementedError(f"{pre}: channel counts must be multiples of {ce} on the accelerated path")
                w_key = pre + '.weight'
                self.param(w_key, (c.cout, k, k, c.cin))
                b_ = self.param(pre + '.bias', (c.cout,))
                y, g, wd = self.conv_fwd(cur, w_key, c.cout, k, s, pd, pre + ':y', bias=b_)

                def conv_back(dy: T, ops, x=cur, g=g, wd=wd, pre=pre, w_key=w_key):
                    dx = self.conv_bwd(ops, x, dy, w_key, g, wd, pre + ':dx')
                    K = g['K']
                    db = self.grad(pre + '.bias', (K,))
                    s2, sq = self.f32(pre + ':dbsum', (2, K)), self.f32(pre + ':dbsq', (K,))          # by-products, unused
                    nblk = bn_partials(dy.M, K)
                    part = self.f32(pre + ':dbpartial', (nblk, 2, K))
                    ops.append(Op(ir.OP_BN_STATS, buf=dict(x=dy.s, partial=part), dim=dict(M=dy.M, C=K, nblk=nblk), note=pre))
                    ops.append(Op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part, dsum=s2, dgamma=sq, dbeta=db, fold=self.fold(pre + ':dbfold', nblk, K)),
                                  dim=dict(nblk=nblk, C=K), note=pre))
                    self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
                    return dx
                self._back.append(conv_back)
                cur = y
            elif c.kind == 'conv':
                k, s, pd = c.args
                xin = self.slot('x', 'input', (self.N, c.cin, self.H, self.W), 'f32')
                named['x'] = xin
                xt = T(xin, self.N, self.H, self.W, c.cin)
                w, b = self.param(pre + '.weight', (c.cout, k, k, c.cin)), self.param(pre + '.bias', (c.cout,))
                CP = 4 if self.fp32 else 8
                s2d = (not self.fp32 and (k, s, pd) == (7, 2, 3) and c.cin <= 4 and self.H % 2 == 0 and self.W % 2 == 0 and c.cout % 16 == 0
                       and os.environ.get('RN_VALU_STEM', '0') != '1' and os.environ.get('RN_NO_S2D_STEM', '0') != '1')
                if s2d:
                    # the ImageNet stem (7 x 7, stride 2, padding 3) as a 4 x 4 / stride-1 / VALID convolution over a space-to-depth image (16 channels per
                    # s2d pixel, stored pre-padded: csrc/misc.hip img_to_s2d): 4 K tiles of 64 instead of the 7 of the padded-channel route below, and a
                    # kernel row is 128 contiguous bytes per output pixel.  Same output tensor, same weight / bias gradients (the [K][4][4][16] weight
                    # gradient is mapped back to [K][7][7][C]).
                    xp = self.act(pre + ':xs2d', self.N, self.H // 2 + 3, self.W // 2 + 3, 16)
                    self.fwd.append(Op(ir.OP_IMG_TO_S2D, buf=dict(x=xin, out=xp.s), dim=dict(N=self.N, C=c.cin, H=self.H, W=self.W), note=pre))
                    wp = self.slot(pre + ':ws2d', 'act', (c.cout, 16, 16), 'T')
                    self.fwd.append(Op(ir.OP_PACK_STEM_W_S2D, buf=dict(w=w, w_s2d=wp), dim=dict(K=c.cout, C=c.cin), note=pre))
                    g = self.geom(xp, c.cout, 4, 1, 0)
                    alg = k * k * c.cin                           # MACs per output of the reference's layer (bench.py's algorithmic FLOPs)
                    y = self.act(pre + ':y', self.N, g['P'], g['Q'], c.cout)
                    part = -1
                    if self.train and self.fuse:
                        rows = conv_stats_rows(g)
                        part = self.f32(pre + ':stats', (rows, 2, c.cout))
                        self._stats_of[y.s] = (part, rows)
                    self.fwd.append(Op(ir.OP_CONV_FWD, buf=dict(x=xp.s, w_fwd=wp, y=y.s, res=-1, stats=part, bias=b), dim=dict(g, res_mode=0, res_C=0, alg_macs=alg), note=pre))
                    self._wants_colsum.add(y.s)

                    def stem_back(dy: T, ops, g=g, xp=xp, pre=pre, cin=c.cin, alg=alg):
                        K = g['K']
                        dwp = self.f32(pre + ':dws2d', (K, 16, 16))
                        ops.append(Op(ir.OP_CONV_WGRAD, buf=dict(x=xp.s, dy=dy.s, dw=dwp, ws=self.ws()), dim=dict(g, alg_macs=alg), note=pre))
                        self._ws_need.append(('wgrad', dict(g)))
                        dw, db = self.grad(pre + '.weight', (K, 7, 7, cin)), self.grad(pre + '.bias', (K,))
                        ops.append(Op(ir.OP_UNPACK_STEM_DW_S2D, buf=dict(dw_s2d=dwp, dw=dw), dim=dict(K=K, C=cin), note=pre))
                        s2, sq = self.f32(pre + ':dbsum', (2, K)), self.f32(pre + ':dbsq', (K,))       # by-products, unused
                        if dy.s in self._colsum_of:
                            part, nblk = self._colsum_of[dy.s]
                        else:
                            nblk = bn_partials(dy.M, K)
                            part = self.f32(pre + ':dbpartial', (nblk, 2, K))
                            ops.append(Op(ir.OP_BN_STATS, buf=dict(x=dy.s, partial=part), dim=dict(M=dy.M, C=K, nblk=nblk), note=pre))
                        ops.append(Op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part, dsum=s2, dgamma=sq, dbeta=db, fold=self.fold(pre + ':dbfold', nblk, K)), dim=dict(nblk=nblk, C=K), note=pre))
                        self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
                        return None
                    self._back.append(stem_back)
                    cur = y
                    idx += 1
                    continue
                if c.cin <= CP and c.cout % 16 == 0 and os.environ.get('RN_VALU_STEM', '0') != '1':
                    # MFMA route: NHWC image with the channels zero-padded to one 16-byte chunk, then the implicit-GEMM
                    # kernels with bias (+ BN statistics) in the epilogue and the im2col wgrad (dy read once)
                    xp = self.act(pre + ':xpad', self.N, self.H, self.W, CP)
                    self.fwd.append(Op(ir.OP_IMG_TO_NHWC, buf=dict(x=xin, out=xp.s), dim=dict(N=self.N, C=c.cin, H=self.H, W=self.W, CP=CP), note=pre))
                    wp = self.slot(pre + ':wpad', 'act', (c.cout, k * k, CP), 'T')
                    self.fwd.append(Op(ir.OP_PACK_STEM_W, buf=dict(w=w, w_padded=wp), dim=dict(K=c.cout, RS=k * k, C=c.cin, CP=CP), note=pre))
                    g = self.geom(xp, c.cout, k, s, pd)
                    y = self.act(pre + ':y', self.N, g['P'], g['Q'], c.cout)
                    part = -1
                    if self.train and self.fuse:
                        rows = conv_stats_rows(g)
                        part = self.f32(pre + ':stats', (rows, 2, c.cout))
                        self._stats_of[y.s] = (part, rows)
                    self.fwd.append(Op(ir.OP_CONV_FWD, buf=dict(x=xp.s, w_fwd=wp, y=y.s, res=-1, stats=part, bias=b), dim=dict(g, res_mode=0, res_C=0), note=pre))
                    self._wants_colsum.add(y.s)           # whoever writes d(y) may leave its per-channel sums (the bias gradient) behind

                    def stem_back(dy: T, ops, g=g, xp=xp, pre=pre, cin=c.cin, CP=CP):
                        K, RS = g['K'], g['R'] * g['S']
                        dwp = self.f32(pre + ':dwpad', (K, RS, CP))
                        ops.append(Op(ir.OP_CONV_WGRAD, buf=dict(x=xp.s, dy=dy.s, dw=dwp, ws=self.ws()), dim=dict(g), note=pre))
                        self._ws_need.append(('wgrad', dict(g)))
                        dw, db = self.grad(pre + '.weight', (K, g['R'], g['S'], cin)), self.grad(pre + '.bias', (K,))
                        ops.append(Op(ir.OP_UNPACK_STEM_DW, buf=dict(dw_padded=dwp, dw=dw), dim=dict(K=K, RS=RS, C=cin, CP=CP), note=pre))
                        s2, sq = self.f32(pre + ':dbsum', (2, K)), self.f32(pre + ':dbsq', (K,))       # by-products, unused
                        if dy.s in self._colsum_of:               # bias gradient = per-channel sum of dy: left behind by the pass that wrote dy ...
                            part, nblk = self._colsum_of[dy.s]
                        else:                                     # ... or one more pass over it
                            nblk = bn_partials(dy.M, K)
                            part = self.f32(pre + ':dbpartial', (nblk, 2, K))
                            ops.append(Op(ir.OP_BN_STATS, buf=dict(x=dy.s, partial=part), dim=dict(M=dy.M, C=K, nblk=nblk), note=pre))
                        ops.append(Op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part, dsum=s2, dgamma=sq, dbeta=db, fold=self.fold(pre + ':dbfold', nblk, K)), dim=dict(nblk=nblk, C=K), note=pre))
                        self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
                        return None
                    self._back.append(stem_back)
                    cur = y
                    idx += 1
                    continue
                g = self.geom(xt, c.cout, k, s, pd)
                y = self.act(pre + ':y', self.N, g['P'], g['Q'], c.cout)
                self.fwd.append(Op(ir.OP_STEM_FWD, buf=dict(x=xin, w=w, bias=b, y=y.s), dim=dict(g), note=pre))

                def stem_back(dy: T, ops, g=g, xin=xin, pre=pre):
                    dw, db = self.grad(pre + '.weight', (g['K'], g['R'], g['S'], g['C'])), self.grad(pre + '.bias', (g['K'],))
                    ops.append(Op(ir.OP_STEM_WGRAD, buf=dict(x=xin, dy=dy.s, dw=dw, db=db, ws=self.ws()), dim=dict(g), note=pre))
                    self._ws_need.append(('stem', dict(g)))
                    self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
                    return None
                self._back.append(stem_back)
                cur = y
            elif c.kind == 'norm':
                relu = idx + 1 < len(comps) and comps[idx + 1].kind == 'act'
                nxt = comps[idx + (2 if relu else 1)] if idx + (2 if relu else 1) < len(comps) else None
                ce = 4 if self.fp32 else 8
                if (nxt is not None and nxt.kind == 'maxpool' and not self.sync and self.fuse_pool and cur.C % ce == 0 and 256 % (cur.C // ce) == 0):
                    # "n [a] mp": BatchNorm-apply (+ReLU) + MaxPool as one pass each way (the ImageNet stems): the normalised activation
                    # (WRN-50-2-B: 3.3 GB at batch 256) and its gradient are never stored
                    k, s, pd = nxt.args
                    mpre = f'_architecture.{idx + (2 if relu else 1)}'
                    coef = self.bn_coef(cur, pre)
                    P, Q = (cur.H + 2 * pd - k) // s + 1, (cur.W + 2 * pd - k) // s + 1
                    y = self.act(mpre + ':y', cur.N, P, Q, cur.C)
                    d = dict(N=cur.N, H=cur.H, W=cur.W, C=cur.C, k=k, stride=s, pad=pd)
                    am = self.slot(mpre + ':argmax', 'u8', (cur.N, P, Q, cur.C), 'u8') if self.need_grad else -1
                    # the window winners' INPUT values: the backward sums are then taken at pooled resolution (2 x 1/4 of the map read instead of the map)
                    xsel = self.act(mpre + ':xsel', cur.N, P, Q, cur.C).s if (self.need_grad and self.train and os.environ.get('RN_POOL_GATHER_SUMS', '0') != '1') else -1
                    fl = ir.F_RELU if relu else 0
                    self.fwd.append(Op(ir.OP_BN_POOL_FWD, buf=dict(x=cur.s, coef=coef, y=y.s, argmax=am, xsel=xsel), dim=d, flags=fl, note=pre))

                    def norm_pool_back(dy: T, ops, x=cur, coef=coef, pre=pre, d=d, am=am, fl=fl, xsel=xsel):
                        C = x.C
                        nblk = min(x.N * x.H, 2048)
                        part = self.f32(pre + ':dpartial', (nblk, 2, C))
                        dsum = self.f32(pre + ':dsum', (2, C))
                        flt = fl | (ir.F_TRAIN if self.train else 0)
                        ops.append(Op(ir.OP_BN_POOL_BWD_REDUCE, buf=dict(dy=dy.s, argmax=am, x=x.s, coef=coef, partial=part, xsel=xsel), dim=dict(d, nblk=nblk, npix=dy.M),
                                      flags=flt, note=pre))
                        dg, db = self.grad(pre + '.weight', (C,)), self.grad(pre + '.bias', (C,))
                        ops.append(Op(ir.OP_BN_BWD_FINALIZE, buf=dict(partial=part, dsum=dsum, dgamma=dg, dbeta=db, fold=self.fold(pre + ':dfold', nblk, C)), dim=dict(nblk=nblk, C=C), note=pre))
                        self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
                        dx = self.act(pre + ':dx', x.N, x.H, x.W, C)
                        sums, rows = -1, 0
                        if x.s in self._wants_colsum and os.environ.get('RN_POOL_NO_COLSUM', '0') != '1':
                            quad = (d['k'], d['stride'], d['pad']) == (3, 2, 1) and x.H % 2 == 0 and x.W % 2 == 0
                            rows = min(x.N * (x.H // 2 if quad else x.H), 8192)         # the pass's own grid: one row of sums per workgroup
                            sums = self.f32(pre + ':dxsums', (rows, 2, C))
                            self._colsum_of[dx.s] = (sums, rows)
                        ops.append(Op(ir.OP_BN_POOL_BWD_APPLY, buf=dict(dy=dy.s, argmax=am, x=x.s, coef=coef, dsum=dsum, dx=dx.s, sums=sums), dim=dict(d, count=x.M, rows=rows),
                                      flags=flt, note=pre))
                        return dx
                    self._back.append(norm_pool_back)
                    cur = y
                    idx += 3 if relu else 2
                    continue
                coef = self.bn_coef(cur, pre)
                out, _ = self.bn_apply(cur, coef, pre + ':out', relu=relu)
                self._tail_bn[out.s] = dict(x=cur, mask=out if relu else None, coef=coef, p=0.0)

                def norm_back(dout: T, ops, x=cur, out=out, coef=coef, pre=pre, relu=relu):
                    dx, _ = self.bn_bwd(ops, dout, x, out if relu else None, coef, pre, pre + ':dx')
                    return dx
                self._back.append(norm_back)
                cur = out
                if relu:
                    idx += 1
            elif c.kind == 'act':
                # an 'a' that follows no 'n' (resnet.py:143-145): ReLU as a pass of its own; the backward goes by the sign of the stored output
                y = self.act(pre + ':y', cur.N, cur.H, cur.W, cur.C)
                n_el = cur.M * cur.C
                nd = dict(n_lo=n_el & 0x7fffffff, n_hi=n_el >> 31)
                self.fwd.append(Op(ir.OP_RELU_FWD, buf=dict(x=cur.s, y=y.s), dim=nd, note=pre))

                def act_back(dy: T, ops, y=y, pre=pre, nd=nd):
                    dx = self.act(pre + ':dx', y.N, y.H, y.W, y.C)
                    ops.append(Op(ir.OP_RELU_BWD, buf=dict(dy=dy.s, y=y.s, dx=dx.s), dim=nd, note=pre))
                    return dx
                self._back.append(act_back)
                cur = y
            elif c.kind == 'maxpool':
                k, s, pd = c.args
                P, Q = (cur.H + 2 * pd - k) // s + 1, (cur.W + 2 * pd - k) // s + 1
                y = self.act(pre + ':y', cur.N, P, Q, cur.C)
                d = dict(N=cur.N, H=cur.H, W=cur.W, C=cur.C, k=k, stride=s, pad=pd)
                am = self.slot(pre + ':argmax', 'u8', (cur.N, P, Q, cur.C), 'u8') if self.need_grad else -1
                self.fwd.append(Op(ir.OP_MAXPOOL_FWD, buf=dict(x=cur.s, y=y.s, argmax=am), dim=d, note=pre))

                def mp_back(dy: T, ops, x=cur, d=d, pre=pre, am=am):
                    dx = self.act(pre + ':dx', x.N, x.H, x.W, x.C)
                    ops.append(Op(ir.OP_MAXPOOL_BWD, buf=dict(dy=dy.s, argmax=am, dx=dx.s), dim=d, note=pre))
                    return dx
                self._back.append(mp_back)
                cur = y
            elif c.kind == 'avgpool':
                k, s, pd = c.args
                nxt = comps[idx + 1] if idx + 1 < len(comps) else None
                if not (nxt is not None and nxt.kind == 'fc' and pd == 0 and k == cur.H == cur.W):
                    # any other AvgPool2d(k, s, p) (resnet.py:77-81): a pooling pass of its own (the global pool in front of 'f' is fused with the classifier below)
                    if 2 * pd > k or cur.H + 2 * pd < k or cur.W + 2 * pd < k:
                        raise ValueError(f"{pre}: AvgPool2d({k}, {s}, {pd}) does not fit a {cur.H} x {cur.W} map")       # torch raises on both
                    P, Q = (cur.H + 2 * pd - k) // s + 1, (cur.W + 2 * pd - k) // s + 1
                    y = self.act(pre + ':y', cur.N, P, Q, cur.C)
                    d = dict(N=cur.N, H=cur.H, W=cur.W, C=cur.C, k=k, stride=s, pad=pd)
                    self.fwd.append(Op(ir.OP_AVGPOOL_FWD, buf=dict(x=cur.s, y=y.s), dim=d, note=pre))

                    def ap_back(dy: T, ops, x=cur, d=d, pre=pre):
                        dx = self.act(pre + ':dx', x.N, x.H, x.W, x.C)
                        ops.append(Op(ir.OP_AVGPOOL_BWD, buf=dict(dy=dy.s, dx=dx.s), dim=d, note=pre))
                        return dx
                    self._back.append(ap_back)
                    cur = y
                    idx += 1
                    continue
                fpre = f'_architecture.{idx + 1}'
                O = nxt.cout
                if nxt.cin != cur.C:
                    raise ValueError(f"fc input width {nxt.cin} does not match {cur.C} pooled channels")
                w, b = self.param(fpre + '.1.weight', (O, cur.C)), self.param(fpre + '.1.bias', (O,))
                feat = self.f32('feat', (cur.N, cur.C))
                logits = self.slot('logits', 'f32', (cur.N, O), 'f32')
                named['logits'] = logits
                d = dict(N=cur.N, HW=cur.H * cur.W, C=cur.C, O=O)
                self.fwd.append(Op(ir.OP_POOL_FC_FWD, buf=dict(x=cur.s, w=w, b=b, feat=feat, logits=logits), dim=d, note=fpre))

                def fc_back(_, ops, x=cur, d=d, feat=feat, w=w, fpre=fpre):
                    dl = named['dlogits']
                    dw, db = self.grad(fpre + '.1.weight', (d['O'], d['C'])), self.grad(fpre + '.1.bias', (d['O'],))
                    dx = self.act(fpre + ':dx', x.N, x.H, x.W, x.C)
                    ops.append(Op(ir.OP_POOL_FC_BWD, buf=dict(dlogits=dl, feat=feat, w=w, dx=dx.s, dw=dw, db=db), dim=d, note=fpre))
                    self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
                    return dx
                self._back.append(fc_back)
                cur = None
                idx += 1
            elif c.kind == 'fc':
                # 'f' on a map that no global 'ap' has pooled (resnet.py:117-120: Flatten() of the NCHW map, then Linear): the classifier kernels with
                # one "pixel" of H*W*C features.  The reference orders the features (c, h, w), the engine's map is (h, w, c): the weight is read through
                # a permuted copy and its gradient is permuted back (nothing to do for a 1 x 1 map).
                if cur is None or idx + 1 != len(comps):
                    raise NotImplementedError("'f' must be the last component, on a feature map")
                HW, Cf, O = cur.H * cur.W, cur.H * cur.W * cur.C, c.cout
                if c.cin != Cf:
                    raise ValueError(f"fc input width {c.cin} does not match the {cur.C} x {cur.H} x {cur.W} map")
                w, b = self.param(pre + '.1.weight', (O, Cf)), self.param(pre + '.1.bias', (O,))
                wp = w
                if HW > 1:
                    wp = self.f32(pre + ':wperm', (O, Cf))
                    self.fwd.append(Op(ir.OP_PERMUTE_F32, buf={'in': w, 'out': wp}, dim=dict(A=O, B=cur.C, C=HW), note=pre))
                feat = self.f32('feat', (cur.N, Cf))
                logits = self.slot('logits', 'f32', (cur.N, O), 'f32')
                named['logits'] = logits
                d = dict(N=cur.N, HW=1, C=Cf, O=O)
                self.fwd.append(Op(ir.OP_POOL_FC_FWD, buf=dict(x=cur.s, w=wp, b=b, feat=feat, logits=logits), dim=d, note=pre))

                def flat_fc_back(_, ops, x=cur, d=d, feat=feat, wp=wp, pre=pre, HW=HW, O=O, Cf=Cf):
                    dl = named['dlogits']
                    dw, db = self.grad(pre + '.1.weight', (O, Cf)), self.grad(pre + '.1.bias', (O,))
                    dwp = self.f32(pre + ':dwperm', (O, Cf)) if HW > 1 else dw
                    dx = self.act(pre + ':dx', x.N, x.H, x.W, x.C)
                    ops.append(Op(ir.OP_POOL_FC_BWD, buf=dict(dlogits=dl, feat=feat, w=wp, dx=dx.s, dw=dwp, db=db), dim=d, note=pre))
                    if HW > 1:
                        ops.append(Op(ir.OP_PERMUTE_F32, buf={'in': dwp, 'out': dw}, dim=dict(A=O, B=HW, C=x.C), note=pre))
                    self.bwd_hooks.append(Hook(len(ops), 'grad_ready', arg=len(self.grad_order) - 1))
                    return dx
                self._back.append(flat_fc_back)
                cur = None
            else:
                for b_ in range(c.depth):
                    cin = c.cin if b_ == 0 else c.cout
                    cur = self.block(f'{pre}.{b_}', c.kind, cin, c.down and b_ == 0, cur)
            idx += 1
        if 'logits' not in named:
            raise NotImplementedError("the network must end in a classifier ('fI,O')")
        n_logits = self.slots[named['logits']].shape
        if self.with_loss:
            named['labels'] = self.slot('labels', 'labels', (n_logits[0],), 'i64')
            named['loss3'] = self.f32('loss3', (4,))
        ops_b: List[Op] = []
        if self.need_grad:
            named['dlogits'] = self.slot('dlogits', 'f32', n_logits, 'f32')
            if self.with_loss:
                self.fwd.append(Op(ir.OP_SOFTMAX_CE, buf=dict(logits=named['logits'], labels=named['labels'], out3=named['loss3'],
                                                              dlogits=named['dlogits']),
                                   dim=dict(N=n_logits[0], O=n_logits[1]), fp=dict(scale=1.0 / n_logits[0])))
            g = None
            for back in reversed(self._back):
                g = back(g, ops_b)
        elif self.with_loss:
            self.fwd.append(Op(ir.OP_SOFTMAX_CE, buf=dict(logits=named['logits'], labels=named['labels'], out3=named['loss3'], dlogits=-1),
                               dim=dict(N=n_logits[0], O=n_logits[1]), fp=dict(scale=1.0 / n_logits[0])))
        npk = len(self.fwd_packs)
        self.fwd = self.fwd_packs + self.fwd
        n_fwd = len(self.fwd)
        hooks = [Hook(h.at + npk, h.action, h.slot, h.arg) for h in self.fwd_hooks] + [Hook(h.at + n_fwd, h.action, h.slot, h.arg) for h in self.bwd_hooks]
        if 'ws' in self._named:
            named['ws'] = self._named['ws']
        return Plan(self.slots, self.fwd + ops_b, n_fwd, hooks, self.grad_order, self.param_keys, named, self.train,
                    meta=dict(ws_need=self._ws_need, N=self.N, H=self.H, W=self.W, fp32=self.fp32))


def lower(spec, preact, use_proj, dropout_prob, N, H, W, **kw) -> Plan:
    return Lowering(spec, preact, use_proj, dropout_prob, N, H, W, **kw).lower()
