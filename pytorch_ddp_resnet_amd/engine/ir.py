"""
Plan IR shared by the lowering (lowering.py), the ctypes binding (executor.py) and the HIP executor
(csrc/plan.cpp).  Constants mirror include/rn_hip.h one to one; tests/test_abi.py checks they agree with the header.

An Op is one fused kernel launch (a few expand to 2-4 launches inside the library).  ``buf`` holds indices into the
plan's slot table, ``dim``/``fp`` are per-kind scalars, documented in OP_FIELDS below.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

RN_F32, RN_BF16, RN_F16 = 0, 1, 2

RES_NONE, RES_SAME, RES_DOWN2PAD, RES_UP2 = 0, 1, 2, 3

(OP_STEM_FWD, OP_PACK_W, OP_CONV_FWD, OP_BN_STATS, OP_BN_FINALIZE, OP_BN_APPLY, OP_DROPOUT_FWD, OP_MAXPOOL_FWD,
 OP_POOL_FC_FWD, OP_POOL_FC_BWD, OP_MAXPOOL_BWD, OP_BN_BWD_REDUCE, OP_BN_BWD_FINALIZE, OP_BN_BWD_APPLY,
 OP_CONV_DGRAD, OP_CONV_WGRAD, OP_STEM_WGRAD, OP_DROPOUT_BWD, OP_SOFTMAX_CE, OP_ZERO, OP_ADD_RES, OP_IMG_TO_NHWC,
 OP_PACK_STEM_W, OP_UNPACK_STEM_DW, OP_BN_POOL_FWD, OP_BN_POOL_BWD_REDUCE, OP_BN_POOL_BWD_APPLY, OP_IMG_TO_S2D, OP_PACK_STEM_W_S2D,
 OP_UNPACK_STEM_DW_S2D, OP_RELU_FWD, OP_RELU_BWD, OP_AVGPOOL_FWD, OP_AVGPOOL_BWD, OP_PERMUTE_F32) = range(1, 36)

OP_NAMES = {v: k for k, v in list(globals().items()) if k.startswith('OP_') and isinstance(v, int)}

F_RELU, F_TRAIN, F_ACCUM, F_WRITE_G, F_NEED_DGRAD_PACK, F_SKIP_FWD_PACK, F_NO_DX, F_MASK_RECOMPUTE, F_FORK, F_DEFER_REDUCE = (1 << i for i in range(10))

OP_NBUF, OP_NDIM = 8, 20
CONV_STATS_ROWS = 128      # RN_CONV_STATS_ROWS: output pixels per partial-sum row of a fused conv epilogue

# buf[] / dim[] / fp[] meaning per kind (geom = N,H,W,C,P,Q,K,R,S,stride,pad in dim[0:11])
OP_FIELDS = {
    OP_STEM_FWD:        ('x w bias y', 'geom', ''),
    OP_PACK_W:          ('w w_fwd w_dgrad', 'K RS C', ''),
    OP_CONV_FWD:        ('x w_fwd y res stats bias', 'geom res_mode res_C alg_macs', ''),      # alg_macs: MACs per output of the reference's layer when the geometry is a regrouping of it (s2d stem; bench.py), else 0
    OP_BN_STATS:        ('x partial', 'M C nblk', ''),
    OP_BN_FINALIZE:     ('partial gamma beta running_mean running_var nbt coef fold', 'nblk count C', 'eps momentum'),
    OP_BN_APPLY:        ('x coef res out', 'N H W C res_mode res_C', 'p'),
    OP_DROPOUT_FWD:     ('x out', 'n_lo n_hi', 'p'),
    OP_MAXPOOL_FWD:     ('x y argmax', 'N H W C k stride pad', ''),
    OP_POOL_FC_FWD:     ('x w b feat logits', 'N HW C O', ''),
    OP_POOL_FC_BWD:     ('dlogits feat w dx dw db', 'N HW C O', ''),
    OP_MAXPOOL_BWD:     ('dy argmax dx', 'N H W C k stride pad', ''),
    OP_BN_BWD_REDUCE:   ('dout x mask coef partial', 'M C nblk', 'gscale p'),
    OP_BN_BWD_FINALIZE: ('partial dsum dgamma dbeta fold', 'nblk C', ''),
    OP_BN_BWD_APPLY:    ('dout x mask coef dsum add dx g_out', 'N H W C add_mode add_C count', 'gscale p'),
    OP_CONV_DGRAD:      ('dy w_dgrad dx res bn_x bn_mask bn_coef bn_partial', 'geom res_mode res_C', 'gscale'),
    OP_CONV_WGRAD:      ('x dy dw ws', 'geom alg_macs', ''),
    OP_STEM_WGRAD:      ('x dy dw db ws', 'geom', ''),
    OP_DROPOUT_BWD:     ('dout out din', 'n_lo n_hi', 'p'),
    OP_SOFTMAX_CE:      ('logits labels out3 dlogits gscale', 'N O', 'scale'),
    OP_ZERO:            ('dst', 'bytes_lo bytes_hi', ''),
    OP_ADD_RES:         ('dst res', 'N H W C res_mode res_C', ''),
    OP_IMG_TO_NHWC:     ('x out', 'N C H W CP', ''),
    OP_PACK_STEM_W:     ('w w_padded', 'K RS C CP', ''),
    OP_UNPACK_STEM_DW:  ('dw_padded dw', 'K RS C CP', ''),
    OP_BN_POOL_FWD:     ('x coef y argmax xsel', 'N H W C k stride pad', ''),
    OP_BN_POOL_BWD_REDUCE: ('dy argmax x coef partial xsel', 'N H W C k stride pad nblk npix', ''),
    OP_BN_POOL_BWD_APPLY:  ('dy argmax x coef dsum dx sums', 'N H W C k stride pad count rows', ''),
    OP_IMG_TO_S2D:         ('x out', 'N C H W', ''),
    OP_PACK_STEM_W_S2D:    ('w w_s2d', 'K C', ''),
    OP_UNPACK_STEM_DW_S2D: ('dw_s2d dw', 'K C', ''),
    OP_RELU_FWD:           ('x y', 'n_lo n_hi', ''),
    OP_RELU_BWD:           ('dy y dx', 'n_lo n_hi', ''),
    OP_AVGPOOL_FWD:        ('x y', 'N H W C k stride pad', ''),
    OP_AVGPOOL_BWD:        ('dy dx', 'N H W C k stride pad', ''),
    OP_PERMUTE_F32:        ('in out', 'A B C', ''),
}

# the buf[] fields an op WRITES (everything else it only reads): rn_op_output_mask in csrc/plan.cpp is the same table in bits (tests/test_abi.py)
OP_OUTPUTS = {
    OP_STEM_FWD: 'y', OP_PACK_W: 'w_fwd w_dgrad', OP_CONV_FWD: 'y stats', OP_BN_STATS: 'partial',
    OP_BN_FINALIZE: 'running_mean running_var nbt coef fold', OP_BN_APPLY: 'out', OP_DROPOUT_FWD: 'out', OP_MAXPOOL_FWD: 'y argmax',
    OP_POOL_FC_FWD: 'feat logits', OP_POOL_FC_BWD: 'dx dw db', OP_MAXPOOL_BWD: 'dx', OP_BN_BWD_REDUCE: 'partial',
    OP_BN_BWD_FINALIZE: 'dsum dgamma dbeta fold', OP_BN_BWD_APPLY: 'dx g_out', OP_CONV_DGRAD: 'dx bn_partial', OP_CONV_WGRAD: 'dw ws',
    OP_STEM_WGRAD: 'dw db ws', OP_DROPOUT_BWD: 'din', OP_SOFTMAX_CE: 'out3 dlogits', OP_ZERO: 'dst', OP_ADD_RES: 'dst', OP_IMG_TO_NHWC: 'out',
    OP_PACK_STEM_W: 'w_padded', OP_UNPACK_STEM_DW: 'dw', OP_BN_POOL_FWD: 'y argmax xsel', OP_BN_POOL_BWD_REDUCE: 'partial',
    OP_BN_POOL_BWD_APPLY: 'dx sums', OP_IMG_TO_S2D: 'out', OP_PACK_STEM_W_S2D: 'w_s2d', OP_UNPACK_STEM_DW_S2D: 'dw', OP_RELU_FWD: 'y',
    OP_RELU_BWD: 'dx', OP_AVGPOOL_FWD: 'y', OP_AVGPOOL_BWD: 'dx', OP_PERMUTE_F32: 'out',
}

GEOM = 'N H W C P Q K R S stride pad'.split()


@dataclass
class Slot:
    """one device buffer of a plan.  role: 'act' (compute dtype NHWC), 'f32', 'fold' (fp32 words of a split finalize's hand-off buffer; its byte size is told to the plan), 'i64', 'param', 'buffer', 'grad',
    'input', 'labels', 'ws'.  key: reference state_dict key for param/buffer/grad slots."""
    name: str
    role: str
    shape: Tuple[int, ...]
    dtype: str            # 'T' (plan compute dtype) | 'f32' | 'i64' | 'u8'
    key: Optional[str] = None

    @property
    def numel(self):
        n = 1
        for d in self.shape:
            n *= int(d)
        return n


@dataclass
class Op:
    kind: int
    buf: dict = field(default_factory=dict)      # field name -> slot index
    dim: dict = field(default_factory=dict)      # field name -> int
    fp: dict = field(default_factory=dict)       # field name -> float
    flags: int = 0
    seed: int = 0
    note: str = ''

    def packed(self):
        """-> (buf[8], dim[20], fp[4]) in header order."""
        bnames, dnames, fnames = (OP_FIELDS[self.kind][i].split() for i in range(3))
        if dnames and dnames[0] == 'geom':
            dnames = GEOM + dnames[1:]
        buf = [self.buf.get(n, -1) for n in bnames] + [-1] * (OP_NBUF - len(bnames))
        dim = [int(self.dim.get(n, 0)) for n in dnames] + [0] * (OP_NDIM - len(dnames))
        fp = [float(self.fp.get(n, 0.0)) for n in fnames] + [0.0] * (4 - len(fnames))
        for n in self.buf:
            assert n in bnames, (OP_NAMES[self.kind], n)
        for n in self.dim:
            assert n in dnames, (OP_NAMES[self.kind], n)
        return buf, dim, fp


@dataclass
class Hook:
    """a host action between ops (the plan is run as ranges around hooks): cross-rank reductions for SyncBN
    ('allreduce_f32' on a slot) and gradient-bucket boundaries ('bucket_ready')."""
    at: int               # runs before op index `at`
    action: str
    slot: int = -1
    arg: int = 0
