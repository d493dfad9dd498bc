"""CPU proof that the GPU parity tests reach every convolution kernel instantiation the BASELINE configurations launch.

The library answers "which kernel would this geometry select" without a GPU (rn_conv_kernel_names runs the launchers'
own selection code in a dry run).  For every convolution of the five full-batch configurations (forward, data gradient,
weight gradient; bf16/fp16 and fp32 engines) the selected names must also be selected by a geometry of
tests/prod_geoms.py or of tests/test_gpu_kernels.py -- i.e. no production tile runs only in bench.py."""
import os

import pytest

from pytorch_ddp_resnet_amd import _lib
from pytorch_ddp_resnet_amd.engine import ir
from pytorch_ddp_resnet_amd.engine.lowering import lower
from prod_geoms import PROD_GEOMS, IGEMM8_GEOMS, R8_GEOMS, STEM8_GEOMS, CONFIGS, geom, resolve

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason='librn_hip.so not built')

PASS_OF = {ir.OP_CONV_FWD: 0, ir.OP_CONV_DGRAD: 1, ir.OP_CONV_WGRAD: 2}


def _small_geoms():
    import importlib
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'test_gpu_kernels.py')).read()
    ns = {}
    start = src.index('CONV_GEOMS = [')
    exec(src[start:src.index(']\n', start) + 2], ns)
    return ns['CONV_GEOMS']


def _tested_names(dtype):
    names = set()
    for g in list(PROD_GEOMS) + list(_small_geoms()):
        ce = 4 if dtype == ir.RN_F32 else 8
        g = resolve(g, dtype == ir.RN_F32)
        if g[3] % ce or g[4] % ce:
            continue
        # operand sets of tests/test_gpu_production_tiles.py::run_conv_case: forward = statistics + identity residual, data gradient = BatchNorm-backward sums
        for p, fl in ((0, 3), (1, 1), (2, 0)):
            names.update(_lib.conv_kernel_names(p, dtype, geom(*g), fused_epilogue=fl))
    for g in STEM8_GEOMS:                      # test_igemm8_stem: the stem convolution with its bias and statistics
        if dtype != ir.RN_F32:
            names.update(_lib.conv_kernel_names(0, dtype, geom(*resolve(g, False)), fused_epilogue=9))
            gs = geom(g[0], g[1] // 2 + 3, g[2] // 2 + 3, 16, g[4], 4, 1, 0)          # test_s2d_stem: the same layers in the space-to-depth form the 16-bit engines lower to
            names.update(_lib.conv_kernel_names(0, dtype, gs, fused_epilogue=9))
            names.update(_lib.conv_kernel_names(2, dtype, gs))
    for g in IGEMM8_GEOMS:                     # test_igemm8_production_operand_sets: every operand set
        if dtype == ir.RN_F32:
            break
        for p, fls in ((0, (1, 3)), (1, (1, 3, 5, 17)), (2, (0,))):            # 17: sums with the mask computed from x (test_igemm8_mask_from_x)
            for fl in fls:
                names.update(_lib.conv_kernel_names(p, dtype, geom(*g), fused_epilogue=fl))
    for g in R8_GEOMS:                         # test_igemm8r_production_operand_sets: every operand set of the row-patch kernel
        if dtype == ir.RN_F32:
            break
        for p, fls in ((0, (1, 3)), (1, (1, 3, 5, 17)), (2, (0,))):
            for fl in fls:
                names.update(_lib.conv_kernel_names(p, dtype, geom(*g), fused_epilogue=fl))
    return names


def _op_flags(op):
    """the operand set of a convolution op as rn_conv_kernel_names takes it: 1 fused sums, 2 identity residual, 4 accumulate, 8 bias, 16 mask from x"""
    b = op.buf
    fl = 1 if (b.get('stats', -1) >= 0 or b.get('bn_x', -1) >= 0) else 0
    if b.get('res', -1) >= 0 and op.dim.get('res_mode', 0) == ir.RES_SAME:
        fl |= 2
    if op.kind == ir.OP_CONV_DGRAD and (op.flags & ir.F_ACCUM):
        fl |= 4
    if b.get('bias', -1) >= 0:
        fl |= 8
    if op.kind == ir.OP_CONV_DGRAD and (op.flags & ir.F_MASK_RECOMPUTE):
        fl |= 16
    return fl


@pytest.mark.parametrize('dtype', [ir.RN_F32, ir.RN_BF16])
@pytest.mark.parametrize('name', list(CONFIGS))
def test_every_production_tile_is_parity_tested(name, dtype):
    cfg = CONFIGS[name]
    plan = lower(cfg['spec'], cfg['preact'], cfg['use_proj'], 0.0, cfg['batch'], cfg['hw'], cfg['hw'], train=True,
                 fp32=dtype == ir.RN_F32)
    tested = _tested_names(dtype)
    missing = {}
    n_conv = 0
    for op in plan.ops:
        if op.kind not in PASS_OF:
            continue
        n_conv += 1
        g = {k: op.dim[k] for k in 'N H W C P Q K R S stride pad'.split()}
        for nm in _lib.conv_kernel_names(PASS_OF[op.kind], dtype, g, fused_epilogue=_op_flags(op)):
            if nm not in tested:
                missing.setdefault(nm, []).append((ir.OP_NAMES[op.kind], tuple(g.values())))
    assert n_conv > 10
    assert not missing, f'{name}: kernels launched by the configuration but by no parity-test geometry: ' + \
        '; '.join(f'{k} e.g. {v[0]}' for k, v in missing.items())


@pytest.mark.parametrize('fp32', [False, True])
@pytest.mark.parametrize('name', list(CONFIGS))
def test_every_op_of_every_configuration_packs(name, fp32):
    """Op.packed() (the C-ABI encoding the executor hands to rn_plan_create) accepts every op the lowering emits -- a buffer or dimension name the op table does
    not declare is an error that would otherwise first show on the GPU box"""
    cfg = CONFIGS[name]
    plan = lower(cfg['spec'], cfg['preact'], cfg['use_proj'], 0.0, 4, cfg['hw'], cfg['hw'], train=True, fp32=fp32)
    for op in plan.ops:
        buf, dim, fp = op.packed()
        assert len(buf) == ir.OP_NBUF and len(dim) == ir.OP_NDIM and len(fp) == 4


def test_dry_run_launches_nothing_and_restores_the_log():
    L = _lib.lib()
    L.rn_kernel_log(1)
    names = _lib.conv_kernel_names(0, ir.RN_BF16, geom(128, 32, 32, 160, 160, 3, 1, 1))
    assert names == ['igemm_dma<128x160>'] or len(names) == 1
    assert L.rn_kernel_log_read() == b''          # a query does not leak into a running log
    L.rn_kernel_log(0)
