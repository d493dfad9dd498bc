"""ctypes binding of librn_hip.so (include/rn_hip.h).  Fails loudly when the library is missing: there is no
CPU or eager-PyTorch fallback for the hot path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'librn_hip.so')

OP_NBUF, OP_NDIM = 8, 20
ABI_VERSION = 11         # include/rn_hip.h RN_ABI_VERSION this binding was written for


class RnOp(C.Structure):
    _fields_ = [('kind', C.c_int32), ('flags', C.c_int32), ('buf', C.c_int32 * OP_NBUF), ('dim', C.c_int32 * OP_NDIM),
                ('fp', C.c_float * 4), ('seed', C.c_uint32), ('pad_', C.c_int32)]


class RnConvGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in 'N H W C P Q K R S stride pad'.split()]


class RnError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RnError(f"{LIB_PATH} not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()' "
                      f"or make -C pytorch_ddp_resnet_amd/csrc). There is no fallback path.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double, C.c_size_t
    L.rn_last_error.restype = C.c_char_p
    L.rn_version.restype = i32
    if L.rn_version() != ABI_VERSION:
        raise RnError(f"{LIB_PATH} reports ABI version {L.rn_version()}, this binding is written for {ABI_VERSION}: rebuild the library "
                      f"(make -C pytorch_ddp_resnet_amd/csrc); a stale library would take shifted arguments")
    L.rn_plan_create.argtypes = [C.POINTER(RnOp), i32, i32, i32, C.POINTER(vp)]
    L.rn_plan_bind.argtypes = [vp, C.POINTER(vp), i32]
    L.rn_plan_set_bytes.argtypes = [vp, i32, sz]
    L.rn_plan_run.argtypes = [vp, i32, i32, u64, vp]
    L.rn_plan_num_ops.argtypes = [vp]
    L.rn_plan_profile.argtypes = [vp, i32]
    L.rn_plan_profile_read.argtypes = [vp, C.POINTER(C.c_float), i32]
    L.rn_plan_destroy.argtypes = [vp]
    L.rn_plan_set_overlap.argtypes = [vp, i32]
    L.rn_plan_join.argtypes = [vp, vp]
    L.rn_plan_side_wait.argtypes = [vp, vp]
    L.rn_plan_defer_reduce.argtypes = [vp]
    L.rn_plan_defer_reduce.restype = C.c_size_t
    L.rn_plan_set_reduce_arena.argtypes = [vp, vp, C.c_size_t]
    L.rn_plan_destroy.restype = None
    L.rn_conv_wgrad_ws_bytes.argtypes = [C.POINTER(RnConvGeom)]
    L.rn_conv_wgrad_ws_bytes.restype = sz
    L.rn_stem_wgrad_ws_bytes.argtypes = [C.POINTER(RnConvGeom)]
    L.rn_stem_wgrad_ws_bytes.restype = sz
    L.rn_softmax_ce.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp, vp]
    L.rn_sgd_step.argtypes = [vp, vp, vp, i64, f32, f32, f32, f32, i32, i32, f32, vp]
    L.rn_sgd_step_amp.argtypes = [vp, vp, vp, i64, f32, f32, f32, f32, i32, i32, vp, vp, vp]
    L.rn_set_variant.argtypes = [i32]
    L.rn_set_variant2.argtypes = [i32]
    L.rn_conv_workspace_bytes.restype = sz
    L.rn_set_conv_workspace.argtypes = [vp, sz]
    L.rn_augment_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.rn_amp_check_unscale.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.rn_kernel_log.argtypes = [i32]
    L.rn_kernel_log.restype = None
    L.rn_kernel_log_read.restype = C.c_char_p
    L.rn_conv_kernel_names.argtypes = [i32, i32, C.POINTER(RnConvGeom), i32, C.c_char_p, sz]
    if os.environ.get('RN_VARIANT'):          # kernel-variant switch for A/B runs (tools/conv_bench.py); unset = shipped configuration
        L.rn_set_variant(int(os.environ['RN_VARIANT']))
    if os.environ.get('RN_VARIANT2'):
        L.rn_set_variant2(int(os.environ['RN_VARIANT2']))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RnError(lib().rn_last_error().decode())


def conv_kernel_names(pass_, dtype, g, fused_epilogue=False):
    """names of the kernels a convolution geometry selects (pass_: 0 forward, 1 dgrad, 2 wgrad), without launching."""
    buf = C.create_string_buffer(512)
    gs = geom_struct(g)
    check(lib().rn_conv_kernel_names(pass_, dtype, C.byref(gs), int(fused_epilogue), buf, 512))
    return buf.value.decode().split(',') if buf.value else []


def geom_struct(d):
    return RnConvGeom(*[int(d[n]) for n in 'N H W C P Q K R S stride pad'.split()])
