// Plan executor: runs the lowered op list of one ResNet forward / backward as a sequence of HIP launches on the
// caller's stream, one host call per range.  Replaces the Python-side module dispatch of ResNet.forward
// (/root/reference/resnet/architectures/resnet.py:165-166) and the autograd engine's backward walk
// (/root/reference/resnet/algos/training.py:100,102).  No allocation, no synchronisation, no host<->device copies
// happen here (hipGraph-capturable); the caller binds every buffer.
#include <stdarg.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <string.h>
#include <cstdlib>
#include <vector>

#include "common.h"

extern "C" int rn_add_res(void* dst, const void* res, int dtype, int N, int H, int W, int C, int res_mode, int res_C, rn_stream s);

static thread_local char g_err[512] = "";

void rn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* rn_last_error(void) { return g_err; }

// ---- kernel log: names of the convolution kernels picked since rn_kernel_log(1) (tests assert which tile ran) ----
static thread_local bool g_log_on = false, g_dry = false;
static thread_local std::string g_log;
void rn_note_kernel(const char* fmt, ...) {
  if (!g_log_on) return;
  char b[96];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(b, sizeof(b), fmt, ap);
  va_end(ap);
  if (!g_log.empty()) g_log += ',';
  g_log += b;
}
bool rn_dry_run() { return g_dry; }
extern "C" void rn_kernel_log(int enable) { g_log_on = enable != 0; g_log.clear(); }
extern "C" const char* rn_kernel_log_read(void) { return g_log.c_str(); }
extern "C" int rn_conv_kernel_names(int pass, int dtype, const rn_conv_geom* g, int fused_epilogue, char* out, size_t n) {
  RN_CHECK_ARG(g && out && n > 0 && pass >= 0 && pass <= 2, "rn_conv_kernel_names: bad argument");
  const bool was_on = g_log_on;
  const std::string keep = g_log;
  g_log_on = true; g_dry = true; g_log.clear();
  static float dummy[8];
  void* p = dummy;                              // never dereferenced: the launchers return before launching
  // fused_epilogue is a set of flags naming the launch's operand set (the eight-phase kernels are specialised per set): 1 = fused BatchNorm sums
  // (forward: statistics of the output; data gradient: the backward sums over x and the mask), 2 = identity residual, 4 = accumulate into dx, 8 = per-channel bias (forward: the stem), 16 = mask_from_x
  rn_conv_epilogue ep{(float*)p, nullptr, nullptr, nullptr, 1.f, nullptr};
  const bool fused = (fused_epilogue & 1) != 0, res = (fused_epilogue & 2) != 0, acc = (fused_epilogue & 4) != 0, bias = (fused_epilogue & 8) != 0;
  ep.mask_from_x = (fused_epilogue & 16) ? 1 : 0;       // 16 = the data gradient's mask may be computed from bn_x (rn_conv_epilogue.mask_from_x)
  int e;
  if (pass == 0) {
    if (bias) ep.bias = (const float*)p;
    if (!fused) ep.partial = nullptr;
    e = rn_conv_fwd(p, p, p, res ? p : nullptr, res ? RN_RES_SAME : RN_RES_NONE, res ? g->K : 0, dtype, g, (fused || bias) ? &ep : nullptr, nullptr);
  }
  else if (pass == 1) {
    ep.bn_x = p; ep.bn_mask = p; ep.bn_coef = (const float*)p;
    e = rn_conv_dgrad(p, p, p, res ? p : nullptr, res ? RN_RES_SAME : RN_RES_NONE, res ? g->C : 0, acc ? RN_F_ACCUM : 0, dtype, g, fused ? &ep : nullptr, nullptr);
  } else e = rn_conv_wgrad(p, p, (float*)p, p, (size_t)-1, 0, dtype, g, nullptr);
  snprintf(out, n, "%s", g_log.c_str());
  g_dry = false; g_log_on = was_on; g_log = keep;
  return e;
}
extern "C" int rn_version(void) { return RN_ABI_VERSION; }

struct rn_plan {
  std::vector<rn_op> ops;
  std::vector<void*> bufs;
  std::vector<size_t> ws_bytes;   // bytes available behind a 'ws' slot (set through rn_plan_set_bytes)
  int dtype;
  bool profile = false;            // per-op hipEvent pairs on the launch stream (bench.py roofline leg)
  std::vector<hipEvent_t> ev;      // 2 per op
  std::vector<char> ev_set;
  // Weight-gradient ops (rn_plan_set_overlap) run on a second, low-priority stream: each is forked off the launch stream by
  // an event at its position in the op list and needs only its layer's x and dy, which nothing overwrites before the join;
  // their results are consumed after rn_plan_join.  They overlap the data-gradient / BatchNorm chain, which is the critical
  // path of the backward and leaves matrix-pipe and HBM idle time in every kernel's prologue, epilogue and tail round.
  bool overlap = false, side_pending = false;
  // Deferred weight-gradient slab sums (rn_plan_defer_reduce): marked CONV_WGRAD ops leave their slabs in a region of their own; the
  // sums of a whole run of ops go out as batched launches (rn_wgrad_reduce_batch) at the end of the range / every 32 layers
  std::vector<long> slab_off;      // per op: byte offset of its slab region in the arena, -1 = reduces immediately
  std::vector<int> slab_splits;
  size_t arena_bytes = 0;
  char* arena = nullptr;           // deferral is on iff it is set
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_peek = nullptr;
  // The slab sums of the forked weight gradients run on a THIRD stream (light HBM-bound launches that share CUs with the next weight gradient and with the data
  // gradients); the side workspace is used in two halves alternately, a launch into a half waits for the sums that last read it (ev_red), and side2 is folded
  // back into the side stream before anything waits for the side stream (fold_side2).
  hipStream_t side2 = nullptr;
  hipEvent_t ev_wg = nullptr, ev_red[2] = {nullptr, nullptr}, ev_s2 = nullptr;
  bool red_rec[2] = {false, false}, side2_pending = false;
  int w_parity = 0;
  ~rn_plan() {
    if (ev_wg) (void)hipEventDestroy(ev_wg);
    if (ev_red[0]) (void)hipEventDestroy(ev_red[0]);
    if (ev_red[1]) (void)hipEventDestroy(ev_red[1]);
    if (ev_s2) (void)hipEventDestroy(ev_s2);
    if (side2) (void)hipStreamDestroy(side2);
    for (auto e : ev) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_peek) (void)hipEventDestroy(ev_peek);
    if (side) (void)hipStreamDestroy(side);
  }
};

extern "C" int rn_plan_create(const rn_op* ops, int n_ops, int n_bufs, int dtype, rn_plan** out) {
  RN_CHECK_ARG(ops && out && n_ops > 0 && n_bufs > 0, "rn_plan_create: bad argument");
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "rn_plan_create: bad dtype %d", dtype);
  for (int i = 0; i < n_ops; ++i) {
    RN_CHECK_ARG(ops[i].kind >= RN_OP_STEM_FWD && ops[i].kind <= RN_OP_PERMUTE_F32, "rn_plan_create: op %d has unknown kind %d", i, ops[i].kind);
    for (int j = 0; j < RN_OP_NBUF; ++j)
      RN_CHECK_ARG(ops[i].buf[j] >= -1 && ops[i].buf[j] < n_bufs, "rn_plan_create: op %d buffer index %d out of range", i, ops[i].buf[j]);
  }
  rn_plan* p = new rn_plan();
  p->ops.assign(ops, ops + n_ops);
  p->bufs.assign(n_bufs, nullptr);
  p->ws_bytes.assign(n_bufs, 0);
  p->dtype = dtype;
  *out = p;
  return 0;
}

extern "C" size_t rn_plan_defer_reduce(rn_plan* plan) {
  if (!plan) return 0;
  const int n = (int)plan->ops.size();
  plan->slab_off.assign(n, -1); plan->slab_splits.assign(n, 0);
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    const rn_op& o = plan->ops[i];
    if (o.kind != RN_OP_CONV_WGRAD || (o.flags & RN_F_FORK)) continue;       // forked ops own a stream and a workspace: they reduce there
    rn_conv_geom g;
    g.N = o.dim[0]; g.H = o.dim[1]; g.W = o.dim[2]; g.C = o.dim[3]; g.P = o.dim[4]; g.Q = o.dim[5]; g.K = o.dim[6];
    g.R = o.dim[7]; g.S = o.dim[8]; g.stride = o.dim[9]; g.pad = o.dim[10];
    const int splits = rn_conv_wgrad_splits(&g, plan->dtype, o.flags);
    if (splits <= 0) continue;
    plan->slab_off[i] = (long)off; plan->slab_splits[i] = splits;
    off += ((size_t)splits * g.K * g.R * g.S * g.C * sizeof(float) + 255) / 256 * 256;
  }
  plan->arena_bytes = off;
  return off;
}
extern "C" int rn_plan_set_reduce_arena(rn_plan* plan, void* arena, size_t bytes) {
  RN_CHECK_ARG(plan != nullptr, "rn_plan_set_reduce_arena: null plan");
  RN_CHECK_ARG(!arena || (!plan->slab_off.empty() && bytes >= plan->arena_bytes), "rn_plan_set_reduce_arena: call rn_plan_defer_reduce first and pass at least the bytes it returned");
  plan->arena = plan->arena_bytes ? reinterpret_cast<char*>(arena) : nullptr;
  return 0;
}

extern "C" int rn_plan_bind(rn_plan* plan, const void* const* device_ptrs, int n_bufs) {
  RN_CHECK_ARG(plan && device_ptrs && n_bufs == (int)plan->bufs.size(), "rn_plan_bind: bad argument");
  for (int i = 0; i < n_bufs; ++i) plan->bufs[i] = const_cast<void*>(device_ptrs[i]);
  return 0;
}

extern "C" int rn_plan_set_bytes(rn_plan* plan, int slot, size_t bytes) {
  RN_CHECK_ARG(plan && slot >= 0 && slot < (int)plan->bufs.size(), "rn_plan_set_bytes: bad slot");
  plan->ws_bytes[slot] = bytes;
  return 0;
}

extern "C" int rn_plan_profile(rn_plan* plan, int enable) {
  RN_CHECK_ARG(plan != nullptr, "rn_plan_profile: null plan");
  if (enable && plan->ev.empty()) {
    plan->ev.resize(2 * plan->ops.size());
    for (auto& e : plan->ev)
      if (hipEventCreate(&e) != hipSuccess) { rn_set_error("rn_plan_profile: hipEventCreate failed"); return 2; }
    plan->ev_set.assign(plan->ops.size(), 0);
  }
  plan->profile = enable != 0;
  if (enable) plan->ev_set.assign(plan->ops.size(), 0);
  return 0;
}

// elapsed milliseconds of every op launched since profiling was (re-)enabled; 0 for ops that did not run. Blocks.
extern "C" int rn_plan_profile_read(rn_plan* plan, float* ms, int n) {
  RN_CHECK_ARG(plan && ms && n == (int)plan->ops.size() && !plan->ev.empty(), "rn_plan_profile_read: bad argument");
  for (int i = 0; i < n; ++i) {
    ms[i] = 0.f;
    if (!plan->ev_set[i]) continue;
    if (hipEventSynchronize(plan->ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms[i], plan->ev[2 * i], plan->ev[2 * i + 1]) != hipSuccess) {
      rn_set_error("rn_plan_profile_read: event query failed at op %d", i);
      return 2;
    }
  }
  return 0;
}

extern "C" int rn_plan_set_overlap(rn_plan* plan, int enable) {
  RN_CHECK_ARG(plan != nullptr, "rn_plan_set_overlap: null plan");
  if (enable && !plan->side) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&plan->side, hipStreamNonBlocking, least) != hipSuccess ||
        hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&plan->ev_peek, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithPriority(&plan->side2, hipStreamNonBlocking, least) != hipSuccess ||
        hipEventCreateWithFlags(&plan->ev_wg, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&plan->ev_red[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&plan->ev_red[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&plan->ev_s2, hipEventDisableTiming) != hipSuccess) {
      rn_set_error("rn_plan_set_overlap: could not create the side streams / events");
      return 2;
    }
  }
  plan->overlap = enable != 0;
  return 0;
}

// the side stream waits for the slab sums issued on side2 so far: whoever then waits for the side stream waits for them too
static int fold_side2(rn_plan* plan) {
  if (!plan->side2_pending) return 0;
  if (hipEventRecord(plan->ev_s2, plan->side2) != hipSuccess || hipStreamWaitEvent(plan->side, plan->ev_s2, 0) != hipSuccess) {
    rn_set_error("rn_plan: folding the slab-sum stream back failed");
    return 2;
  }
  plan->side2_pending = false;
  return 0;
}

// makes `stream` wait for every weight-gradient op forked so far (no-op when none is pending)
extern "C" int rn_plan_join(rn_plan* plan, rn_stream stream) {
  RN_CHECK_ARG(plan != nullptr, "rn_plan_join: null plan");
  if (!plan->side_pending) return 0;
  if (int e = fold_side2(plan)) return e;
  if (hipEventRecord(plan->ev_join, plan->side) != hipSuccess || hipStreamWaitEvent(as_stream(stream), plan->ev_join, 0) != hipSuccess) {
    rn_set_error("rn_plan_join: event record / wait failed");
    return 2;
  }
  plan->side_pending = false;
  return 0;
}

// makes `stream` (a communication stream) wait for the weight-gradient ops forked so far WITHOUT joining them into the launch stream:
// the launch stream keeps issuing the data-gradient chain; rn_plan_join still joins everything at the end of the range
extern "C" int rn_plan_side_wait(rn_plan* plan, rn_stream stream) {
  RN_CHECK_ARG(plan != nullptr, "rn_plan_side_wait: null plan");
  if (!plan->side_pending) return 0;
  if (int e = fold_side2(plan)) return e;
  if (hipEventRecord(plan->ev_peek, plan->side) != hipSuccess || hipStreamWaitEvent(as_stream(stream), plan->ev_peek, 0) != hipSuccess) {
    rn_set_error("rn_plan_side_wait: event record / wait failed");
    return 2;
  }
  return 0;
}

extern "C" int rn_plan_num_ops(const rn_plan* plan) { return plan ? (int)plan->ops.size() : 0; }
extern "C" void rn_plan_destroy(rn_plan* plan) { delete plan; }

// which buf[] entries of an op kind are WRITTEN (bit b = buf[b]; the field lists are in rn_hip.h / engine/ir.py OP_FIELDS).  The lowering gives every
// activation and gradient a slot of its own, so nothing rewrites a queued weight gradient's operands today; this table keeps the deferral correct for
// any plan a host builds
extern "C" unsigned rn_op_output_mask(int kind) {
  switch (kind) {
    case RN_OP_STEM_FWD: return 1u << 3;
    case RN_OP_PACK_W: return (1u << 1) | (1u << 2);
    case RN_OP_CONV_FWD: return (1u << 2) | (1u << 4);
    case RN_OP_BN_STATS: return 1u << 1;
    case RN_OP_BN_FINALIZE: return (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 7);
    case RN_OP_BN_APPLY: return 1u << 3;
    case RN_OP_DROPOUT_FWD: return 1u << 1;
    case RN_OP_MAXPOOL_FWD: return (1u << 1) | (1u << 2);
    case RN_OP_POOL_FC_FWD: return (1u << 3) | (1u << 4);
    case RN_OP_POOL_FC_BWD: return (1u << 3) | (1u << 4) | (1u << 5);
    case RN_OP_MAXPOOL_BWD: return 1u << 2;
    case RN_OP_BN_BWD_REDUCE: return 1u << 4;
    case RN_OP_BN_BWD_FINALIZE: return (1u << 1) | (1u << 2) | (1u << 3) | (1u << 4);
    case RN_OP_BN_BWD_APPLY: return (1u << 6) | (1u << 7);
    case RN_OP_CONV_DGRAD: return (1u << 2) | (1u << 7);
    case RN_OP_CONV_WGRAD: return (1u << 2) | (1u << 3);
    case RN_OP_STEM_WGRAD: return (1u << 2) | (1u << 3) | (1u << 4);
    case RN_OP_DROPOUT_BWD: return 1u << 2;
    case RN_OP_SOFTMAX_CE: return (1u << 2) | (1u << 3);
    case RN_OP_ZERO: case RN_OP_ADD_RES: return 1u << 0;
    case RN_OP_IMG_TO_NHWC: case RN_OP_PACK_STEM_W: case RN_OP_UNPACK_STEM_DW: case RN_OP_IMG_TO_S2D: case RN_OP_PACK_STEM_W_S2D:
    case RN_OP_UNPACK_STEM_DW_S2D: case RN_OP_RELU_FWD: case RN_OP_AVGPOOL_FWD: case RN_OP_AVGPOOL_BWD: case RN_OP_PERMUTE_F32: return 1u << 1;
    case RN_OP_BN_POOL_FWD: return (1u << 2) | (1u << 3) | (1u << 4);
    case RN_OP_BN_POOL_BWD_REDUCE: return 1u << 4;
    case RN_OP_BN_POOL_BWD_APPLY: return (1u << 5) | (1u << 6);
    case RN_OP_RELU_BWD: return 1u << 2;
    default: return ~0u;                                   // unknown kind: treat every buffer as written
  }
}

static inline rn_conv_geom geom_of(const rn_op& o) {
  rn_conv_geom g;
  g.N = o.dim[0]; g.H = o.dim[1]; g.W = o.dim[2]; g.C = o.dim[3]; g.P = o.dim[4]; g.Q = o.dim[5]; g.K = o.dim[6];
  g.R = o.dim[7]; g.S = o.dim[8]; g.stride = o.dim[9]; g.pad = o.dim[10];
  return g;
}

static int run_op(rn_plan* p, int idx, uint64_t step_seed, rn_stream s) {
  const rn_op& o = p->ops[idx];
  const int dt = p->dtype;
  auto B = [&](int i) -> void* { return o.buf[i] >= 0 ? p->bufs[o.buf[i]] : nullptr; };
  auto need = [&](int n) -> bool {
    for (int i = 0; i < n; ++i)
      if (o.buf[i] >= 0 && p->bufs[o.buf[i]] == nullptr) return false;
    return true;
  };
  if (!need(RN_OP_NBUF)) {
    rn_set_error("rn_plan_run: op %d (kind %d) uses an unbound buffer", idx, o.kind);
    return 1;
  }
  const int* d = o.dim;
  switch (o.kind) {
    case RN_OP_STEM_FWD: {
      rn_conv_geom g = geom_of(o);
      return rn_stem_conv_fwd((const float*)B(0), (const float*)B(1), (const float*)B(2), B(3), dt, &g, s);
    }
    case RN_OP_PACK_W:
      return rn_pack_weights((const float*)B(0), B(1), B(2), dt, d[0], d[1], d[2], s);
    case RN_OP_CONV_FWD: {
      rn_conv_geom g = geom_of(o);
      rn_conv_epilogue ep{(float*)B(4), nullptr, nullptr, nullptr, 1.f, (const float*)B(5)};
      return rn_conv_fwd(B(0), B(1), B(2), B(3), d[11], d[12], dt, &g, (o.buf[4] >= 0 || o.buf[5] >= 0) ? &ep : nullptr, s);
    }
    case RN_OP_BN_STATS:
      return rn_bn_stats(B(0), (float*)B(1), d[2], dt, d[0], d[1], s);
    case RN_OP_BN_FINALIZE:            /* partial gamma beta running_mean running_var nbt coef fold | nblk count C */
      if (B(7))
        return rn_bn_finalize_split((const float*)B(0), d[0], (double)d[1], (const float*)B(1), (const float*)B(2), (float*)B(3), (float*)B(4),
                                    (int64_t*)B(5), (float*)B(6), d[2], o.fp[0], o.fp[1], o.flags, B(7), p->ws_bytes[o.buf[7]], s);
      return rn_bn_finalize((const float*)B(0), d[0], (double)d[1], (const float*)B(1), (const float*)B(2), (float*)B(3), (float*)B(4),
                            (int64_t*)B(5), (float*)B(6), d[2], o.fp[0], o.fp[1], o.flags, s);
    case RN_OP_BN_APPLY:
      return rn_bn_apply(B(0), (const float*)B(1), B(2), B(3), dt, d[0], d[1], d[2], d[3], d[4], d[5], o.flags, o.fp[0], o.seed, step_seed, s);
    case RN_OP_DROPOUT_FWD:
      return rn_dropout_fwd(B(0), B(1), dt, ((int64_t)d[1] << 31) | (int64_t)d[0], o.fp[0], o.seed, step_seed, s);
    case RN_OP_MAXPOOL_FWD:
      return rn_maxpool_fwd(B(0), B(1), (unsigned char*)B(2), dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], s);
    case RN_OP_POOL_FC_FWD:
      return rn_pool_fc_fwd(B(0), (const float*)B(1), (const float*)B(2), (float*)B(3), (float*)B(4), dt, d[0], d[1], d[2], d[3], s);
    case RN_OP_POOL_FC_BWD:
      return rn_pool_fc_bwd((const float*)B(0), (const float*)B(1), (const float*)B(2), B(3), (float*)B(4), (float*)B(5), dt, d[0], d[1], d[2], d[3],
                            o.flags, s);
    case RN_OP_MAXPOOL_BWD:
      return rn_maxpool_bwd(B(0), (const unsigned char*)B(1), B(2), dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], s);
    case RN_OP_BN_BWD_REDUCE:
      return rn_bn_bwd_reduce(B(0), B(1), B(2), (const float*)B(3), (float*)B(4), d[2], dt, d[0], d[1], o.flags, o.fp[0], o.fp[1], o.seed, step_seed, s);
    case RN_OP_BN_BWD_FINALIZE:        /* partial dsum dgamma dbeta fold | nblk C */
      if (B(4))
        return rn_bn_bwd_finalize_split((const float*)B(0), d[0], (float*)B(1), (float*)B(2), (float*)B(3), d[1], o.flags, B(4), p->ws_bytes[o.buf[4]], s);
      return rn_bn_bwd_finalize((const float*)B(0), d[0], (float*)B(1), (float*)B(2), (float*)B(3), d[1], o.flags, s);
    case RN_OP_BN_BWD_APPLY:
      return rn_bn_bwd_apply(B(0), B(1), B(2), (const float*)B(3), (const float*)B(4), B(5), B(6), B(7), dt, d[0], d[1], d[2], d[3], d[4], d[5],
                             o.flags, o.fp[0], (double)d[6], o.fp[1], o.seed, step_seed, s);
    case RN_OP_CONV_DGRAD: {
      rn_conv_geom g = geom_of(o);
      rn_conv_epilogue ep{(float*)B(7), B(4), B(5), (const float*)B(6), o.fp[0], nullptr, (o.flags & RN_F_MASK_RECOMPUTE) ? 1 : 0};
      return rn_conv_dgrad(B(0), B(1), B(2), B(3), d[11], d[12], o.flags & ~RN_F_RELU, dt, &g, o.buf[7] >= 0 ? &ep : nullptr, s);
    }
    case RN_OP_CONV_WGRAD: {
      rn_conv_geom g = geom_of(o);
      return rn_conv_wgrad(B(0), B(1), (float*)B(2), B(3), o.buf[3] >= 0 ? p->ws_bytes[o.buf[3]] : 0, o.flags, dt, &g, s);
    }
    case RN_OP_STEM_WGRAD: {
      rn_conv_geom g = geom_of(o);
      if (o.buf[4] < 0 || p->ws_bytes[o.buf[4]] < rn_stem_wgrad_ws_bytes(&g)) {
        rn_set_error("rn_plan_run: op %d: stem wgrad workspace too small", idx);
        return 1;
      }
      return rn_stem_conv_wgrad((const float*)B(0), B(1), dt, (float*)B(2), (float*)B(3), B(4), (o.flags & RN_F_ACCUM) ? 1 : 0, &g, s);
    }
    case RN_OP_DROPOUT_BWD:
      return rn_dropout_bwd(B(0), B(2), dt, ((int64_t)d[1] << 31) | (int64_t)d[0], o.fp[0], o.seed, step_seed, s);
    case RN_OP_SOFTMAX_CE:
      return rn_softmax_ce((const float*)B(0), (const int64_t*)B(1), (float*)B(2), (float*)B(3), d[0], d[1], o.fp[0], (const float*)B(4), s);
    case RN_OP_ZERO: {
      size_t bytes = ((size_t)(uint32_t)d[1] << 31) | (size_t)(uint32_t)d[0];
      hipError_t e = hipMemsetAsync(B(0), 0, bytes, as_stream(s));
      if (e != hipSuccess) { rn_set_error("rn_plan_run: memset failed: %s", hipGetErrorString(e)); return 2; }
      return 0;
    }
    case RN_OP_ADD_RES:
      return rn_add_res(B(0), B(1), dt, d[0], d[1], d[2], d[3], d[4], d[5], s);
    case RN_OP_IMG_TO_NHWC:
      return rn_img_to_nhwc((const float*)B(0), B(1), dt, d[0], d[1], d[2], d[3], d[4], s);
    case RN_OP_PACK_STEM_W:
      return rn_pack_stem_w((const float*)B(0), B(1), dt, d[0], d[1], d[2], d[3], s);
    case RN_OP_BN_POOL_FWD:            /* x coef y argmax xsel | N H W C k stride pad */
      if (B(4))
        return rn_bn_pool_fwd_sel(B(0), (const float*)B(1), B(2), (unsigned char*)B(3), B(4), dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], o.flags, s);
      return rn_bn_pool_fwd(B(0), (const float*)B(1), B(2), (unsigned char*)B(3), dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], o.flags, s);
    case RN_OP_BN_POOL_BWD_REDUCE:     /* dy argmax x coef partial xsel | N H W C k stride pad nblk npix */
      if (B(5))
        return rn_bn_pool_bwd_reduce_sel(B(0), B(5), (const float*)B(3), (float*)B(4), d[7], dt, (long)d[8], d[3], o.flags, s);
      return rn_bn_pool_bwd_reduce(B(0), (const unsigned char*)B(1), B(2), (const float*)B(3), (float*)B(4), d[7], dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], o.flags, s);
    case RN_OP_BN_POOL_BWD_APPLY:      /* dy argmax x coef dsum dx sums | N H W C k stride pad count rows */
      if (B(6))
        return rn_bn_pool_bwd_apply_sums(B(0), (const unsigned char*)B(1), B(2), (const float*)B(3), (const float*)B(4), B(5), (float*)B(6), d[8], dt, d[0], d[1], d[2], d[3],
                                         d[4], d[5], d[6], o.flags, (double)d[7], s);
      return rn_bn_pool_bwd_apply(B(0), (const unsigned char*)B(1), B(2), (const float*)B(3), (const float*)B(4), B(5), dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6],
                                  o.flags, (double)d[7], s);
    case RN_OP_IMG_TO_S2D:             /* x out | N C H W */
      return rn_img_to_s2d((const float*)B(0), B(1), dt, d[0], d[1], d[2], d[3], s);
    case RN_OP_PACK_STEM_W_S2D:        /* w w_s2d | K C */
      return rn_pack_stem_w_s2d((const float*)B(0), B(1), dt, d[0], d[1], s);
    case RN_OP_UNPACK_STEM_DW_S2D:     /* dw_s2d dw | K C */
      return rn_unpack_stem_dw_s2d((const float*)B(0), (float*)B(1), d[0], d[1], (o.flags & RN_F_ACCUM) ? 1 : 0, s);
    case RN_OP_UNPACK_STEM_DW:
      return rn_unpack_stem_dw((const float*)B(0), (float*)B(1), d[0], d[1], d[2], d[3], (o.flags & RN_F_ACCUM) ? 1 : 0, s);
    case RN_OP_RELU_FWD:               /* x y | n_lo n_hi */
      return rn_relu_fwd(B(0), B(1), dt, ((int64_t)d[1] << 31) | (int64_t)d[0], s);
    case RN_OP_RELU_BWD:               /* dy y dx | n_lo n_hi */
      return rn_relu_bwd(B(0), B(1), B(2), dt, ((int64_t)d[1] << 31) | (int64_t)d[0], s);
    case RN_OP_AVGPOOL_FWD:            /* x y | N H W C k stride pad */
      return rn_avgpool_fwd(B(0), B(1), dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], s);
    case RN_OP_AVGPOOL_BWD:            /* dy dx | N H W C k stride pad */
      return rn_avgpool_bwd(B(0), B(1), dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], s);
    case RN_OP_PERMUTE_F32:            /* in out | A B C */
      return rn_permute_f32((const float*)B(0), (float*)B(1), d[0], d[1], d[2], s);
    default:
      rn_set_error("rn_plan_run: unknown op kind %d", o.kind);
      return 1;
  }
}

extern "C" int rn_plan_run(rn_plan* plan, int first, int last, uint64_t step_seed, rn_stream stream) {
  RN_CHECK_ARG(plan != nullptr, "rn_plan_run: null plan");
  RN_CHECK_ARG(first >= 0 && last <= (int)plan->ops.size() && first <= last, "rn_plan_run: bad range [%d, %d)", first, last);
  // a range that forks weight gradients onto the side stream: the chain's BatchNorm-backward kernels take the forms that fit beside them on a CU
  // -- where those are the 12-wave kernels of the 160-channel family (three waves of 136 registers per SIMD leave room for a fourth); the eight-phase kernels of
  // the 256-channel family hold every register of a CU, nothing fits beside them and the narrow BatchNorm forms only lose bandwidth (WRN-50-2-B: 52.9-53.1 vs 52.4-52.5 ms)
  bool forks = false;
  if (plan->overlap && !plan->profile)
    for (int i = first; i < last && !forks; ++i)
      if ((plan->ops[i].flags & RN_F_FORK) && plan->ops[i].kind == RN_OP_CONV_WGRAD) {
        const rn_conv_geom g = geom_of(plan->ops[i]);
        forks = rn_conv_wgrad8r_ok(&g, plan->dtype) != 0;
      }
  struct SideFriendly { bool on; explicit SideFriendly(bool f) : on(f) { if (on) rn_bn_side_friendly(1); } ~SideFriendly() { if (on) rn_bn_side_friendly(0); } } side_friendly(forks);
  rn_reduce_desc pending[RN_REDUCE_BATCH_MAX];
  int pending_slot[RN_REDUCE_BATCH_MAX];                  // the dw buffer of each pending sum: an op that touches one forces the flush first
  int n_pending = 0;
  // the slab-writing launches of those weight gradients wait, too, grouped by tile shape (rn_conv_wgrad_batch_key): every group goes out as ONE grid
  // when it is full or with the sums.  Their operands -- a layer's input and the gradient of its output -- are slots of their own that nothing
  // rewrites inside a backward, so running them later in the range reads the same values.  RN_NO_WGRAD_BATCH=1: every launch on its own (A/B).
  static const bool batch_wgrads = !(getenv("RN_NO_WGRAD_BATCH") && atoi(getenv("RN_NO_WGRAD_BATCH")) == 1);
  constexpr int WQ_KEYS = 6;
  struct WQueue { int key = 0, n = 0; rn_wgrad_desc d[RN_WGRAD_BATCH_MAX]; int xs[RN_WGRAD_BATCH_MAX], dys[RN_WGRAD_BATCH_MAX]; } wq[WQ_KEYS];
  auto launch_queue = [&](WQueue& q) -> int {
    if (!q.n) return 0;
    const int e = rn_conv_wgrad_batch(q.d, q.n, plan->dtype, stream);
    q.n = 0; q.key = 0;
    return e;
  };
  // Wide layers of the 160-channel family (WRN-28-10): the FORKED weight gradients the 320 x 160 kernel takes wait as well, per geometry, and go to the
  // side stream as ONE launch of up to w8r_batch layers (rn_conv_wgrad8r_batch: 1 / n of the pixel splits and slab traffic per layer, one ramp and tail).
  // RN_W8R_BATCH=<n>: the largest batch; the batch of a geometry is rn_conv_wgrad8r_best_batch(<= n).  Default 1 (every layer on its own) since the chain's
  // BatchNorm kernels run BESIDE the forked launches (rn_bn_side_friendly): short launches interleave with them layer by layer -- 6.06 ms per step against 6.17-6.19
  // (2, 4 layers) / 6.26 (8) / 6.35 (12) on one box; with the wide BatchNorm kernels it was the other way round (6.64 / 6.55 / 6.51 / 6.50 for 2 / 4 / 8 / 12).
  static const int w8r_batch = getenv("RN_W8R_BATCH") ? std::max(1, std::min(RN_WGRAD8R_BATCH_MAX, atoi(getenv("RN_W8R_BATCH")))) : 1;
  static const int w8r_fork_grid = getenv("RN_W8_FORK_GRID") ? atoi(getenv("RN_W8_FORK_GRID")) : 256;
  static const bool use_side2 = !(getenv("RN_NO_SIDE2") && atoi(getenv("RN_NO_SIDE2")) == 1);      // (A/B: the slab sums behind their kernel on the side stream)
  struct W8Queue { int n = 0, target = 1; rn_wgrad8r_desc d[RN_WGRAD8R_BATCH_MAX]; int xs[RN_WGRAD8R_BATCH_MAX], dys[RN_WGRAD8R_BATCH_MAX], dws[RN_WGRAD8R_BATCH_MAX]; } w8q;
  auto launch_w8q = [&]() -> int {
    if (!w8q.n) return 0;
    if (hipEventRecord(plan->ev_fork, as_stream(stream)) != hipSuccess || hipStreamWaitEvent(plan->side, plan->ev_fork, 0) != hipSuccess) {
      rn_set_error("rn_plan_run: fork onto the side stream failed (batched weight gradients)");
      return 2;
    }
    plan->side_pending = true;
    // every record names the ONE side workspace (the lowering's slot): with room for two regions the slab sums move to side2
    bool two = use_side2 && plan->side2 != nullptr;
    for (int i = 0; i < w8q.n && two; ++i) two = w8q.d[i].ws == w8q.d[0].ws && w8q.d[i].ws_bytes == w8q.d[0].ws_bytes && w8q.d[i].ws_bytes >= (1u << 20);
    int e;
    if (two) {
      const int p = plan->w_parity;
      plan->w_parity ^= 1;
      const size_t half = (w8q.d[0].ws_bytes / 2) & ~(size_t)255;
      for (int i = 0; i < w8q.n; ++i) { w8q.d[i].ws = static_cast<char*>(w8q.d[i].ws) + (size_t)p * half; w8q.d[i].ws_bytes = half; }
      if (plan->red_rec[p] && hipStreamWaitEvent(plan->side, plan->ev_red[p], 0) != hipSuccess) { rn_set_error("rn_plan_run: wait for the slab sums failed"); return 2; }
      e = rn_conv_wgrad8r_batch2(w8q.d, w8q.n, plan->dtype, w8r_fork_grid, reinterpret_cast<rn_stream>(plan->side), reinterpret_cast<rn_stream>(plan->side2), plan->ev_wg);
      if (!e && hipEventRecord(plan->ev_red[p], plan->side2) != hipSuccess) { rn_set_error("rn_plan_run: event record failed"); e = 2; }
      plan->red_rec[p] = true;
      plan->side2_pending = true;
    } else {
      if (int e2 = fold_side2(plan)) return e2;
      e = rn_conv_wgrad8r_batch(w8q.d, w8q.n, plan->dtype, w8r_fork_grid, reinterpret_cast<rn_stream>(plan->side));
    }
    w8q.n = 0;
    return e;
  };
  auto flush = [&]() -> int {
    if (int e = launch_w8q()) return e;
    for (auto& q : wq)
      if (int e = launch_queue(q)) return e;
    if (!n_pending) return 0;
    const int e = rn_wgrad_reduce_batch(pending, n_pending, stream);
    n_pending = 0;
    return e;
  };
  for (int i = first; i < last; ++i) {
    if (plan->arena && !plan->profile && plan->slab_off[i] >= 0) {      // a weight gradient whose slab sum is deferred to the end of the range
      const rn_op& o = plan->ops[i];
      auto P = [&](int b) -> void* { return o.buf[b] >= 0 ? plan->bufs[o.buf[b]] : nullptr; };
      rn_conv_geom g = geom_of(o);
      const size_t nel = (size_t)g.K * g.R * g.S * g.C;
      char* slabs = plan->arena + plan->slab_off[i];
      int e = 0;
      const int key = batch_wgrads ? rn_conv_wgrad_batch_key(&g, plan->dtype, o.flags) : 0;
      if (!(P(0) && P(1) && P(2))) {
        rn_set_error("rn_plan_run: op %d (kind %d) uses an unbound buffer", i, o.kind);
        e = 1;
      } else if (key > 0) {                                // queued: launched with the others of its tile shape
        WQueue* q = nullptr;
        for (auto& c : wq) if (c.n && c.key == key) { q = &c; break; }
        if (!q) for (auto& c : wq) if (!c.n) { q = &c; break; }
        if (!q) {                                            // every queue holds another shape: the fullest one goes out
          q = &wq[0];
          for (auto& c : wq) if (c.n > q->n) q = &c;
          e = launch_queue(*q);
        }
        if (!e) {
          q->key = key;
          q->xs[q->n] = o.buf[0]; q->dys[q->n] = o.buf[1];
          q->d[q->n++] = rn_wgrad_desc{P(0), P(1), reinterpret_cast<float*>(slabs), g, o.flags, plan->slab_splits[i], (uint64_t)plan->slab_splits[i] * nel * sizeof(float)};
          if (q->n == RN_WGRAD_BATCH_MAX) e = launch_queue(*q);
        }
      } else {
        e = rn_conv_wgrad(P(0), P(1), (float*)P(2), slabs, (size_t)plan->slab_splits[i] * nel * sizeof(float), o.flags | RN_F_DEFER_REDUCE, plan->dtype, &g, stream);
      }
      if (e) { std::string msg = g_err; rn_set_error("op %d (kind %d): %s", i, o.kind, msg.c_str()); return e; }
      pending_slot[n_pending] = o.buf[2];
      pending[n_pending++] = rn_reduce_desc{reinterpret_cast<const float*>(slabs), (float*)P(2), (int64_t)nel, plan->slab_splits[i], (o.flags & RN_F_ACCUM) ? 1 : 0};
      if (n_pending == RN_REDUCE_BATCH_MAX)
        if (int e2 = flush()) return e2;
      continue;
    }
    if (w8q.n) {                                           // an op that writes an operand of a queued wide weight gradient, or touches its dw, sends the queue out first
      const unsigned outs = rn_op_output_mask(plan->ops[i].kind);
      bool hit = false;
      for (int b = 0; b < RN_OP_NBUF && !hit; ++b) {
        const int sl = plan->ops[i].buf[b];
        if (sl < 0) continue;
        for (int j = 0; j < w8q.n && !hit; ++j)
          hit = sl == w8q.dws[j] || (((outs >> b) & 1) && (sl == w8q.xs[j] || sl == w8q.dys[j]));
      }
      if (hit)
        if (int e2 = launch_w8q()) return e2;
    }
    {                                                      // an op that WRITES a slot a queued weight gradient still has to read sends that queue out first
      const unsigned outs = rn_op_output_mask(plan->ops[i].kind);
      for (auto& q : wq) {
        bool hit = false;
        for (int b = 0; b < RN_OP_NBUF && !hit && q.n; ++b) {
          const int sl = plan->ops[i].buf[b];
          if (!((outs >> b) & 1) || sl < 0) continue;
          for (int j = 0; j < q.n; ++j)
            if (q.xs[j] == sl || q.dys[j] == sl) { hit = true; break; }
        }
        if (hit)
          if (int e2 = launch_queue(q)) return e2;
      }
    }
    if (n_pending) {                                       // e.g. the stem: UNPACK_STEM_DW reads the padded weight gradient right behind its CONV_WGRAD
      bool touched = false;
      for (int b = 0; b < RN_OP_NBUF && !touched; ++b)
        for (int q = 0; q < n_pending; ++q)
          if (plan->ops[i].buf[b] >= 0 && plan->ops[i].buf[b] == pending_slot[q]) { touched = true; break; }
      if (touched)
        if (int e2 = flush()) return e2;
    }
    if (!plan->profile && plan->ops[i].kind == RN_OP_PACK_W) {         // a run of weight packs becomes one launch
      rn_pack_desc descs[RN_PACK_BATCH_MAX];
      int n = 0;
      while (i + n < last && n < RN_PACK_BATCH_MAX && plan->ops[i + n].kind == RN_OP_PACK_W && !(plan->ops[i + n].flags & RN_F_FORK)) {
        const rn_op& o = plan->ops[i + n];
        auto P = [&](int b) -> void* { return o.buf[b] >= 0 ? plan->bufs[o.buf[b]] : nullptr; };
        descs[n] = rn_pack_desc{(const float*)P(0), P(1), P(2), o.dim[0], o.dim[1], o.dim[2]};
        ++n;
      }
      if (n > 1) {
        if (int e = rn_pack_weights_batch(descs, n, plan->dtype, stream)) {
          std::string msg = g_err;
          rn_set_error("op %d (kind %d): %s", i, plan->ops[i].kind, msg.c_str());
          return e;
        }
        i += n - 1;
        continue;
      }
    }
    const bool forked = plan->overlap && !plan->profile && (plan->ops[i].flags & RN_F_FORK);   // flagged ops own a workspace of their own
    if (forked && w8r_batch > 1 && plan->ops[i].kind == RN_OP_CONV_WGRAD) {
      const rn_op& o = plan->ops[i];
      rn_conv_geom g = geom_of(o);
      auto P = [&](int b) -> void* { return o.buf[b] >= 0 ? plan->bufs[o.buf[b]] : nullptr; };
      if (P(0) && P(1) && P(2) && P(3) && rn_conv_wgrad8r_ok(&g, plan->dtype)) {
        if (w8q.n && memcmp(&w8q.d[0].g, &g, sizeof(g)) != 0)
          if (int e2 = launch_w8q()) return e2;
        if (!w8q.n) w8q.target = rn_conv_wgrad8r_best_batch(&g, plan->dtype, w8r_batch);      // the count that fills whole rounds of the chip best
        w8q.xs[w8q.n] = o.buf[0]; w8q.dys[w8q.n] = o.buf[1]; w8q.dws[w8q.n] = o.buf[2];
        w8q.d[w8q.n++] = rn_wgrad8r_desc{P(0), P(1), (float*)P(2), P(3), plan->ws_bytes[o.buf[3]], g, o.flags};
        if (w8q.n >= w8q.target)
          if (int e2 = launch_w8q()) return e2;
        continue;
      }
    }
    rn_stream s = stream;
    if (forked) {
      if (hipEventRecord(plan->ev_fork, as_stream(stream)) != hipSuccess || hipStreamWaitEvent(plan->side, plan->ev_fork, 0) != hipSuccess) {
        rn_set_error("rn_plan_run: fork onto the side stream failed at op %d", i);
        return 2;
      }
      if (int e2 = fold_side2(plan)) return e2;             // (it may use the side workspace from its base: the pending slab sums read it)
      s = reinterpret_cast<rn_stream>(plan->side);
      plan->side_pending = true;
    }
    if (plan->profile) (void)hipEventRecord(plan->ev[2 * i], as_stream(stream));
    int e = run_op(plan, i, step_seed, s);
    if (plan->profile) { (void)hipEventRecord(plan->ev[2 * i + 1], as_stream(stream)); plan->ev_set[i] = 1; }
    if (e) {
      std::string msg = g_err;
      rn_set_error("op %d (kind %d): %s", i, plan->ops[i].kind, msg.c_str());
      return e;
    }
  }
  return flush();
}
