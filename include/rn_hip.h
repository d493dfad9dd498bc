/*
 * rn_hip.h -- C ABI of librn_hip.so: the MI355X (gfx950) engine for the ResNet forward/backward hot path.
 *
 * Boundary: the reference has no FFI; its hot path sits behind the torch.nn.Module
 * contract of ResNet (/root/reference/resnet/architectures/resnet.py:25-32 constructor, :165-166 forward) and the
 * autograd backward started at /root/reference/resnet/algos/training.py:100,102.  This library replaces what
 * that forward/backward dispatches to (ATen Conv2d / BatchNorm2d / ReLU / Dropout / pools / Linear kernels and their
 * autograd formulas) with hand-written HIP.  Plain pointers + sizes only: no torch types cross this boundary.
 * The caller owns every buffer (device memory, allocated through its own allocator); every entry point is
 * asynchronous on the given hipStream_t and returns 0 on success, non-zero on a rejected argument or launch
 * failure (message: rn_last_error()).
 *
 * Layouts: activations NHWC, element type `dtype` (RN_F32 | RN_BF16); convolution weights KRSC (that is the
 * physical order of a torch channels_last [K,C,R,S] tensor); statistics / BN coefficients / weight gradients /
 * logits fp32; network input NCHW fp32 as the reference's data loader hands it over (training.py:94).
 */
#ifndef RN_HIP_H
#define RN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rn_stream;                 /* a hipStream_t */

enum { RN_F32 = 0, RN_BF16 = 1, RN_F16 = 2 };   /* RN_F16: IEEE half storage + f16 MFMA, fp32 accumulate; needs loss scaling (rn_amp_*) */

/* residual / merge operand addressing, shared by conv epilogues, bn_apply and bn_bwd_apply */
enum {
  RN_RES_NONE = 0,
  RN_RES_SAME = 1,      /* res[n,h,w,c]                                    identity shortcut (residual_block.py:96)          */
  RN_RES_DOWN2PAD = 2,  /* c < res_C ? res[n,2h,2w,c] : 0                  AvgPool(1,2) + zero-pad shortcut (:49,90,94)     */
  RN_RES_UP2 = 3        /* (h,w even) ? res[n,h/2,w/2,c] : 0, c < res_C    its adjoint (backward of the line above)         */
};

/* ---- op kinds of a plan (one fused kernel launch each, a few expand to 2-4 launches) ---- */
enum {
  RN_OP_STEM_FWD = 1,       /* resnet.py:69-75   Conv2d(3->K,k,s,p,bias) on the NCHW fp32 image -> NHWC                       */
  RN_OP_PACK_W = 2,         /* fp32 KRSC master -> compute-dtype KRSC (forward) and CRSK (dgrad) copies                       */
  RN_OP_CONV_FWD = 3,       /* residual_block.py:34-57,129-159 Conv2d(bias=False) 3x3/1x1, stride 1|2, + residual epilogue   */
  RN_OP_BN_STATS = 4,       /* BatchNorm2d train: per-channel partial (sum, sumsq)                                            */
  RN_OP_BN_FINALIZE = 5,    /* -> coef = (scale, shift, mean, invstd); running-stat update (train) or running stats (eval)    */
  RN_OP_BN_APPLY = 6,       /* y = [relu](x*scale+shift [+ res]) [* dropout]   (residual_block.py:70-72,81-83,96-98)          */
  RN_OP_DROPOUT_FWD = 7,    /* standalone Dropout in front of a v1 block's first conv (residual_block.py:80)                  */
  RN_OP_MAXPOOL_FWD = 8,    /* resnet.py:83-87                                                                                  */
  RN_OP_POOL_FC_FWD = 9,    /* resnet.py:77-81 global AvgPool2d + :117-120 Flatten+Linear                                     */
  RN_OP_POOL_FC_BWD = 10,
  RN_OP_MAXPOOL_BWD = 11,
  RN_OP_BN_BWD_REDUCE = 12, /* per-channel partial (sum g, sum g*xhat), g = dout * relu/dropout mask                          */
  RN_OP_BN_BWD_FINALIZE = 13, /* partials -> (sum g, sum g*xhat) and the dgamma/dbeta gradients                               */
  RN_OP_BN_BWD_APPLY = 14,  /* dx = scale*(g - mean(g) - xhat*mean(g*xhat)) [+ add operand]; optional copy of g              */
  RN_OP_CONV_DGRAD = 15,
  RN_OP_CONV_WGRAD = 16,
  RN_OP_STEM_WGRAD = 17,
  RN_OP_DROPOUT_BWD = 18,
  RN_OP_SOFTMAX_CE = 19,    /* metrics.py:10-29: mean CE loss, top-1/top-5 error counts, dlogits                             */
  RN_OP_ZERO = 20,          /* memset of a buffer slot (split-K / accumulation targets)                                       */
  RN_OP_ADD_RES = 21,       /* dst += res (mapped): gradient merge behind a standalone v1 dropout                             */
  RN_OP_IMG_TO_NHWC = 22,   /* NCHW fp32 image -> NHWC compute dtype, channels zero-padded to one 16-byte chunk (MFMA stem)    */
  RN_OP_PACK_STEM_W = 23,   /* stem weights [K][RS][C] fp32 -> [K][RS][CP] compute dtype (padded)                              */
  RN_OP_UNPACK_STEM_DW = 24,/* stem weight gradient [K][RS][CP] fp32 -> [K][RS][C] (drops the padding)                          */
  RN_OP_BN_POOL_FWD = 25,        /* y, argmax = maxpool([relu](x * scale + shift)): BN_APPLY + MAXPOOL_FWD of a top-level "n a mp" in one pass */
  RN_OP_BN_POOL_BWD_REDUCE = 26, /* BN_BWD_REDUCE over g = [bn(x) > 0] * maxpool_bwd(dy, argmax), without materialising it */
  RN_OP_BN_POOL_BWD_APPLY = 27,  /* BN_BWD_APPLY over the same g */
  RN_OP_IMG_TO_S2D = 28,         /* NCHW fp32 image -> space-to-depth NHWC [N][H/2+3][W/2+3][16] (the 7x7 / stride-2 stem as a 4x4 VALID convolution)  */
  RN_OP_PACK_STEM_W_S2D = 29,    /* stem weights [K][7][7][C] fp32 -> [K][4][4][16] compute dtype                                                       */
  RN_OP_UNPACK_STEM_DW_S2D = 30, /* stem weight gradient [K][4][4][16] fp32 -> [K][7][7][C]                                                             */
  /* grammar corners of resnet.py:122-158 (any token sequence is a network; no shipped config uses these): */
  RN_OP_RELU_FWD = 31,           /* resnet.py:143-145 a top-level 'a' that follows no 'n': x y | n_lo n_hi                                               */
  RN_OP_RELU_BWD = 32,           /* dy y dx | n_lo n_hi                                                                                                  */
  RN_OP_AVGPOOL_FWD = 33,        /* resnet.py:77-81 AvgPool2d(k, s, p) other than the global pool in front of 'f': x y | N H W C k stride pad            */
  RN_OP_AVGPOOL_BWD = 34,        /* dy dx | N H W C k stride pad                                                                                         */
  RN_OP_PERMUTE_F32 = 35         /* out[a][c][b] = in[a][b][c]: Linear weights / gradients between NCHW-flattened and NHWC feature order: in out | A B C */
};

/* flags */
enum {
  RN_F_RELU = 1 << 0,
  RN_F_TRAIN = 1 << 1,        /* BN uses batch statistics                                   */
  RN_F_ACCUM = 1 << 2,        /* destination += result                                      */
  RN_F_WRITE_G = 1 << 3,      /* bn_bwd_apply also writes g = masked dout (v1 shortcut)     */
  RN_F_NEED_DGRAD_PACK = 1 << 4,
  RN_F_SKIP_FWD_PACK = 1 << 5,
  RN_F_NO_DX = 1 << 6,        /* pool_fc_bwd etc.: input gradient not needed                */
  RN_F_MASK_RECOMPUTE = 1 << 7, /* bn_bwd_*: mask = [x*scale+shift > 0] & dropout hash, recomputed instead of read (mask_src NULL) */
  RN_F_FORK = 1 << 8,         /* plan executor: run this weight-gradient op on the side stream (rn_plan_set_overlap) */
  RN_F_DEFER_REDUCE = 1 << 9  /* rn_conv_wgrad: write the split-K slabs only; the caller sums them later (rn_wgrad_reduce_batch) */
};

#define RN_OP_NBUF 8
#define RN_OP_NDIM 20

/* One op of a plan.  buf[] are indices into the plan's buffer table (-1 = absent); dim[] and fp[] are per-kind
 * (documented next to each rn_* launcher below: the executor forwards them 1:1 to that launcher). */
typedef struct rn_op {
  int32_t kind;
  int32_t flags;
  int32_t buf[RN_OP_NBUF];
  int32_t dim[RN_OP_NDIM];
  float fp[4];
  uint32_t seed;     /* dropout site id: mask = hash(step_seed, seed, element index) */
  int32_t pad_;
} rn_op;

typedef struct rn_plan rn_plan;

const char* rn_last_error(void);
/* ABI version: bumped with EVERY change of an entry point's signature or meaning; a binding refuses a library of another version (a stale
 * librn_hip.so would otherwise take shifted pointer / integer arguments).  3: round 3 (operand-set flags of rn_conv_kernel_names, workspaces).
 * 9: round 4 (rn_set_variant2).  10: rn_conv_wgrad8r_best_batch.  11: rn_bn_side_friendly, rn_conv_wgrad8r_batch2, rn_wgrad_desc.splits / slab_bytes. */
#define RN_ABI_VERSION 11
int rn_version(void);
/* kernel-variant switch for A/B measurements (tools/conv_bench.py; bits documented in csrc/conv_igemm.hip); 0 = shipped */
void rn_set_variant(int v);
/* second switch word (round 4; bits documented in csrc/conv_igemm8r.hip): 1 = never take the row-patch 256 x 160 convolution kernel (igemm8r), 2 = take it on any
 * grid size (tests, tools/conv_bench.py); 0 = shipped */
void rn_set_variant2(int v);
/* Stream-K workspace of the 256-row convolution kernels (csrc/conv_igemm8.hip): device memory of rn_conv_workspace_bytes() bytes whose first 4 KiB are
 * zero (tile tickets; the kernels leave them zero), owned by the caller, used by one stream at a time.  NULL (default): whole tiles only. */
size_t rn_conv_workspace_bytes(void);
int rn_set_conv_workspace(void* p, size_t bytes);
/* which convolution kernel ran: rn_kernel_log(1) starts (and clears) a per-thread log of the instantiations the conv launchers
 * pick ("igemm_dma<128x160>", "wgrad<160x160>", ...), rn_kernel_log_read() returns them comma-separated, rn_kernel_log(0) stops.
 * rn_conv_kernel_names: the names a geometry WOULD select (pass 0 forward, 1 dgrad, 2 wgrad), without launching anything
 * (host-only: works without a GPU) -- the tests use it to prove that every tile a BASELINE config selects is parity-tested.
 * fused_epilogue names the launch's operand set (kernels may be specialised per set): 1 = fused BatchNorm sums (forward: statistics of the
 * output; data gradient: the backward sums over x and the mask), 2 = identity residual, 4 = accumulate into dx, 8 = per-channel bias (the stem convolution), 16 = rn_conv_epilogue.mask_from_x (data gradient) */
void rn_kernel_log(int enable);
const char* rn_kernel_log_read(void);
struct rn_conv_geom;
int rn_conv_kernel_names(int pass, int dtype, const struct rn_conv_geom* g, int fused_epilogue, char* out, size_t n);
/* diagnostic builds only: device buffer [grid][16] of u64 s_memtime stamps written by the implicit-GEMM kernels (NULL = off) */
void rn_set_stamp_buffer(void* device_u64);

/* ---- plan executor: the per-batch forward / backward of ResNet.forward as ONE host call each ---- */
int rn_plan_create(const rn_op* ops, int n_ops, int n_bufs, int dtype, rn_plan** out);
int rn_plan_bind(rn_plan* plan, const void* const* device_ptrs, int n_bufs);
/* bytes available behind a workspace slot (wgrad split slabs): checked before every launch that uses it */
int rn_plan_set_bytes(rn_plan* plan, int slot, size_t bytes);
/* runs ops [first, last) in order on `stream`; step_seed feeds the dropout hash (same value forward and backward) */
int rn_plan_run(rn_plan* plan, int first, int last, uint64_t step_seed, rn_stream stream);
/* weight-gradient ops of the plan on a second (low-priority) stream, forked by events at their list position; the
 * caller joins before it consumes weight gradients (end of the backward, or before reducing a gradient bucket) */
int rn_plan_set_overlap(rn_plan* plan, int enable);
int rn_plan_join(rn_plan* plan, rn_stream stream);
/* `stream` (e.g. the gradient all-reduce stream) waits for the weight-gradient ops forked so far; nothing is joined into the launch
 * stream (rn_plan_join at the end of the range still does that) */
int rn_plan_side_wait(rn_plan* plan, rn_stream stream);
int rn_plan_num_ops(const rn_plan* plan);
/* Deferred weight-gradient slab sums inside a plan: rn_plan_defer_reduce marks the (not forked) CONV_WGRAD ops whose reduction is of the
 * deferrable kind, gives each its own slab region in an arena and returns the arena bytes (0: none); rn_plan_set_reduce_arena hands the
 * arena over (NULL: off).  rn_plan_run then sums the slabs of a whole range in batched launches at the END of the range (and every
 * RN_REDUCE_BATCH_MAX layers): a host-side consumer of the gradients (bucket hooks) sits between ranges and sees them complete. */
size_t rn_plan_defer_reduce(rn_plan* plan);
int rn_plan_set_reduce_arena(rn_plan* plan, void* arena, size_t bytes);
/* per-op hipEvent pairs on the launch stream; rn_plan_profile_read blocks and returns ms per op (0 = not run) */
int rn_plan_profile(rn_plan* plan, int enable);
int rn_plan_profile_read(rn_plan* plan, float* ms, int n);
void rn_plan_destroy(rn_plan* plan);

/* ---- geometry of a (transposed) convolution as an implicit GEMM ---- */
typedef struct rn_conv_geom {
  int32_t N, H, W, C;      /* input  [N,H,W,C]  */
  int32_t P, Q, K;         /* output [N,P,Q,K]  */
  int32_t R, S, stride, pad;
} rn_conv_geom;

/* optional fused epilogue of rn_conv_fwd / rn_conv_dgrad (NULL = plain store).  The convolution reduces, per tile of
 * RN_CONV_STATS_ROWS output pixels, two per-channel sums of what it stores, so the adjacent BatchNorm needs no pass of
 * its own over the tensor:
 *   forward (bn_x == NULL): partial[row][0][k] = sum y, partial[row][1][k] = sum y^2        -> rn_bn_finalize
 *   dgrad   (bn_x != NULL): g = dx * gscale * [bn_mask > 0];  partial[row] = (sum g, sum g*xhat), xhat from bn_x, bn_coef
 *                                                                                             -> rn_bn_bwd_finalize
 * partial has rn_conv_stats_rows(g, is_dgrad) rows of [2][channels]. */
#define RN_CONV_STATS_ROWS 128
typedef struct rn_conv_epilogue {
  float* partial;
  const void* bn_x;
  const void* bn_mask;     /* NULL: no ReLU/dropout mask */
  const float* bn_coef;    /* [4][C] of that BatchNorm */
  float gscale;            /* 1/(1-p) of the dropout behind that BatchNorm, else 1 */
  const float* bias;       /* forward only: per-output-channel bias (the stem Conv2d, resnet.py:69-75), or NULL */
  int mask_from_x;         /* the caller's promise that bn_mask > 0 exactly where bn_x * scale + shift > 0 (a plain BatchNorm + ReLU: no residual added before the
                              ReLU, no dropout): a kernel may then test bn_x, which it reads for the sums anyway, and leave the mask tensor unread */
} rn_conv_epilogue;
int rn_conv_stats_rows(const rn_conv_geom* g, int is_dgrad);

/* ---- single kernels (also the executor's building blocks; tests call these through one-op plans or directly) ---- */

/* y[n,p,q,k] = bias[k] + sum x[n,c,p*s+r-pad,q*s+t-pad] * w[k,r,t,c];  x NCHW fp32, w KRSC fp32, y NHWC dtype */
int rn_stem_conv_fwd(const float* x_nchw, const float* w_krsc, const float* bias, void* y, int dtype,
                     const rn_conv_geom* g, rn_stream s);
/* dw[k,r,t,c], db[k] (fp32, overwritten or accumulated); ws: >= rn_stem_wgrad_ws_bytes(g) bytes */
int rn_stem_conv_wgrad(const float* x_nchw, const void* dy, int dtype, float* dw_krsc, float* dbias, void* ws,
                       int accumulate, const rn_conv_geom* g, rn_stream s);
size_t rn_stem_wgrad_ws_bytes(const rn_conv_geom* g);

/* MFMA route of the stem (kernel 7x7: GEMM-K = 147 pays on the matrix cores): the image and the weights are brought
 * to NHWC with the 3 channels zero-padded to CP (8 bf16 / 4 fp32 = one 16-byte chunk), then rn_conv_fwd / rn_conv_wgrad run
 * with C = CP; rn_unpack_stem_dw drops the padding of the weight gradient */
int rn_img_to_nhwc(const float* x_nchw, void* out, int dtype, int N, int C, int H, int W, int CP, rn_stream s);
int rn_pack_stem_w(const float* w_krsc, void* w_padded, int dtype, int K, int RS, int C, int CP, rn_stream s);
int rn_unpack_stem_dw(const float* dw_padded, float* dw_krsc, int K, int RS, int C, int CP, int accumulate, rn_stream s);
/* The ImageNet stem Conv2d(C <= 4 -> K, 7 x 7, stride 2, padding 3) (resnet.py:69-75 under "c3,K,7,2,3") as a 4 x 4 / stride-1 / VALID convolution over a
 * space-to-depth image: s2d pixel (i, j) = image pixels (2i + dy, 2j + dx) as 16 channels ((dy, dx, c of 4), zero-filled), stored with 2 zero s2d pixels before
 * and 1 after in both directions: out [N][H/2 + 3][W/2 + 3][16] (H, W even).  Filter tap (r', s'), channel (dy, dx, c) = the 7 x 7 weight at
 * (2r' + dy - 1, 2s' + dx - 1), zero outside: w_s2d [K][4][4][16].  rn_conv_fwd / rn_conv_wgrad on that geometry (C = 16, R = S = 4, stride 1, pad 0) give the
 * stem's output and a [K][4][4][16] weight gradient, which rn_unpack_stem_dw_s2d maps back.  4 K tiles of 64 instead of 7, and contiguous 128-byte rows. */
int rn_img_to_s2d(const float* x_nchw, void* out, int dtype, int N, int C, int H, int W, rn_stream s);
int rn_pack_stem_w_s2d(const float* w_k77c, void* w_s2d, int dtype, int K, int C, rn_stream s);
int rn_unpack_stem_dw_s2d(const float* dw_s2d, float* dw_k77c, int K, int C, int accumulate, rn_stream s);

/* fp32 KRSC master -> w_fwd [K][R*S][C] and w_dgrad [C][R*S][K] in dtype (either may be NULL) */
int rn_pack_weights(const float* w_krsc, void* w_fwd, void* w_dgrad, int dtype, int K, int RS, int C, rn_stream s);
/* the same for up to RN_PACK_BATCH_MAX weights in ONE launch (descriptors are host memory, consumed before the call returns) */
#define RN_PACK_BATCH_MAX 32
typedef struct rn_pack_desc {
  const float* w;   /* fp32 KRSC master */
  void* w_fwd;      /* [K][R*S][C] in dtype, or NULL */
  void* w_dgrad;    /* [C][R*S][K] in dtype, or NULL */
  int32_t K, RS, C;
} rn_pack_desc;
int rn_pack_weights_batch(const rn_pack_desc* descs, int n, int dtype, rn_stream s);

/* y = conv(x, w_fwd) [+ res];  MFMA implicit GEMM, M = N*P*Q, N = K, K = R*S*C.  C % 8 == 0, K % 16 == 0 */
int rn_conv_fwd(const void* x, const void* w_fwd, void* y, const void* res, int res_mode, int res_C, int dtype,
                const rn_conv_geom* g, const rn_conv_epilogue* ep, rn_stream s);
/* dx = conv_transpose(dy, w) [+ res] ; flags: RN_F_ACCUM (dx += ...).  With an epilogue descriptor the BatchNorm-backward sums are
 * taken over the value finally stored in dx, so with RN_F_ACCUM the call must be the last accumulation into dx */
int rn_conv_dgrad(const void* dy, const void* w_dgrad, void* dx, const void* res, int res_mode, int res_C,
                  int flags, int dtype, const rn_conv_geom* g, const rn_conv_epilogue* ep, rn_stream s);
/* dw[k,r,s,c] (fp32 KRSC) = sum_{n,p,q} dy * x ; split over pixels into ws, then reduced. flags: RN_F_ACCUM */
int rn_conv_wgrad(const void* x, const void* dy, float* dw_krsc, void* ws, size_t ws_bytes, int flags, int dtype,
                  const rn_conv_geom* g, rn_stream s);
size_t rn_conv_wgrad_ws_bytes(const rn_conv_geom* g);
/* Deferred slab sums.  A thin layer's weight gradient is a ~10 us kernel followed by a ~5 us launch that only adds its split-K slabs;
 * with RN_F_DEFER_REDUCE rn_conv_wgrad leaves the slabs in `ws` (which must then be a region of its OWN until they are summed) and
 * rn_wgrad_reduce_batch sums the slabs of up to RN_REDUCE_BATCH_MAX layers in ONE launch -- same per-output summation order as the
 * immediate reduction (bitwise identical results).  rn_conv_wgrad_splits: the slab count [splits][K*R*S*C] that launch writes
 * (0: it writes dw directly, nothing to sum; < 0: this geometry's reduction is not of the deferrable kind). */
#define RN_REDUCE_BATCH_MAX 32
typedef struct rn_reduce_desc {
  const float* slabs;   /* [splits][n] */
  float* dw;            /* [n] */
  int64_t n;
  int32_t splits;
  int32_t accumulate;   /* dw += sum */
} rn_reduce_desc;
int rn_conv_wgrad_splits(const rn_conv_geom* g, int dtype, int flags);
int rn_wgrad_reduce_batch(const rn_reduce_desc* descs, int n, rn_stream s);
/* Thin networks (ResNet-20 / ResNet-v2-164: ~165 weight gradients of ~10 us per step): the slab-writing launches of up to RN_WGRAD_BATCH_MAX layers
 * that share a tile shape as ONE grid.  rn_conv_wgrad_batch_key: > 0 when this geometry's weight gradient may ride in such a launch, with others of
 * the same key (3 x 3 / 1 x 1 block convolutions whose slab sums are deferrable, not forked); 0: it launches alone.  Every record writes
 * [rn_conv_wgrad_splits][K*R*S*C] slabs at `slabs`, exactly as rn_conv_wgrad(..., RN_F_DEFER_REDUCE) does (bit-identical); rn_wgrad_reduce_batch sums them. */
#define RN_WGRAD_BATCH_MAX 16
typedef struct rn_wgrad_desc {
  const void* x;        /* the layer's input,  NHWC compute dtype */
  const void* dy;       /* the gradient of its output             */
  float* slabs;
  rn_conv_geom g;
  int32_t flags;
  int32_t splits;       /* the split count the slab region was sized for (rn_conv_wgrad_splits when the plan was built): the launch refuses another one --
                         * rn_set_variant between planning and running would otherwise write past the region or sum the wrong number of slabs; 0 = unchecked */
  uint64_t slab_bytes;  /* bytes of the region at `slabs` (0 = unchecked) */
} rn_wgrad_desc;
int rn_conv_wgrad_batch_key(const rn_conv_geom* g, int dtype, int flags);
int rn_conv_wgrad_batch(const rn_wgrad_desc* descs, int n, int dtype, rn_stream s);
/* Wide layers of the 160-channel family (WRN-28-10: 3x3 convolutions with 160 n channels): the weight gradients of up to RN_WGRAD8R_BATCH_MAX layers of ONE
 * geometry as one launch of the 320 x 160 kernel (csrc/conv_wgrad8r.hip).  A 160-channel layer has five output tiles, so a launch of its own cuts the pixels
 * ~51 ways (47 MB of fp32 slabs per layer); n layers in one launch take ~1/n of those splits each.  Every record's slabs go to its own workspace (ws_bytes >=
 * what rn_conv_wgrad_ws_bytes gives the layer is always enough) and are summed into its dw (+= with RN_F_ACCUM) by the fixed-order reductions right behind,
 * on the same stream: the gradients of a batch are bitwise reproducible for a given n.  rn_conv_wgrad8r_ok: 1 when the geometry is one the kernel takes. */
#define RN_WGRAD8R_BATCH_MAX 12
typedef struct rn_wgrad8r_desc {
  const void* x;
  const void* dy;
  float* dw;
  void* ws;
  size_t ws_bytes;
  rn_conv_geom g;
  int32_t flags;
} rn_wgrad8r_desc;
int rn_conv_wgrad8r_ok(const rn_conv_geom* g, int dtype);
int rn_conv_wgrad8r_batch(const rn_wgrad8r_desc* descs, int n, int dtype, int max_grid, rn_stream s);
/* the same with the slab sums on s_reduce behind `ev` (a hipEvent_t of the caller, recorded on s behind the kernel): light launches that share CUs with the next
 * weight gradient; the caller orders the workspace's re-use and folds s_reduce back before the gradients are consumed (csrc/plan.cpp does both) */
int rn_conv_wgrad8r_batch2(const rn_wgrad8r_desc* descs, int n, int dtype, int max_grid, rn_stream s, rn_stream s_reduce, void* ev);
/* how many layers of geometry g are worth collecting for one rn_conv_wgrad8r_batch launch (1..max_n): the count whose tiles fill whole rounds of the chip best
 * (modelled time per layer; csrc/conv_wgrad9.hip) */
int rn_conv_wgrad8r_best_batch(const rn_conv_geom* g, int dtype, int max_n);
unsigned rn_op_output_mask(int kind);    /* bit b set: an op of this kind WRITES buf[b] (rn_plan_run sends a queued weight gradient out before an op that
                                          * would rewrite one of its operands); mirrored by engine/ir.py OP_OUTPUTS, checked by tests/test_abi.py */
long rn_wgrad_batch_launches(void);      /* diagnostic: batched launches this process has issued (the kernel log keeps the per-record tile names) */

/* BatchNorm over a [M, C] view.  partial: [nblk][2][C] fp32 (sum, sum of squares) of nblk row slabs; the caller picks
 * nblk (one workgroup per slab) */
int rn_bn_stats(const void* x, float* partial, int nblk, int dtype, int64_t M, int C, rn_stream s);
/* coef: [4][C] = scale, shift, mean, invstd.  train: batch stats from partial (count = number of rows over ALL
 * ranks that contributed to `partial`), running_mean/var (momentum, unbiased var) and num_batches_tracked updated
 * in place; eval: coef from running stats.  */
int rn_bn_finalize(const float* partial, int nblk, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* coef, int C,
                   float eps, float momentum, int flags, rn_stream s);
/* The two finalize kernels with the partial ROWS split over workgroups, for the thousands of rows the fused conv epilogues leave on the ImageNet nets
 * (one workgroup per 16 channels reads them at a fraction of HBM speed).  rn_bn_fold_bytes(nblk, C): bytes of the caller-owned `fold` buffer such a
 * launch needs, 0 = this size is not split (call the plain function).  `fold` must be zero before its first use and belong to ONE layer (launches that
 * may overlap must not share it); the sums are added in a fixed order (bitwise reproducible).  Training mode only for the forward form. */
size_t rn_bn_fold_bytes(int nblk, int C);
int rn_bn_finalize_split(const float* partial, int nblk, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* coef, int C,
                         float eps, float momentum, int flags, void* fold, size_t fold_bytes, rn_stream s);
int rn_bn_bwd_finalize_split(const float* partial, int nblk, float* dsum, float* dgamma, float* dbeta, int C, int flags, void* fold, size_t fold_bytes,
                             rn_stream s);
/* out = [relu](x*scale+shift [+ res]) [dropout(p)] ; geometry N,H,W of x for the residual mapping */
int rn_bn_apply(const void* x, const float* coef, const void* res, void* out, int dtype, int N, int H, int W, int C,
                int res_mode, int res_C, int flags, float drop_p, uint32_t site, uint64_t step_seed, rn_stream s);
/* g = dout * gscale * [mask_src > 0] (mask only with RN_F_RELU; gscale = 1/(1-p)); partial [nblk][2][C] */
int rn_bn_bwd_reduce(const void* dout, const void* x, const void* mask_src, const float* coef, float* partial,
                     int nblk, int dtype, int64_t M, int C, int flags, float gscale, float drop_p, uint32_t site,
                     uint64_t step_seed, rn_stream s);
/* partial -> dsum [2][C]; dgamma/dbeta written (or accumulated with RN_F_ACCUM) */
/* 1: the calling thread's following BatchNorm-backward launches (finalize, apply) take the forms that fit beside the forked weight gradients' persistent
 * workgroups (256-thread finalize, two rows in flight); rn_plan_run sets it for ranges that fork onto the side stream and clears it again */
void rn_bn_side_friendly(int on);
int rn_bn_bwd_finalize(const float* partial, int nblk, float* dsum, float* dgamma, float* dbeta, int C, int flags,
                       rn_stream s);
/* dx = scale*(g - dsum0/count - xhat*dsum1/count) [train] | scale*g [eval]  [+ add operand];  optional g_out */
int rn_bn_bwd_apply(const void* dout, const void* x, const void* mask_src, const float* coef, const float* dsum,
                    const void* add, void* dx, void* g_out, int dtype, int N, int H, int W, int C, int add_mode,
                    int add_C, int flags, float gscale, double count, float drop_p, uint32_t site, uint64_t step_seed,
                    rn_stream s);

int rn_dropout_fwd(const void* x, void* out, int dtype, int64_t n, float p, uint32_t site, uint64_t step_seed, rn_stream s);
/* din = dout * [keep(site, step_seed, index)] / (1-p): the mask is recomputed from the counter hash of the forward */
int rn_dropout_bwd(const void* dout, void* din, int dtype, int64_t n, float p, uint32_t site, uint64_t step_seed, rn_stream s);
/* dst[n,h,w,c] += res (RN_RES_* mapping) */
int rn_add_res(void* dst, const void* res, int dtype, int N, int H, int W, int C, int res_mode, int res_C, rn_stream s);

/* grammar corners (resnet.py:122-158 builds any token sequence; none of the shipped configs uses these).
 * rn_relu_*: nn.ReLU as a layer of its own (:143-145), backward by the sign of the stored output; n elements, a multiple of the 16-byte chunk.
 * rn_avgpool_*: nn.AvgPool2d(k, s, p) (:77-81; zero padding counts in the divisor, floor output size), NHWC; the backward is a gather.
 * rn_permute_f32: out[a][c][b] = in[a][b][c] -- Flatten() of an NCHW map with more than one pixel orders the features (c, h, w) (:117-120), the
 * engine's maps are (h, w, c): the Linear weight goes [O][C][HW] -> [O][HW][C] in front of rn_pool_fc_fwd (HW = 1), its gradient comes back. */
int rn_relu_fwd(const void* x, void* y, int dtype, int64_t n, rn_stream s);
int rn_relu_bwd(const void* dy, const void* y, void* dx, int dtype, int64_t n, rn_stream s);
int rn_avgpool_fwd(const void* x, void* y, int dtype, int N, int H, int W, int C, int k, int stride, int pad, rn_stream s);
int rn_avgpool_bwd(const void* dy, void* dx, int dtype, int N, int H, int W, int C, int k, int stride, int pad, rn_stream s);
int rn_permute_f32(const float* in, float* out, int A, int B, int C, rn_stream s);

/* argmax: one byte per output element (window position r*k+s of the first maximum), consumed by the backward */
int rn_maxpool_fwd(const void* x, void* y, unsigned char* argmax, int dtype, int N, int H, int W, int C, int k, int stride,
                   int pad, rn_stream s);
int rn_maxpool_bwd(const void* dy, const unsigned char* argmax, void* dx, int dtype, int N, int H, int W, int C, int k,
                   int stride, int pad, rn_stream s);
/* BatchNorm-apply (+ReLU with RN_F_RELU) + MaxPool fused (the "n a mp3,2,1" stem of the ImageNet nets, resnet.py:111-115, 83-87): the
 * normalised activation and its gradient are never stored.  C / (16-byte chunk) must divide 256.  The backward pair replaces
 * rn_maxpool_bwd + rn_bn_bwd_reduce / rn_bn_bwd_apply (mask recomputed from x and coef): partial = [nblk][2][C] rows for rn_bn_bwd_finalize. */
int rn_bn_pool_fwd(const void* x, const float* coef, void* y, unsigned char* argmax, int dtype, int N, int H, int W, int C, int k, int stride,
                   int pad, int flags, rn_stream s);
int rn_bn_pool_bwd_reduce(const void* dy, const unsigned char* argmax, const void* x, const float* coef, float* partial, int nblk, int dtype, int N,
                          int H, int W, int C, int k, int stride, int pad, int flags, rn_stream s);
int rn_bn_pool_bwd_apply(const void* dy, const unsigned char* argmax, const void* x, const float* coef, const float* dsum, void* dx, int dtype, int N,
                         int H, int W, int C, int k, int stride, int pad, int flags, double count, rn_stream s);
/* rn_bn_pool_fwd that also keeps xsel[N][P][Q][C] = the input element that won each window; with it the backward sums are taken at pooled
 * resolution (dy and xsel, 2 x 1/4 of the map, instead of the map + dy + argmax): partial rows for rn_bn_bwd_finalize as rn_bn_pool_bwd_reduce
 * leaves them (1 <= nblk <= 8192, npix = N * P * Q).  Windows are added unrounded, the gather form rounds each pixel's summed gradient to the
 * compute dtype first: equal in fp32 up to summation order, within the dtype's rounding otherwise. */
int rn_bn_pool_fwd_sel(const void* x, const float* coef, void* y, unsigned char* argmax, void* xsel, int dtype, int N, int H, int W, int C, int k,
                       int stride, int pad, int flags, rn_stream s);
int rn_bn_pool_bwd_reduce_sel(const void* dy, const void* xsel, const float* coef, float* partial, int nblk, int dtype, long npix, int C, int flags,
                              rn_stream s);
/* the same pass, also leaving sums_partial[rows][2][C] = per-workgroup (sum dx, 0) of the stored gradient (rows = its grid, 1..8192):
 * the bias gradient of a biased producer (the ImageNet stem convolution, resnet.py:111) through rn_bn_bwd_finalize, without another pass over dx */
int rn_bn_pool_bwd_apply_sums(const void* dy, const unsigned char* argmax, const void* x, const float* coef, const float* dsum, void* dx, float* sums_partial,
                              int rows, int dtype, int N, int H, int W, int C, int k, int stride, int pad, int flags, double count, rn_stream s);

/* logits[n,o] = b[o] + sum_c W[o,c] * mean_{hw} x[n,hw,c];  feat: [N][C] fp32 scratch kept for backward */
int rn_pool_fc_fwd(const void* x, const float* w, const float* b, float* feat, float* logits, int dtype, int N, int HW,
                   int C, int O, rn_stream s);
int rn_pool_fc_bwd(const float* dlogits, const float* feat, const float* w, void* dx, float* dw, float* db, int dtype,
                   int N, int HW, int C, int O, int flags, rn_stream s);

/* out3 = (sum nll, #top1 wrong, #top5 wrong) over the batch (fp32, overwritten); dlogits (may be NULL) = (softmax - onehot) *
 * scale * (scale_dev ? *scale_dev : 1): scale_dev is a DEVICE scalar -- the upstream gradient of the loss, i.e. the AMP loss scale of
 * scaler.scale(loss).backward() (training.py:100) -- so no host synchronisation is needed to apply it */
int rn_softmax_ce(const float* logits, const int64_t* labels, float* out3, float* dlogits, int N, int O, float scale,
                  const float* scale_dev, rn_stream s);

/* Input pipeline on the device: the data_aug_train chain of the reference's shipped configs (transform_util.py:36-205, order
 * config.yaml:6-14) for a whole batch in ONE launch: ToTensor (u8 HWC / 255) -> whitening ((x - mean) [/ stddev], per pixel and
 * channel, [C,H,W] images; stddev NULL = ZeroMeanWhiteningTransform) -> horizontal flip (per-sample byte) -> padding (pad pixels,
 * zero or mirror = F.pad reflect) -> crop (per-sample top / left, crop x crop).  The random draws are inputs.  Outputs (either
 * may be NULL): out_nchw fp32 [N,C,crop,crop] (what the reference's loader hands to classifier(x)) and out_nhwc [N,crop,crop,CP]
 * in `dtype` with the channels zero-padded to CP (the engine's stem input: skips rn_img_to_nhwc). */
int rn_augment_batch(const unsigned char* x_nhwc_u8, const float* mean_chw, const float* stddev_chw, const unsigned char* flip,
                     const int32_t* top, const int32_t* left, float* out_nchw, void* out_nhwc, int dtype, int N, int H, int W, int C,
                     int pad, int pad_mirror, int crop, int CP, rn_stream s);

/* fused multi-tensor SGD over one flat fp32 buffer (torch.optim.SGD rule; optim_util.py:11-18, config.yaml:22-28) */
int rn_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                float dampening, float weight_decay, int nesterov, int first_step, float grad_scale, rn_stream s);
/* GradScaler's gradient inspection over one flat buffer (training.py:104-110: scaler.step -> unscale_ / inf check): grads *= *inv_scale_dev
 * (skipped when it is 1: the check-only call) and *found_inf_dev = 1 if any element is inf or nan (never reset here) */
int rn_amp_check_unscale(float* grads, int64_t n, const float* inv_scale_dev, float* found_inf_dev, rn_stream s);
/* the same under AMP (GradScaler.step, training.py:104-110): gradients are divided by *loss_scale_dev and the whole step is skipped
 * when *found_inf_dev != 0 (both device scalars, either may be NULL) */
int rn_sgd_step_amp(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                    float weight_decay, int nesterov, int first_step, const float* loss_scale_dev, const float* found_inf_dev,
                    rn_stream s);

#ifdef __cplusplus
}
#endif
#endif
