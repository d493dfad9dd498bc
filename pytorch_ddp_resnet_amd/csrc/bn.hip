// BatchNorm2d (+ReLU, +Dropout, +residual) forward/backward as HBM-bound passes over an NHWC [M, C] view, gfx950.
// Reference call sites: resnet.py:111-115 (top-level n a), residual_block.py:58-65,70-72,76-77,81-87,96-98 (basic),
// :160-171,176-203,212-214 (bottleneck).  torch semantics kept: eps 1e-5, momentum 0.1, biased variance for
// normalisation, unbiased variance into running_var, num_batches_tracked += 1.
//
// Every kernel moves 16 bytes per lane per access (8 bf16 / 4 f32), consecutive lanes on consecutive channel chunks
// of one pixel row, so a wave reads/writes whole 1 KiB rows-of-rows; statistics are wave/LDS-reduced per workgroup
// into [nblk][2][C] partial sums and combined in double by the finalize kernels (no float atomics: bitwise stable).
#include "common.h"
#include <string.h>

namespace {

constexpr int NT = 256;

// ---- per-channel partial reductions over a slab of rows --------------------------------------------------
// MODE 0: (sum x, sum x^2)            MODE 1: (sum g, sum g*xhat), g = dout*gscale*[mask>0]
template <typename T, int MODE, int UNR = 4, int UM = -1>
__global__ __launch_bounds__(NT) void bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dout, const T* __restrict__ mask,
                                                       const float* __restrict__ coef, float* __restrict__ partial, int M, int C,
                                                       int rows_per_blk, int use_mask_rt, float gscale, uint32_t key, uint32_t thr) {
  const int use_mask = UM >= 0 ? UM : use_mask_rt;         // (UM: the mask mode at compile time -- the light form: the recompute path's coefficients and hash leave the registers)
  // use_mask: 0 none, 1 read `mask`, 2 recompute [x*scale+shift > 0] (and the dropout keep hash when thr != 0)
  constexpr int CE = Elem<T>::CE;
  __shared__ float red[2][NT][CE + 1];
  const int CC = C / CE;
  const int tid = threadIdx.x;
  const int r_begin = blockIdx.x * rows_per_blk;
  const int r_end = min(M, r_begin + rows_per_blk);
  const int lanes = CC >= NT ? 1 : NT / CC;        // row lanes per chunk column
  const int cols_per_pass = CC >= NT ? NT : CC;
  for (int cbase = 0; cbase < CC; cbase += cols_per_pass) {
    const int cg = cbase + (tid % cols_per_pass);
    const int rl = tid / cols_per_pass;
    float s0[CE], s1[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) s0[e] = s1[e] = 0.f;
    const bool active = rl < lanes && cg < CC;
    if (active) {
      float mean[CE], invstd[CE], sc[CE], sh[CE];
      if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          mean[e] = coef[2 * C + cg * CE + e]; invstd[e] = coef[3 * C + cg * CE + e];
          sc[e] = coef[cg * CE + e]; sh[e] = coef[C + cg * CE + e];
        }
      }
#pragma unroll UNR
      for (int r = r_begin + rl; r < r_end; r += lanes) {
        const size_t off = (size_t)r * C + (size_t)cg * CE;
        Chunk<T> cx = load_chunk<T>(x + off);
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < CE; ++e) { float v = Elem<T>::to_f(cx.e[e]); s0[e] += v; s1[e] += v * v; }
        } else {
          Chunk<T> cd = load_chunk<T>(dout + off);
          Chunk<T> cm;
          if (use_mask == 1) cm = load_chunk<T>(mask + off);
          const uint32_t kp = (use_mask == 2 && thr) ? rn_keep_chunk<CE>(key, (uint32_t)off, thr) : 0xFFFFFFFFu;
#pragma unroll
          for (int e = 0; e < CE; ++e) {
            float g = Elem<T>::to_f(cd.e[e]) * gscale;
            const float xv = Elem<T>::to_f(cx.e[e]);
            if (use_mask == 1 && !(Elem<T>::to_f(cm.e[e]) > 0.f)) g = 0.f;
            if (use_mask == 2 && (!(fmaf(xv, sc[e], sh[e]) > 0.f) || !((kp >> e) & 1u))) g = 0.f;
            float xh = (xv - mean[e]) * invstd[e];
            s0[e] += g; s1[e] += g * xh;
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < CE; ++e) { red[0][tid][e] = s0[e]; red[1][tid][e] = s1[e]; }
    __syncthreads();
    if (tid < cols_per_pass && cg < CC) {
      float t0[CE], t1[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) t0[e] = t1[e] = 0.f;
      for (int l = 0; l < lanes; ++l) {
        const int src = l * cols_per_pass + tid;
#pragma unroll
        for (int e = 0; e < CE; ++e) { t0[e] += red[0][src][e]; t1[e] += red[1][src][e]; }
      }
      float* p0 = partial + ((size_t)blockIdx.x * 2 + 0) * C + cg * CE;
      float* p1 = partial + ((size_t)blockIdx.x * 2 + 1) * C + cg * CE;
#pragma unroll
      for (int e = 0; e < CE; ++e) { p0[e] = t0[e]; p1[e] = t1[e]; }
    }
    __syncthreads();
  }
}

// The backward's BatchNorm kernels in forms that fit BESIDE the forked weight gradients' persistent workgroups (three waves of 136 registers per SIMD leave
// 104 of 512): 256-thread finalize workgroups, two rows of loads in flight in the apply pass.  The plan executor switches it on for ranges that fork
// (rn_bn_side_friendly); RN_BN_LIGHT=0 keeps the wide forms (A/B).  Measured on one box, WRN-28-10: 6.131 / 6.128 / 6.122 -> 5.974 / 5.979 / 6.002 ms per step.
static thread_local int t_bn_side_friendly = 0;
inline bool bn_light() {
  static const bool off = getenv("RN_BN_LIGHT") && atoi(getenv("RN_BN_LIGHT")) == 0;
  return t_bn_side_friendly && !off;
}

// reduces [nblk][2][C] partials in double: 16 channels x 64 slab-lanes per workgroup (the fused conv epilogues write one
// partial row per 128 output pixels: up to 1024 rows)
constexpr int FC = 16, FL = 64;
template <int FLT = FL>
__device__ inline void reduce_partials(const float* __restrict__ partial, int nblk, int C, int c, int bl, double& s, double& ss, double (*red)[FLT][FC], int b_lo = 0) {
  constexpr int FL = FLT;                                  // (shadows the namespace constant: the row lanes of THIS instantiation)
  // fixed-order tree (bitwise reproducible): rows b = bl, bl+64, ... per thread; the 4 row lanes of a wave by shuffles; the 16
  // waves through LDS, 4 per lane group of wave 0, then shuffles again.  The result is valid in threads 0..15 (bl == 0).
  // (a serial 64-step LDS loop here cost ~2.8 us of a 6 us kernel)
  // a thread's rows in row order (bitwise reproducible), the loads of eight rows issued together: one row at a time, every row was a
  // memory round trip of its own (16 rows per thread on the 16-channel layers of ResNet-v2-164: 8.1 us for a one-workgroup launch)
  s = 0.0; ss = 0.0;
  if (c < C) {
    int b = b_lo + bl;                                     // rows [b_lo, nblk)
    for (; b + 7 * FL < nblk; b += 8 * FL) {
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v0[u] = partial[((size_t)(b + u * FL) * 2) * C + c]; v1[u] = partial[((size_t)(b + u * FL) * 2 + 1) * C + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += (double)v0[u]; ss += (double)v1[u]; }
    }
    if (b < nblk) {
      // the remaining 1-7 rows of this thread in ONE round trip: every load issued (a row past the end re-reads the last row and counts as zero -- a
      // branch around a load would make each one a round trip of its own).  Pairs and singles here took 3-4 dependent trips on the 98-392-row layers
      // (19-24 us per launch for 1.6-6.4 MB).  Same rows, same order of adds as before.
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int row = min(b + u * FL, nblk - 1);
        v0[u] = partial[((size_t)row * 2) * C + c]; v1[u] = partial[((size_t)row * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const bool in = b + u * FL < nblk;
        s += in ? (double)v0[u] : 0.0; ss += in ? (double)v1[u] : 0.0;
      }
    }
  }
  static_assert(FC == 16 && (FL == 64 || FL == 16), "reduction tree is written for 16 channels x 64 (or 16) row lanes");
  const int tid = threadIdx.x, cl = tid % FC, wave = tid >> 6, lane = tid & 63;
  s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
  s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
  if (lane < FC) { red[0][wave][cl] = s; red[1][wave][cl] = ss; }
  __syncthreads();
  if (wave == 0) {
    const int g = lane >> 4;
    s = 0.0; ss = 0.0;
#pragma unroll
    for (int w = 0; w < FL / 16; ++w) { s += red[0][g + 4 * w][cl]; ss += red[1][g + 4 * w][cl]; }
    s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
    s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
  }
}

// FLT = 16 (256-thread workgroups, a quarter of the registers per SIMD): the form launched beside forked weight gradients, whose persistent workgroups leave
// ~100 registers per SIMD lane free -- a 1,024-thread workgroup does not fit beside them and waited for a whole CU
template <int FLT>
__global__ __launch_bounds__(FC * FLT) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   long long* __restrict__ nbt, float* __restrict__ coef, int C, float eps, float momentum, int train) {
  __shared__ double red[2][FLT][FC];
  const int c = blockIdx.x * FC + threadIdx.x % FC, bl = threadIdx.x / FC;
  if (blockIdx.x == 0 && threadIdx.x == 0 && train && nbt) nbt[0] += 1;
  double mean = 0.0, var = 0.0;
  if (train) {
    double s, ss;
    reduce_partials<FLT>(partial, nblk, C, c, bl, s, ss, red);
    if (bl != 0 || c >= C) return;
    mean = s / count;
    var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
    rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
    rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbiased);
  } else {
    if (bl != 0 || c >= C) return;
    mean = (double)rmean[c];
    var = (double)rvar[c];
  }
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double scale = (double)gamma[c] * invstd;
  coef[c] = (float)scale;
  coef[C + c] = (float)((double)beta[c] - mean * scale);
  coef[2 * C + c] = (float)mean;
  coef[3 * C + c] = (float)invstd;
}

template <int FLT>
__global__ __launch_bounds__(FC * FLT) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dsum, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int C, int accum) {
  __shared__ double red[2][FLT][FC];
  const int c = blockIdx.x * FC + threadIdx.x % FC, bl = threadIdx.x / FC;
  double s, ss;
  reduce_partials<FLT>(partial, nblk, C, c, bl, s, ss, red);
  if (bl != 0 || c >= C) return;
  dsum[c] = (float)s;
  dsum[C + c] = (float)ss;
  if (accum) { dbeta[c] += (float)s; dgamma[c] += (float)ss; }
  else { dbeta[c] = (float)s; dgamma[c] = (float)ss; }
}

// ---- the same two kernels with the ROWS split over workgroups.  One workgroup per 16 channels is 8-128 workgroups; on the ImageNet nets the fused
// conv epilogues leave 1,568-25,088 rows per layer (WRN-50-2-B: 780 MB of partial sums per step), which those few workgroups read at ~1.5 TB/s
// (17 us per launch, 1.7 ms per step).  Here workgroup (channel group cb, split sp) sums rows [sp, sp + 1) * nblk / S as above, publishes its 2 x 16
// doubles and draws a ticket; the LAST arriver of a channel group adds the S parts in split order (fixed order: bitwise reproducible, whoever arrives
// last) and does the finalize arithmetic.  Nobody waits: no spin, no residency assumption.  fold = [counters: ncb ints, padded to 64 B][cb][sp][2][16] doubles,
// caller-owned, zero before the first launch (the last arriver leaves its counter at zero).  Hand-off as in conv_igemm8.hip: write-through (sc1) stores,
// drained by the storing wave before its lane 0 draws the ticket; the finisher acquires once and reads the parts with sc1 loads.
__device__ inline __amdgpu_buffer_rsrc_t fold_rsrc(void* fold) { return __builtin_amdgcn_make_buffer_rsrc(fold, 0, 0x7FFFFFF0, 0x00020000); }
__host__ __device__ inline int fold_header_bytes(int ncb) { return ((ncb * 4 + 63) / 64) * 64; }

// returns true in EVERY thread of the one workgroup per channel group that finishes; then s / ss hold the totals in threads 0..15
__device__ inline bool fold_rows(const float* __restrict__ partial, int nblk, int C, void* fold, int S, double& s, double& ss, double (*red)[FL][FC], int* flag) {
  const int ncb = (C + FC - 1) / FC;
  const int cb = blockIdx.x % ncb, sp = blockIdx.x / ncb;  // neighbouring workgroups: neighbouring channel groups of the same rows (the two halves of a 128-byte line)
  const int c = cb * FC + threadIdx.x % FC, bl = threadIdx.x / FC;
  const int b_lo = (int)((long)sp * nblk / S), b_hi = (int)((long)(sp + 1) * nblk / S);
  reduce_partials(partial, b_hi, C, c, bl, s, ss, red, b_lo);
  const __amdgpu_buffer_rsrc_t r = fold_rsrc(fold);
  const int base = fold_header_bytes(ncb) + ((cb * S + sp) * 2) * FC * 8;
  typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
  if (threadIdx.x < FC) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, s), r, base + (int)threadIdx.x * 8, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, ss), r, base + (FC + (int)threadIdx.x) * 8, 0, 16);
  }
  if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the storing wave drains; its lane 0 signals
  int* cnt = reinterpret_cast<int*>(fold) + cb;
  if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*flag != S - 1) return false;                        // workgroup-uniform
  if (threadIdx.x == 0) {
    __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // every part has arrived: clean for the next launch
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  s = 0.0; ss = 0.0;
  if (threadIdx.x < FC) {
    const int first = fold_header_bytes(ncb) + (cb * S * 2) * FC * 8 + (int)threadIdx.x * 8;
    int q = 0;
    for (; q + 4 <= S; q += 4) {                           // four parts in flight, added in split order
      v2u32 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = __builtin_amdgcn_raw_buffer_load_b64(r, first + (q + u) * 2 * FC * 8, 0, 16);
        b[u] = __builtin_amdgcn_raw_buffer_load_b64(r, first + ((q + u) * 2 + 1) * FC * 8, 0, 16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s += __builtin_bit_cast(double, a[u]); ss += __builtin_bit_cast(double, b[u]); }
    }
    for (; q < S; ++q) {
      s += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, first + q * 2 * FC * 8, 0, 16));
      ss += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, first + (q * 2 + 1) * FC * 8, 0, 16));
    }
  }
  return true;
}

__global__ __launch_bounds__(FC * FL) void bn_finalize_split_kernel(const float* __restrict__ partial, int nblk, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   long long* __restrict__ nbt, float* __restrict__ coef, int C, float eps, float momentum, void* fold, int S) {
  __shared__ double red[2][FL][FC];
  __shared__ int flag;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
  double s, ss;
  if (!fold_rows(partial, nblk, C, fold, S, s, ss, red, &flag)) return;
  const int c = (blockIdx.x % ((C + FC - 1) / FC)) * FC + threadIdx.x;
  if (threadIdx.x >= FC || c >= C) return;
  const double mean = s / count;
  double var = ss / count - mean * mean;                  // the arithmetic of bn_finalize_kernel
  if (var < 0.0) var = 0.0;
  const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
  rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
  rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbiased);
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double scale = (double)gamma[c] * invstd;
  coef[c] = (float)scale;
  coef[C + c] = (float)((double)beta[c] - mean * scale);
  coef[2 * C + c] = (float)mean;
  coef[3 * C + c] = (float)invstd;
}

__global__ __launch_bounds__(FC * FL) void bn_bwd_finalize_split_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dsum, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int C, int accum, void* fold, int S) {
  __shared__ double red[2][FL][FC];
  __shared__ int flag;
  double s, ss;
  if (!fold_rows(partial, nblk, C, fold, S, s, ss, red, &flag)) return;
  const int c = (blockIdx.x % ((C + FC - 1) / FC)) * FC + threadIdx.x;
  if (threadIdx.x >= FC || c >= C) return;
  dsum[c] = (float)s;
  dsum[C + c] = (float)ss;
  if (accum) { dbeta[c] += (float)s; dgamma[c] += (float)ss; }
  else { dbeta[c] = (float)s; dgamma[c] = (float)ss; }
}

// rows split S ways when one workgroup per 16 channels leaves the chip idle: >= 256 rows per part, ~1024 workgroups at most.  Measured
// (tools/finalize_bench.py, us per launch plain -> split): 25.7 MB of partials 29.9 -> 17.7, 33.6 MB 37.1 -> 21.4, 102.8 MB 103 -> 38, 6.4 MB in 8 channel
// groups 14.4 -> 11.0; but 3.2 MB 7.6 -> 9.7, 8.4 MB 8.5 -> 9.5, 12.8 MB 10.9 -> 10.6 (the ticket and the finisher cost ~3 us): split from 16 MB, or from 4 MB
// when the plain kernel would be 8 workgroups or fewer.
static int fold_splits(int nblk, int C) {
  const int ncb = cdiv(C, FC);
  const long bytes = (long)nblk * 2 * C * 4;
  if (nblk < 512 || !(bytes >= (16l << 20) || (ncb <= 8 && bytes >= (4l << 20)))) return 1;
  int S = std::min(nblk / 256, std::max(1, 1024 / ncb));
  S = std::min(S, 64);
  return S < 2 ? 1 : S;
}
extern "C" size_t rn_bn_fold_bytes(int nblk, int C) {
  if (nblk <= 0 || C <= 0) return 0;
  const int S = fold_splits(nblk, C);
  if (S == 1) return 0;
  const int ncb = cdiv(C, FC);
  return (size_t)fold_header_bytes(ncb) + (size_t)ncb * S * 2 * FC * 8;
}

// ---- elementwise passes.  A thread owns ONE 16-byte channel chunk (its per-channel coefficients live in registers for
// the whole launch) and streams the rows of its workgroup's slab; consecutive lanes sit on consecutive chunks of a row.
struct Slab {
  int cols, lanes, cg, rl, r_begin, r_end;
  bool active;
};
__device__ inline Slab make_slab(int CC, int M, int rows_per_blk, int cbase) {
  Slab s;
  s.cols = CC >= NT ? NT : CC;
  s.lanes = CC >= NT ? 1 : NT / CC;
  s.cg = cbase + (int)(threadIdx.x % s.cols);
  s.rl = threadIdx.x / s.cols;
  s.r_begin = blockIdx.x * rows_per_blk;
  s.r_end = min(M, s.r_begin + rows_per_blk);
  s.active = s.rl < s.lanes && s.cg < CC;
  return s;
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ coef, ResDesc res, T* __restrict__ out,
                                                      int M, int H, int W, int C, int rows_per_blk, int relu, float inv_keep, uint32_t key, uint32_t thr) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  for (int cbase = 0; cbase < CC; cbase += NT) {
    const Slab s = make_slab(CC, M, rows_per_blk, cbase);
    if (!s.active) continue;
    const int c0 = s.cg * CE;
    float sc[CE], sh[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { sc[e] = coef[c0 + e]; sh[e] = coef[C + c0 + e]; }
    const bool res_same = res.mode == RN_RES_SAME;
    auto one = [&](int r, size_t off, const Chunk<T>& cx, const Chunk<T>& cr) {
      float v[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) v[e] = fmaf(Elem<T>::to_f(cx.e[e]), sc[e], sh[e]);
      if (res_same) {
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] += Elem<T>::to_f(cr.e[e]);
      } else if (res.mode != RN_RES_NONE) {
        const int hw = H * W;
        const int n = r / hw, rem = r - n * hw;
        const int h = rem / W, w = rem - h * W;
        res_add_chunk<T>(res, n, h, w, c0, v);
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (thr) {
        const uint32_t kp = rn_keep_chunk<CE>(key, (uint32_t)off, thr);
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] = ((kp >> e) & 1u) ? v[e] * inv_keep : 0.f;
      }
      Chunk<T> co;
#pragma unroll
      for (int e = 0; e < CE; ++e) co.e[e] = Elem<T>::from_f(v[e]);
      store_chunk<T>(out + off, co);
    };
    // four rows of loads in flight (a thread streams >= 4 rows: slab_rows): row by row, every row was a memory round trip of its own
    int r = s.r_begin + s.rl;
    for (; r + 3 * s.lanes < s.r_end; r += 4 * s.lanes) {
      Chunk<T> cx[4], cr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t off = (size_t)(r + u * s.lanes) * C + c0;
        cx[u] = load_chunk<T>(x + off);
        if (res_same) cr[u] = load_chunk<T>(reinterpret_cast<const T*>(res.ptr) + off);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r + u * s.lanes, (size_t)(r + u * s.lanes) * C + c0, cx[u], cr[u]);
    }
    for (; r < s.r_end; r += s.lanes) {
      const size_t off = (size_t)r * C + c0;
      Chunk<T> cx = load_chunk<T>(x + off), cr;
      if (res_same) cr = load_chunk<T>(reinterpret_cast<const T*>(res.ptr) + off);
      one(r, off, cx, cr);
    }
  }
}

// dx = a*g + b*x + c per channel:  a = scale, b = -scale*invstd*m1, c = scale*(invstd*m1*mean - m0), m = dsum/count
// Streaming form of the backward apply.  MASK / ADD are compile-time (no branch inside the row loop, so the loads of the
// UNR unrolled rows are issued together), the per-channel coefficients are fetched as 16-byte vectors, and a thread
// streams >= 4 rows (slab_rows_stream) so that fetch amortises.
template <typename T, int MASK, int ADD, int UNR>
__global__ __launch_bounds__(NT) void bn_bwd_apply_stream_kernel(const T* __restrict__ dout, const T* __restrict__ x, const T* __restrict__ mask,
                                                                 const float* __restrict__ coef, const float* __restrict__ dsum, ResDesc add,
                                                                 T* __restrict__ dx, T* __restrict__ g_out, int M, int H, int W, int C, int rows_per_blk,
                                                                 int train, float gscale, float inv_count, uint32_t key, uint32_t thr) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  for (int cbase = 0; cbase < CC; cbase += NT) {
    const Slab s = make_slab(CC, M, rows_per_blk, cbase);
    if (!s.active) continue;
    const int c0 = s.cg * CE;
    float ka[CE], kb[CE], kc[CE], sc[CE], sh[CE];
#pragma unroll
    for (int e = 0; e < CE; e += 4) {
      const float4 vs = *reinterpret_cast<const float4*>(coef + c0 + e), vh = *reinterpret_cast<const float4*>(coef + C + c0 + e);
      float4 vm = make_float4(0.f, 0.f, 0.f, 0.f), vi = vm, d0 = vm, d1 = vm;
      if (train) {
        vm = *reinterpret_cast<const float4*>(coef + 2 * C + c0 + e); vi = *reinterpret_cast<const float4*>(coef + 3 * C + c0 + e);
        d0 = *reinterpret_cast<const float4*>(dsum + c0 + e); d1 = *reinterpret_cast<const float4*>(dsum + C + c0 + e);
      }
      const float a_s[4] = {vs.x, vs.y, vs.z, vs.w}, a_h[4] = {vh.x, vh.y, vh.z, vh.w}, a_m[4] = {vm.x, vm.y, vm.z, vm.w};
      const float a_i[4] = {vi.x, vi.y, vi.z, vi.w}, a_0[4] = {d0.x, d0.y, d0.z, d0.w}, a_1[4] = {d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // explicit roundings (no compiler-chosen contraction): every instantiation derives bit-identical coefficients
        const float scale = a_s[q];
        sc[e + q] = scale; sh[e + q] = a_h[q];
        ka[e + q] = __fmul_rn(scale, gscale);
        const float m0 = __fmul_rn(a_0[q], inv_count), m1 = __fmul_rn(a_1[q], inv_count);
        const float im1 = __fmul_rn(a_i[q], m1);
        kb[e + q] = train ? -__fmul_rn(scale, im1) : 0.f;
        kc[e + q] = train ? __fmul_rn(scale, __fmaf_rn(im1, a_m[q], -m0)) : 0.f;
      }
    }
    const int hw = H * W;
    auto one = [&](size_t off, int r, const Chunk<T>& cd, const Chunk<T>& cx, const Chunk<T>& cm, const Chunk<T>& cr) {
      float v[CE], g[CE];
      const uint32_t kp = (MASK == 2 && thr) ? rn_keep_chunk<CE>(key, (uint32_t)off, thr) : 0xFFFFFFFFu;
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        float gg = Elem<T>::to_f(cd.e[e]);
        const float xv = Elem<T>::to_f(cx.e[e]);
        if (MASK == 1 && !(Elem<T>::to_f(cm.e[e]) > 0.f)) gg = 0.f;
        if (MASK == 2 && (!(fmaf(xv, sc[e], sh[e]) > 0.f) || !((kp >> e) & 1u))) gg = 0.f;
        g[e] = __fmul_rn(gg, gscale);
        v[e] = __fmaf_rn(ka[e], gg, __fmaf_rn(kb[e], xv, kc[e]));
      }
      if (ADD == 1) {
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] += Elem<T>::to_f(cr.e[e]);
      } else if (ADD == 2) {
        const int n = r / hw, rem = r - n * hw;
        const int h = rem / W, w = rem - h * W;
        res_add_chunk<T>(add, n, h, w, c0, v);
      }
      Chunk<T> co;
#pragma unroll
      for (int e = 0; e < CE; ++e) co.e[e] = Elem<T>::from_f(v[e]);
      store_chunk<T>(dx + off, co);
      if (g_out) {
#pragma unroll
        for (int e = 0; e < CE; ++e) co.e[e] = Elem<T>::from_f(g[e]);
        store_chunk<T>(g_out + off, co);
      }
    };
    int r = s.r_begin + s.rl;
    for (; r + (UNR - 1) * s.lanes < s.r_end; r += UNR * s.lanes) {
      Chunk<T> cd[UNR], cx[UNR], cm[UNR], cr[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const size_t off = (size_t)(r + u * s.lanes) * C + c0;
        cd[u] = load_chunk<T>(dout + off);
        cx[u] = load_chunk<T>(x + off);
        if (MASK == 1) cm[u] = load_chunk<T>(mask + off);
        if (ADD == 1) cr[u] = load_chunk<T>(reinterpret_cast<const T*>(add.ptr) + off);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) one((size_t)(r + u * s.lanes) * C + c0, r + u * s.lanes, cd[u], cx[u], cm[u], cr[u]);
    }
    for (; r < s.r_end; r += s.lanes) {
      const size_t off = (size_t)r * C + c0;
      Chunk<T> cd = load_chunk<T>(dout + off), cx = load_chunk<T>(x + off), cm, cr;
      if (MASK == 1) cm = load_chunk<T>(mask + off);
      if (ADD == 1) cr = load_chunk<T>(reinterpret_cast<const T*>(add.ptr) + off);
      one(off, r, cd, cx, cm, cr);
    }
  }
}


template <typename T>
__global__ __launch_bounds__(NT) void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, long nchunks, float inv_keep, uint32_t key,
                                                         uint32_t thr) {
  constexpr int CE = Elem<T>::CE;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < nchunks; i += (long)gridDim.x * NT) {
    Chunk<T> c = load_chunk<T>(x + i * CE);
#pragma unroll
    for (int e = 0; e < CE; ++e) c.e[e] = Elem<T>::from_f(rn_keep(key, (uint32_t)(i * CE + e), thr) ? Elem<T>::to_f(c.e[e]) * inv_keep : 0.f);
    store_chunk<T>(out + i * CE, c);
  }
}

// the keep mask is RECOMPUTED from the counter hash (site, step seed, element index), never inferred from the forward output: a kept
// element whose input was exactly 0 (a stem without ReLU in front, fp16 underflow) must still pass its gradient
template <typename T>
__global__ __launch_bounds__(NT) void dropout_bwd_kernel(const T* __restrict__ dout, T* __restrict__ din, long nchunks, float inv_keep, uint32_t key,
                                                         uint32_t thr) {
  constexpr int CE = Elem<T>::CE;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < nchunks; i += (long)gridDim.x * NT) {
    Chunk<T> d = load_chunk<T>(dout + i * CE);
#pragma unroll
    for (int e = 0; e < CE; ++e) d.e[e] = Elem<T>::from_f(rn_keep(key, (uint32_t)(i * CE + e), thr) ? Elem<T>::to_f(d.e[e]) * inv_keep : 0.f);
    store_chunk<T>(din + i * CE, d);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void add_res_kernel(T* __restrict__ dst, ResDesc res, int H, int W, int C, long nchunks) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < nchunks; i += (long)gridDim.x * NT) {
    const long pix = i / CC;
    const int c0 = (int)(i - pix * CC) * CE;
    Chunk<T> c = load_chunk<T>(dst + i * CE);
    float v[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) v[e] = Elem<T>::to_f(c.e[e]);
    const int hw = H * W;
    const int n = (int)(pix / hw), rem = (int)(pix - (long)n * hw);
    const int h = rem / W, w = rem - h * W;
    res_add_chunk<T>(res, n, h, w, c0, v);
#pragma unroll
    for (int e = 0; e < CE; ++e) c.e[e] = Elem<T>::from_f(v[e]);
    store_chunk<T>(dst + i * CE, c);
  }
}

// rows of a slab pass per workgroup: every thread should see >= 4 rows so its coefficient loads amortise
inline int slab_rows(long M, int C, int ce) {
  const int CC = C / ce;
  const int lanes = CC >= NT ? 1 : NT / CC;
  long blocks = (M + (long)lanes * 4 - 1) / ((long)lanes * 4);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  long rows = (M + blocks - 1) / blocks;
  rows = (rows + lanes - 1) / lanes * lanes;
  return (int)rows;
}

// streaming passes: 4..16 rows per thread: one batch of four rows of loads on small tensors (thin networks: a second batch is a second
// memory round trip on a 6 us kernel; ResNet-v2-164 +2.3 % against a floor of 8), up to 16 when that still leaves >= 1024 workgroups
inline int slab_rows_stream(long M, int C, int ce) {
  const int CC = C / ce;
  const int lanes = CC >= NT ? 1 : NT / CC;
  long rpt = M / ((long)lanes * 1024);
  if (rpt > 16) rpt = 16;
  if (rpt < 4) rpt = 4;
  long rows = (long)lanes * rpt;
  if (rows > M) rows = (M + lanes - 1) / lanes * lanes;
  return (int)rows;
}

template <typename T, int MASK, int ADD>
void launch_bwd_apply_stream(int grid, hipStream_t st, const void* dout, const void* x, const void* mask, const float* coef, const float* dsum, const ResDesc& r,
                             void* dx, void* g, int M, int H, int W, int C, int rows, int train, float gscale, float inv_count, uint32_t key, uint32_t thr) {
  if constexpr (sizeof(T) == 2) {
    if (bn_light()) {       // two rows of loads in flight instead of four: <= 104 registers, a wave per SIMD fits beside a forked weight gradient's three
      hipLaunchKernelGGL((bn_bwd_apply_stream_kernel<T, MASK, ADD, 2>), dim3(grid), dim3(NT), 0, st, (const T*)dout, (const T*)x, (const T*)mask, coef, dsum, r, (T*)dx, (T*)g,
                         M, H, W, C, rows, train, gscale, inv_count, key, thr);
      return;
    }
  }
  hipLaunchKernelGGL((bn_bwd_apply_stream_kernel<T, MASK, ADD, 4>), dim3(grid), dim3(NT), 0, st, (const T*)dout, (const T*)x, (const T*)mask, coef, dsum, r, (T*)dx, (T*)g,
                     M, H, W, C, rows, train, gscale, inv_count, key, thr);
}
template <typename T>
void dispatch_bwd_apply_stream(int use_mask, int add_kind, int grid, hipStream_t st, const void* dout, const void* x, const void* mask, const float* coef,
                               const float* dsum, const ResDesc& r, void* dx, void* g, int M, int H, int W, int C, int rows, int train, float gscale,
                               float inv_count, uint32_t key, uint32_t thr) {
#define BA_CASE(MK, AD) if (use_mask == MK && add_kind == AD) return launch_bwd_apply_stream<T, MK, AD>(grid, st, dout, x, mask, coef, dsum, r, dx, g, M, H, W, C, rows, train, gscale, inv_count, key, thr);
  BA_CASE(0, 0) BA_CASE(0, 1) BA_CASE(0, 2) BA_CASE(1, 0) BA_CASE(1, 1) BA_CASE(1, 2) BA_CASE(2, 0) BA_CASE(2, 1) BA_CASE(2, 2)
#undef BA_CASE
}

inline int ew_grid(long nchunks) {
  long b = (nchunks + NT - 1) / NT;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

inline void make_res(ResDesc& r, const void* p, int mode, int res_C, int H, int W, int C) {
  r.ptr = p;
  r.mode = p ? mode : RN_RES_NONE;
  if (r.mode == RN_RES_SAME) { r.C = C; r.H = H; r.W = W; }
  else if (r.mode == RN_RES_DOWN2PAD) { r.C = res_C; r.H = 2 * H; r.W = 2 * W; }
  else if (r.mode == RN_RES_UP2) { r.C = res_C; r.H = (H + 1) / 2; r.W = (W + 1) / 2; }
  else { r.C = r.H = r.W = 0; }
}

inline int check_mc(int dtype, long M, int C, const char* who) {
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "%s: bad dtype %d", who, dtype);
  RN_CHECK_ARG(M > 0 && C > 0 && C % (dtype == RN_F32 ? 4 : 8) == 0, "%s: bad shape M=%ld C=%d", who, M, C);
  return 0;
}

}  // namespace

extern "C" int rn_bn_stats(const void* x, float* partial, int nblk, int dtype, int64_t M, int C, rn_stream s) {
  if (int e = check_mc(dtype, M, C, "rn_bn_stats")) return e;
  RN_CHECK_ARG(x && partial && nblk > 0 && M < (1L << 31), "rn_bn_stats: bad argument");
  const int rows = (int)((M + nblk - 1) / nblk);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_reduce_kernel<T_, 0>), dim3(nblk), dim3(NT), 0, as_stream(s), (const T_*)x, nullptr, nullptr, nullptr, partial, (int)M, C, rows, 0, 1.f, 0u, 0u));
  RN_CHECK_LAUNCH("bn_stats");
  return 0;
}

extern "C" int rn_bn_finalize(const float* partial, int nblk, double count, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, int64_t* nbt, float* coef, int C, float eps, float momentum, int flags, rn_stream s) {
  const int train = (flags & RN_F_TRAIN) ? 1 : 0;
  RN_CHECK_ARG(gamma && beta && running_mean && running_var && coef && C > 0, "rn_bn_finalize: null pointer");
  RN_CHECK_ARG(!train || (partial && nblk > 0 && count > 0), "rn_bn_finalize: train mode needs partial sums");
  hipLaunchKernelGGL(bn_finalize_kernel<FL>, dim3(cdiv(C, FC)), dim3(FC * FL), 0, as_stream(s), partial, nblk, count, gamma, beta, running_mean, running_var,
                     (long long*)nbt, coef, C, eps, momentum, train);
  RN_CHECK_LAUNCH("bn_finalize");
  return 0;
}

extern "C" int rn_bn_finalize_split(const float* partial, int nblk, double count, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, int64_t* nbt, float* coef, int C, float eps, float momentum, int flags, void* fold, size_t fold_bytes,
                                    rn_stream s) {
  RN_CHECK_ARG(gamma && beta && running_mean && running_var && coef && C > 0, "rn_bn_finalize_split: null pointer");
  RN_CHECK_ARG((flags & RN_F_TRAIN) && partial && nblk > 0 && count > 0, "rn_bn_finalize_split: training mode with partial sums only");
  const size_t need = rn_bn_fold_bytes(nblk, C);
  RN_CHECK_ARG(need > 0, "rn_bn_finalize_split: %d rows x %d channels is not split (rn_bn_fold_bytes says 0): call rn_bn_finalize", nblk, C);
  RN_CHECK_ARG(fold && fold_bytes >= need, "rn_bn_finalize_split: fold buffer of %zu bytes, %zu needed", fold_bytes, need);
  const int S = fold_splits(nblk, C);
  hipLaunchKernelGGL(bn_finalize_split_kernel, dim3(cdiv(C, FC) * S), dim3(FC * FL), 0, as_stream(s), partial, nblk, count, gamma, beta, running_mean, running_var,
                     (long long*)nbt, coef, C, eps, momentum, fold, S);
  RN_CHECK_LAUNCH("bn_finalize_split");
  return 0;
}

extern "C" int rn_bn_apply(const void* x, const float* coef, const void* res, void* out, int dtype, int N, int H, int W, int C, int res_mode,
                           int res_C, int flags, float drop_p, uint32_t site, uint64_t step_seed, rn_stream s) {
  const long M = (long)N * H * W;
  if (int e = check_mc(dtype, M, C, "rn_bn_apply")) return e;
  RN_CHECK_ARG(x && coef && out && drop_p >= 0.f && drop_p < 1.f, "rn_bn_apply: bad argument");
  ResDesc r;
  make_res(r, res, res_mode, res_C, H, W, C);
  const int ce = dtype == RN_F32 ? 4 : 8;
  const long nchunks = M * (C / ce);
  RN_CHECK_ARG(drop_p == 0.f || M * C < (1L << 32), "rn_bn_apply: dropout index space exceeds 2^32 elements");
  const uint32_t thr = drop_p > 0.f ? rn_drop_threshold(drop_p) : 0u;
  const uint32_t key = rn_drop_key(site, step_seed);
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int relu = (flags & RN_F_RELU) ? 1 : 0;
  RN_CHECK_ARG(M < (1L << 31), "rn_bn_apply: too many pixels");
  (void)nchunks;
  // (measured, round 4: 10 rows per thread = one round of 1,092 workgroups instead of 4 rows = 1.33 rounds of 2,731: 0.343 vs 0.322 ms per step -- the short form stays)
  const int rows = slab_rows(M, C, ce);
  const int grid = cdiv(M, rows);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_kernel<T_>), dim3(grid), dim3(NT), 0, as_stream(s), (const T_*)x, coef, r, (T_*)out, (int)M, H, W, C, rows, relu, inv_keep, key, thr));
  RN_CHECK_LAUNCH("bn_apply");
  return 0;
}

static inline int mask_mode(int flags) { return (flags & RN_F_MASK_RECOMPUTE) ? 2 : ((flags & RN_F_RELU) ? 1 : 0); }

extern "C" int rn_bn_bwd_reduce(const void* dout, const void* x, const void* mask_src, const float* coef, float* partial, int nblk, int dtype,
                                int64_t M, int C, int flags, float gscale, float drop_p, uint32_t site, uint64_t step_seed, rn_stream s) {
  if (int e = check_mc(dtype, M, C, "rn_bn_bwd_reduce")) return e;
  const int use_mask = mask_mode(flags);
  RN_CHECK_ARG(dout && x && coef && partial && nblk > 0 && (use_mask != 1 || mask_src) && M < (1L << 31), "rn_bn_bwd_reduce: bad argument");
  RN_CHECK_ARG(drop_p == 0.f || (use_mask == 2 && M * C < (1L << 32)), "rn_bn_bwd_reduce: dropout recompute needs RN_F_MASK_RECOMPUTE and < 2^32 elements");
  const uint32_t thr = drop_p > 0.f ? rn_drop_threshold(drop_p) : 0u, key = rn_drop_key(site, step_seed);
  const int rows = (int)((M + nblk - 1) / nblk);
  if (bn_light() && use_mask == 1) {      // two rows in flight, the mask mode fixed: fits beside the forked weight gradients
    RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_reduce_kernel<T_, 1, 2, 1>), dim3(nblk), dim3(NT), 0, as_stream(s), (const T_*)x, (const T_*)dout, (const T_*)mask_src, coef, partial, (int)M, C, rows, use_mask, gscale, key, thr));
  } else {
    RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_reduce_kernel<T_, 1>), dim3(nblk), dim3(NT), 0, as_stream(s), (const T_*)x, (const T_*)dout, (const T_*)mask_src, coef, partial, (int)M, C, rows, use_mask, gscale, key, thr));
  }
  RN_CHECK_LAUNCH("bn_bwd_reduce");
  return 0;
}

extern "C" void rn_bn_side_friendly(int on) { t_bn_side_friendly = on; }

extern "C" int rn_bn_bwd_finalize(const float* partial, int nblk, float* dsum, float* dgamma, float* dbeta, int C, int flags, rn_stream s) {
  RN_CHECK_ARG(partial && dsum && dgamma && dbeta && nblk > 0 && C > 0, "rn_bn_bwd_finalize: bad argument");
  if (bn_light() && nblk <= 1024) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3(cdiv(C, FC)), dim3(FC * 16), 0, as_stream(s), partial, nblk, dsum, dgamma, dbeta, C, (flags & RN_F_ACCUM) ? 1 : 0);
    RN_CHECK_LAUNCH("bn_bwd_finalize");
    return 0;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel<FL>, dim3(cdiv(C, FC)), dim3(FC * FL), 0, as_stream(s), partial, nblk, dsum, dgamma, dbeta, C, (flags & RN_F_ACCUM) ? 1 : 0);
  RN_CHECK_LAUNCH("bn_bwd_finalize");
  return 0;
}

extern "C" int rn_bn_bwd_finalize_split(const float* partial, int nblk, float* dsum, float* dgamma, float* dbeta, int C, int flags, void* fold, size_t fold_bytes,
                                        rn_stream s) {
  RN_CHECK_ARG(partial && dsum && dgamma && dbeta && nblk > 0 && C > 0, "rn_bn_bwd_finalize_split: bad argument");
  const size_t need = rn_bn_fold_bytes(nblk, C);
  RN_CHECK_ARG(need > 0, "rn_bn_bwd_finalize_split: %d rows x %d channels is not split (rn_bn_fold_bytes says 0): call rn_bn_bwd_finalize", nblk, C);
  RN_CHECK_ARG(fold && fold_bytes >= need, "rn_bn_bwd_finalize_split: fold buffer of %zu bytes, %zu needed", fold_bytes, need);
  const int S = fold_splits(nblk, C);
  hipLaunchKernelGGL(bn_bwd_finalize_split_kernel, dim3(cdiv(C, FC) * S), dim3(FC * FL), 0, as_stream(s), partial, nblk, dsum, dgamma, dbeta, C,
                     (flags & RN_F_ACCUM) ? 1 : 0, fold, S);
  RN_CHECK_LAUNCH("bn_bwd_finalize_split");
  return 0;
}

extern "C" int rn_bn_bwd_apply(const void* dout, const void* x, const void* mask_src, const float* coef, const float* dsum, const void* add,
                               void* dx, void* g_out, int dtype, int N, int H, int W, int C, int add_mode, int add_C, int flags, float gscale,
                               double count, float drop_p, uint32_t site, uint64_t step_seed, rn_stream s) {
  const long M = (long)N * H * W;
  if (int e = check_mc(dtype, M, C, "rn_bn_bwd_apply")) return e;
  const int use_mask = mask_mode(flags), train = (flags & RN_F_TRAIN) ? 1 : 0;
  RN_CHECK_ARG(dout && x && coef && dx && (!train || dsum) && (use_mask != 1 || mask_src) && count > 0, "rn_bn_bwd_apply: bad argument");
  RN_CHECK_ARG(drop_p == 0.f || (use_mask == 2 && M * C < (1L << 32)), "rn_bn_bwd_apply: dropout recompute needs RN_F_MASK_RECOMPUTE and < 2^32 elements");
  const uint32_t thr = drop_p > 0.f ? rn_drop_threshold(drop_p) : 0u, key = rn_drop_key(site, step_seed);
  RN_CHECK_ARG(!(flags & RN_F_WRITE_G) || g_out, "rn_bn_bwd_apply: RN_F_WRITE_G without g_out");
  ResDesc r;
  make_res(r, add, add_mode, add_C, H, W, C);
  const int ce = dtype == RN_F32 ? 4 : 8;
  const long nchunks = M * (C / ce);
  void* g = (flags & RN_F_WRITE_G) ? g_out : nullptr;
  const float inv_count = (float)(1.0 / count);
  RN_CHECK_ARG(M < (1L << 31), "rn_bn_bwd_apply: too many pixels");
  (void)nchunks;
  const int rows = slab_rows_stream(M, C, ce);
  const int grid = cdiv(M, rows);
  const int add_kind = r.mode == RN_RES_NONE ? 0 : (r.mode == RN_RES_SAME ? 1 : 2);
  RN_BY_DTYPE(dtype, dispatch_bwd_apply_stream<T_>(use_mask, add_kind, grid, as_stream(s), dout, x, mask_src, coef, dsum, r, dx, g, (int)M, H, W, C, rows, train, gscale, inv_count, key, thr));
  RN_CHECK_LAUNCH("bn_bwd_apply");
  return 0;
}

extern "C" int rn_dropout_fwd(const void* x, void* out, int dtype, int64_t n, float p, uint32_t site, uint64_t step_seed, rn_stream s) {
  RN_CHECK_ARG(x && out && p > 0.f && p < 1.f && n > 0 && n < (1L << 32), "rn_dropout_fwd: bad argument");
  const int ce = dtype == RN_F32 ? 4 : 8;
  RN_CHECK_ARG(n % ce == 0, "rn_dropout_fwd: n must be a multiple of %d", ce);
  const long nchunks = n / ce;
  const uint32_t thr = rn_drop_threshold(p), key = rn_drop_key(site, step_seed);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((dropout_fwd_kernel<T_>), dim3(ew_grid(nchunks)), dim3(NT), 0, as_stream(s), (const T_*)x, (T_*)out, nchunks, 1.f / (1.f - p), key, thr));
  RN_CHECK_LAUNCH("dropout_fwd");
  return 0;
}

extern "C" int rn_dropout_bwd(const void* dout, void* din, int dtype, int64_t n, float p, uint32_t site, uint64_t step_seed, rn_stream s) {
  RN_CHECK_ARG(dout && din && p > 0.f && p < 1.f && n > 0 && n < (1L << 32), "rn_dropout_bwd: bad argument");
  const int ce = dtype == RN_F32 ? 4 : 8;
  RN_CHECK_ARG(n % ce == 0, "rn_dropout_bwd: n must be a multiple of %d", ce);
  const long nchunks = n / ce;
  const uint32_t thr = rn_drop_threshold(p), key = rn_drop_key(site, step_seed);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((dropout_bwd_kernel<T_>), dim3(ew_grid(nchunks)), dim3(NT), 0, as_stream(s), (const T_*)dout, (T_*)din, nchunks, 1.f / (1.f - p), key, thr));
  RN_CHECK_LAUNCH("dropout_bwd");
  return 0;
}

extern "C" int rn_add_res(void* dst, const void* res, int dtype, int N, int H, int W, int C, int res_mode, int res_C, rn_stream s) {
  const long M = (long)N * H * W;
  if (int e = check_mc(dtype, M, C, "rn_add_res")) return e;
  RN_CHECK_ARG(dst && res && res_mode != RN_RES_NONE, "rn_add_res: bad argument");
  ResDesc r;
  make_res(r, res, res_mode, res_C, H, W, C);
  const int ce = dtype == RN_F32 ? 4 : 8;
  const long nchunks = M * (C / ce);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((add_res_kernel<T_>), dim3(ew_grid(nchunks)), dim3(NT), 0, as_stream(s), (T_*)dst, r, H, W, C, nchunks));
  RN_CHECK_LAUNCH("add_res");
  return 0;
}
