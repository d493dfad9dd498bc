// Weight gradient of the block convolutions (autograd of residual_block.py:34-57,129-159) on MFMA, gfx950.
//
//   dw[k][t][c] = sum_{m=(n,p,q)} dy[m][k] * x[n, p*stride+dh[t], q*stride+dw[t], c]          (fp32, KRSC)
//
// GEMM view per tap t: rows = K output channels (from dy), columns = C input channels (from x), reduction = the
// N*P*Q output pixels.  Both operands are pixel-major in HBM (NHWC), i.e. the reduction index is the SLOW index of
// both tiles, so the MFMA fragments are transposed reads of a [pixel][channel] LDS image:
//   bf16: ds_read_b64_tr_b16 (gfx950 hardware transpose) feeding v_mfma_f32_16x16x32_bf16,
//   f32 : ds_read_b32 feeding v_mfma_f32_16x16x4_f32 (one value per lane, rows of a tile are consecutive floats).
// Work split: (K tile x C tile x tap) x pixel-splits; each block reduces its pixel range into an fp32 slab, a
// second kernel sums the slabs in a fixed order (bitwise reproducible; no float atomics).
#include "common.h"
#include <stdlib.h>

extern int g_rn_variant;   // conv_igemm.hip (rn_set_variant)
// conv_wgrad8.hip: the eight-phase kernel for channel counts that are multiples of 256 (stream-K, sums in-kernel: no slabs, no reduction launch)
int rn_wgrad8_splits(const rn_conv_geom* g, int dtype);
int rn_launch_wgrad8(const void* x, const void* dy, float* out, int splits, int dtype, const rn_conv_geom* g, int max_grid, hipStream_t s);
// conv_wgrad8r.hip: 320 x 160 tiles for channel counts that are multiples of 160 (the WRN-28-10 family)
int rn_wgrad8r_splits(const rn_conv_geom* g, int dtype);
int rn_launch_wgrad8r(const void* x, const void* dy, float* out, int splits, int accumulate, int dtype, const rn_conv_geom* g, int max_grid, hipStream_t s);
// conv_wgrad9.hip: 3x3 stride-1 layers with 160 n output channels, all nine taps from one staged input patch (288 x 160 tiles)
int rn_wgrad9_splits(const rn_conv_geom* g, int dtype);
int rn_launch_wgrad9(const void* x, const void* dy, float* out, int splits, int dtype, const rn_conv_geom* g, int max_grid, hipStream_t s);

namespace {

constexpr int MAX_TAPS = 49;        // 7x7 stem
template <typename T> struct Bp { static constexpr int v = sizeof(T) == 4 ? 16 : 64; };  // pixels per LDS tile (bf16: two 32-pixel MFMA k-steps)

struct WgradArgs {
  const void* x;
  const void* dy;
  float* out;        // slab base: [splits][K][RS][C] (or dw itself when splits == 1 and no accumulate)
  int N, H, W, C, P, Q, K;
  int stride, RS, nt;
  int M;             // N*P*Q
  int splits, rows_per_split;
  int kt, ct;        // tiles along K and C
  unsigned magic_pq, magic_q;   // floor(2^32 / (P*Q)), floor(2^32 / Q): division by multiply-high + one correction
  int xcd_remap;     // 1: blockIdx -> work item through the bijective XCD remap
  int im2col;        // 1: the taps are folded into the column dimension (stem: C = one chunk per tap), dy is read once per column tile
  int dh[MAX_TAPS], dw[MAX_TAPS];
};

// 16-bit element types: the transposed LDS read and the MFMA of each
template <typename T> struct H16;
template <> struct H16<bf16_t> {
  typedef bf16x8 v8; typedef bf16x4 v4;
  __device__ static inline v4 tr(const char* p) { typedef __attribute__((address_space(3))) bf16x4* lp; return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p)); }
  __device__ static inline f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct H16<f16_t> {
  typedef f16x8 v8; typedef f16x4 v4;
  __device__ static inline v4 tr(const char* p) {          // the builtin is declared on __fp16 vectors: same bits as _Float16
    typedef __fp16 h4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) h4* lp;
    const h4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p));
    return __builtin_bit_cast(v4, r);
  }
  __device__ static inline f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// bf16 / fp16: A fragment of a 16-channel group = 8 pixels per lane via two transposed 4x16 block reads
template <typename T, int TI, int TJ> struct WTile {
  typedef typename H16<T>::v8 v8;
  typedef typename H16<T>::v4 v4;
  __device__ static inline v8 frag(const char* tile, int rowb, int ch0, int lane) {
    // One transposed read serves, per 32-lane half, pixels {a..a+3} and {a+8..a+11}; 8 row strides are a multiple of 64 banks
    // for every 32-byte-aligned stride, so those two pixel groups would hit the same banks (measured: a third of all LDS
    // cycles were conflict cycles).  The 32-byte channel octets of a row are therefore swapped pairwise in rows whose pixel
    // index has bit 3 set (store_tile writes them that way): the two groups land on octets of different parity.
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const char* a0 = tile + (size_t)(8 * g + q) * rowb + ((ch0 ^ ((g & 1) << 4)) + 4 * p) * 2;
    v4 lo = H16<T>::tr(a0);
    v4 hi = H16<T>::tr(a0 + 4 * rowb);
    v8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
  __device__ static inline void run(const char* ta, int rowa, int cha, const char* tb, int rowbb, int chb, int lane, f32x4 (&acc)[TI][TJ]) {
    v8 fa[2][TI], fb[2][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) fa[0][i] = frag(ta, rowa, cha + 16 * i, lane);
#pragma unroll
    for (int j = 0; j < TJ; ++j) fb[0][j] = frag(tb, rowbb, chb + 16 * j, lane);
    constexpr int NS = Bp<T>::v / 32;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < NS) {
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[nxt][i] = frag(ta + (size_t)(s + 1) * 32 * rowa, rowa, cha + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < TJ; ++j) fb[nxt][j] = frag(tb + (size_t)(s + 1) * 32 * rowbb, rowbb, chb + 16 * j, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = H16<T>::mfma(fa[cur][i], fb[cur][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

template <int TI, int TJ> struct WTile<float, TI, TJ> {
  __device__ static inline void run(const char* ta, int rowa, int cha, const char* tb, int rowbb, int chb, int lane, f32x4 (&acc)[TI][TJ]) {
    const int kk = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int s = 0; s < Bp<float>::v / 4; ++s) {
      float fa[TI], fb[TJ];
      const int pix = 4 * s + kk;
#pragma unroll
      for (int i = 0; i < TI; ++i) fa[i] = *reinterpret_cast<const float*>(ta + (size_t)pix * rowa + (cha + 16 * i + i16) * 4);
#pragma unroll
      for (int j = 0; j < TJ; ++j) fb[j] = *reinterpret_cast<const float*>(tb + (size_t)pix * rowbb + (chb + 16 * j + i16) * 4);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
};

constexpr int padded_row(int bytes) { return bytes % 128 == 64 ? bytes : bytes + ((64 - bytes % 128) + 128) % 128; }

// block tile (2*TI*16) x (2*TJ*16): 4 waves as 2x2, each TI x TJ MFMA tiles of 16x16
// IC: im2col mode (the taps are folded into the column dimension; see WgradArgs::im2col)
// (the body is a device function over the argument record and the workgroup's index: the batched launch below runs it over one record of several)
template <typename T, int TI, int TJ, bool IC, typename ARGS>
__device__ __forceinline__ void wgrad_tile(const ARGS& a, const int wg, const int nwg) {
  constexpr int CE = Elem<T>::CE;
  constexpr int BK_ = 2 * TI * 16, BC_ = 2 * TJ * 16;
  constexpr int ROWA = padded_row(BK_ * (int)sizeof(T)), ROWB = padded_row(BC_ * (int)sizeof(T));
  constexpr int CHA = BK_ / CE, CHB = BC_ / CE;          // 16-byte chunks per pixel row
  static_assert(CE != 8 || (CHA % 4 == 0 && CHB % 4 == 0), "octet swap needs whole octet pairs");
  constexpr int BP = Bp<T>::v, TPR = 256 / BP;          // pixels per tile, threads staging one pixel row
  constexpr int NJA = (CHA + TPR - 1) / TPR, NJB = (CHB + TPR - 1) / TPR;
  __shared__ __attribute__((aligned(16))) char lds[2][BP * (ROWA + ROWB)];   // the only LDS object: 2 workgroups per CU at the largest tile
  static_assert(2 * BP * (ROWA + ROWB) >= 2 * MAX_TAPS * (int)sizeof(int), "tap tables are parked in the staging buffer during setup");
  int* tdh = reinterpret_cast<int*>(&lds[0][0]);
  int* tdw = tdh + MAX_TAPS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (IC) {
    if (tid < MAX_TAPS) { tdh[tid] = tid < a.RS ? a.dh[tid] : 0; tdw[tid] = tid < a.RS ? a.dw[tid] : 0; }
    __syncthreads();
  }
  // work order: tap fastest, then C tile, K tile, pixel split slowest -- the workgroups that read the same dy / x rows are
  // neighbours.  With the XCD remap (consecutive work items on ONE XCD instead of round-robin over the 8) their re-reads
  // hit that XCD's L2 instead of going to the memory side 9 times (one per tap).
  int b = wg;
  if (a.xcd_remap) {
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int t = b % a.nt; b /= a.nt;
  const int ctile = b % a.ct; b /= a.ct;
  const int ktile = b % a.kt;
  const int split = b / a.kt;
  const int k0 = ktile * BK_, c0 = ctile * BC_;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);

  const int m_begin = split * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const int niter = m_end > m_begin ? (m_end - m_begin + BP - 1) / BP : 0;

  // staging role: pixel row tid/TPR of the tile, chunks tid%TPR + TPR*j
  const int prow = tid / TPR, cl = tid % TPR;
  const int pq = a.P * a.Q;
  const int dht = a.dh[t], dwt = a.dw[t];

  // buffer descriptors rebased on this block's first image: 32-bit byte offsets, out-of-range -> zeros in hardware
  constexpr int ES = (int)sizeof(T);
  const int n_first = m_begin / pq;
  const size_t ximg = (size_t)a.H * a.W * a.C * ES, yimg = (size_t)pq * a.K * ES;
  const size_t xleft = ((size_t)a.N - n_first) * ximg, yleft = ((size_t)a.N - n_first) * yimg;
  const __amdgpu_buffer_rsrc_t xdesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(X)) + (size_t)n_first * ximg, (short)0,
                                                                        (int)(xleft > 0xFFFFFFE0ull ? 0xFFFFFFE0u : (unsigned)xleft), 0x00020000);
  const __amdgpu_buffer_rsrc_t ydesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(DY)) + (size_t)n_first * yimg, (short)0,
                                                                        (int)(yleft > 0xFFFFFFE0ull ? 0xFFFFFFE0u : (unsigned)yleft), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  };
  unsigned cha_off[NJA], chb_off[NJB];
  int chb_dh[IC ? NJB : 1], chb_dw[IC ? NJB : 1];   // im2col: tap of each B chunk
  const int ncols = IC ? a.RS * a.C : a.C;
#pragma unroll
  for (int j = 0; j < NJA; ++j) { const int ch = cl + TPR * j; cha_off[j] = (ch < CHA && k0 + ch * CE < a.K) ? (unsigned)((k0 + ch * CE) * ES) : OOB; }
#pragma unroll
  for (int j = 0; j < NJB; ++j) {
    const int ch = cl + TPR * j;
    const bool ok = ch < CHB && c0 + ch * CE < ncols;
    if constexpr (IC) {                       // column chunk (c0/CE + ch) = (tap, chunk of the pixel): C / CE chunks per tap (1 for the padded stem, 2 for the s2d stem)
      const int cpt = a.C / CE, cc = ok ? c0 / CE + ch : 0;
      const int tap = cc / cpt;
      chb_dh[j] = tdh[tap]; chb_dw[j] = tdw[tap];
      chb_off[j] = ok ? (unsigned)((cc - tap * cpt) * 16) : OOB;
    } else {
      chb_off[j] = ok ? (unsigned)((c0 + ch * CE) * ES) : OOB;
    }
  }
  if (IC) __syncthreads();                    // the tables are dead: the staging buffer may be written

  uint4 ra[NJA], rb[NJB];
  auto load_tile = [&](int it) {
    const int m = m_begin + it * BP + prow;
    const bool mv = m < m_end;
    unsigned xoff = OOB, yoff = OOB;
    int n = 0, h0 = 0, w0 = 0;
    if (mv) {
      n = (int)__umulhi((unsigned)m, a.magic_pq);
      int rem = m - n * pq;
      if (rem >= pq) { ++n; rem -= pq; }
      int p = (int)__umulhi((unsigned)rem, a.magic_q), q = rem - p * a.Q;
      if (q >= a.Q) { ++p; q -= a.Q; }
      h0 = p * a.stride; w0 = q * a.stride;
      const int h = h0 + dht, w = w0 + dwt;
      yoff = (unsigned)(((size_t)(m - n_first * pq)) * a.K * ES);
      if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) xoff = (unsigned)((((size_t)(n - n_first) * a.H + h) * a.W + w) * a.C * ES);
    }
#pragma unroll
    for (int j = 0; j < NJA; ++j) ra[j] = bload(ydesc, (yoff != OOB && cha_off[j] != OOB) ? yoff + cha_off[j] : OOB);
    if constexpr (!IC) {
#pragma unroll
      for (int j = 0; j < NJB; ++j) rb[j] = bload(xdesc, (xoff != OOB && chb_off[j] != OOB) ? xoff + chb_off[j] : OOB);
    } else {
#pragma unroll
      for (int j = 0; j < NJB; ++j) {
        const int h = h0 + chb_dh[j], w = w0 + chb_dw[j];
        const bool ok = mv && chb_off[j] != OOB && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
        rb[j] = bload(xdesc, ok ? (unsigned)((((size_t)(n - n_first) * a.H + h) * a.W + w) * a.C * ES) + chb_off[j] : OOB);
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* ta = lds[buf];
    char* tb = lds[buf] + BP * ROWA;
    const int sw = CE == 8 ? ((prow >> 3) & 1) << 1 : 0;       // bf16: octet swap of the transposed-read layout (WTile::frag)
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
      const int ch = cl + TPR * j;
      if (ch < CHA) *reinterpret_cast<uint4*>(ta + prow * ROWA + (ch ^ sw) * 16) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
      const int ch = cl + TPR * j;
      if (ch < CHB) *reinterpret_cast<uint4*>(tb + prow * ROWB + (ch ^ sw) * 16) = rb[j];
    }
  };

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wi = wave >> 1, wj = wave & 1;
  const int cha = wi * TI * 16, chb = wj * TJ * 16;   // channel offsets of this wave inside the block tile

  if (niter > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    const int buf = it & 1;
    if (it + 1 < niter) load_tile(it + 1);
    WTile<T, TI, TJ>::run(lds[buf], ROWA, cha, lds[buf] + BP * ROWA, ROWB, chb, lane, acc);
    if (it + 1 < niter) store_tile(buf ^ 1);
    __syncthreads();
  }

  // C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* __restrict__ out = a.out + (size_t)split * a.K * a.RS * a.C;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + cha + 16 * i + (lane >> 4) * 4 + r;
      if (k >= a.K) continue;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int c = c0 + chb + 16 * j + (lane & 15);
        if (IC) { if (c < a.RS * a.C) out[(size_t)k * a.RS * a.C + c] = acc[i][j][r]; }
        else if (c < a.C) out[((size_t)k * a.RS + t) * a.C + c] = acc[i][j][r];
      }
    }
}

template <typename T, int TI, int TJ, bool IC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) { wgrad_tile<T, TI, TJ, IC>(a, (int)blockIdx.x, (int)gridDim.x); }

// Batched launch for thin networks (ResNet-20, ResNet-v2-164: ~165 weight gradients per step, each a ~10 us launch whose ramp, tail round and launch gap
// are half of it): the weight gradients of up to WB_MAX layers that share a tile shape run as ONE grid.  Record i owns the block range
// [first_block[i], first_block[i] + nwg[i]) -- starts are multiples of 8, so blockIdx & 7 is still the workgroup's XCD inside a record and the XCD remap
// of the body holds; the padding blocks exit.  Same body, same splits, same slabs as the single launches: bit-identical gradients.  The records are the
// 3 x 3 / 1 x 1 block convolutions: nine taps, so sixteen records fit the 4 KiB kernel-argument segment.
constexpr int WB_MAX = 16, WB_TAPS = 9;
struct WgradArgsS {
  const void* x;
  const void* dy;
  float* out;
  int N, H, W, C, P, Q, K;
  int stride, RS, nt;
  int M;
  int splits, rows_per_split;
  int kt, ct;
  unsigned magic_pq, magic_q;
  int xcd_remap, im2col;
  int dh[WB_TAPS], dw[WB_TAPS];
};
struct WgradBatch {
  int n;
  int first_block[WB_MAX], nwg[WB_MAX];
  WgradArgsS a[WB_MAX];
};
static_assert(sizeof(WgradBatch) <= 4096, "kernel-argument segment");

template <typename T, int TI, int TJ>
__global__ __launch_bounds__(256) void wgrad_batch_kernel(const WgradBatch wb) {
  int i = 0;
  while (i + 1 < wb.n && (int)blockIdx.x >= wb.first_block[i + 1]) ++i;         // wave-uniform
  const int wg = (int)blockIdx.x - wb.first_block[i];
  if (wg >= wb.nwg[i]) return;
  wgrad_tile<T, TI, TJ, false>(wb.a[i], wg, wb.nwg[i]);
}

// (A three-taps-per-workgroup form -- dy loaded once for a row of taps, x as one run of 66 pixels read shifted, padding by redirecting
// the transposed read to a zero row: 28-40 % fewer bytes per FLOP -- was built, parity-tested and measured 10-30 % SLOWER on the
// WRN-28-10 shapes (253 registers, a pixel decode and address selects in every k-step); DESIGN.md 6e.  It is gone.)

__device__ inline void add4(float4& s, const float4& v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }

// dw = slab 0 + slab 1 + ... in slab order.  (Issuing the loads of eight slabs together made this SLOWER, 14.7 -> 22 us on WRN-28-10's
// 320 -> 320 layers: the slabs of one output are a multiple of 16 KiB apart and land on the same memory channels.)
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n, int splits, int accum) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n4 = n >> 2;
  for (; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(ws)[i];
    for (int k = 1; k < splits; ++k) add4(s, reinterpret_cast<const float4*>(ws + (size_t)k * n)[i]);
    if (accum) add4(s, reinterpret_cast<float4*>(dw)[i]);
    reinterpret_cast<float4*>(dw)[i] = s;
  }
}

// the same in-order sum (slab 0 + slab 1 + ... : the same bits) for the LARGE gradients of the eight-phase weight-gradient kernel (up to 9.4 M floats per
// slab): one thread per output chunk and four slabs' loads in flight before the four in-order adds -- the loop above walks the slabs one dependent
// load at a time and was 81 us per launch on average in the WRN-50-2 profile (41 launches per step)
__global__ __launch_bounds__(256) void wgrad_reduce_mlp_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n, int splits, int accum) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (n >> 2)) return;
  float4 s = reinterpret_cast<const float4*>(ws)[i];
  int k = 1;
  for (; k + 3 < splits; k += 4) {
    float4 v[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) v[l] = reinterpret_cast<const float4*>(ws + (size_t)(k + l) * n)[i];
#pragma unroll
    for (int l = 0; l < 4; ++l) add4(s, v[l]);
  }
  for (; k < splits; ++k) add4(s, reinterpret_cast<const float4*>(ws + (size_t)k * n)[i]);
  if (accum) add4(s, reinterpret_cast<float4*>(dw)[i]);
  reinterpret_cast<float4*>(dw)[i] = s;
}

// The sums of reduce_wide_body below (sixteen interleaved partial sums per output, then those in order: the SAME bits) by ONE thread per
// output chunk with sixteen accumulators: sixteen independent, fully coalesced loads per step.  For outputs large enough to fill the chip
// with a thread each (WRN-28-10's 160 -> 160 3x3: 57,600 chunks x 43 slabs took 34 us as 3,600 workgroups of 16 x 16 threads with two
// barriers each).
__global__ __launch_bounds__(256) void wgrad_reduce_lanes_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n, int splits, int accum) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 acc[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) acc[l] = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 15 < splits; k += 16) {
      float4 v[16];
#pragma unroll
      for (int l = 0; l < 16; ++l) v[l] = reinterpret_cast<const float4*>(ws + (size_t)(k + l) * n)[i];
#pragma unroll
      for (int l = 0; l < 16; ++l) add4(acc[l], v[l]);
    }
#pragma unroll
    for (int l = 0; l < 16; ++l)
      if (k + l < splits) add4(acc[l], reinterpret_cast<const float4*>(ws + (size_t)(k + l) * n)[i]);
    float4 t = acc[0];
#pragma unroll
    for (int l = 1; l < 16; ++l) add4(t, acc[l]);
    if (accum) add4(t, reinterpret_cast<float4*>(dw)[i]);
    reinterpret_cast<float4*>(dw)[i] = t;
  }
}

// Small outputs split many ways (1x1 convolutions of thin layers: 1,024 outputs x 512 slabs): one thread per output would walk
// the slabs serially (measured 43 us).  Here 16 threads share an output chunk, each sums every 16th slab with 4 loads in
// flight, and the 16 partial sums are combined through LDS in a fixed order (bitwise reproducible).
__device__ inline void reduce_wide_body(const float* __restrict__ ws, float* __restrict__ dw, long n, int splits, int accum, int blk, int nblk,
                                        float4 (*part)[17]) {
  const long n4 = n >> 2;
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  for (long base = (long)blk * 16; base < n4; base += (long)nblk * 16) {
    const long i = base + o;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      int k = sl;
      for (; k + 48 < splits; k += 64) {
        const float4 v0 = reinterpret_cast<const float4*>(ws + (size_t)k * n)[i];
        const float4 v1 = reinterpret_cast<const float4*>(ws + (size_t)(k + 16) * n)[i];
        const float4 v2 = reinterpret_cast<const float4*>(ws + (size_t)(k + 32) * n)[i];
        const float4 v3 = reinterpret_cast<const float4*>(ws + (size_t)(k + 48) * n)[i];
        s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
        s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
        s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
        s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
      }
      for (; k < splits; k += 16) {
        const float4 v = reinterpret_cast<const float4*>(ws + (size_t)k * n)[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    part[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < n4) {
      float4 t = part[0][o];
#pragma unroll
      for (int l = 1; l < 16; ++l) { const float4 v = part[l][o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      if (accum) { const float4 v = reinterpret_cast<float4*>(dw)[i]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      reinterpret_cast<float4*>(dw)[i] = t;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n, int splits, int accum) {
  __shared__ float4 part[16][17];
  reduce_wide_body(ws, dw, n, splits, accum, (int)blockIdx.x, (int)gridDim.x, part);
}
// the same sums for several layers in one launch (deferred reductions of thin networks: 165 launches of ~5 us per ResNet-v2-164 step); a
// workgroup finds its layer by the block prefix sums, like pack_w_batch_kernel.  Per output the summation order is that of the kernel above.
struct ReduceBatch {
  rn_reduce_desc d[RN_REDUCE_BATCH_MAX];
  int first_block[RN_REDUCE_BATCH_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const ReduceBatch rb) {
  __shared__ float4 part[16][17];
  int l = 0;
  while (l + 1 < rb.n && (int)blockIdx.x >= rb.first_block[l + 1]) ++l;
  const rn_reduce_desc& d = rb.d[l];
  reduce_wide_body(d.slabs, d.dw, (long)d.n, d.splits, d.accumulate, (int)blockIdx.x - rb.first_block[l], rb.first_block[l + 1] - rb.first_block[l], part);
}
inline int reduce_wide_blocks(long n4) {
  int blocks = (int)((n4 + 15) / 16);
  return blocks > 4096 ? 4096 : blocks;
}
inline bool reduce_is_wide(int splits, long n4) { return splits >= 32 && n4 < 65536; }      // few outputs, many slabs: the slab walk is split over 16 threads

struct WCfg { int bk, bc; };

inline int pick_tile(int n) {       // block tile edge from {160,128,64,32}: fewest tiles, then least padding
  const int opts[4] = {160, 128, 64, 32};
  int best = 32, best_tiles = 1 << 30, best_pad = 1 << 30;
  for (int o : opts) {
    int tiles = (n + o - 1) / o, pad = tiles * o - n;
    long cost = (long)tiles * o;
    if (cost < (long)best_tiles * best || (cost == (long)best_tiles * best && tiles < best_tiles)) { best = o; best_tiles = tiles; best_pad = pad; }
  }
  (void)best_pad;
  return best;
}

constexpr int IM2COL_BC = 160;
// one-chunk inputs (the stems: image channels padded to a chunk), and the space-to-depth ImageNet stem (16 channels, 4 x 4 taps, no padding: 256 columns)
inline bool s2d_stem(const rn_conv_geom* g, int dtype_ce) { return dtype_ce == 8 && g->C == 16 && g->R == 4 && g->S == 4 && g->pad == 0 && g->stride == 1; }
inline bool use_im2col(const rn_conv_geom* g, int dtype_ce) { return (g->C == dtype_ce && g->R * g->S >= 4) || s2d_stem(g, dtype_ce); }
inline int col_tile(const rn_conv_geom* g, bool ic, int bk) { return ic ? (((g->R * g->S * g->C) % 128 == 0 && bk == 128) ? 128 : IM2COL_BC) : pick_tile(g->C); }

// workgroups of a (bk x bc) tile that are resident at once: 256 CUs x what LDS (two 64- / 16-pixel stages of both operands) and the
// accumulator registers allow -- 2 per CU at 160 x 160, 8 at 32 x 32.  Thin layers (ResNet-20 / v2-164: 16..64 channels) are a chain of
// ~1 us load -> LDS -> barrier steps per workgroup; sized for 512 workgroups like the wide tiles they ran 37 such steps each at an eighth
// of the occupancy the tile allows (33 us for 8 MB of input).
inline int wgrad_capacity(int bk, int bc, int ce) {
  const int es = ce == 8 ? 2 : 4, bp = ce == 8 ? 64 : 16;
  const int lds = 2 * bp * (padded_row(bk * es) + padded_row(bc * es));
  const int tiles16 = (bk / 32) * (bc / 32);             // 16x16 accumulator tiles per wave
  const int by_regs = tiles16 <= 1 ? 8 : (tiles16 <= 4 ? 5 : (tiles16 <= 10 ? 3 : 2));
  int per_cu = 160 * 1024 / lds;
  if (per_cu > by_regs) per_cu = by_regs;
  if (per_cu < 1) per_cu = 1;
  return 256 * per_cu;
}

int wgrad_splits(const rn_conv_geom* g, int bk, int bc, bool im2col, int capacity) {
  const long M = (long)g->N * g->P * g->Q;
  const int tiles = im2col ? cdiv(g->K, bk) * cdiv(g->R * g->S * g->C, bc) : cdiv(g->K, bk) * cdiv(g->C, bc) * g->R * g->S;
  int splits = capacity / tiles;                         // one resident round (7/8 of one for the forked launches, see rn_conv_wgrad)
  const int max_by_rows = (int)((M + 255) / 256);      // at least 8 K-steps per block
  if (splits > max_by_rows) splits = max_by_rows;
  if (splits < 1) splits = 1;
  if (splits > 256) splits = 256;
  return splits;
}

template <typename T, int TI, int TJ, bool IC = false>
int launch_w(const WgradArgs& a, hipStream_t s) {
  int grid = a.kt * a.ct * a.nt * a.splits;
  rn_note_kernel("wgrad%s<%dx%d>", IC ? "_im2col" : "", 2 * TI * 16, 2 * TJ * 16);
  if (rn_dry_run()) return 0;
  hipLaunchKernelGGL((wgrad_kernel<T, TI, TJ, IC>), dim3(grid), dim3(256), 0, s, a);
  RN_CHECK_LAUNCH("wgrad");
  return 0;
}

template <typename T> int dispatch_w(const WgradArgs& a, int bk, int bc, hipStream_t s) {
  if (a.im2col) {                                        // column tile fixed at IM2COL_BC, or 128 when the columns are a multiple of it (the s2d stem's 256)
    if (bc == 128) return launch_w<T, 4, 4, true>(a, s);      // (col_tile picks 128 columns only beside 128 rows)
    if (bk == 160) return launch_w<T, 5, IM2COL_BC / 32, true>(a, s);
    if (bk == 128) return launch_w<T, 4, IM2COL_BC / 32, true>(a, s);
    if (bk == 64) return launch_w<T, 2, IM2COL_BC / 32, true>(a, s);
    return launch_w<T, 1, IM2COL_BC / 32, true>(a, s);
  }
#define W_CASE(BK, BC) if (bk == BK && bc == BC) return launch_w<T, BK / 32, BC / 32>(a, s);
  W_CASE(160, 160) W_CASE(160, 128) W_CASE(160, 64) W_CASE(160, 32)
  W_CASE(128, 160) W_CASE(128, 128) W_CASE(128, 64) W_CASE(128, 32)
  W_CASE(64, 160) W_CASE(64, 128) W_CASE(64, 64) W_CASE(64, 32)
  W_CASE(32, 160) W_CASE(32, 128) W_CASE(32, 64) W_CASE(32, 32)
#undef W_CASE
  rn_set_error("wgrad: no tile %dx%d", bk, bc);
  return 1;
}

template <typename T> int dispatch_wb(const WgradBatch& wb, int grid, int bk, int bc, hipStream_t s) {
#define WB_CASE(BK, BC) if (bk == BK && bc == BC) { hipLaunchKernelGGL((wgrad_batch_kernel<T, BK / 32, BC / 32>), dim3(grid), dim3(256), 0, s, wb); RN_CHECK_LAUNCH("wgrad_batch"); return 0; }
  WB_CASE(160, 160) WB_CASE(160, 128) WB_CASE(160, 64) WB_CASE(160, 32)
  WB_CASE(128, 160) WB_CASE(128, 128) WB_CASE(128, 64) WB_CASE(128, 32)
  WB_CASE(64, 160) WB_CASE(64, 128) WB_CASE(64, 64) WB_CASE(64, 32)
  WB_CASE(32, 160) WB_CASE(32, 128) WB_CASE(32, 64) WB_CASE(32, 32)
#undef WB_CASE
  rn_set_error("wgrad batch: no tile %dx%d", bk, bc);
  return 1;
}

}  // namespace

// the kernel choice and the argument record of one weight gradient (everything but the output pointer): rn_conv_wgrad and the batched launch share it
struct WgradSel { int w8, w9, w8r, bk, bc; bool ic; };
static void wgrad_fill(WgradArgs& a, WgradSel& sel, const void* x, const void* dy, const rn_conv_geom* g, int dtype, int flags) {
  const int ce = dtype == RN_F32 ? 4 : 8;
  sel.w8 = rn_wgrad8_splits(g, dtype);                  // > 0: the eight-phase kernel with that many pixel splits (its slabs take the same reduction)
  sel.w9 = sel.w8 > 0 ? 0 : rn_wgrad9_splits(g, dtype);       // > 0: the nine-tap 288 x 160 kernel (3x3 stride 1, 160 n output channels), likewise
  sel.w8r = (sel.w8 > 0 || sel.w9 > 0) ? 0 : rn_wgrad8r_splits(g, dtype);     // > 0: the 320 x 160 kernel of the 160-channel family, likewise
  const bool ic = use_im2col(g, ce);
  const int bk = pick_tile(g->K), bc = col_tile(g, ic, bk);
  sel.ic = ic; sel.bk = bk; sel.bc = bc;
  a.im2col = ic ? 1 : 0;
  a.xcd_remap = (g_rn_variant & 4) ? 0 : 1;
  {
    const unsigned long long pq = (unsigned long long)g->P * g->Q;
    a.magic_pq = pq <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / pq);
    a.magic_q = g->Q <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned)g->Q);
  }
  a.x = x; a.dy = dy;
  a.N = g->N; a.H = g->H; a.W = g->W; a.C = g->C; a.P = g->P; a.Q = g->Q; a.K = g->K;
  a.stride = g->stride; a.RS = g->R * g->S; a.nt = ic ? 1 : a.RS;
  for (int r = 0; r < g->R; ++r)
    for (int t = 0; t < g->S; ++t) { a.dh[r * g->S + t] = r - g->pad; a.dw[r * g->S + t] = t - g->pad; }
  a.M = g->N * g->P * g->Q;
  // A FORKED weight gradient (side stream, beside the data-gradient / BatchNorm chain) is sized to 7/8 of a resident round: a
  // one-round grid retires no workgroup until it ends, so the chain's tiny finalize kernels waited ~19 us each for a slot
  // (26 per WRN-28-10 step); with an eighth of the slots left free they dispatch at once.  rn_set_variant 1 << 24: full round.
  const int round = (g_rn_variant & (1 << 23)) ? 512 : wgrad_capacity(bk, bc, ce);       // 1 << 23: the fixed 512-workgroup round (A/B)
  const int capacity = ((flags & RN_F_FORK) && !(g_rn_variant & (1 << 24))) ? round / 8 * 7 : round;
  a.splits = sel.w8 > 0 ? sel.w8 : (sel.w9 > 0 ? sel.w9 : (sel.w8r > 0 ? sel.w8r : wgrad_splits(g, bk, bc, ic, capacity)));
  a.rows_per_split = ((a.M + a.splits - 1) / a.splits + 31) / 32 * 32;
  a.kt = cdiv(g->K, bk); a.ct = cdiv(ic ? g->R * g->S * g->C : g->C, bc);
}

// dw (+)= sum of `splits` slabs of n floats, in slab order per output (the kernel depends on the shape only: bitwise reproducible)
int rn_wgrad_reduce_slabs(const float* ws, float* dw_krsc, long n, int splits, int accum, int eight_phase, hipStream_t s) {
  const long n4 = n / 4;
  if (reduce_is_wide(splits, n4) && n4 >= 16384) {   // a thread per output chunk fills the chip: same sums, no LDS step
    hipLaunchKernelGGL(wgrad_reduce_lanes_kernel, dim3((int)((n4 + 255) / 256)), dim3(256), 0, s, ws, dw_krsc, n, splits, accum);
  } else if (reduce_is_wide(splits, n4)) {
    const int blocks = reduce_wide_blocks(n4);
    hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(blocks), dim3(256), 0, s, ws, dw_krsc, n, splits, accum);
  } else if (eight_phase && n4 >= 65536) {           // the eight-phase kernels' large slabs: a thread per chunk, four slabs in flight (same sums)
    hipLaunchKernelGGL(wgrad_reduce_mlp_kernel, dim3((int)((n4 + 255) / 256)), dim3(256), 0, s, ws, dw_krsc, n, splits, accum);
  } else {
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, dw_krsc, n, splits, accum);
  }
  RN_CHECK_LAUNCH("wgrad_reduce");
  return 0;
}

extern "C" size_t rn_conv_wgrad_ws_bytes(const rn_conv_geom* g) {
  if (!g) return 0;
  size_t best = 0;
  for (int ce : {4, 8}) {                     // the workspace is sized before the dtype is known: take the larger need
    const bool ic = use_im2col(g, ce);
    const int bk = pick_tile(g->K), bc = col_tile(g, ic, bk);
    const size_t need = (size_t)wgrad_splits(g, bk, bc, ic, wgrad_capacity(bk, bc, ce)) * g->K * g->R * g->S * g->C * sizeof(float);
    if (need > best) best = need;
  }
  const int w8 = rn_wgrad8_splits(g, RN_F16);             // the eight-phase kernel's own split count (16-bit engines)
  size_t need8 = w8 > 0 ? (size_t)w8 * g->K * g->R * g->S * g->C * sizeof(float) : 0;
  const int w9 = w8 > 0 ? 0 : rn_wgrad9_splits(g, RN_F16);
  const int w8r = (w8 > 0 || w9 > 0) ? 0 : rn_wgrad8r_splits(g, RN_F16);
  if (w9 > 0) need8 = (size_t)w9 * g->K * g->R * g->S * g->C * sizeof(float);
  if (w8r > 0) need8 = (size_t)w8r * g->K * g->R * g->S * g->C * sizeof(float);
  return need8 > best ? need8 : best;
}

extern "C" int rn_conv_wgrad(const void* x, const void* dy, float* dw_krsc, void* ws, size_t ws_bytes, int flags, int dtype,
                             const rn_conv_geom* g, rn_stream s) {
  RN_CHECK_ARG(g && x && dy && dw_krsc, "rn_conv_wgrad: null pointer");
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "rn_conv_wgrad: bad dtype");
  const int ce = dtype == RN_F32 ? 4 : 8;
  RN_CHECK_ARG(g->C % ce == 0 && g->K % ce == 0, "rn_conv_wgrad: C=%d, K=%d must be multiples of %d", g->C, g->K, ce);
  RN_CHECK_ARG(g->R == g->S && g->R * g->S <= MAX_TAPS, "rn_conv_wgrad: kernel %dx%d unsupported", g->R, g->S);
  RN_CHECK_ARG((long)g->N * g->P * g->Q < (1L << 31), "rn_conv_wgrad: too many pixels");
  WgradSel sel;
  WgradArgs a{};
  wgrad_fill(a, sel, x, dy, g, dtype, flags);
  const int w8 = sel.w8, w9 = sel.w9, w8r = sel.w8r, bk = sel.bk, bc = sel.bc;
  const size_t n = (size_t)g->K * a.RS * g->C;
  const bool direct = a.splits == 1 && !(flags & RN_F_ACCUM);
  if (!direct) {
    RN_CHECK_ARG(ws != nullptr && ws_bytes >= (size_t)a.splits * n * sizeof(float), "rn_conv_wgrad: workspace too small (%zu < %zu)",
                 ws_bytes, (size_t)a.splits * n * sizeof(float));
  }
  a.out = direct ? dw_krsc : reinterpret_cast<float*>(ws);
  int e = 0;
  // rn_set_variant low byte of bits 8..15 is taken elsewhere; RN_W8_FORK_GRID (environment, read once) sizes a forked eight-phase launch for A/B runs
  static const int w8_fork_grid = getenv("RN_W8_FORK_GRID") ? atoi(getenv("RN_W8_FORK_GRID")) : 256;
  if (w8 > 0) e = rn_launch_wgrad8(x, dy, a.out, a.splits, dtype, g, (flags & RN_F_FORK) ? w8_fork_grid : 256, as_stream(s));
  else if (w9 > 0) e = rn_launch_wgrad9(x, dy, a.out, a.splits, dtype, g, (flags & RN_F_FORK) ? w8_fork_grid : 256, as_stream(s));
  else if (w8r > 0) e = rn_launch_wgrad8r(x, dy, a.out, a.splits, 0, dtype, g, (flags & RN_F_FORK) ? w8_fork_grid : 256, as_stream(s));
  else RN_BY_DTYPE(dtype, e = dispatch_w<T_>(a, bk, bc, as_stream(s)));
  if (e) return e;
  if (!direct) {
    const long n4 = (long)(n / 4);
    rn_note_kernel(reduce_is_wide(a.splits, n4) ? "wgrad_reduce_wide" : "wgrad_reduce");
    if (rn_dry_run()) return 0;
    if (flags & RN_F_DEFER_REDUCE) {                     // the caller sums the slabs later (rn_wgrad_reduce_batch): only for the wide kind
      RN_CHECK_ARG(reduce_is_wide(a.splits, n4), "rn_conv_wgrad: RN_F_DEFER_REDUCE on a geometry whose reduction is not deferrable (rn_conv_wgrad_splits < 0)");
      return 0;
    }
    return rn_wgrad_reduce_slabs(reinterpret_cast<const float*>(ws), dw_krsc, (long)n, a.splits, (flags & RN_F_ACCUM) ? 1 : 0, (w8 > 0 || w9 > 0 || w8r > 0) ? 1 : 0, as_stream(s));
  }
  return 0;
}

extern "C" int rn_conv_wgrad_splits(const rn_conv_geom* g, int dtype, int flags) {
  if (!g || !RN_DTYPE_OK(dtype)) return -1;
  const int ce = dtype == RN_F32 ? 4 : 8;
  const bool ic = use_im2col(g, ce);
  const int bk = pick_tile(g->K), bc = col_tile(g, ic, bk);
  const int round = (g_rn_variant & (1 << 23)) ? 512 : wgrad_capacity(bk, bc, ce);
  const int capacity = ((flags & RN_F_FORK) && !(g_rn_variant & (1 << 24))) ? round / 8 * 7 : round;
  const int w8 = rn_wgrad8_splits(g, dtype), w9 = w8 > 0 ? 0 : rn_wgrad9_splits(g, dtype), w8r = (w8 > 0 || w9 > 0) ? 0 : rn_wgrad8r_splits(g, dtype);
  const int splits = w8 > 0 ? w8 : (w9 > 0 ? w9 : (w8r > 0 ? w8r : wgrad_splits(g, bk, bc, ic, capacity)));
  if (splits == 1 && !(flags & RN_F_ACCUM)) return 0;
  const long n4 = (long)g->K * g->R * g->S * g->C / 4;
  return reduce_is_wide(splits, n4) ? splits : -1;
}

static long g_wgrad_batch_launches = 0;                  // diagnostic (tests): batched launches issued by this process
extern "C" long rn_wgrad_batch_launches(void) { return g_wgrad_batch_launches; }

// > 0: the weight gradient of this geometry may ride in a batched launch with others of the same key (the tile shape); 0: it launches alone
extern "C" int rn_conv_wgrad_batch_key(const rn_conv_geom* g, int dtype, int flags) {
  if (!g || !RN_DTYPE_OK(dtype) || (flags & RN_F_FORK) || (g_rn_variant & (1 << 17))) return 0;        // 1 << 17: never (A/B)
  const int ce = dtype == RN_F32 ? 4 : 8;
  if (g->R != g->S || g->R * g->S > WB_TAPS || g->C % ce || g->K % ce) return 0;
  if (use_im2col(g, ce) || rn_wgrad8_splits(g, dtype) > 0 || rn_wgrad9_splits(g, dtype) > 0 || rn_wgrad8r_splits(g, dtype) > 0) return 0;
  if (rn_conv_wgrad_splits(g, dtype, flags) <= 0) return 0;                                           // only slabs whose sums are deferrable
  const int bk = pick_tile(g->K), bc = col_tile(g, false, bk);
  return bk * 1024 + bc;
}

// the slab-writing launches of up to RN_WGRAD_BATCH_MAX weight gradients of ONE key as one grid (thin networks: their ~10 us launches are half ramp, tail
// and gap); every record writes its [splits][K*R*S*C] slabs exactly as rn_conv_wgrad(..., RN_F_DEFER_REDUCE) would -- the caller sums them
// (rn_wgrad_reduce_batch)
extern "C" int rn_conv_wgrad_batch(const rn_wgrad_desc* descs, int n, int dtype, rn_stream s) {
  static_assert(RN_WGRAD_BATCH_MAX == WB_MAX, "header and kernel disagree");
  RN_CHECK_ARG(descs && n > 0 && n <= RN_WGRAD_BATCH_MAX, "rn_conv_wgrad_batch: n=%d out of range (1..%d)", n, RN_WGRAD_BATCH_MAX);
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "rn_conv_wgrad_batch: bad dtype");
  const int key = rn_conv_wgrad_batch_key(&descs[0].g, dtype, descs[0].flags);
  RN_CHECK_ARG(key > 0, "rn_conv_wgrad_batch: record 0 is not batchable (rn_conv_wgrad_batch_key)");
  WgradBatch wb{};
  wb.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const rn_wgrad_desc& d = descs[i];
    RN_CHECK_ARG(d.x && d.dy && d.slabs, "rn_conv_wgrad_batch: record %d: null pointer", i);
    RN_CHECK_ARG(rn_conv_wgrad_batch_key(&d.g, dtype, d.flags) == key, "rn_conv_wgrad_batch: record %d has another key", i);
    WgradSel sel;
    WgradArgs a{};
    wgrad_fill(a, sel, d.x, d.dy, &d.g, dtype, d.flags);
    RN_CHECK_ARG(d.splits == 0 || d.splits == a.splits, "rn_conv_wgrad_batch: record %d was planned with %d pixel splits, the launch would take %d (rn_set_variant changed in between?)", i,
                 d.splits, a.splits);
    RN_CHECK_ARG(d.slab_bytes == 0 || d.slab_bytes >= (uint64_t)a.splits * d.g.K * d.g.R * d.g.S * d.g.C * sizeof(float), "rn_conv_wgrad_batch: record %d: slab region of %llu bytes is too small for %d splits", i,
                 (unsigned long long)d.slab_bytes, a.splits);
    WgradArgsS& o = wb.a[i];
    o.x = a.x; o.dy = a.dy; o.out = d.slabs;
    o.N = a.N; o.H = a.H; o.W = a.W; o.C = a.C; o.P = a.P; o.Q = a.Q; o.K = a.K;
    o.stride = a.stride; o.RS = a.RS; o.nt = a.nt; o.M = a.M;
    o.splits = a.splits; o.rows_per_split = a.rows_per_split; o.kt = a.kt; o.ct = a.ct;
    o.magic_pq = a.magic_pq; o.magic_q = a.magic_q; o.xcd_remap = a.xcd_remap; o.im2col = 0;
    for (int t = 0; t < WB_TAPS; ++t) { o.dh[t] = t < a.RS ? a.dh[t] : 0; o.dw[t] = t < a.RS ? a.dw[t] : 0; }
    wb.first_block[i] = blocks;
    wb.nwg[i] = a.kt * a.ct * a.nt * a.splits;
    blocks += (wb.nwg[i] + 7) / 8 * 8;                   // a record starts on a multiple of 8: blockIdx & 7 stays the XCD inside it
    rn_note_kernel("wgrad<%dx%d>", sel.bk, sel.bc);
    rn_note_kernel("wgrad_reduce_wide");
  }
  if (rn_dry_run()) return 0;
  ++g_wgrad_batch_launches;
  int e = 0;
  RN_BY_DTYPE(dtype, e = dispatch_wb<T_>(wb, blocks, key / 1024, key % 1024, as_stream(s)));
  return e;
}

extern "C" int rn_wgrad_reduce_batch(const rn_reduce_desc* descs, int n, rn_stream s) {
  RN_CHECK_ARG(descs && n > 0 && n <= RN_REDUCE_BATCH_MAX, "rn_wgrad_reduce_batch: n=%d out of range (1..%d)", n, RN_REDUCE_BATCH_MAX);
  ReduceBatch rb;
  rb.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    RN_CHECK_ARG(descs[i].slabs && descs[i].dw && descs[i].n > 0 && descs[i].n % 4 == 0 && descs[i].splits > 0, "rn_wgrad_reduce_batch: bad descriptor %d", i);
    rb.d[i] = descs[i];
    rb.first_block[i] = blocks;
    blocks += reduce_wide_blocks(descs[i].n / 4);
  }
  rb.first_block[n] = blocks;
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, as_stream(s), rb);
  RN_CHECK_LAUNCH("wgrad_reduce_batch");
  return 0;
}
