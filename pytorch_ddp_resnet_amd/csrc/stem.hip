// Stem convolution (resnet.py:69-75: Conv2d(3 -> K, k, s, p) WITH bias) and its weight/bias gradient, gfx950.
// C_in = 3 gives a GEMM K-dimension of 27 (3x3) or 147 (7x7): MFMA does not pay, and the op is bound by writing the
// K-channel output, so this is a direct convolution on the vector ALU that reads the NCHW fp32 image as the data
// loader delivers it (training.py:94) and writes the engine's NHWC compute dtype -- the layout change is free.
// The input has no gradient (it is the image), so the backward is wgrad + dbias only.
#include "common.h"

namespace {

constexpr int NT = 256;

struct StemArgs {
  const float* x;      // [N, C, H, W]
  const float* w;      // [K, R, S, C]
  const float* bias;   // [K]
  void* y;             // [N, P, Q, K] (forward: output; wgrad: dy)
  float* out;          // wgrad: partial slabs [nblk][K][RSC + 1]
  int N, C, H, W, K, R, S, stride, pad, P, Q;
  int KT;              // channels per block (multiple of CE)
  int M;               // N*P*Q
};

// forward: block = (pixels-per-block) x (KT/CE chunk columns); weights of the K tile transposed in LDS [RSC][KT]
template <typename T>
__global__ __launch_bounds__(NT) void stem_fwd_kernel(const StemArgs a) {
  constexpr int CE = Elem<T>::CE;
  extern __shared__ float wsm[];          // [RSC][KT]
  const int RSC = a.R * a.S * a.C;
  const int k0 = blockIdx.y * a.KT;
  const int kt = min(a.KT, a.K - k0);
  for (int i = threadIdx.x; i < RSC * a.KT; i += NT) {
    const int tap = i / a.KT, kk = i - tap * a.KT;
    wsm[i] = kk < kt ? a.w[(size_t)(k0 + kk) * RSC + tap] : 0.f;
  }
  __syncthreads();
  const int ccb = a.KT / CE;
  const int ppb = NT / ccb;
  const int kc = threadIdx.x % ccb, pl = threadIdx.x / ccb;
  if (pl >= ppb || kc * CE >= kt) return;
  const int pq = a.P * a.Q;
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  for (int m = blockIdx.x * ppb + pl; m < a.M; m += gridDim.x * ppb) {
    const int n = m / pq, rem = m - n * pq;
    const int p = rem / a.Q, q = rem - p * a.Q;
    float acc[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] = a.bias[k0 + kc * CE + e];
    for (int r = 0; r < a.R; ++r) {
      const int h = p * a.stride + r - a.pad;
      if ((unsigned)h >= (unsigned)a.H) continue;
      for (int s = 0; s < a.S; ++s) {
        const int w = q * a.stride + s - a.pad;
        if ((unsigned)w >= (unsigned)a.W) continue;
        for (int c = 0; c < a.C; ++c) {
          const float v = a.x[(((size_t)n * a.C + c) * a.H + h) * a.W + w];
          const float* wr = wsm + ((r * a.S + s) * a.C + c) * a.KT + kc * CE;
#pragma unroll
          for (int e = 0; e < CE; ++e) acc[e] = fmaf(v, wr[e], acc[e]);
        }
      }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < CE; ++e) o.e[e] = Elem<T>::from_f(acc[e]);
    store_chunk<T>(y + (size_t)m * a.K + k0 + kc * CE, o);
  }
}

// wgrad + dbias as a skinny GEMM  [K x M] . [M x (RSC+1)]  (last column = ones -> bias gradient) on the vector ALU.
// A workgroup walks pixel tiles of 64: the dy tile [64][KT] and the gathered image patches [64][RSC+1] are staged in
// LDS; thread (k, tap group) accumulates its taps over the tile with broadcast LDS reads of the patch row.  One partial
// [K][RSC+1] slab per workgroup, summed in a fixed order by stem_wgrad_reduce_kernel.
constexpr int WPIX = 64;        // pixels per tile
constexpr int MAXTG = 16;       // taps per thread (register accumulators)

template <typename T>
__global__ __launch_bounds__(NT) void stem_wgrad_kernel(const StemArgs a) {
  extern __shared__ float smem[];
  const int RSC = a.R * a.S * a.C, NCOL = RSC + 1;
  const int NCP = (NCOL + 3) / 4 * 4;                    // padded patch row (float4 reads)
  float* patch = smem;                                   // [WPIX][NCP]
  float* dyt = smem + WPIX * NCP;                        // [WPIX][KT]
  const int k0 = blockIdx.y * a.KT;
  const int kt = min(a.KT, a.K - k0);
  const int tg_count = NT / a.KT;                        // tap groups
  const int tpg = (NCOL + tg_count - 1) / tg_count;      // taps per group (<= MAXTG, checked on the host)
  const int kk = threadIdx.x % a.KT, tg = threadIdx.x / a.KT;
  const int t0 = tg * tpg;
  const bool active = tg < tg_count && kk < kt;
  const int pq = a.P * a.Q;
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.y);
  float acc[MAXTG];
#pragma unroll
  for (int i = 0; i < MAXTG; ++i) acc[i] = 0.f;
  const int ntiles = (a.M + WPIX - 1) / WPIX;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * WPIX;
    __syncthreads();
    for (int i = threadIdx.x; i < WPIX * NCP; i += NT) {
      const int pl = i / NCP, col = i - pl * NCP;
      const int m = m0 + pl;
      float v = 0.f;
      if (m < a.M && col < NCOL) {
        if (col == RSC) v = 1.f;
        else {
          const int n = m / pq, rem = m - n * pq;
          const int p = rem / a.Q, q = rem - p * a.Q;
          const int r = col / (a.S * a.C), sc = col - r * (a.S * a.C);
          const int s_ = sc / a.C, c = sc - s_ * a.C;
          const int h = p * a.stride + r - a.pad, w = q * a.stride + s_ - a.pad;
          if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) v = a.x[(((size_t)n * a.C + c) * a.H + h) * a.W + w];
        }
      }
      patch[i] = v;
    }
    {   // dy tile: 16-byte chunks (CE channels) per thread, converted to fp32 on the way into LDS
      constexpr int CE = Elem<T>::CE;
      const int cpr = a.KT / CE;                                   // KT is a multiple of 8
      for (int i = threadIdx.x; i < WPIX * cpr; i += NT) {
        const int pl = i / cpr, ch = i - pl * cpr;
        const int m = m0 + pl;
        float* d = dyt + pl * a.KT + ch * CE;
        if (m < a.M && ch * CE < kt) {
          Chunk<T> c = load_chunk<T>(dy + (size_t)m * a.K + k0 + ch * CE);
#pragma unroll
          for (int e = 0; e < CE; ++e) d[e] = Elem<T>::to_f(c.e[e]);
        } else {
#pragma unroll
          for (int e = 0; e < CE; ++e) d[e] = 0.f;
        }
      }
    }
    __syncthreads();
    if (active) {
      for (int pl = 0; pl < WPIX; ++pl) {
        const float d = dyt[pl * a.KT + kk];
        const float* pr = patch + pl * NCP + t0;
#pragma unroll
        for (int i = 0; i < MAXTG; ++i)
          if (i < tpg) acc[i] = fmaf(d, pr[i], acc[i]);
      }
    }
  }
  if (active) {
    float* __restrict__ out = a.out + (size_t)blockIdx.x * a.K * NCOL + (size_t)(k0 + kk) * NCOL;
#pragma unroll
    for (int i = 0; i < MAXTG; ++i)
      if (i < tpg && t0 + i < NCOL) out[t0 + i] = acc[i];
  }
}

// sums the per-workgroup slabs in a fixed order: 16 outputs x 16 slab-lanes per workgroup, double accumulation
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ dw, float* __restrict__ db, int K,
                                                                int RSC, int accum) {
  __shared__ double red[16][17];
  const int n = K * (RSC + 1);
  const int ol = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ol;
  double s = 0.0;
  if (i < n)
    for (int b = bl; b < nblk; b += 16) s += (double)ws[(size_t)b * n + i];
  red[bl][ol] = s;
  __syncthreads();
  if (bl != 0 || i >= n) return;
  for (int l = 1; l < 16; ++l) s += red[l][ol];
  const int k = i / (RSC + 1), col = i - k * (RSC + 1);
  float* dst = col == RSC ? db + k : dw + (size_t)k * RSC + col;
  *dst = accum ? *dst + (float)s : (float)s;
}

inline int stem_kt(const rn_conv_geom* g, int ce) {
  const int rsc = g->R * g->S * g->C;
  int kt = 49152 / (4 * rsc);
  kt = kt / 32 * 32;
  if (kt > 256) kt = 256;
  if (kt < 32) kt = 32;
  int kr = (g->K + ce - 1) / ce * ce;
  if (kt > kr) kt = kr;
  return kt;
}

inline int stem_wgrad_blocks(const rn_conv_geom* g) {
  long M = (long)g->N * g->P * g->Q;
  long b = (M + WPIX - 1) / WPIX;
  if (b > 256) b = 256;
  if (b < 1) b = 1;
  return (int)b;
}

// channels per workgroup so that (256 / KT) tap groups cover RSC+1 columns with <= MAXTG taps each
inline int stem_wgrad_kt(const rn_conv_geom* g) {
  const int ncol = g->R * g->S * g->C + 1;
  int kt = 128;
  while (kt > 8 && ((ncol + (NT / kt) - 1) / (NT / kt) > MAXTG || kt > g->K)) kt /= 2;
  return kt;          // power of two >= 8: whole 16-byte chunks of either dtype
}

int check_stem(const rn_conv_geom* g, int dtype, const char* who) {
  RN_CHECK_ARG(g != nullptr, "%s: null geometry", who);
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "%s: bad dtype", who);
  RN_CHECK_ARG(g->C > 0 && g->C <= 4 && g->S * g->C <= 24, "%s: stem path needs C_in <= 4 and S*C <= %d (got C=%d S=%d)", who, 24, g->C, g->S);
  RN_CHECK_ARG(g->K % 8 == 0, "%s: K=%d must be a multiple of 8", who, g->K);
  RN_CHECK_ARG(g->P == (g->H + 2 * g->pad - g->R) / g->stride + 1 && g->Q == (g->W + 2 * g->pad - g->S) / g->stride + 1, "%s: inconsistent output size", who);
  RN_CHECK_ARG((long)g->N * g->P * g->Q < (1L << 31), "%s: too many pixels", who);
  return 0;
}

void fill(StemArgs& a, const rn_conv_geom* g) {
  a.N = g->N; a.C = g->C; a.H = g->H; a.W = g->W; a.K = g->K; a.R = g->R; a.S = g->S; a.stride = g->stride; a.pad = g->pad;
  a.P = g->P; a.Q = g->Q; a.M = g->N * g->P * g->Q;
}

}  // namespace

extern "C" int rn_stem_conv_fwd(const float* x_nchw, const float* w_krsc, const float* bias, void* y, int dtype, const rn_conv_geom* g, rn_stream s) {
  if (int e = check_stem(g, dtype, "rn_stem_conv_fwd")) return e;
  RN_CHECK_ARG(x_nchw && w_krsc && bias && y, "rn_stem_conv_fwd: null pointer");
  StemArgs a{};
  fill(a, g);
  a.x = x_nchw; a.w = w_krsc; a.bias = bias; a.y = y;
  const int ce = dtype == RN_F32 ? 4 : 8;
  a.KT = stem_kt(g, ce);
  const int ppb = NT / (a.KT / ce);
  int gx = cdiv(a.M, ppb);
  if (gx > 8192) gx = 8192;
  const size_t smem = (size_t)g->R * g->S * g->C * a.KT * sizeof(float);
  dim3 grid(gx, cdiv(g->K, a.KT));
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((stem_fwd_kernel<T_>), grid, dim3(NT), smem, as_stream(s), a));
  RN_CHECK_LAUNCH("stem_fwd");
  return 0;
}

extern "C" size_t rn_stem_wgrad_ws_bytes(const rn_conv_geom* g) {
  if (!g) return 0;
  return (size_t)stem_wgrad_blocks(g) * g->K * (g->R * g->S * g->C + 1) * sizeof(float);
}

extern "C" int rn_stem_conv_wgrad(const float* x_nchw, const void* dy, int dtype, float* dw_krsc, float* dbias, void* ws, int accumulate,
                                  const rn_conv_geom* g, rn_stream s) {
  if (int e = check_stem(g, dtype, "rn_stem_conv_wgrad")) return e;
  RN_CHECK_ARG(x_nchw && dy && dw_krsc && dbias && ws, "rn_stem_conv_wgrad: null pointer");
  StemArgs a{};
  fill(a, g);
  a.x = x_nchw; a.y = const_cast<void*>(dy); a.out = reinterpret_cast<float*>(ws);
  a.KT = stem_wgrad_kt(g);
  const int ncol = g->R * g->S * g->C + 1;
  RN_CHECK_ARG((ncol + (NT / a.KT) - 1) / (NT / a.KT) <= MAXTG, "rn_stem_conv_wgrad: %d patch columns do not fit the tap groups", ncol);
  const int nblk = stem_wgrad_blocks(g);
  dim3 grid(nblk, cdiv(g->K, a.KT));
  const size_t smem = (size_t)WPIX * (((ncol + 3) / 4 * 4) + a.KT) * sizeof(float);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((stem_wgrad_kernel<T_>), grid, dim3(NT), smem, as_stream(s), a));
  RN_CHECK_LAUNCH("stem_wgrad");
  const int n = g->K * (g->R * g->S * g->C + 1);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(cdiv(n, 16)), dim3(256), 0, as_stream(s), reinterpret_cast<const float*>(ws), nblk, dw_krsc, dbias,
                     g->K, g->R * g->S * g->C, accumulate);
  RN_CHECK_LAUNCH("stem_wgrad_reduce");
  return 0;
}
