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

// wgrad: thread = (4 output channels) x (pixel lane); per tap-row group accumulates dy*x in registers over its pixels,
// block-reduces through LDS and writes one partial slab per block.
constexpr int WCH = 4;          // channels per thread
constexpr int MAXG = 24;        // taps per register group (S*C <= 24: 3x3x3 -> 9, 7x7x3 -> 21)

template <typename T>
__global__ __launch_bounds__(NT) void stem_wgrad_kernel(const StemArgs a) {
  __shared__ float red[NT][WCH + 1];
  const int RSC = a.R * a.S * a.C, SC = a.S * a.C;
  const int k0 = blockIdx.y * a.KT;
  const int kt = min(a.KT, a.K - k0);
  const int ccb = a.KT / WCH;
  const int ppb = NT / ccb;
  const int kc = threadIdx.x % ccb, pl = threadIdx.x / ccb;
  const bool active = pl < ppb && kc * WCH < kt;
  const int pq = a.P * a.Q;
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.y);
  float* __restrict__ out = a.out + (size_t)blockIdx.x * a.K * (RSC + 1);

  for (int r = 0; r <= a.R; ++r) {            // r == R: the bias gradient pass
    float acc[MAXG][WCH];
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
#pragma unroll
      for (int e = 0; e < WCH; ++e) acc[g][e] = 0.f;
    if (active) {
      for (int m = blockIdx.x * ppb + pl; m < a.M; m += gridDim.x * ppb) {
        const int n = m / pq, rem = m - n * pq;
        const int p = rem / a.Q, q = rem - p * a.Q;
        float d[WCH];
#pragma unroll
        for (int e = 0; e < WCH; ++e) d[e] = Elem<T>::to_f(dy[(size_t)m * a.K + k0 + kc * WCH + e]);
        if (r == a.R) {
#pragma unroll
          for (int e = 0; e < WCH; ++e) acc[0][e] += d[e];
          continue;
        }
        const int h = p * a.stride + r - a.pad;
        if ((unsigned)h >= (unsigned)a.H) continue;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
          if (g < SC) {
            const int s = g / a.C, c = g - s * a.C;
            const int w = q * a.stride + s - a.pad;
            const float v = (unsigned)w < (unsigned)a.W ? a.x[(((size_t)n * a.C + c) * a.H + h) * a.W + w] : 0.f;
#pragma unroll
            for (int e = 0; e < WCH; ++e) acc[g][e] = fmaf(v, d[e], acc[g][e]);
          }
        }
      }
    }
    const int ng = r == a.R ? 1 : SC;
    for (int g = 0; g < ng; ++g) {
#pragma unroll
      for (int e = 0; e < WCH; ++e) red[threadIdx.x][e] = acc[g][e];
      __syncthreads();
      if (threadIdx.x < ccb && threadIdx.x * WCH < kt) {
        float t[WCH] = {0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < ppb; ++l)
#pragma unroll
          for (int e = 0; e < WCH; ++e) t[e] += red[l * ccb + threadIdx.x][e];
        const int col = r == a.R ? RSC : r * SC + g;
#pragma unroll
        for (int e = 0; e < WCH; ++e) out[(size_t)(k0 + threadIdx.x * WCH + e) * (RSC + 1) + col] = t[e];
      }
      __syncthreads();
    }
  }
}

__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ dw, float* __restrict__ db, int K, int RSC,
                                         int accum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = K * (RSC + 1);
  if (i >= n) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)ws[(size_t)b * n + i];
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
  long b = M / 64;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

int check_stem(const rn_conv_geom* g, int dtype, const char* who) {
  RN_CHECK_ARG(g != nullptr, "%s: null geometry", who);
  RN_CHECK_ARG(dtype == RN_F32 || dtype == RN_BF16, "%s: bad dtype", who);
  RN_CHECK_ARG(g->C > 0 && g->C <= 4 && g->S * g->C <= MAXG, "%s: stem path needs C_in <= 4 and S*C <= %d (got C=%d S=%d)", who, MAXG, g->C, g->S);
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
  if (dtype == RN_F32) hipLaunchKernelGGL((stem_fwd_kernel<float>), grid, dim3(NT), smem, as_stream(s), a);
  else hipLaunchKernelGGL((stem_fwd_kernel<bf16_t>), grid, dim3(NT), smem, as_stream(s), a);
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
  a.KT = g->K < 128 ? (g->K + 3) / 4 * 4 : 128;
  const int nblk = stem_wgrad_blocks(g);
  dim3 grid(nblk, cdiv(g->K, a.KT));
  if (dtype == RN_F32) hipLaunchKernelGGL((stem_wgrad_kernel<float>), grid, dim3(NT), 0, as_stream(s), a);
  else hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t>), grid, dim3(NT), 0, as_stream(s), a);
  RN_CHECK_LAUNCH("stem_wgrad");
  const int n = g->K * (g->R * g->S * g->C + 1);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(s), reinterpret_cast<const float*>(ws), nblk, dw_krsc, dbias,
                     g->K, g->R * g->S * g->C, accumulate);
  RN_CHECK_LAUNCH("stem_wgrad_reduce");
  return 0;
}
