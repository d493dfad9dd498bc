// Small kernels around the conv/BN core: weight repack, MaxPool2d (resnet.py:83-87), global AvgPool2d + Flatten +
// Linear (resnet.py:77-81,117-120), softmax cross-entropy + top-k error (metrics.py:10-29) and a fused flat SGD step
// (torch.optim.SGD rule; optim_util.py:11-18, config.yaml:22-28).  gfx950.
#include "common.h"
#include <algorithm>
#include <float.h>

extern int g_rn_variant;   // conv_igemm.hip (rn_set_variant): 1 << 25 = per-pixel gather in the fused stem backward (A/B)

namespace {

constexpr int NT = 256;

// ---- weights: fp32 KRSC master -> [K][RS][C] and [C][RS][K] in the compute dtype ---------------------------
template <typename T>
__global__ void pack_w_fwd_kernel(const float* __restrict__ w, T* __restrict__ wf, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) wf[i] = Elem<T>::from_f(w[i]);
}

// w[k][rs][c] -> wf[k][rs][c] and wd[c][rs][k] (either may be NULL): one 64(k) x 64(c) tile of one tap per workgroup, transposed
// through LDS; 16-byte reads along c, 16-byte (bf16: 8 k) stores along k of the transposed copy.  K and C are multiples of
// 8 (bf16) / 4 (fp32) (conv geometry check), so every 4-wide / 8-wide group is whole.
template <typename T>
__device__ inline void pack_tile(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int K, int RS, int C, int b, float (*tile)[65]) {
  constexpr int CE = Elem<T>::CE;
  const int ct = (C + 63) / 64;
  const int rs = b % RS; b /= RS;
  const int c0 = (b % ct) * 64, k0 = (b / ct) * 64;
  const int t = threadIdx.x;
  {
    const int c4 = (t & 15) * 4, r = t >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + r + 16 * j, c = c0 + c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool ok = k < K && c < C;
      if (ok) {
        const size_t off = ((size_t)k * RS + rs) * C + c;
        v = *reinterpret_cast<const float4*>(w + off);
        if (wf) {                                  // forward copy from the same read
          if constexpr (CE == 8) {
            typedef T t4 __attribute__((ext_vector_type(4)));
            t4 o = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
            *reinterpret_cast<t4*>(wf + off) = o;
          } else {
            *reinterpret_cast<float4*>(wf + off) = v;
          }
        }
      }
      tile[r + 16 * j][c4] = v.x; tile[r + 16 * j][c4 + 1] = v.y; tile[r + 16 * j][c4 + 2] = v.z; tile[r + 16 * j][c4 + 3] = v.w;
    }
  }
  if (!wd) return;                                 // uniform per workgroup
  __syncthreads();
  {
    const int k8 = (t & 7) * 8, cc = t >> 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + cc + 32 * j, k = k0 + k8;
      if (c >= C || k >= K) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = tile[k8 + i][cc + 32 * j];
      T* dst = wd + ((size_t)c * RS + rs) * K + k;
      if constexpr (CE == 8) {
        Chunk<T> o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.e[i] = Elem<T>::from_f(v[i]);
        store_chunk<T>(dst, o);
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        if (k + 4 < K) *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_w_dgrad_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int K, int RS, int C) {
  __shared__ float tile[64][65];
  pack_tile<T>(w, wf, wd, K, RS, C, blockIdx.x, tile);
}

// every conv weight of a network in ONE launch (rn_pack_weights_batch): the descriptors travel as kernel arguments, a
// workgroup finds its layer by a scan of the block prefix sums (26 launches of 3-14 us each were launch-latency bound)
struct PackBatch {
  rn_pack_desc d[RN_PACK_BATCH_MAX];
  int first_block[RN_PACK_BATCH_MAX + 1];
  int n;
};
template <typename T>
__global__ __launch_bounds__(256) void pack_w_batch_kernel(const PackBatch pb) {
  __shared__ float tile[64][65];
  int l = 0;
  while (l + 1 < pb.n && (int)blockIdx.x >= pb.first_block[l + 1]) ++l;
  const rn_pack_desc& d = pb.d[l];
  pack_tile<T>(d.w, reinterpret_cast<T*>(d.w_fwd), reinterpret_cast<T*>(d.w_dgrad), d.K, d.RS, d.C, (int)blockIdx.x - pb.first_block[l], tile);
}

// ---- MFMA stem route: image / weights / weight-gradient between the reference layouts and the padded NHWC operands ----
template <typename T>
__global__ __launch_bounds__(NT) void img_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int C, int H, int W, int CP) {
  const long n = (long)N * H * W;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const long hw = (long)H * W;
    const long nn = i / hw, r = i - nn * hw;
    T* o = out + i * CP;
    for (int c = 0; c < CP; ++c) o[c] = Elem<T>::from_f(c < C ? x[(nn * C + c) * hw + r] : 0.f);
  }
}

template <typename T>
__global__ void pack_stem_w_kernel(const float* __restrict__ w, T* __restrict__ wp, long n_rows, int C, int CP) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * CP; i += (long)gridDim.x * blockDim.x) {
    const long row = i / CP;
    const int c = (int)(i - row * CP);
    wp[i] = Elem<T>::from_f(c < C ? w[row * C + c] : 0.f);
  }
}

__global__ void unpack_stem_dw_kernel(const float* __restrict__ dwp, float* __restrict__ dw, long n_rows, int C, int CP, int accum) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * C; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    const float v = dwp[row * CP + c];
    dw[i] = accum ? dw[i] + v : v;
  }
}

// ---- the ImageNet stem Conv2d(3 -> K, 7 x 7, stride 2, padding 3) (resnet.py:69-75 under the "c3,K,7,2,3" specs) as a 4 x 4 / stride-1 / VALID convolution
// over a space-to-depth image.  s2d pixel (i, j) holds the 2 x 2 block of image pixels (2i + dy, 2j + dx) as 16 channels (dy, dx, c of 4: three image
// channels and a zero); output (p, q) reads s2d rows p - 2 .. p + 1 and columns q - 2 .. q + 1, so the image is stored with 2 zero s2d pixels before and 1
// after in both directions ([N][H/2 + 3][W/2 + 3][16]) and the convolution needs no padding at all.  Kernel tap (r', s'), channel (dy, dx, c) is the 7 x 7
// weight at (2r' + dy - 1, 2s' + dx - 1), zero outside.  Against one 16-byte chunk per image pixel (C padded 3 -> 8, K = 49 taps x 8 -> 7 K tiles of 64) the
// contraction is 16 taps x 16 channels = 4 K tiles, and the 4 taps of a kernel row are 128 contiguous bytes: the eight-phase kernel's ordinary row copy.
template <typename T>
__global__ __launch_bounds__(NT) void img_to_s2d_kernel(const float* __restrict__ x, T* __restrict__ out, int N, int C, int H, int W, int HP, int WP) {
  const long n = (long)N * HP * WP;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const int wp = (int)(i % WP), hp = (int)((i / WP) % HP);
    const long nn = i / ((long)WP * HP);
    T* o = out + i * 16;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int h = 2 * (hp - 2) + dy, w = 2 * (wp - 2) + dx;
        const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[(dy * 2 + dx) * 4 + c] = Elem<T>::from_f((in && c < C) ? x[((nn * C + c) * H + h) * (long)W + w] : 0.f);
      }
  }
}
// index of 7 x 7 weight element (r, s, c) inside a [4][4][16] s2d filter, or of filter element j the (r, s, c) it holds (r < 0: a structural zero)
__device__ inline int s2d_of_rsc(int r, int s, int c) { return ((r + 1) >> 1) * 64 + ((s + 1) >> 1) * 16 + ((((r + 1) & 1) << 1) | ((s + 1) & 1)) * 4 + c; }
template <typename T>
__global__ void pack_stem_w_s2d_kernel(const float* __restrict__ w, T* __restrict__ wp, int K, int C) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)K * 256; i += (long)gridDim.x * blockDim.x) {
    const long k = i >> 8;
    const int j = (int)(i & 255), rp = j >> 6, sp = (j >> 4) & 3, dy = (j >> 3) & 1, dx = (j >> 2) & 1, c = j & 3;
    const int r = 2 * rp + dy - 1, t = 2 * sp + dx - 1;
    wp[i] = Elem<T>::from_f(((unsigned)r < 7u && (unsigned)t < 7u && c < C) ? w[((k * 7 + r) * 7 + t) * C + c] : 0.f);
  }
}
__global__ void unpack_stem_dw_s2d_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int K, int C, int accum) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)K * 49 * C; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long rs = i / C, k = rs / 49;
    const int r = (int)((rs % 49) / 7), t = (int)(rs % 7);
    const float v = dwp[k * 256 + s2d_of_rsc(r, t, c)];
    dw[i] = accum ? dw[i] + v : v;
  }
}

// ---- MaxPool2d(k, s, p), NHWC, -inf padding.  The forward also stores the argmax (window position r*k+s, first maximum
// in scan order wins -- torch's rule) as one byte per output element; the backward is then a gather: an input element
// sums dy of the (at most ceil(k/s)^2) windows whose stored argmax points at it.  No atomics, no zero-fill pass.
template <typename T>
__global__ __launch_bounds__(NT) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx, int N, int H, int W,
                                                         int C, int P, int Q, int k, int stride, int pad) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  const long n = (long)N * P * Q * CC;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const int cg = (int)(i % CC);
    long pix = i / CC;
    const int q = (int)(pix % Q); pix /= Q;
    const int p = (int)(pix % P);
    const int nn = (int)(pix / P);
    float m[CE];
    __attribute__((aligned(8))) unsigned char am[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { m[e] = -FLT_MAX; am[e] = 255; }
    for (int r = 0; r < k; ++r) {
      const int h = p * stride + r - pad;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int s = 0; s < k; ++s) {
        const int w = q * stride + s - pad;
        if ((unsigned)w >= (unsigned)W) continue;
        Chunk<T> c = load_chunk<T>(x + (((size_t)nn * H + h) * W + w) * C + cg * CE);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          const float v = Elem<T>::to_f(c.e[e]);
          if (v > m[e] || am[e] == 255) { m[e] = v; am[e] = (unsigned char)(r * k + s); }
        }
      }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < CE; ++e) o.e[e] = Elem<T>::from_f(m[e]);
    store_chunk<T>(y + i * CE, o);
    if (idx) {                                             // the chunk's argmax bytes leave in ONE store (the backward reads them the same way)
      if constexpr (CE == 8) *reinterpret_cast<uint2*>(idx + i * CE) = *reinterpret_cast<const uint2*>(am);
      else *reinterpret_cast<unsigned*>(idx + i * CE) = *reinterpret_cast<const unsigned*>(am);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx, T* __restrict__ dx, int N, int H,
                                                         int W, int C, int P, int Q, int k, int stride, int pad) {
  // gather form: an input pixel collects from the <= ceil(k/stride)^2 windows that cover it.  One workgroup walks image
  // rows (n, h); threads walk (w, channel chunk) with 32-bit index math; the argmax bytes of a chunk come in ONE load.
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  const int per_row = W * CC;
  for (int row = blockIdx.x; row < N * H; row += gridDim.x) {
    const int nn = row / H, h = row - nn * H;
    const int p_lo = max(0, (h + pad - k + stride) / stride), p_hi = min(P - 1, (h + pad) / stride);
    for (int j = threadIdx.x; j < per_row; j += NT) {
      const int w = j / CC, cg = j - w * CC;
      const int q_lo = max(0, (w + pad - k + stride) / stride), q_hi = min(Q - 1, (w + pad) / stride);
      float g[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) g[e] = 0.f;
      auto take = [&](int p, int q, const Chunk<T>& d, const unsigned char* ib) {
        const unsigned me = (unsigned)((h - (p * stride - pad)) * k + (w - (q * stride - pad)));     // my position inside window (p, q)
#pragma unroll
        for (int e = 0; e < CE; ++e) if (ib[e] == me) g[e] += Elem<T>::to_f(d.e[e]);
      };
      auto fetch = [&](int p, int q, Chunk<T>& d, unsigned char* ib) {
        const size_t o = (((size_t)nn * P + p) * Q + q) * C + cg * CE;
        d = load_chunk<T>(dy + o);
        if constexpr (CE == 8) { *reinterpret_cast<uint2*>(ib) = *reinterpret_cast<const uint2*>(idx + o); }
        else { *reinterpret_cast<unsigned*>(ib) = *reinterpret_cast<const unsigned*>(idx + o); }
      };
      if (p_hi - p_lo <= 1 && q_hi - q_lo <= 1) {            // <= 2 x 2 windows (3x3 stride 2): all loads issued before the first use
        Chunk<T> d[4];
        unsigned char ib[4][CE];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = p_lo + (u >> 1), q = q_lo + (u & 1);
          ok[u] = p <= p_hi && q <= q_hi;
          if (ok[u]) fetch(p, q, d[u], ib[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ok[u]) take(p_lo + (u >> 1), q_lo + (u & 1), d[u], ib[u]);
      } else {
        for (int p = p_lo; p <= p_hi; ++p)
          for (int q = q_lo; q <= q_hi; ++q) {
            Chunk<T> d;
            unsigned char ib[CE];
            fetch(p, q, d, ib);
            take(p, q, d, ib);
          }
      }
      Chunk<T> o;
#pragma unroll
      for (int e = 0; e < CE; ++e) o.e[e] = Elem<T>::from_f(g[e]);
      store_chunk<T>(dx + ((size_t)row * W + w) * C + cg * CE, o);
    }
  }
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  A pooling pass re-reads its neighbours' rows (windows overlap),
// so give every XCD a contiguous band of the launch's work instead of every 8th piece: the re-read then hits the L2 that fetched it.
__device__ inline int xcd_band(int b, int G) { return (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3); }

// ---- BatchNorm-apply + ReLU + MaxPool as ONE pass each way (the stem of the ImageNet nets, resnet.py:111-115 + 83-87: "n a mp3,2,1").
// Unfused, the 112x112x512 activation of WRN-50-2-B (3.3 GB at batch 256) is written by bn_apply, read by maxpool, and in the backward
// written by maxpool_bwd, read by bn_bwd_reduce and read again by bn_bwd_apply.  Here the normalised activation and its gradient never
// exist in memory: the forward pools relu(x * scale + shift) on the fly (each value rounded to the compute dtype BEFORE the comparison,
// so the argmax is the one the unfused pair picks), the backward gathers the pooled gradient through the stored argmax bytes where the
// unfused pair would read maxpool_bwd's output.
// Waves per SIMD the two register-heavy forms are asked to fit (second __launch_bounds__ argument; 1 = the compiler's own choice).  fp16 only, where it
// was measured on the 112 x 112 x 512 map of WRN-50-2-B: the forward that also keeps the winners sat at 130 registers (3 waves; 2.16 ms), at 128 with two
// spilled dwords it runs 4 waves (1.99 ms); the apply pass with column sums 177 registers (2 waves; 1.98 ms) -> 168 + two spilled dwords, 3 waves (1.75 ms).
// bf16 needs more temporaries (44 bytes of scratch under the same request, 2.6 ms): left to the compiler.
template <typename T> struct PoolWaves { static constexpr int FWD = 1, APPLY_SUMS = 1; };
template <> struct PoolWaves<f16_t> { static constexpr int FWD = 4, APPLY_SUMS = 3; };

template <typename T, int KS>                               // KS = 3: the 3 x 3 window unrolled, its nine loads issued before the first comparison
__global__ __launch_bounds__(NT, (KS == 3 ? PoolWaves<T>::FWD : 1)) void bn_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ coef, T* __restrict__ y, unsigned char* __restrict__ idx,
                                                         T* __restrict__ xsel, int N, int H, int W, int C, int P, int Q, int k, int stride, int pad, int relu) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  const long n = (long)N * P * Q * CC;
  for (long i = (long)xcd_band(blockIdx.x, gridDim.x) * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const int cg = (int)(i % CC);
    long pix = i / CC;
    const int q = (int)(pix % Q); pix /= Q;
    const int p = (int)(pix % P);
    const int nn = (int)(pix / P);
    float sc[CE], sh[CE], m[CE];
    __attribute__((aligned(8))) unsigned char am[CE];
    Chunk<T> win;                                          // the input element that won each window (xsel: the backward's sums read it instead of the whole map)
#pragma unroll
    for (int e = 0; e < CE; ++e) { sc[e] = coef[cg * CE + e]; sh[e] = coef[C + cg * CE + e]; m[e] = -FLT_MAX; am[e] = 255; win.e[e] = Elem<T>::from_f(0.f); }
    auto offer = [&](const Chunk<T>& c, int pos) {
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        float v = fmaf(Elem<T>::to_f(c.e[e]), sc[e], sh[e]);
        if (relu) v = fmaxf(v, 0.f);
        v = Elem<T>::to_f(Elem<T>::from_f(v));              // what bn_apply would have stored
        if (v > m[e] || am[e] == 255) { m[e] = v; am[e] = (unsigned char)pos; win.e[e] = c.e[e]; }
      }
    };
    if constexpr (KS == 3) {
      Chunk<T> c[9];
      bool ok[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int h = p * stride + t / 3 - pad, w = q * stride + t % 3 - pad;
        ok[t] = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        if (ok[t]) c[t] = load_chunk<T>(x + (((size_t)nn * H + h) * W + w) * C + cg * CE);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (ok[t]) offer(c[t], t);
    } else {
      for (int r = 0; r < k; ++r) {
        const int h = p * stride + r - pad;
        if ((unsigned)h >= (unsigned)H) continue;
        for (int s = 0; s < k; ++s) {
          const int w = q * stride + s - pad;
          if ((unsigned)w >= (unsigned)W) continue;
          offer(load_chunk<T>(x + (((size_t)nn * H + h) * W + w) * C + cg * CE), r * k + s);
        }
      }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < CE; ++e) o.e[e] = Elem<T>::from_f(m[e]);
    store_chunk<T>(y + i * CE, o);
    if (idx) {
      if constexpr (CE == 8) *reinterpret_cast<uint2*>(idx + i * CE) = *reinterpret_cast<const uint2*>(am);
      else *reinterpret_cast<unsigned*>(idx + i * CE) = *reinterpret_cast<const unsigned*>(am);
    }
    if (xsel) store_chunk<T>(xsel + i * CE, win);
  }
}

// backward, gather form (as maxpool_bwd_kernel: one workgroup walks image rows, threads walk (w, channel chunk); NT % (C / CE) == 0, so a
// thread keeps ONE channel chunk).  g = [x * scale + shift > 0] * sum of dy over the windows whose argmax is this element.
// APPLY 0: partial[blockIdx][2][C] = (sum g, sum g * xhat);   APPLY 1: dx = scale * (g - dsum0 / count - xhat * dsum1 / count)
// APPLY 2: as 1, and partial[blockIdx][2][C] = (sum dx, 0) of the STORED values -- the per-channel sums a biased producer (the
// ImageNet stem convolution) needs for its bias gradient, which otherwise cost one more pass over the 3.3 GB gradient (bn_stats)
template <typename T, int APPLY>
struct BnPoolBack {                                         // what a thread does with one input pixel's chunk once its gathered gradient is known
  static constexpr int CE = Elem<T>::CE;
  float sc[CE], sh[CE], mean[CE], invstd[CE], ka[CE], kb[CE], kc[CE], s0[CE], s1[CE];
  __device__ void init(const float* __restrict__ coef, const float* __restrict__ dsum, int C, int cg, int train, float inv_count) {
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      const int c = cg * CE + e;
      sc[e] = coef[c]; sh[e] = coef[C + c]; mean[e] = coef[2 * C + c]; invstd[e] = coef[3 * C + c];
      s0[e] = s1[e] = 0.f;
      if (APPLY) {                                         // the coefficients of bn_bwd_apply_stream_kernel
        const float m0 = train ? __fmul_rn(dsum[c], inv_count) : 0.f, m1 = train ? __fmul_rn(dsum[C + c], inv_count) : 0.f;
        const float im1 = __fmul_rn(invstd[e], m1);
        ka[e] = sc[e];
        kb[e] = train ? -__fmul_rn(sc[e], im1) : 0.f;      // eval-mode BatchNorm: dx = scale * g
        kc[e] = train ? __fmul_rn(sc[e], __fmaf_rn(im1, mean[e], -m0)) : 0.f;
      }
    }
  }
  __device__ void pixel(const Chunk<T>& cx, const float (&g)[CE], int relu, T* __restrict__ dx_at) {
    Chunk<T> co;
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      const float xv = Elem<T>::to_f(cx.e[e]);
      // maxpool_bwd stores its output in the compute dtype before bn_bwd reads it: the same rounding here
      float gg = Elem<T>::to_f(Elem<T>::from_f(g[e]));
      if (relu && !(fmaf(xv, sc[e], sh[e]) > 0.f)) gg = 0.f;
      if (APPLY) {
        co.e[e] = Elem<T>::from_f(__fmaf_rn(ka[e], gg, __fmaf_rn(kb[e], xv, kc[e])));
        if (APPLY == 2) s0[e] += Elem<T>::to_f(co.e[e]);   // (the second row of the partials stays 0: nobody reads a sum of squares of this gradient)
      } else {
        const float xh = (xv - mean[e]) * invstd[e];
        s0[e] += gg; s1[e] += gg * xh;
      }
    }
    if (APPLY) store_chunk<T>(dx_at, co);
  }
  __device__ void window(const Chunk<T>& cx, const Chunk<T>& cd, int relu) {     // one pooling window: its winner's input value and its gradient (APPLY 0)
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      const float xv = Elem<T>::to_f(cx.e[e]);
      float gg = Elem<T>::to_f(cd.e[e]);
      if (relu && !(fmaf(xv, sc[e], sh[e]) > 0.f)) gg = 0.f;
      s0[e] += gg; s1[e] += gg * ((xv - mean[e]) * invstd[e]);
    }
  }
  __device__ void flush(float* __restrict__ partial, int C, int CC, int cg) {      // the workgroup's row of partial sums: lanes of one chunk through LDS
    __shared__ float red[2][NT][CE + 1];
#pragma unroll
    for (int e = 0; e < CE; ++e) { red[0][threadIdx.x][e] = s0[e]; red[1][threadIdx.x][e] = s1[e]; }
    __syncthreads();
    if ((int)threadIdx.x < CC) {
      float t0[CE], t1[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) t0[e] = t1[e] = 0.f;
      for (int l = threadIdx.x; l < NT; l += CC) {
#pragma unroll
        for (int e = 0; e < CE; ++e) { t0[e] += red[0][l][e]; t1[e] += red[1][l][e]; }
      }
      float* p0 = partial + ((size_t)blockIdx.x * 2) * C + cg * CE;
#pragma unroll
      for (int e = 0; e < CE; ++e) { p0[e] = t0[e]; p0[C + e] = t1[e]; }
    }
  }
};
template <typename T> struct PoolTap {                      // a pooled position's gradient chunk and argmax bytes
  static constexpr int CE = Elem<T>::CE;
  Chunk<T> d;
  __attribute__((aligned(8))) unsigned char ib[CE];
  __device__ void fetch(const T* __restrict__ dy, const unsigned char* __restrict__ idx, size_t o) {
    d = load_chunk<T>(dy + o);
    if constexpr (CE == 8) *reinterpret_cast<uint2*>(ib) = *reinterpret_cast<const uint2*>(idx + o);
    else *reinterpret_cast<unsigned*>(ib) = *reinterpret_cast<const unsigned*>(idx + o);
  }
  __device__ void take(unsigned me, float (&g)[CE]) const {
#pragma unroll
    for (int e = 0; e < CE; ++e) if (ib[e] == me) g[e] += Elem<T>::to_f(d.e[e]);
  }
};

template <typename T, int APPLY>
__global__ __launch_bounds__(NT) void bn_pool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx, const T* __restrict__ x,
                                                         const float* __restrict__ coef, const float* __restrict__ dsum, float* __restrict__ partial,
                                                         T* __restrict__ dx, int N, int H, int W, int C, int P, int Q, int k, int stride, int pad, int relu,
                                                         int train, float inv_count) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  const int per_row = W * CC;
  const int cg = threadIdx.x % CC;                         // fixed: NT is a multiple of CC
  BnPoolBack<T, APPLY> bk;
  bk.init(coef, dsum, C, cg, train, inv_count);
  for (int row = blockIdx.x; row < N * H; row += gridDim.x) {
    const int nn = row / H, h = row - nn * H;
    const int p_lo = max(0, (h + pad - k + stride) / stride), p_hi = min(P - 1, (h + pad) / stride);
    for (int j = threadIdx.x; j < per_row; j += NT) {
      const int w = j / CC;
      const int q_lo = max(0, (w + pad - k + stride) / stride), q_hi = min(Q - 1, (w + pad) / stride);
      const size_t xo = ((size_t)row * W + w) * C + cg * CE;
      const Chunk<T> cx = load_chunk<T>(x + xo);
      float g[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) g[e] = 0.f;
      for (int p = p_lo; p <= p_hi; ++p)
        for (int q = q_lo; q <= q_hi; ++q) {
          PoolTap<T> t;
          t.fetch(dy, idx, (((size_t)nn * P + p) * Q + q) * C + cg * CE);
          t.take((unsigned)((h - (p * stride - pad)) * k + (w - (q * stride - pad))), g);
        }
      bk.pixel(cx, g, relu, dx + xo);
    }
  }
  if (APPLY != 1) bk.flush(partial, C, CC, cg);
}

// the geometry every shipped spec has ("mp3,2,1" on an even map): a thread owns the 2 x 2 input pixels (2a..2a+1, 2b..2b+1) of one channel
// chunk, which are covered by exactly the pooled positions (a..a+1, b..b+1) -- 4 gradient + 4 argmax loads for 4 pixels where the
// per-pixel gather issues up to 4 for one, and all 12 loads are in flight before the first use.  A pixel adds its windows in the same
// order (p major, q minor) as the per-pixel kernel, so both give the same bits.
template <typename T, int APPLY>
__global__ __launch_bounds__(NT, (APPLY == 2 ? PoolWaves<T>::APPLY_SUMS : 1)) void bn_pool_bwd_quad_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx, const T* __restrict__ x,
                                                              const float* __restrict__ coef, const float* __restrict__ dsum, float* __restrict__ partial,
                                                              T* __restrict__ dx, int N, int H, int W, int C, int relu, int train, float inv_count) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE, P = H >> 1, Q = W >> 1;
  const int per_row = Q * CC;
  const int cg = threadIdx.x % CC;
  BnPoolBack<T, APPLY> bk;
  bk.init(coef, dsum, C, cg, train, inv_count);
  for (int row = xcd_band(blockIdx.x, gridDim.x); row < N * P; row += gridDim.x) {
    const int nn = row / P, a = row - nn * P;
    const bool down = a + 1 < P;
    for (int j = threadIdx.x; j < per_row; j += NT) {
      const int b = j / CC;
      const bool right = b + 1 < Q;
      const size_t x00 = (((size_t)nn * H + 2 * a) * W + 2 * b) * C + cg * CE, x10 = x00 + (size_t)W * C;
      const size_t o00 = ((size_t)row * Q + b) * C + cg * CE, o10 = o00 + (size_t)Q * C;
      Chunk<T> cx[4] = {load_chunk<T>(x + x00), load_chunk<T>(x + x00 + C), load_chunk<T>(x + x10), load_chunk<T>(x + x10 + C)};
      PoolTap<T> t[4];
      t[0].fetch(dy, idx, o00);
      if (right) t[1].fetch(dy, idx, o00 + C);
      if (down) t[2].fetch(dy, idx, o10);
      if (down && right) t[3].fetch(dy, idx, o10 + C);
      float g[4][CE];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < CE; ++e) g[u][e] = 0.f;
      // position of a pixel inside a 3 x 3 window whose top-left is (2p - 1, 2q - 1): row 2a -> r = 1 in p = a; row 2a + 1 -> r = 2 in a, r = 0 in a + 1
      t[0].take(4, g[0]);
      t[0].take(5, g[1]); if (right) t[1].take(3, g[1]);
      t[0].take(7, g[2]); if (down) t[2].take(1, g[2]);
      t[0].take(8, g[3]); if (right) t[1].take(6, g[3]);
      if (down) t[2].take(2, g[3]);
      if (down && right) t[3].take(0, g[3]);
      bk.pixel(cx[0], g[0], relu, dx + x00);
      bk.pixel(cx[1], g[1], relu, dx + x00 + C);
      bk.pixel(cx[2], g[2], relu, dx + x10);
      bk.pixel(cx[3], g[3], relu, dx + x10 + C);
    }
  }
  if (APPLY != 1) bk.flush(partial, C, CC, cg);
}

// the backward sums at POOLED resolution: with the winning input element of every window kept by the forward (xsel), sum g and sum g * xhat are sums
// over the windows -- g of an input element is the sum of the pooled gradients whose argmax it is, masked by ITS sign, and every one of those windows
// stored that same element -- so the pass reads the pooled gradient and xsel (2 x 1/4 of the map) instead of the map, the gradient and the argmax
// bytes.  (The gather kernels round a pixel's summed gradient to the compute dtype before using it, as the unfused chain stores it; here the windows
// are added unrounded: a difference below the dtype's rounding per element, none in fp32.)  Row b of partial = windows b, b + nblk, ... in blocks of
// NT / (C / CE) pixels; NT % (C / CE) == 0, so a thread keeps one channel chunk.
template <typename T>
__global__ __launch_bounds__(NT) void bn_pool_bwd_reduce_sel_kernel(const T* __restrict__ dy, const T* __restrict__ xsel, const float* __restrict__ coef,
                                                                    float* __restrict__ partial, long npix, int C, int relu) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE, cg = threadIdx.x % CC, lanes = NT / CC;
  BnPoolBack<T, 0> bk;
  bk.init(coef, nullptr, C, cg, 1, 0.f);
  const long step = (long)gridDim.x * lanes;
  long p = (long)xcd_band(blockIdx.x, gridDim.x) * lanes + threadIdx.x / CC;
  for (; p + 3 * step < npix; p += 4 * step) {                // four windows in flight per thread (adds in window order)
    Chunk<T> d[4], xs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t o = (size_t)(p + u * step) * C + cg * CE; d[u] = load_chunk<T>(dy + o); xs[u] = load_chunk<T>(xsel + o); }
#pragma unroll
    for (int u = 0; u < 4; ++u) bk.window(xs[u], d[u], relu);
  }
  for (; p < npix; p += step) {
    const size_t o = (size_t)p * C + cg * CE;
    bk.window(load_chunk<T>(xsel + o), load_chunk<T>(dy + o), relu);
  }
  bk.flush(partial, C, CC, cg);
}

// ---- global average pool: feat[n][c] = mean_{hw} x[n][hw][c] ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void gap_kernel(const T* __restrict__ x, float* __restrict__ feat, int HW, int C) {
  constexpr int CE = Elem<T>::CE;
  __shared__ float red[NT][CE + 1];
  const int CC = C / CE;
  const int n = blockIdx.x;
  const int cols = CC >= NT ? NT : CC, lanes = CC >= NT ? 1 : NT / CC;
  for (int cbase = 0; cbase < CC; cbase += cols) {
    const int cg = cbase + threadIdx.x % cols, rl = threadIdx.x / cols;
    float s[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) s[e] = 0.f;
    if (rl < lanes && cg < CC)
#pragma unroll 8
      for (int p = rl; p < HW; p += lanes) {               // unrolled: the loads of eight pixels in flight (adds stay in pixel order)
        Chunk<T> c = load_chunk<T>(x + ((size_t)n * HW + p) * C + cg * CE);
#pragma unroll
        for (int e = 0; e < CE; ++e) s[e] += Elem<T>::to_f(c.e[e]);
      }
#pragma unroll
    for (int e = 0; e < CE; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    if (threadIdx.x < cols && cg < CC) {
      float t[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) t[e] = 0.f;
      for (int l = 0; l < lanes; ++l)
#pragma unroll
        for (int e = 0; e < CE; ++e) t[e] += red[l * cols + threadIdx.x][e];
#pragma unroll
      for (int e = 0; e < CE; ++e) feat[(size_t)n * C + cg * CE + e] = t[e] / (float)HW;
    }
    __syncthreads();
  }
}

// one wave per (n, o): logits[n][o] = b[o] + <W[o,:], feat[n,:]>
__global__ __launch_bounds__(NT) void fc_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ b,
                                                    float* __restrict__ logits, int N, int C, int O) {
  const long wid = ((long)blockIdx.x * NT + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wid >= (long)N * O) return;
  const int n = (int)(wid / O), o = (int)(wid - (long)n * O);
  float s = 0.f;
#pragma unroll 4
  for (int c = lane; c < C; c += 64) s = fmaf(feat[(size_t)n * C + c], w[(size_t)o * C + c], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) logits[(size_t)n * O + o] = s + b[o];
}

// dW[o][c] = sum_n dl[n][o] * feat[n][c]; db[o] = sum_n dl[n][o].  A workgroup owns 64 (o, c) outputs (c fastest -> coalesced); its four
// waves each take a quarter of the batch, eight rows in flight at a time, and meet in LDS in wave order (a CIFAR head is 640-6400 outputs:
// one thread per output walking the whole batch left the chip idle for 28 us, waiting on 128 dependent round trips)
__global__ __launch_bounds__(NT) void fc_wgrad_kernel(const float* __restrict__ dl, const float* __restrict__ feat, float* __restrict__ dw,
                                                      float* __restrict__ db, int N, int C, int O, int accum) {
  __shared__ float red[2][NT / 64][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  const bool live = i < (long)O * C;
  const int o = live ? (int)(i / C) : 0, c = live ? (int)(i - (long)o * C) : 0;
  const int per = (N + NT / 64 - 1) / (NT / 64), n_lo = part * per, n_hi = min(N, n_lo + per);
  float s = 0.f, sb = 0.f;
#pragma unroll 8
  for (int n = n_lo; n < n_hi; ++n) {
    const float d = dl[(size_t)n * O + o];
    s = fmaf(d, feat[(size_t)n * C + c], s);
    sb += d;
  }
  red[0][part][lane] = s; red[1][part][lane] = sb;
  __syncthreads();
  if (part == 0 && live) {
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) { s += red[0][w][lane]; sb += red[1][w][lane]; }
    dw[i] = accum ? dw[i] + s : s;
    if (c == 0) db[o] = accum ? db[o] + sb : sb;
  }
}

// dx[n][hw][c] = (sum_o dl[n][o] * W[o][c]) / HW, broadcast over hw
template <typename T>
__global__ __launch_bounds__(NT) void fc_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ w, T* __restrict__ dx, int HW, int C, int O) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  const int n = blockIdx.x;
  for (int cg = threadIdx.x; cg < CC; cg += NT) {
    float s[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) s[e] = 0.f;
#pragma unroll 4
    for (int o = 0; o < O; ++o) {
      const float d = dl[(size_t)n * O + o];
#pragma unroll
      for (int e = 0; e < CE; ++e) s[e] = fmaf(d, w[(size_t)o * C + cg * CE + e], s[e]);
    }
    Chunk<T> c;
#pragma unroll
    for (int e = 0; e < CE; ++e) c.e[e] = Elem<T>::from_f(s[e] / (float)HW);
    for (int p = 0; p < HW; ++p) store_chunk<T>(dx + ((size_t)n * HW + p) * C + cg * CE, c);
  }
}

// ---- wide classifier heads (ImageNet: O = 1000, C = 2048): the three products above as one LDS-tiled fp32 GEMM --------------------
// D[m][n] = sum_k A(m, k) * B(n, k), 64 x 64 tile per workgroup, 4 x 4 per thread, 16-deep k steps through LDS with a register prefetch
// of the next step.  AK / BK: the operand is k-contiguous (row stride ld over m / n) or m- / n-contiguous (row stride ld over k).
// The wave-per-output kernels above re-read an operand per output (4 GB of L2 traffic for 256 x 1000 x 2048) and stay for O < 128.
constexpr int FG_T = 64, FG_K = 16, FG_LD = FG_T + 4;
template <bool KC>
__device__ inline void fg_load(const float* __restrict__ p, int ld, int t0, int k0, int TD, int K, float (&r)[4]) {
  if constexpr (KC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + (threadIdx.x & 15), t = t0 + (threadIdx.x >> 4) + 16 * i;
      r[i] = (t < TD && k < K) ? p[(size_t)t * ld + k] : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + (threadIdx.x & 63), k = k0 + (threadIdx.x >> 6) + 4 * i;
      r[i] = (t < TD && k < K) ? p[(size_t)k * ld + t] : 0.f;
    }
  }
}
template <bool KC>
__device__ inline void fg_store(float (*sm)[FG_LD], const float (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (KC) sm[threadIdx.x & 15][(threadIdx.x >> 4) + 16 * i] = r[i];
    else sm[(threadIdx.x >> 6) + 4 * i][threadIdx.x & 63] = r[i];
  }
}
template <typename T> struct alignas(sizeof(T) * 4) Quad { T e[4]; };
// MODE 0: logits[m][n] = D + bias[n]          (A = feat [N][C], B = W [O][C]; M = batch, N = O, K = C)
// MODE 1: dx[m][p][n] = D / HW for all p      (A = dl [N][O],  B = W [O][C] n-contiguous; M = batch, N = C, K = O)
// MODE 2: dw[m][n] (+)= D, db[m] (+)= sum_k A (A = dl m-contiguous, B = feat n-contiguous; M = O, N = C, K = batch)
template <typename T, int MODE, bool AK, bool BK>
__global__ __launch_bounds__(NT) void fc_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, int M, int N, int K, int lda, int ldb,
                                                     void* __restrict__ out, const float* __restrict__ bias, float* __restrict__ db, int HW, int accum) {
  __shared__ float As[FG_K][FG_LD], Bs[FG_K][FG_LD];
  const int m0 = blockIdx.y * FG_T, n0 = blockIdx.x * FG_T;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[4][4] = {};
  float ra[4], rb[4];
  fg_load<AK>(A, lda, m0, 0, M, K, ra);
  fg_load<BK>(B, ldb, n0, 0, N, K, rb);
  for (int k0 = 0; k0 < K; k0 += FG_K) {
    __syncthreads();
    fg_store<AK>(As, ra);
    fg_store<BK>(Bs, rb);
    __syncthreads();
    if (k0 + FG_K < K) {
      fg_load<AK>(A, lda, m0, k0 + FG_K, M, K, ra);
      fg_load<BK>(B, ldb, n0, k0 + FG_K, N, K, rb);
    }
#pragma unroll
    for (int k = 0; k < FG_K; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
  }
  const int n = n0 + tx * 4;                    // N % 4 == 0 (checked by the launcher): a thread's four columns are all in or all out
  if (n < N) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + ty * 4 + i;
      if (m >= M) continue;
      if constexpr (MODE == 0) {
        float* o = reinterpret_cast<float*>(out) + (size_t)m * N + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = acc[i][j] + bias[n + j];
      } else if constexpr (MODE == 1) {
        Quad<T> q;
#pragma unroll
        for (int j = 0; j < 4; ++j) q.e[j] = Elem<T>::from_f(acc[i][j] / (float)HW);
        T* o = reinterpret_cast<T*>(out) + (size_t)m * HW * N + n;
        for (int p = 0; p < HW; ++p) *reinterpret_cast<Quad<T>*>(o + (size_t)p * N) = q;
      } else {
        float* o = reinterpret_cast<float*>(out) + (size_t)m * N + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = accum ? o[j] + acc[i][j] : acc[i][j];
      }
    }
  }
  if constexpr (MODE == 2) {
    if (blockIdx.x == 0 && threadIdx.x < FG_T && m0 + (int)threadIdx.x < M) {      // db[o] = sum_n dl[n][o], rows of A in order
      const int m = m0 + threadIdx.x;
      float sb = 0.f;
      for (int k = 0; k < K; ++k) sb += A[(size_t)k * lda + m];
      db[m] = accum ? db[m] + sb : sb;
    }
  }
}
static inline bool fc_wide(int C, int O) { return O >= 128 && C % 4 == 0; }   // CIFAR heads (O = 10, 100) are a few microseconds either way

// ---- softmax cross-entropy, top-1 / top-5 error counts, dlogits (single block: N is a batch, a few hundred rows) ----
// rank rule for ties (torch.topk leaves it implementation-defined): an entry outranks the label iff it is strictly
// greater, or equal with a lower index.
constexpr int SM_NT = 1024, SM_NW = SM_NT / 64;
template <typename V> __device__ inline V wave_all(V v, V (*f)(V, V)) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = f(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ inline float f_add(float a, float b) { return a + b; }
__device__ inline float f_max(float a, float b) { return fmaxf(a, b); }
__device__ inline int i_add(int a, int b) { return a + b; }
// one workgroup of 16 waves, a wave per row, lanes stride the classes (coalesced).  Rows of up to 1024 classes are held in registers: one
// batch of 16 independent loads, then the maximum, the exponentials, the rank of the label and dlogits all come from registers.  The sums
// over rows stay in one workgroup, in a fixed order, so the three outputs are deterministic.
constexpr int SM_R = 16;
__global__ __launch_bounds__(SM_NT) void softmax_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, float* __restrict__ out3,
                                                           float* __restrict__ dlogits, int N, int O, float scale, const float* __restrict__ scale_dev) {
  __shared__ float red[3][SM_NW];
  if (scale_dev) scale *= scale_dev[0];          // loss scale (AMP) / upstream gradient of the loss, a device scalar: no host sync
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool in_regs = O <= 64 * SM_R;
  float nll = 0.f, e1 = 0.f, e5 = 0.f;
  if (O <= 32) {                                 // CIFAR heads: a row takes G = 4..32 lanes, a wave holds 64 / G rows at once
    int G = 4;
    while (G < O) G <<= 1;
    const int sub = lane & (G - 1), per_wave = 64 / G;
    for (int n0 = wave * per_wave; n0 < N; n0 += SM_NW * per_wave) {
      const int n = n0 + lane / G;
      const bool live = n < N, mine = live && sub < O;
      // a label outside [0, O) (torch's cross_entropy device-asserts there) must not become an out-of-range read: the row's loss turns into NaN instead
      const long long lab_ll = live ? labels[n] : 0;
      const bool bad = (unsigned long long)lab_ll >= (unsigned long long)O;
      const int lab = bad ? 0 : (int)lab_ll;
      const float zl = !live ? 0.f : bad ? __builtin_nanf("") : logits[(size_t)n * O + lab];
      const float z = mine ? logits[(size_t)n * O + sub] : -FLT_MAX;
      float mx = z;
      for (int off = G >> 1; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float ez = mine ? expf(z - mx) : 0.f;
      float se = ez;
      int ahead = mine && ((z > zl) || (z == zl && sub < lab));
      for (int off = G >> 1; off > 0; off >>= 1) { se += __shfl_xor(se, off, 64); ahead += __shfl_xor(ahead, off, 64); }
      if (live && sub == 0) {
        nll += logf(se) - (zl - mx);
        e1 += ahead >= 1 ? 1.f : 0.f;
        e5 += ahead >= (O < 5 ? O : 5) ? 1.f : 0.f;
      }
      if (dlogits && mine) dlogits[(size_t)n * O + sub] = (ez / se - (sub == lab ? 1.f : 0.f)) * scale;
    }
    for (int off = 32; off >= G; off >>= 1) {    // the rows of a wave: only lanes with sub == 0 carry a value, and xor by a multiple of G keeps sub
      nll += __shfl_xor(nll, off, 64); e1 += __shfl_xor(e1, off, 64); e5 += __shfl_xor(e5, off, 64);
    }
    N = 0;                                       // the wide loop below has nothing left
  }
  for (int n = wave; n < N; n += SM_NW) {
    const float* row = logits + (size_t)n * O;
    const long long lab_ll = labels[n];
    const bool bad = (unsigned long long)lab_ll >= (unsigned long long)O;          // see above: NaN loss, never an out-of-range read
    const int lab = bad ? 0 : (int)lab_ll;
    const float zl = bad ? __builtin_nanf("") : row[lab];
    float z[SM_R];
    float mx = -FLT_MAX, se = 0.f;
    int ahead = 0;
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < SM_R; ++i) { const int o = lane + 64 * i; z[i] = o < O ? row[o] : -FLT_MAX; }
#pragma unroll
      for (int i = 0; i < SM_R; ++i) mx = fmaxf(mx, z[i]);
      mx = wave_all(mx, f_max);
#pragma unroll
      for (int i = 0; i < SM_R; ++i) {
        const int o = lane + 64 * i;
        if (o < O) {
          ahead += (z[i] > zl) || (z[i] == zl && o < lab);
          z[i] = expf(z[i] - mx);
          se += z[i];
        }
      }
    } else {
      for (int o = lane; o < O; o += 64) mx = fmaxf(mx, row[o]);
      mx = wave_all(mx, f_max);
      for (int o = lane; o < O; o += 64) {
        const float v = row[o];
        se += expf(v - mx);
        ahead += (v > zl) || (v == zl && o < lab);
      }
    }
    se = wave_all(se, f_add);
    ahead = wave_all(ahead, i_add);
    nll += logf(se) - (zl - mx);
    e1 += ahead >= 1 ? 1.f : 0.f;
    e5 += ahead >= (O < 5 ? O : 5) ? 1.f : 0.f;
    if (dlogits) {
      const float inv = 1.f / se;
      if (in_regs) {
#pragma unroll
        for (int i = 0; i < SM_R; ++i) {
          const int o = lane + 64 * i;
          if (o < O) dlogits[(size_t)n * O + o] = (z[i] * inv - (o == lab ? 1.f : 0.f)) * scale;
        }
      } else {
        for (int o = lane; o < O; o += 64) dlogits[(size_t)n * O + o] = (expf(row[o] - mx) * inv - (o == lab ? 1.f : 0.f)) * scale;
      }
    }
  }
  if (lane == 0) { red[0][wave] = nll; red[1][wave] = e1; red[2][wave] = e5; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int i = 0; i < SM_NW; ++i) s += red[threadIdx.x][i];
    out3[threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(NT) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, float lr, float momentum,
                                                 float dampening, float wd, int nesterov, int first, float gscale, const float* __restrict__ loss_scale_dev,
                                                 const float* __restrict__ found_inf_dev) {
  // AMP (training.py:104-110 scaler.step): gradients carry the loss scale, a device scalar; a step with non-finite gradients
  // is skipped entirely (parameters and momentum untouched) -- decided on the device, no host sync
  if (found_inf_dev && found_inf_dev[0] != 0.f) return;
  if (loss_scale_dev) gscale /= loss_scale_dev[0];
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    float d = g[i] * gscale + wd * p[i];
    if (momentum != 0.f) {
      float b = first ? d : momentum * buf[i] + (1.f - dampening) * d;
      buf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] -= lr * d;
  }
}

// ---- input pipeline on the device: the data_aug_train chain of the shipped configs as ONE pass over a batch -------
// reference, per sample on the host (transform_util.py): ToTensorTransform :36-47 (u8 HWC -> float CHW / 255), ZeroMean / Standardize
// whitening :50-109 ((x - mean) / stddev with per-pixel per-channel statistics), FlipTransform :156-166 (tc.flip(x, dims=(2,)) = the W
// axis), PaddingTransform :169-187 (F.pad reflect | constant 0), RandomCropTransform :190-205 (x[:, t:t+s, l:l+s]); order config.yaml:6-14.
// One thread per output pixel composes the index maps backwards: crop -> padding (reflect / zero) -> flip -> whitening at the SOURCE
// coordinate (whitening precedes the flip).  The random draws (flip bit, crop offsets) are inputs.
template <typename T>
__global__ __launch_bounds__(NT) void augment_kernel(const unsigned char* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                     const unsigned char* __restrict__ flip, const int* __restrict__ top, const int* __restrict__ left,
                                                     float* __restrict__ out_nchw, T* __restrict__ out_nhwc, int N, int H, int W, int C, int pad,
                                                     int mirror, int crop, int CP) {
  const long n_out = (long)N * crop * crop;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n_out; i += (long)gridDim.x * NT) {
    const int n = (int)(i / (crop * crop)), r = (int)(i - (long)n * crop * crop);
    const int oi = r / crop, oj = r - oi * crop;
    int hi = top[n] + oi - pad, wj = left[n] + oj - pad;   // coordinates in the (flipped) unpadded image
    bool inside = true;
    if (mirror) {                                          // F.pad(mode='reflect'): -1 -> 1, H -> H - 2 (the edge is not repeated)
      hi = hi < 0 ? -hi : (hi >= H ? 2 * (H - 1) - hi : hi);
      wj = wj < 0 ? -wj : (wj >= W ? 2 * (W - 1) - wj : wj);
      hi = min(max(hi, 0), H - 1);                         // offsets outside RandomCrop's range (caller error) must not leave the image
      wj = min(max(wj, 0), W - 1);
    } else {
      inside = (unsigned)hi < (unsigned)H && (unsigned)wj < (unsigned)W;
    }
    const int ws = flip[n] ? W - 1 - wj : wj;              // source column before the flip
    for (int c = 0; c < (out_nhwc ? CP : C); ++c) {
      float v = 0.f;
      if (c < C && inside) {
        const long hw = (long)hi * W + ws;
        v = (float)x[((long)n * H * W + hw) * C + c] / 255.f;        // ToTensor
        const long st = (long)c * H * W + hw;
        v = v - mean[st];
        if (inv_std) v = v / inv_std[st];                  // inv_std holds the stddev image: the reference DIVIDES (kept bit-exact)
      }
      if (out_nchw && c < C) out_nchw[(((long)n * C + c) * crop + oi) * crop + oj] = v;
      if (out_nhwc) out_nhwc[i * CP + c] = Elem<T>::from_f(v);
    }
  }
}

// ---- GradScaler's non-finite check (and unscale) over ONE flat gradient buffer (torch walks the parameter list and launches
// _amp_foreach_non_finite_check_and_unscale_ per device / dtype group: the check alone re-WRITES every gradient) -------------------------
__global__ __launch_bounds__(NT) void amp_check_unscale_kernel(float* __restrict__ g, long n, const float* __restrict__ inv_scale, float* __restrict__ found_inf) {
  const float sc = *inv_scale;
  const bool scale = sc != 1.f;                // the check-only call passes 1: nothing is written back
  bool bad = false;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
    float4 v = reinterpret_cast<float4*>(g)[i];
    bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
    if (scale) { v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; reinterpret_cast<float4*>(g)[i] = v; }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float v = g[i];
    bad |= !isfinite(v);
    if (scale) g[i] = v * sc;
  }
  if (bad) *found_inf = 1.f;                   // every writer stores the same value
}

// ---- grammar corners of the architecture spec (resnet.py:122-158 accepts any token sequence; none of the shipped configs uses these) ----
// a top-level 'a' that follows no 'n' (resnet.py:143-145, nn.ReLU): y = max(x, 0); backward by the sign of the stored output
template <typename T>
__global__ __launch_bounds__(NT) void relu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long nchunks) {
  constexpr int CE = Elem<T>::CE;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < nchunks; i += (long)gridDim.x * NT) {
    Chunk<T> c = load_chunk<T>(x + i * CE);
#pragma unroll
    for (int e = 0; e < CE; ++e) c.e[e] = Elem<T>::from_f(fmaxf(Elem<T>::to_f(c.e[e]), 0.f));
    store_chunk<T>(y + i * CE, c);
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long nchunks) {
  constexpr int CE = Elem<T>::CE;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < nchunks; i += (long)gridDim.x * NT) {
    Chunk<T> d = load_chunk<T>(dy + i * CE);
    const Chunk<T> o = load_chunk<T>(y + i * CE);
#pragma unroll
    for (int e = 0; e < CE; ++e) d.e[e] = Elem<T>::to_f(o.e[e]) > 0.f ? d.e[e] : Elem<T>::from_f(0.f);
    store_chunk<T>(dx + i * CE, d);
  }
}

// AvgPool2d(k, s, p) that is not the global pool in front of the classifier (resnet.py:77-81; torch defaults: zero padding counted in the divisor,
// floor output size), NHWC.  Forward: a thread per (output pixel, channel chunk); backward: gather form, an input pixel sums dy / k^2 of the windows
// that cover it (no atomics, no zero-fill pass), like maxpool_bwd_kernel.
template <typename T>
__global__ __launch_bounds__(NT) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int P, int Q, int k, int stride, int pad) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  const long n = (long)N * P * Q * CC;
  const float div = (float)(k * k);
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const int cg = (int)(i % CC);
    long pix = i / CC;
    const int q = (int)(pix % Q); pix /= Q;
    const int p = (int)(pix % P);
    const int nn = (int)(pix / P);
    float acc[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] = 0.f;
    for (int r = 0; r < k; ++r) {
      const int h = p * stride + r - pad;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int t = 0; t < k; ++t) {
        const int w = q * stride + t - pad;
        if ((unsigned)w >= (unsigned)W) continue;
        const Chunk<T> c = load_chunk<T>(x + (((size_t)nn * H + h) * W + w) * C + cg * CE);
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] += Elem<T>::to_f(c.e[e]);
      }
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < CE; ++e) o.e[e] = Elem<T>::from_f(acc[e] / div);
    store_chunk<T>(y + i * CE, o);
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int P, int Q, int k, int stride, int pad) {
  constexpr int CE = Elem<T>::CE;
  const int CC = C / CE;
  const long n = (long)N * H * W * CC;
  const float div = (float)(k * k);
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const int cg = (int)(i % CC);
    long pix = i / CC;
    const int w = (int)(pix % W); pix /= W;
    const int h = (int)(pix % H);
    const int nn = (int)(pix / H);
    const int p_lo = max(0, (h + pad - k + stride) / stride), p_hi = min(P - 1, (h + pad) / stride);
    const int q_lo = max(0, (w + pad - k + stride) / stride), q_hi = min(Q - 1, (w + pad) / stride);
    float g[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) g[e] = 0.f;
    for (int p = p_lo; p <= p_hi; ++p)
      for (int q = q_lo; q <= q_hi; ++q) {
        const Chunk<T> d = load_chunk<T>(dy + (((size_t)nn * P + p) * Q + q) * C + cg * CE);
#pragma unroll
        for (int e = 0; e < CE; ++e) g[e] += Elem<T>::to_f(d.e[e]);
      }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < CE; ++e) o.e[e] = Elem<T>::from_f(g[e] / div);
    store_chunk<T>(dx + i * CE, o);
  }
}

// out[a][c][b] = in[a][b][c]: the Linear weight of an 'f' that flattens an NCHW map of more than one pixel ([O][C][H*W] in the reference's feature
// order, resnet.py:117-120) against this engine's NHWC features ([O][H*W][C]), and its gradient back
__global__ __launch_bounds__(NT) void permute_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int A, int B, int Cc) {
  const long n = (long)A * B * Cc;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {       // i walks the OUTPUT: coalesced stores
    const int b = (int)(i % B);
    const long r = i / B;
    const int c = (int)(r % Cc);
    const long a = r / Cc;
    out[i] = in[(a * B + b) * Cc + c];
  }
}

inline int ew_grid(long n) {
  long b = (n + NT - 1) / NT;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int rn_pack_weights(const float* w_krsc, void* w_fwd, void* w_dgrad, int dtype, int K, int RS, int C, rn_stream s) {
  RN_CHECK_ARG(w_krsc && (w_fwd || w_dgrad) && K > 0 && RS > 0 && C > 0, "rn_pack_weights: bad argument");
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "rn_pack_weights: bad dtype");
  RN_CHECK_ARG(!w_dgrad || (C % (dtype == RN_F32 ? 4 : 8) == 0 && K % (dtype == RN_F32 ? 4 : 8) == 0), "rn_pack_weights: C=%d and K=%d must be multiples of %d for this dtype", C, K, dtype == RN_F32 ? 4 : 8);
  const long n = (long)K * RS * C;
  if (w_fwd && !w_dgrad) {
    RN_BY_DTYPE(dtype, hipLaunchKernelGGL((pack_w_fwd_kernel<T_>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), w_krsc, (T_*)w_fwd, n));
    RN_CHECK_LAUNCH("pack_weights_fwd");
  }
  if (w_dgrad) {
    const int grid = cdiv(K, 64) * cdiv(C, 64) * RS;
    RN_BY_DTYPE(dtype, hipLaunchKernelGGL((pack_w_dgrad_kernel<T_>), dim3(grid), dim3(256), 0, as_stream(s), w_krsc, (T_*)w_fwd, (T_*)w_dgrad, K, RS, C));
  }
  RN_CHECK_LAUNCH("pack_weights");
  return 0;
}

extern "C" int rn_pack_weights_batch(const rn_pack_desc* descs, int n, int dtype, rn_stream s) {
  RN_CHECK_ARG(descs && n > 0 && n <= RN_PACK_BATCH_MAX, "rn_pack_weights_batch: n=%d out of range (1..%d)", n, RN_PACK_BATCH_MAX);
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "rn_pack_weights_batch: bad dtype");
  const int ce = dtype == RN_F32 ? 4 : 8;
  PackBatch pb;
  pb.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const rn_pack_desc& d = descs[i];
    RN_CHECK_ARG(d.w && (d.w_fwd || d.w_dgrad) && d.K > 0 && d.RS > 0 && d.C > 0, "rn_pack_weights_batch: bad descriptor %d", i);
    RN_CHECK_ARG(d.C % ce == 0 && d.K % ce == 0, "rn_pack_weights: C=%d and K=%d must be multiples of %d for this dtype", d.C, d.K, ce);
    pb.d[i] = d;
    pb.first_block[i] = blocks;
    blocks += cdiv(d.K, 64) * cdiv(d.C, 64) * d.RS;
  }
  pb.first_block[n] = blocks;
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((pack_w_batch_kernel<T_>), dim3(blocks), dim3(256), 0, as_stream(s), pb));
  RN_CHECK_LAUNCH("pack_weights_batch");
  return 0;
}

extern "C" int rn_img_to_nhwc(const float* x_nchw, void* out, int dtype, int N, int C, int H, int W, int CP, rn_stream s) {
  RN_CHECK_ARG(x_nchw && out && N > 0 && C > 0 && C <= CP && RN_DTYPE_OK(dtype) && CP == (dtype == RN_F32 ? 4 : 8),
               "rn_img_to_nhwc: bad argument (C=%d CP=%d)", C, CP);
  const long n = (long)N * H * W;
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((img_to_nhwc_kernel<T_>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), x_nchw, (T_*)out, N, C, H, W, CP));
  RN_CHECK_LAUNCH("img_to_nhwc");
  return 0;
}

extern "C" int rn_pack_stem_w(const float* w_krsc, void* w_padded, int dtype, int K, int RS, int C, int CP, rn_stream s) {
  RN_CHECK_ARG(w_krsc && w_padded && K > 0 && RS > 0 && C > 0 && C <= CP && RN_DTYPE_OK(dtype), "rn_pack_stem_w: bad argument");
  const long rows = (long)K * RS;
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((pack_stem_w_kernel<T_>), dim3(ew_grid(rows * CP)), dim3(NT), 0, as_stream(s), w_krsc, (T_*)w_padded, rows, C, CP));
  RN_CHECK_LAUNCH("pack_stem_w");
  return 0;
}

extern "C" int rn_img_to_s2d(const float* x_nchw, void* out, int dtype, int N, int C, int H, int W, rn_stream s) {
  RN_CHECK_ARG(x_nchw && out && N > 0 && C > 0 && C <= 4 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && RN_DTYPE_OK(dtype), "rn_img_to_s2d: bad argument (C=%d H=%d W=%d)", C, H, W);
  const long n = (long)N * (H / 2 + 3) * (W / 2 + 3);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((img_to_s2d_kernel<T_>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), x_nchw, (T_*)out, N, C, H, W, H / 2 + 3, W / 2 + 3));
  RN_CHECK_LAUNCH("img_to_s2d");
  return 0;
}
extern "C" int rn_pack_stem_w_s2d(const float* w_k77c, void* w_s2d, int dtype, int K, int C, rn_stream s) {
  RN_CHECK_ARG(w_k77c && w_s2d && K > 0 && C > 0 && C <= 4 && RN_DTYPE_OK(dtype), "rn_pack_stem_w_s2d: bad argument");
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((pack_stem_w_s2d_kernel<T_>), dim3(ew_grid((long)K * 256)), dim3(NT), 0, as_stream(s), w_k77c, (T_*)w_s2d, K, C));
  RN_CHECK_LAUNCH("pack_stem_w_s2d");
  return 0;
}
extern "C" int rn_unpack_stem_dw_s2d(const float* dw_s2d, float* dw_k77c, int K, int C, int accumulate, rn_stream s) {
  RN_CHECK_ARG(dw_s2d && dw_k77c && K > 0 && C > 0 && C <= 4, "rn_unpack_stem_dw_s2d: bad argument");
  hipLaunchKernelGGL(unpack_stem_dw_s2d_kernel, dim3(ew_grid((long)K * 49 * C)), dim3(NT), 0, as_stream(s), dw_s2d, dw_k77c, K, C, accumulate);
  RN_CHECK_LAUNCH("unpack_stem_dw_s2d");
  return 0;
}

extern "C" int rn_unpack_stem_dw(const float* dw_padded, float* dw_krsc, int K, int RS, int C, int CP, int accumulate, rn_stream s) {
  RN_CHECK_ARG(dw_padded && dw_krsc && K > 0 && RS > 0 && C > 0 && C <= CP, "rn_unpack_stem_dw: bad argument");
  const long rows = (long)K * RS;
  hipLaunchKernelGGL(unpack_stem_dw_kernel, dim3(ew_grid(rows * C)), dim3(NT), 0, as_stream(s), dw_padded, dw_krsc, rows, C, CP, accumulate);
  RN_CHECK_LAUNCH("unpack_stem_dw");
  return 0;
}

static int check_pool(int dtype, int N, int H, int W, int C, int k, int stride, int pad, const char* who) {
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "%s: bad dtype", who);
  RN_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % (dtype == RN_F32 ? 4 : 8) == 0 && k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k,
               "%s: bad shape", who);
  return 0;
}

extern "C" int rn_relu_fwd(const void* x, void* y, int dtype, int64_t n, rn_stream s) {
  RN_CHECK_ARG(RN_DTYPE_OK(dtype) && x && y && n > 0 && n % (dtype == RN_F32 ? 4 : 8) == 0, "rn_relu_fwd: bad argument");
  const long nc = n / (dtype == RN_F32 ? 4 : 8);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((relu_fwd_kernel<T_>), dim3(ew_grid(nc)), dim3(NT), 0, as_stream(s), (const T_*)x, (T_*)y, nc));
  RN_CHECK_LAUNCH("relu_fwd");
  return 0;
}

extern "C" int rn_relu_bwd(const void* dy, const void* y, void* dx, int dtype, int64_t n, rn_stream s) {
  RN_CHECK_ARG(RN_DTYPE_OK(dtype) && dy && y && dx && n > 0 && n % (dtype == RN_F32 ? 4 : 8) == 0, "rn_relu_bwd: bad argument");
  const long nc = n / (dtype == RN_F32 ? 4 : 8);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((relu_bwd_kernel<T_>), dim3(ew_grid(nc)), dim3(NT), 0, as_stream(s), (const T_*)dy, (const T_*)y, (T_*)dx, nc));
  RN_CHECK_LAUNCH("relu_bwd");
  return 0;
}

extern "C" int rn_avgpool_fwd(const void* x, void* y, int dtype, int N, int H, int W, int C, int k, int stride, int pad, rn_stream s) {
  if (int e = check_pool(dtype, N, H, W, C, k, stride, pad, "rn_avgpool_fwd")) return e;
  RN_CHECK_ARG(x && y && H + 2 * pad >= k && W + 2 * pad >= k, "rn_avgpool_fwd: bad argument");
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const long n = (long)N * P * Q * (C / (dtype == RN_F32 ? 4 : 8));
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((avgpool_fwd_kernel<T_>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), (const T_*)x, (T_*)y, N, H, W, C, P, Q, k, stride, pad));
  RN_CHECK_LAUNCH("avgpool_fwd");
  return 0;
}

extern "C" int rn_avgpool_bwd(const void* dy, void* dx, int dtype, int N, int H, int W, int C, int k, int stride, int pad, rn_stream s) {
  if (int e = check_pool(dtype, N, H, W, C, k, stride, pad, "rn_avgpool_bwd")) return e;
  RN_CHECK_ARG(dy && dx && H + 2 * pad >= k && W + 2 * pad >= k, "rn_avgpool_bwd: bad argument");
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const long n = (long)N * H * W * (C / (dtype == RN_F32 ? 4 : 8));
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((avgpool_bwd_kernel<T_>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), (const T_*)dy, (T_*)dx, N, H, W, C, P, Q, k, stride, pad));
  RN_CHECK_LAUNCH("avgpool_bwd");
  return 0;
}

extern "C" int rn_permute_f32(const float* in, float* out, int A, int B, int C, rn_stream s) {
  RN_CHECK_ARG(in && out && in != out && A > 0 && B > 0 && C > 0, "rn_permute_f32: bad argument");
  hipLaunchKernelGGL(permute_f32_kernel, dim3(ew_grid((long)A * B * C)), dim3(NT), 0, as_stream(s), in, out, A, B, C);
  RN_CHECK_LAUNCH("permute_f32");
  return 0;
}

extern "C" int rn_maxpool_fwd(const void* x, void* y, unsigned char* argmax, int dtype, int N, int H, int W, int C, int k, int stride, int pad, rn_stream s) {
  if (int e = check_pool(dtype, N, H, W, C, k, stride, pad, "rn_maxpool_fwd")) return e;
  RN_CHECK_ARG(x && y && k * k < 255, "rn_maxpool_fwd: bad argument");
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const long n = (long)N * P * Q * (C / (dtype == RN_F32 ? 4 : 8));
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T_>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), (const T_*)x, (T_*)y, argmax, N, H, W, C, P, Q, k, stride, pad));
  RN_CHECK_LAUNCH("maxpool_fwd");
  return 0;
}

extern "C" int rn_maxpool_bwd(const void* dy, const unsigned char* argmax, void* dx, int dtype, int N, int H, int W, int C, int k, int stride, int pad, rn_stream s) {
  if (int e = check_pool(dtype, N, H, W, C, k, stride, pad, "rn_maxpool_bwd")) return e;
  RN_CHECK_ARG(dy && argmax && dx, "rn_maxpool_bwd: null pointer");
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const int rows_grid = (int)std::min<long>((long)N * H, 8192);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((maxpool_bwd_kernel<T_>), dim3(rows_grid), dim3(NT), 0, as_stream(s), (const T_*)dy, argmax, (T_*)dx, N, H, W, C, P, Q, k, stride, pad));
  RN_CHECK_LAUNCH("maxpool_bwd");
  return 0;
}

// fused BatchNorm-apply (+ReLU) + MaxPool: C / CE must divide 256; nblk (backward reduce) = workgroups = partial rows
static int bn_pool_fwd(const void* x, const float* coef, void* y, unsigned char* argmax, void* xsel, int dtype, int N, int H, int W, int C, int k, int stride,
                       int pad, int flags, rn_stream s) {
  if (int e = check_pool(dtype, N, H, W, C, k, stride, pad, "rn_bn_pool_fwd")) return e;
  RN_CHECK_ARG(x && coef && y && k * k < 255, "rn_bn_pool_fwd: bad argument");
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const long n = (long)N * P * Q * (C / (dtype == RN_F32 ? 4 : 8));
  if (k == 3) {
    RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_fwd_kernel<T_, 3>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), (const T_*)x, coef, (T_*)y, argmax, (T_*)xsel, N, H, W, C, P, Q,
                                          k, stride, pad, (flags & RN_F_RELU) ? 1 : 0));
  } else {
    RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_fwd_kernel<T_, 0>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), (const T_*)x, coef, (T_*)y, argmax, (T_*)xsel, N, H, W, C, P, Q,
                                          k, stride, pad, (flags & RN_F_RELU) ? 1 : 0));
  }
  RN_CHECK_LAUNCH("bn_pool_fwd");
  return 0;
}
extern "C" int rn_bn_pool_fwd(const void* x, const float* coef, void* y, unsigned char* argmax, int dtype, int N, int H, int W, int C, int k, int stride,
                              int pad, int flags, rn_stream s) {
  return bn_pool_fwd(x, coef, y, argmax, nullptr, dtype, N, H, W, C, k, stride, pad, flags, s);
}
extern "C" int rn_bn_pool_fwd_sel(const void* x, const float* coef, void* y, unsigned char* argmax, void* xsel, int dtype, int N, int H, int W, int C, int k,
                                  int stride, int pad, int flags, rn_stream s) {
  RN_CHECK_ARG(xsel != nullptr, "rn_bn_pool_fwd_sel: null xsel");
  return bn_pool_fwd(x, coef, y, argmax, xsel, dtype, N, H, W, C, k, stride, pad, flags, s);
}
extern "C" int rn_bn_pool_bwd_reduce_sel(const void* dy, const void* xsel, const float* coef, float* partial, int nblk, int dtype, long npix, int C, int flags,
                                         rn_stream s) {
  RN_CHECK_ARG(dy && xsel && coef && partial && npix > 0 && nblk > 0 && nblk <= 8192, "rn_bn_pool_bwd_reduce_sel: bad argument");
  RN_CHECK_ARG(RN_DTYPE_OK(dtype) && C > 0 && C % (dtype == RN_F32 ? 4 : 8) == 0 && NT % (C / (dtype == RN_F32 ? 4 : 8)) == 0,
               "rn_bn_pool_bwd_reduce_sel: C=%d (C / chunk must divide %d)", C, NT);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_reduce_sel_kernel<T_>), dim3(nblk), dim3(NT), 0, as_stream(s), (const T_*)dy, (const T_*)xsel, coef, partial, npix, C,
                                        (flags & RN_F_RELU) ? 1 : 0));
  RN_CHECK_LAUNCH("bn_pool_bwd_reduce_sel");
  return 0;
}
static int bn_pool_bwd(const void* dy, const unsigned char* argmax, const void* x, const float* coef, const float* dsum, float* partial, int nblk, void* dx,
                       int dtype, int N, int H, int W, int C, int k, int stride, int pad, int flags, double count, rn_stream s, const char* who) {
  if (int e = check_pool(dtype, N, H, W, C, k, stride, pad, who)) return e;
  const int cc = C / (dtype == RN_F32 ? 4 : 8);
  RN_CHECK_ARG(dy && argmax && x && coef && NT % cc == 0, "%s: bad argument (C / chunk = %d must divide %d)", who, cc, NT);
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const int relu = (flags & RN_F_RELU) ? 1 : 0;
  const bool quad = k == 3 && stride == 2 && pad == 1 && H % 2 == 0 && W % 2 == 0 && !(g_rn_variant & (1 << 25));
  if (partial && !dx) {
    RN_CHECK_ARG(nblk > 0 && nblk <= N * H, "%s: nblk out of range", who);
    if (quad) {
      RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_quad_kernel<T_, 0>), dim3(nblk), dim3(NT), 0, as_stream(s), (const T_*)dy, argmax, (const T_*)x, coef, nullptr,
                                            partial, (T_*)nullptr, N, H, W, C, relu, 1, 0.f));
    } else {
      RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_kernel<T_, 0>), dim3(nblk), dim3(NT), 0, as_stream(s), (const T_*)dy, argmax, (const T_*)x, coef, nullptr,
                                            partial, (T_*)nullptr, N, H, W, C, P, Q, k, stride, pad, relu, 1, 0.f));
    }
  } else {
    const int train = (flags & RN_F_TRAIN) ? 1 : 0;
    RN_CHECK_ARG((dsum || !train) && dx && count > 0, "%s: bad argument", who);
    RN_CHECK_ARG(!partial || (nblk > 0 && nblk <= 8192), "%s: %d rows of column sums (1..8192)", who, nblk);
    // with column sums the caller's row count IS the grid (one partial row per workgroup); the loops are grid-stride, any grid is correct
    if (quad) {
      const int grid = partial ? nblk : (int)std::min<long>((long)N * (H / 2), 8192);
      if (partial) {
        RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_quad_kernel<T_, 2>), dim3(grid), dim3(NT), 0, as_stream(s), (const T_*)dy, argmax, (const T_*)x, coef, dsum,
                                              partial, (T_*)dx, N, H, W, C, relu, train, (float)(1.0 / count)));
      } else {
        RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_quad_kernel<T_, 1>), dim3(grid), dim3(NT), 0, as_stream(s), (const T_*)dy, argmax, (const T_*)x, coef, dsum,
                                              nullptr, (T_*)dx, N, H, W, C, relu, train, (float)(1.0 / count)));
      }
    } else {
      const int grid = partial ? nblk : (int)std::min<long>((long)N * H, 8192);
      if (partial) {
        RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_kernel<T_, 2>), dim3(grid), dim3(NT), 0, as_stream(s), (const T_*)dy, argmax, (const T_*)x, coef, dsum, partial,
                                              (T_*)dx, N, H, W, C, P, Q, k, stride, pad, relu, train, (float)(1.0 / count)));
      } else {
        RN_BY_DTYPE(dtype, hipLaunchKernelGGL((bn_pool_bwd_kernel<T_, 1>), dim3(grid), dim3(NT), 0, as_stream(s), (const T_*)dy, argmax, (const T_*)x, coef, dsum, nullptr,
                                              (T_*)dx, N, H, W, C, P, Q, k, stride, pad, relu, train, (float)(1.0 / count)));
      }
    }
  }
  RN_CHECK_LAUNCH(who);
  return 0;
}
extern "C" int rn_bn_pool_bwd_reduce(const void* dy, const unsigned char* argmax, const void* x, const float* coef, float* partial, int nblk, int dtype, int N,
                                     int H, int W, int C, int k, int stride, int pad, int flags, rn_stream s) {
  RN_CHECK_ARG(partial != nullptr, "rn_bn_pool_bwd_reduce: null partial");
  return bn_pool_bwd(dy, argmax, x, coef, nullptr, partial, nblk, nullptr, dtype, N, H, W, C, k, stride, pad, flags, 1.0, s, "rn_bn_pool_bwd_reduce");
}
extern "C" int rn_bn_pool_bwd_apply(const void* dy, const unsigned char* argmax, const void* x, const float* coef, const float* dsum, void* dx, int dtype, int N,
                                    int H, int W, int C, int k, int stride, int pad, int flags, double count, rn_stream s) {
  return bn_pool_bwd(dy, argmax, x, coef, dsum, nullptr, 0, dx, dtype, N, H, W, C, k, stride, pad, flags, count, s, "rn_bn_pool_bwd_apply");
}
extern "C" int rn_bn_pool_bwd_apply_sums(const void* dy, const unsigned char* argmax, const void* x, const float* coef, const float* dsum, void* dx, float* sums_partial,
                                         int rows, int dtype, int N, int H, int W, int C, int k, int stride, int pad, int flags, double count, rn_stream s) {
  RN_CHECK_ARG(sums_partial != nullptr, "rn_bn_pool_bwd_apply_sums: null sums_partial");
  return bn_pool_bwd(dy, argmax, x, coef, dsum, sums_partial, rows, dx, dtype, N, H, W, C, k, stride, pad, flags, count, s, "rn_bn_pool_bwd_apply_sums");
}

extern "C" int rn_pool_fc_fwd(const void* x, const float* w, const float* b, float* feat, float* logits, int dtype, int N, int HW, int C, int O, rn_stream s) {
  RN_CHECK_ARG(x && w && b && feat && logits && N > 0 && HW > 0 && O > 0, "rn_pool_fc_fwd: bad argument");
  RN_CHECK_ARG(RN_DTYPE_OK(dtype) && C % (dtype == RN_F32 ? 4 : 8) == 0, "rn_pool_fc_fwd: bad dtype / C=%d", C);
  RN_BY_DTYPE(dtype, hipLaunchKernelGGL((gap_kernel<T_>), dim3(N), dim3(NT), 0, as_stream(s), (const T_*)x, feat, HW, C));
  RN_CHECK_LAUNCH("gap");
  if (fc_wide(C, O) && O % 4 == 0)
    hipLaunchKernelGGL((fc_gemm_kernel<float, 0, true, true>), dim3(cdiv(O, FG_T), cdiv(N, FG_T)), dim3(NT), 0, as_stream(s), feat, w, N, O, C, C, C,
                       (void*)logits, b, (float*)nullptr, 1, 0);
  else
    hipLaunchKernelGGL(fc_fwd_kernel, dim3(cdiv((long)N * O * 64, NT)), dim3(NT), 0, as_stream(s), feat, w, b, logits, N, C, O);
  RN_CHECK_LAUNCH("fc_fwd");
  return 0;
}

extern "C" int rn_pool_fc_bwd(const float* dlogits, const float* feat, const float* w, void* dx, float* dw, float* db, int dtype, int N, int HW, int C,
                              int O, int flags, rn_stream s) {
  RN_CHECK_ARG(dlogits && feat && w && dw && db && N > 0 && HW > 0 && O > 0, "rn_pool_fc_bwd: bad argument");
  RN_CHECK_ARG(RN_DTYPE_OK(dtype) && C % (dtype == RN_F32 ? 4 : 8) == 0, "rn_pool_fc_bwd: bad dtype / C=%d", C);
  const bool wide = fc_wide(C, O);
  if (wide)
    hipLaunchKernelGGL((fc_gemm_kernel<float, 2, false, false>), dim3(cdiv(C, FG_T), cdiv(O, FG_T)), dim3(NT), 0, as_stream(s), dlogits, feat, O, C, N, O, C,
                       (void*)dw, (const float*)nullptr, db, 1, (flags & RN_F_ACCUM) ? 1 : 0);
  else
    hipLaunchKernelGGL(fc_wgrad_kernel, dim3(cdiv((long)O * C, 64)), dim3(NT), 0, as_stream(s), dlogits, feat, dw, db, N, C, O, (flags & RN_F_ACCUM) ? 1 : 0);
  RN_CHECK_LAUNCH("fc_wgrad");
  if (!(flags & RN_F_NO_DX)) {
    RN_CHECK_ARG(dx != nullptr, "rn_pool_fc_bwd: dx is null");
    if (wide) {
      RN_BY_DTYPE(dtype, hipLaunchKernelGGL((fc_gemm_kernel<T_, 1, true, false>), dim3(cdiv(C, FG_T), cdiv(N, FG_T)), dim3(NT), 0, as_stream(s), dlogits, w, N, C,
                                            O, O, C, dx, (const float*)nullptr, (float*)nullptr, HW, 0));
    } else {
      RN_BY_DTYPE(dtype, hipLaunchKernelGGL((fc_dgrad_kernel<T_>), dim3(N), dim3(NT), 0, as_stream(s), dlogits, w, (T_*)dx, HW, C, O));
    }
    RN_CHECK_LAUNCH("fc_dgrad");
  }
  return 0;
}

extern "C" int rn_augment_batch(const unsigned char* x_nhwc_u8, const float* mean_chw, const float* stddev_chw, const unsigned char* flip, const int32_t* top,
                                const int32_t* left, float* out_nchw, void* out_nhwc, int dtype, int N, int H, int W, int C, int pad, int pad_mirror,
                                int crop, int CP, rn_stream s) {
  RN_CHECK_ARG(x_nhwc_u8 && mean_chw && flip && top && left && (out_nchw || out_nhwc), "rn_augment_batch: null pointer");
  RN_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && pad >= 0 && crop > 0 && crop <= H + 2 * pad && crop <= W + 2 * pad, "rn_augment_batch: bad shape");
  RN_CHECK_ARG(!pad_mirror || (pad < H && pad < W), "rn_augment_batch: reflect padding needs pad < H, W");
  RN_CHECK_ARG(!out_nhwc || (RN_DTYPE_OK(dtype) && CP >= C), "rn_augment_batch: bad NHWC output (dtype %d, CP %d)", dtype, CP);
  const long n = (long)N * crop * crop;
  if (!out_nhwc) {
    hipLaunchKernelGGL((augment_kernel<float>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), x_nhwc_u8, mean_chw, stddev_chw, flip, top, left, out_nchw,
                       (float*)nullptr, N, H, W, C, pad, pad_mirror ? 1 : 0, crop, CP);
  } else {
    RN_BY_DTYPE(dtype, hipLaunchKernelGGL((augment_kernel<T_>), dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), x_nhwc_u8, mean_chw, stddev_chw, flip, top, left,
                                          out_nchw, (T_*)out_nhwc, N, H, W, C, pad, pad_mirror ? 1 : 0, crop, CP));
  }
  RN_CHECK_LAUNCH("augment_batch");
  return 0;
}

extern "C" int rn_amp_check_unscale(float* grads, int64_t n, const float* inv_scale_dev, float* found_inf_dev, rn_stream s) {
  RN_CHECK_ARG(grads && n > 0 && inv_scale_dev && found_inf_dev, "rn_amp_check_unscale: bad argument");
  hipLaunchKernelGGL(amp_check_unscale_kernel, dim3(ew_grid(n >> 2)), dim3(NT), 0, as_stream(s), grads, (long)n, inv_scale_dev, found_inf_dev);
  RN_CHECK_LAUNCH("amp_check_unscale");
  return 0;
}

extern "C" int rn_softmax_ce(const float* logits, const int64_t* labels, float* out3, float* dlogits, int N, int O, float scale, const float* scale_dev,
                             rn_stream s) {
  RN_CHECK_ARG(logits && labels && out3 && N > 0 && O > 0, "rn_softmax_ce: bad argument");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(SM_NT), 0, as_stream(s), logits, (const long long*)labels, out3, dlogits, N, O, scale, scale_dev);
  RN_CHECK_LAUNCH("softmax_ce");
  return 0;
}

extern "C" int rn_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                           float weight_decay, int nesterov, int first_step, float grad_scale, rn_stream s) {
  RN_CHECK_ARG(param && grad && n > 0 && (momentum == 0.f || momentum_buf), "rn_sgd_step: bad argument");
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), param, grad, momentum_buf, (long)n, lr, momentum, dampening, weight_decay,
                     nesterov, first_step, grad_scale, (const float*)nullptr, (const float*)nullptr);
  RN_CHECK_LAUNCH("sgd");
  return 0;
}

extern "C" int rn_sgd_step_amp(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                               float weight_decay, int nesterov, int first_step, const float* loss_scale_dev, const float* found_inf_dev, rn_stream s) {
  RN_CHECK_ARG(param && grad && n > 0 && (momentum == 0.f || momentum_buf), "rn_sgd_step_amp: bad argument");
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n)), dim3(NT), 0, as_stream(s), param, grad, momentum_buf, (long)n, lr, momentum, dampening, weight_decay,
                     nesterov, first_step, 1.f, loss_scale_dev, found_inf_dev);
  RN_CHECK_LAUNCH("sgd_amp");
  return 0;
}
