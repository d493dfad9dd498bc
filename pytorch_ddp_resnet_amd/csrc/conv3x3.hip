// 3x3 / stride-1 / pad-1 convolution (and its data gradient) for the CIFAR-sized maps of the reference
// (residual_block.py:34-47: W in {32,16,8}) with the input patch RESIDENT in LDS: gfx950.
//
// The generic implicit GEMM (conv_igemm.hip) re-stages the activation tile once per filter tap, i.e. 9x, which puts
// ~57 B/clk/CU on the L2->LDS path -- at the L2 limit.  Here a workgroup owns a spatial tile (G images x TH rows x W
// columns = BM output pixels), stages the (TH+2) x (W+2) halo patch of one 64-byte channel chunk ONCE (zero-filled
// border, XOR-swizzled 64-byte pixel rows) and slides the 9 taps over it: the A fragment of tap (r,s) is the same
// ds_read_b128 at a constant LDS offset.  Only the weights stream per tap.  L2->LDS traffic drops ~3x, LDS stores ~6x.
//
//   out[n,h,w,k] (+)= sum_{t} sum_c in[n, h+dh[t], w+dw[t], c] * wt[k][widx[t]][c]  (+ res)
//
// forward: dh = r-1, wt = KRSC.  dgrad (stride 1): dh = 1-r, wt = CRSK.   Tile: BM x BN, BM/32 waves, each wave 32
// pixels (one MFMA row block) x BN channels; K loop = channel chunks x taps, TPI taps per barrier.
#include "common.h"

namespace {

constexpr int CPR = 4;  // 16-byte chunks per 64-byte pixel row

struct P3Args {
  const void* src;
  const void* wt;
  void* dst;
  ResDesc res;
  int N, H, W, C, K;       // src [N,H,W,C] -> dst [N,H,W,K]
  int G, TH;               // tile = G images x TH rows x W cols
  int PH, PW, PP;          // patch rows, cols, pixels (G*PH*PW)
  int tiles_h;             // H / TH
  int nchunks;             // C / elements-per-64B
  int accum;
  int poff[9];             // patch offset of tap t: (dh+1)*PW + (dw+1)
  int widx[9];             // weight tap index of tap t
};

template <typename T> struct Mfma3;
template <> struct Mfma3<float> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    const float* af = reinterpret_cast<const float*>(&a);
    const float* bf = reinterpret_cast<const float*>(&b);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], c, 0, 0, 0);
  }
};
template <> struct Mfma3<bf16_t> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
  }
};

__device__ inline int swz3(int row, int chunk) { return row * CPR + (chunk ^ ((row >> 2) & 3)); }

template <typename T, int BM, int BN, int TPI, int PCAP>
__global__ __launch_bounds__(BM * 2) void conv3x3_patch_kernel(const P3Args a) {
  constexpr int CE = Elem<T>::CE;
  constexpr int NTH = BM * 2;                 // 64 threads per 32 pixels
  constexpr int TN = BN / 32;
  constexpr int KE = CPR * CE;                // channels per 64-byte chunk row
  constexpr int PSLOTS = (PCAP * CPR + NTH - 1) / NTH;
  constexpr int WCH = TPI * BN * CPR;         // weight chunks per iteration
  constexpr int WSLOTS = (WCH + NTH - 1) / NTH;
  constexpr int NIT = 9 / TPI;                // iterations per channel chunk
  static_assert(9 % TPI == 0, "TPI");
  __shared__ uint4 patch[2][PCAP * CPR];
  __shared__ uint4 wbuf[2][WCH];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  const T* __restrict__ src = reinterpret_cast<const T*>(a.src);
  const T* __restrict__ wt = reinterpret_cast<const T*>(a.wt);

  // ---- which tile ----
  const int mtiles = (a.N / a.G) * a.tiles_h;
  const int mt = blockIdx.x % mtiles, nt = blockIdx.x / mtiles;
  const int n0 = (mt / a.tiles_h) * a.G, h0 = (mt % a.tiles_h) * a.TH;
  const int k0 = nt * BN;

  // ---- patch staging slots: patch pixel pp = slot / 4, chunk = slot % 4 ----
  long psrc[PSLOTS];
  int pdst[PSLOTS];
#pragma unroll
  for (int i = 0; i < PSLOTS; ++i) {
    const int slot = tid + i * NTH;
    const int pp = slot >> 2, ch = slot & 3;
    pdst[i] = -1;
    psrc[i] = -1;
    if (pp < a.PP) {
      const int g = pp / (a.PH * a.PW), rem = pp - g * (a.PH * a.PW);
      const int hy = rem / a.PW, hx = rem - hy * a.PW;
      const int h = h0 + hy - 1, w = hx - 1;
      pdst[i] = swz3(pp, ch);
      if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) psrc[i] = (((long)(n0 + g) * a.H + h) * a.W + w) * a.C + ch * CE;
    }
  }
  // ---- weight staging slots: row = tl*BN + n, chunk ----
  int wrow[WSLOTS], wchk[WSLOTS];
#pragma unroll
  for (int i = 0; i < WSLOTS; ++i) {
    const int slot = tid + i * NTH;
    wrow[i] = slot < WCH ? slot >> 2 : -1;
    wchk[i] = slot & 3;
  }

  uint4 rp[PSLOTS], rw[WSLOTS];
  auto load_patch = [&](int cc) {
#pragma unroll
    for (int i = 0; i < PSLOTS; ++i) {
      rp[i] = make_uint4(0, 0, 0, 0);
      if (psrc[i] >= 0) rp[i] = *reinterpret_cast<const uint4*>(src + psrc[i] + (long)cc * KE);
    }
  };
  auto store_patch = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PSLOTS; ++i)
      if (pdst[i] >= 0) patch[buf][pdst[i]] = rp[i];
  };
  auto load_w = [&](int cc, int itl) {          // itl: iteration inside the chunk (taps itl*TPI .. +TPI-1)
#pragma unroll
    for (int i = 0; i < WSLOTS; ++i) {
      rw[i] = make_uint4(0, 0, 0, 0);
      if (wrow[i] >= 0) {
        const int tl = wrow[i] / BN, n = wrow[i] - tl * BN;
        const int k = k0 + n;
        if (k < a.K) rw[i] = *reinterpret_cast<const uint4*>(wt + ((long)k * 9 + a.widx[itl * TPI + tl]) * a.C + (long)cc * KE + wchk[i] * CE);
      }
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WSLOTS; ++i)
      if (wrow[i] >= 0) wbuf[buf][swz3(wrow[i], wchk[i])] = rw[i];
  };

  // ---- this lane's A row: output pixel m = 32*wave + lr of the tile -> patch pixel of tap (0,0) offset ----
  const int m = 32 * wave + lr;
  const int tw = a.W;
  const int g_ = m / (a.TH * tw), rem_ = m - g_ * (a.TH * tw);
  const int py = rem_ / tw, px = rem_ - py * tw;
  const int q0 = (g_ * a.PH + py) * a.PW + px;

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int total = a.nchunks * NIT;
  load_patch(0);
  load_w(0, 0);
  store_patch(0);
  store_w(0);
  __syncthreads();
  int cc = 0, itl = 0;
  for (int it = 0; it < total; ++it) {
    const int wb = it & 1, pb = cc & 1;
    // next iteration's coordinates
    int ncc = cc, nitl = itl + 1;
    if (nitl == NIT) { nitl = 0; ncc = cc + 1; }
    const bool more = it + 1 < total;
    if (more) load_w(ncc, nitl);
    const bool next_patch = more && itl == 0 && cc + 1 < a.nchunks;      // prefetch the next chunk's patch early in this chunk
    if (next_patch) load_patch(cc + 1);
#pragma unroll
    for (int tl = 0; tl < TPI; ++tl) {
      const int q = q0 + a.poff[itl * TPI + tl];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = 2 * ks + lh;
        const uint4 fa = patch[pb][swz3(q, ch)];
        uint4 fb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = wbuf[wb][swz3(tl * BN + 32 * j + lr, ch)];
#pragma unroll
        for (int j = 0; j < TN; ++j) Mfma3<T>::run(fa, fb[j], acc[j]);
      }
    }
    if (more) store_w(wb ^ 1);
    if (next_patch) store_patch(pb ^ 1);
    __syncthreads();
    cc = ncc; itl = nitl;
  }

  // ---- epilogue (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) ----
  T* __restrict__ dst = reinterpret_cast<T*>(a.dst);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mm = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int g2 = mm / (a.TH * tw), rem2 = mm - g2 * (a.TH * tw);
    const int y2 = rem2 / tw, x2 = rem2 - y2 * tw;
    const int n = n0 + g2, h = h0 + y2;
    const size_t pix = ((size_t)n * a.H + h) * a.W + x2;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int k = k0 + 32 * j + lr;
      if (k >= a.K) continue;
      float v = acc[j][r];
      const size_t off = pix * a.K + k;
      if (a.res.mode == RN_RES_SAME) v += Elem<T>::to_f(reinterpret_cast<const T*>(a.res.ptr)[off]);
      else if (a.res.mode != RN_RES_NONE) v += res_load1<T>(a.res, n, h, x2, k);
      if (a.accum) v += Elem<T>::to_f(dst[off]);
      dst[off] = Elem<T>::from_f(v);
    }
  }
}

template <typename T, int BM, int BN, int TPI, int PCAP>
int launch_p3(const P3Args& a, hipStream_t s) {
  const int mtiles = (a.N / a.G) * a.tiles_h;
  const int ntiles = cdiv(a.K, BN);
  hipLaunchKernelGGL((conv3x3_patch_kernel<T, BM, BN, TPI, PCAP>), dim3(mtiles * ntiles), dim3(BM * 2), 0, s, a);
  RN_CHECK_LAUNCH("conv3x3_patch");
  return 0;
}

template <typename T, int BM, int TPI, int PCAP>
int dispatch_bn(const P3Args& a, hipStream_t s) {
  const int K = a.K;
  if (K % 160 == 0) return launch_p3<T, BM, 160, TPI, PCAP>(a, s);
  if (K % 128 == 0) return launch_p3<T, BM, 128, TPI, PCAP>(a, s);
  if (K > 32) return launch_p3<T, BM, 64, TPI, PCAP>(a, s);
  return launch_p3<T, BM, 32, TPI, PCAP>(a, s);
}

bool tile_geom(int N, int H, int W, int BM, int& G, int& TH) {
  if (BM % W != 0) return false;
  const int rows = BM / W;
  if (rows <= H) { G = 1; TH = rows; return H % TH == 0; }
  if (rows % H != 0) return false;
  G = rows / H; TH = H;
  return N % G == 0;
}

}  // namespace

// returns -1 when the shape is not covered (caller falls back to the generic implicit GEMM), else the launch status
int rn_conv3x3_patch(const void* src, const void* wt, void* dst, const ResDesc& res, int accum, int dtype, int N, int H, int W, int C, int K,
                     bool flip, rn_stream s) {
  const int ke = dtype == RN_F32 ? 16 : 32;
  if (!(W == 32 || W == 16 || W == 8) || C % ke != 0) return -1;
  P3Args a{};
  int G = 0, TH = 0, BM = 256;
  const int ntn = K % 160 == 0 ? K / 160 : (K % 128 == 0 ? K / 128 : (K > 32 ? cdiv(K, 64) : 1));
  bool ok = tile_geom(N, H, W, 256, G, TH);
  if (ok && (long)(N / G) * (H / TH) * ntn < 256) ok = false;      // too few workgroups for 256 CUs: use the smaller tile
  if (!ok) {
    BM = 128;
    if (!tile_geom(N, H, W, 128, G, TH)) return -1;
  }
  a.src = src; a.wt = wt; a.dst = dst; a.res = res;
  a.N = N; a.H = H; a.W = W; a.C = C; a.K = K;
  a.G = G; a.TH = TH; a.PH = TH + 2; a.PW = W + 2; a.PP = G * a.PH * a.PW;
  a.tiles_h = H / TH; a.nchunks = C / ke; a.accum = accum;
  for (int r = 0; r < 3; ++r)
    for (int t = 0; t < 3; ++t) {
      const int i = r * 3 + t;
      const int dh = flip ? 1 - r : r - 1, dw = flip ? 1 - t : t - 1;
      a.poff[i] = (dh + 1) * a.PW + (dw + 1);
      a.widx[i] = i;
    }
  hipStream_t st = as_stream(s);
  if (BM == 256) {
    if (a.PP > 400) return -1;
    return dtype == RN_F32 ? dispatch_bn<float, 256, 3, 400>(a, st) : dispatch_bn<bf16_t, 256, 3, 400>(a, st);
  }
  if (a.PP > 208) return -1;
  return dtype == RN_F32 ? dispatch_bn<float, 128, 1, 208>(a, st) : dispatch_bn<bf16_t, 128, 1, 208>(a, st);
}
