// Weight gradient on the EIGHT-PHASE schedule (gfx950, 16-bit element types), for layers whose channel counts are multiples of 256:
//
//   dw[k][t][c] = sum_{m=(n,p,q)} dy[m][k] * x[n, p*stride+dh[t], q*stride+dw[t], c]          (fp32, KRSC)
//
// Output tile = 256 input channels (MFMA rows) x 256 output channels (MFMA columns) of ONE tap; the reduction runs over the N*P*Q output pixels in K
// tiles of 64 pixels.  Same pipeline as conv_igemm8.hip (8 waves of 128 x 64, four phases per K tile, two wave groups one barrier apart, LDS-DMA in
// flight across barriers behind a counted vmcnt, persistent workgroups) with the operand roles of a weight gradient:
//   * both operands are pixel-major in HBM, i.e. the reduction index is the slow index of both LDS tiles ([64 pixels][256 channels], a half-tile =
//     the 128 channels one quadrant row / column of every wave reads): the fragments are TRANSPOSED reads, two ds_read_b64_tr_b16 per operand and
//     32-pixel k-step; 16-byte chunks of a pixel row are XOR-swizzled by the pixel ((pix & 3) << 1 | ((pix >> 3) & 1) << 3, on the DMA source) so the
//     eight pixel rows one transposed read touches land on sixteen distinct 16-byte slots;
//   * the pixel walk is the K loop: every K tile a lane decodes the two pixels it copies (magic division; tap shift and padding become out-of-range
//     DMA offsets = zeros); the dy rows are dense;
//   * the reduction is long (M / 64 K tiles) and the tiles are few (K/256 x C/256 x taps): the pixels are cut into S equal splits (S from a cost model:
//     rounds of 256 workgroups x K tiles per split vs the slab traffic), work item = (split, tile), split-major, so the workgroups that run side by
//     side on one XCD read the SAME pixels for different tiles and share them through its L2 (with contiguous K-tile ranges per workgroup -- a stream-K
//     split, tried first -- every workgroup streams its own pixels and the kernel is bound by the Infinity Cache: 2.7 us per K tile instead of 1.4);
//     each item writes its fp32 tile into slab [split] in the gradient's own layout and the fixed-order reduction kernels of conv_wgrad.hip sum the
//     slabs (bitwise reproducible; S = 1 writes the gradient directly).
// Epilogue: the products are taken with x as the MFMA row operand, so a lane holds four consecutive input channels of one output channel: one
// 16-byte fp32 store (or load-add-store when accumulating) per accumulator tile.
#include "igemm_shared.h"

namespace {

template <int N> __device__ inline void wait_lgkm8() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void raw_barrier8() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

struct Wg8Args {
  const void* x;
  const void* dy;
  float* dw;
  int N, H, W, C, P, Q, K;
  int stride, pad, S, RS;
  int M, nk;               // output pixels; K tiles of 64 pixels
  int nct, ntiles;         // column (input-channel) tiles; tiles = K/256 * C/256 * taps
  unsigned magic_pq, magic_q;
  int dense;               // 1x1, stride 1, no padding (pixel m reads x pixel m)
  int splits, per;         // pixel splits, K tiles per split; slab s at dw + s * slab_stride
  long slab_stride;        // K * RS * C floats (0 when splits == 1: dw is the gradient itself)
};

template <typename T> struct Tr16;
template <> struct Tr16<bf16_t> {
  __device__ static inline uint2 rd(const char* p) {
    typedef __attribute__((address_space(3))) bf16x4* lp;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p)));
  }
};
template <> struct Tr16<f16_t> {
  __device__ static inline uint2 rd(const char* p) {
    typedef __fp16 h4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) h4* lp;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p)));
  }
};

struct Rows8 { unsigned x[2], dy[2]; };      // byte offsets of the two pixels a lane copies in one K tile (OOB = zeros)

template <typename T>
__global__ __launch_bounds__(512, 2) void wgrad8_kernel(const Wg8Args a) {
  constexpr int ES = 2, RT = 8, CT = 4, QR = 4, QC = 2, AI = 2, BI = 2;
  constexpr int STGB = 65536;                                // bytes per stage: A0 | A1 | B0 | B1 (16 KiB each); the stage toggles by XOR
  constexpr int HALF = 16384;
  __shared__ uint4 smem[2 * STGB / 16];                      // two stages
  const char* lds = reinterpret_cast<const char*>(&smem[0]);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l16 = lane & 15, lq = lane >> 4;
  const int nk = a.nk, pq = a.P * a.Q;
  const v4i32 x_desc = make_desc(a.x, (size_t)a.N * a.H * a.W * a.C * ES);
  const v4i32 dy_desc = make_desc(a.dy, (size_t)a.M * a.K * ES);
  const unsigned crow = (unsigned)(a.C * ES), krow = (unsigned)(a.K * ES);

  // ---- DMA roles: instruction q of a half-tile = pixels 4q .. 4q+3 x 256 B; lane = (pixel 4q + lane / 16, physical chunk lane % 16) ----
  const int fsw = ((lq & 3) << 1) | ((wave & 1) << 3);       // f(pixel) of both pixels of this lane (see the header): pixel = 8 wave + 4 jj + lq
  const int c16 = (lane & 15) ^ fsw;                         // logical chunk held by this lane's physical chunk
  // A half h, LDS chunk c -> input channel 64 h + 8 (c & 7) + 128 (c >> 3);  B half h, LDS chunk c -> output channel 64 (c >> 2) + 32 h + 8 (c & 3)
  const unsigned a_ch = (unsigned)((8 * (c16 & 7) + 128 * (c16 >> 3)) * ES), b_ch = (unsigned)((64 * (c16 >> 2) + 8 * (c16 & 3)) * ES);

  // ---- fragment addresses (bytes inside a stage).  Transposed read: lane t = 4q + p of a 16-lane group supplies row q (pixel), columns 4p .. 4p+3 ----
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  const int frow = (8 * lq + tq) * 256 + 8 * (tp & 1);       // pixel 8 lq + tq (+ 32 ks + 4 u), second half of the chunk for odd p
  const int ff = (tq << 1) | ((lq & 1) << 3);                // f(pixel) of the pixels this lane addresses
  int fa[QR], fb[QC];
#pragma unroll
  for (int i = 0; i < QR; ++i) fa[i] = frow + ((((8 * wm + 2 * i) ^ ff) | (tp >> 1)) << 4);
#pragma unroll
  for (int j = 0; j < QC; ++j) fb[j] = 2 * HALF + frow + ((((4 * wn + 2 * j) ^ ff) | (tp >> 1)) << 4);

  // ---- work list: item = (split, tile), split-major; workgroup b takes items remap(b), remap(b + G), ... where remap hands each XCD class (b % 8) a
  // contiguous eighth of the item order, its workgroups interleaved: XCD-mates run neighbouring items = the same split, different tiles ----
  const int G = gridDim.x, nitems = a.ntiles * a.splits;
  int dh = 0, dw_ = 0;                                       // the current tile's tap
  unsigned a_base = 0, b_base = 0;                           // + channel offsets of the tile
  auto decode = [&](int g, Rows8& r) {                       // the lane's two pixels of K tile g
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int m = 64 * g + 8 * wave + 4 * jj + lq;
      unsigned xo = OOB, yo = OOB;
      if (m < a.M) {
        yo = (unsigned)m * krow + b_base;
        if (a.dense) {
          xo = (unsigned)m * crow + a_base;
        } else {
          int n = (int)__umulhi((unsigned)m, a.magic_pq);
          int rem = m - n * pq;
          if (rem >= pq) { ++n; rem -= pq; }
          int p = (int)__umulhi((unsigned)rem, a.magic_q);
          int q = rem - p * a.Q;
          if (q >= a.Q) { ++p; q -= a.Q; }
          const int hi = p * a.stride + dh, wi = q * a.stride + dw_;
          if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) xo = (unsigned)((n * a.H + hi) * a.W + wi) * crow + a_base;
        }
      }
      r.x[jj] = xo; r.dy[jj] = yo;
    }
  };
  auto issue_a = [&](int h, unsigned stage_lds, const Rows8& r) {
    const unsigned keep = m0_save();
#pragma unroll
    for (int jj = 0; jj < AI; ++jj)
      dma16(x_desc, r.x[jj] != OOB ? r.x[jj] + (unsigned)(h * 64 * ES) : OOB, stage_lds + (unsigned)(h * HALF + (wave * AI + jj) * 1024));
    m0_restore(keep);
  };
  auto issue_b = [&](int h, unsigned stage_lds, const Rows8& r) {
    const unsigned keep = m0_save();
#pragma unroll
    for (int jj = 0; jj < BI; ++jj)
      dma16(dy_desc, r.dy[jj] != OOB ? r.dy[jj] + (unsigned)(h * 32 * ES) : OOB, stage_lds + (unsigned)(2 * HALF + h * HALF + (wave * BI + jj) * 1024));
    m0_restore(keep);
  };

  f32x4 acc[RT][CT];
  uint4 af[QR][2], b0[QC][2], b1[QC][2];
  auto rd = [&](int off) {                                   // one MFMA operand: pixels 8 lq .. 8 lq + 7 of a 32-pixel k-step, two transposed reads
    const uint2 lo = Tr16<T>::rd(lds + off), hi = Tr16<T>::rd(lds + off + 4 * 256);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  auto ktile = [&](auto mode_tag, int sx, const Rows8& prev, const Rows8& cur) {
    constexpr int MODE = decltype(mode_tag)::value;
    const unsigned mine = lds0 + (unsigned)sx, other = lds0 + (unsigned)(sx ^ STGB);
    // ---- phase 1: B0, A0 -> quadrant (0, 0); stage A1 of K tile kt+1 ----
#pragma unroll
    for (int j = 0; j < QC; ++j) { b0[j][0] = rd(sx + fb[j]); b0[j][1] = rd(sx + fb[j] + 32 * 256); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < QR; ++i) { af[i][0] = rd(sx + fa[i]); af[i][1] = rd(sx + fa[i] + 32 * 256); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE >= 1) issue_a(1, other, prev);
    wait_lgkm8<15>();                                     // lgkmcnt is a 4-bit counter: at most 15 of the 16 A reads stay outstanding, i.e. the 8 B0 reads
                                                          // (issued first) are back: B0 may be restaged in phase 2
    raw_barrier8();
    wait_lgkm8<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(af[i][ks], b0[j][ks], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier8();
    // ---- phase 2: B1 -> quadrant (0, 1); stage B0 of K tile kt+2 ----
#pragma unroll
    for (int j = 0; j < QC; ++j) { b1[j][0] = rd(sx + HALF + fb[j]); b1[j][1] = rd(sx + HALF + fb[j] + 32 * 256); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 2) issue_b(0, mine, cur);
    raw_barrier8();
    wait_lgkm8<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(af[i][ks], b1[j][ks], acc[i][QC + j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier8();
    // ---- phase 3: A1 -> quadrant (1, 1); stage A0 of K tile kt+2 ----
#pragma unroll
    for (int i = 0; i < QR; ++i) { af[i][0] = rd(sx + HALF + fa[i]); af[i][1] = rd(sx + HALF + fa[i] + 32 * 256); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 2) issue_a(0, mine, cur);
    raw_barrier8();
    wait_lgkm8<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(af[i][ks], b1[j][ks], acc[QR + i][QC + j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier8();
    // ---- phase 4: quadrant (1, 0) from registers; stage B1 of K tile kt+2; K tile kt+1 has landed behind this wait ----
    if constexpr (MODE == 2) { issue_b(1, mine, cur); wait_vmcnt<AI + 2 * BI>(); }
    else if constexpr (MODE == 1) wait_vmcnt<0>();
    raw_barrier8();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(af[i][ks], b0[j][ks], acc[QR + i][j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier8();
  };

  int ptile = -1, psplit = 0;                                // the item whose accumulators are still in registers
  for (int it = 0;; ++it) {
    const int vb = it * G + (int)blockIdx.x;
    const bool more = vb < nitems;                           // wave-uniform
    int tile = 0, split = 0, kb = 0, ke = 0;
    if (more) {
      const int xcd = vb & 7, q = nitems >> 3, r = nitems & 7;
      const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
      split = item / a.ntiles; tile = item - split * a.ntiles;
      kb = split * a.per; ke = min(nk, kb + a.per);
    }
    const int nseg = ke - kb;                                // >= 1 (launcher: per * (splits - 1) < nk)
    Rows8 r0{}, r1{};
    if (more) {
      const int t = tile % a.RS, rest = tile / a.RS, ct = rest % a.nct, kt_ = rest / a.nct;
      const int tr = t / a.S;
      dh = tr - a.pad; dw_ = (t - tr * a.S) - a.pad;
      a_base = (unsigned)(ct * 256 * ES) + a_ch; b_base = (unsigned)(kt_ * 256 * ES) + b_ch;
      decode(kb, r0);
      issue_b(0, lds0, r0); issue_a(0, lds0, r0); issue_b(1, lds0, r0); issue_a(1, lds0, r0);
      if (nseg > 1) { decode(kb + 1, r1); issue_b(0, lds0 + STGB, r1); issue_a(0, lds0 + STGB, r1); issue_b(1, lds0 + STGB, r1); }
    }
    if (ptile >= 0) {        // acc[i][j][r]: input channel c0 + 128 wm + 16 i + 4 lq + r, output channel k0 + 64 wn + 16 j + l16
      const int t = ptile % a.RS, rest = ptile / a.RS, ct = rest % a.nct, kt_ = rest / a.nct;
      const int c0 = ct * 256 + 128 * wm + 4 * lq, k0 = kt_ * 256 + 64 * wn + l16;
      float* out = a.dw + (size_t)psplit * a.slab_stride;
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        float* row = out + ((size_t)(k0 + 16 * j) * a.RS + t) * a.C + c0;
#pragma unroll
        for (int i = 0; i < RT; ++i)
          *reinterpret_cast<float4*>(row + 16 * i) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    if (!more) break;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    if (nseg > 1) wait_vmcnt<AI + 2 * BI>(); else wait_vmcnt<0>();
    raw_barrier8();
    if (wave >= 4) raw_barrier8();

    int sx = 0;
    Rows8 prev = r1, cur;
    for (int kt = 0; kt + 2 < nseg; ++kt) {
      decode(kb + kt + 2, cur);
      ktile(std::integral_constant<int, 2>{}, sx, prev, cur);
      prev = cur;
      sx ^= STGB;
    }
    if (nseg > 1) {
      ktile(std::integral_constant<int, 1>{}, sx, prev, prev);
      sx ^= STGB;
    }
    ktile(std::integral_constant<int, 0>{}, sx, prev, prev);
    if (wave < 4) raw_barrier8();
    ptile = tile; psplit = split;
  }
}

}  // namespace

// pixel splits for the eight-phase kernel: minimise rounds x K tiles per split (1.4 us each) + slab traffic (written once, read once, ~4 TB/s)
static int wgrad8_pick_splits(long ntiles, long nk, double n_floats) {
  int best = 1;
  double best_cost = 1e30;
  const long smax = nk / 4 < 128 ? (nk / 4 < 1 ? 1 : nk / 4) : 128;
  for (long S = 1; S <= smax; ++S) {
    const long per = (nk + S - 1) / S;
    if (per * (S - 1) >= nk) continue;                                       // an empty last split
    const long rounds = (ntiles * S + 255) / 256;
    const double cost = rounds * per * 1.4 + (S > 1 ? S * n_floats * 8.0 / 4.0e6 : 0.0) + (S > 1 ? 6.0 : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = (int)S; }
  }
  return best;
}

// 0: rn_conv_wgrad does not take the eight-phase kernel for this geometry; S >= 1: it does, with S pixel splits (S > 1: slabs + the reduction kernels)
int rn_wgrad8_splits(const rn_conv_geom* g, int dtype) {
  if (dtype != RN_BF16 && dtype != RN_F16) return 0;
  if (g_rn_variant & (1 << 29)) return 0;                                      // A/B: never
  if (g->C % 256 || g->K % 256 || g->R != g->S) return 0;
  const long M = (long)g->N * g->P * g->Q;
  if ((double)g->N * g->H * g->W * g->C * 2 >= 4.0e9 || (double)M * g->K * 2 >= 4.0e9) return 0;      // 32-bit DMA offsets
  const long nk = (M + 63) / 64, ntiles = (long)(g->K / 256) * (g->C / 256) * g->R * g->S;
  if (ntiles * nk < 8L * 256 && !(g_rn_variant & (1 << 30))) return 0;        // too small to fill the chip (1 << 30: any size, tests)
  return wgrad8_pick_splits(ntiles, nk, (double)g->K * g->R * g->S * g->C);
}

// out: the gradient itself (splits == 1) or the slab region [splits][K][RS][C]
int rn_launch_wgrad8(const void* x, const void* dy, float* out, int splits, int dtype, const rn_conv_geom* g, int max_grid, hipStream_t s) {
  Wg8Args a{};
  a.x = x; a.dy = dy; a.dw = out;
  a.N = g->N; a.H = g->H; a.W = g->W; a.C = g->C; a.P = g->P; a.Q = g->Q; a.K = g->K;
  a.stride = g->stride; a.pad = g->pad; a.S = g->S; a.RS = g->R * g->S;
  a.M = g->N * g->P * g->Q; a.nk = (a.M + 63) / 64;
  a.nct = g->C / 256; a.ntiles = (g->K / 256) * a.nct * a.RS;
  const unsigned long long pq = (unsigned long long)g->P * g->Q;
  a.magic_pq = pq <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / pq);
  a.magic_q = g->Q <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned)g->Q);
  a.dense = (a.RS == 1 && g->stride == 1 && g->pad == 0 && g->H == g->P && g->W == g->Q) ? 1 : 0;
  a.splits = splits; a.per = (a.nk + splits - 1) / splits;
  a.slab_stride = splits > 1 ? (long)g->K * a.RS * g->C : 0;
  rn_note_kernel("wgrad8<256x256>");
  if (rn_dry_run()) return 0;
  const long items = (long)a.ntiles * splits;
  // max_grid < 256: a FORKED launch (side stream, beside the data-gradient / BatchNorm chain) leaves CUs free -- a persistent workgroup of this kernel
  // holds every vector register of its CU until its last item, so behind a full grid the chain's small kernels wait for whole workgroup lifetimes
  const int cap = max_grid > 0 && max_grid < 256 ? max_grid : 256;
  const int grid = items < cap ? (int)items : cap;
  if (dtype == RN_BF16) hipLaunchKernelGGL((wgrad8_kernel<bf16_t>), dim3(grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((wgrad8_kernel<f16_t>), dim3(grid), dim3(512), 0, s, a);
  RN_CHECK_LAUNCH("wgrad8");
  return 0;
}
