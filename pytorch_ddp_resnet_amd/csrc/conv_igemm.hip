// Implicit-GEMM convolution on MFMA (gfx950): forward conv and data-gradient (transposed conv) of the block
// convolutions of the reference (residual_block.py:34-57 3x3 s1|s2 p1 and 1x1 projections, :129-159 bottleneck
// 1x1/3x3/1x1; bias=False everywhere).  One kernel covers all of them through a "tap table":
//
//   dst[n, p*ds+oh, q*ds+ow, k] (+)= sum_t sum_c src[n, p*ss+dh[t], q*ss+dw[t], c] * wt[k][widx[t]][c]   (+ res)
//
// GEMM view: M = N*Pc*Qc compute-grid pixels (MFMA rows), N = Kd output channels (MFMA columns), K = taps*Cs.
// forward:  ss=stride, taps=(r-pad, s-pad), ds=1.   dgrad stride 1: ss=1, taps=(pad-r, pad-s), wt=CRSK pack.
// dgrad stride 2: one launch per output parity class (ds=2, (oh,ow) in {0,1}^2) with that class's tap subset, so no
// MFMA work is spent on structural zeros.
//
// Tiling: 128 x BN block tile (BN in {160,128,96,64,32}), 32x32 MFMA tiles per wave; K streamed in 128-byte rows (64 bf16 /
// 32 f32) global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) into a ring of XOR-swizzled stages, counted s_waitcnt vmcnt
// and raw s_barrier (one per K tile).  Two schedules: igemm_dma_kernel (4 waves load and multiply, 2 stages, two workgroups
// per CU) and igemm_ws_kernel (4 consumer + 4 loader waves, 3 stages, one workgroup per CU); launch_cfg picks by tile count.
// f32 uses v_mfma_f32_32x32x2_f32 (exact f32 FMA chain), bf16 v_mfma_f32_32x32x16_bf16, both fed by one 16-byte
// ds_read_b128 per operand per k-step.  The epilogue (igemm_epilogue) stages the accumulators through LDS and fuses bias,
// residual, accumulate and the BatchNorm sums.  DESIGN.md sections 4-6 hold the measurements behind each choice.
#include "igemm_shared.h"

int g_rn_variant = 0;   // tuning switch (tools/conv_bench.py): 64 DMA source-window timing probe,
                        // 128 force / 512 forbid the wave-specialised kernel, 8192.. epilogue timing probes (fill_ep)
extern "C" void rn_set_variant(int v) { g_rn_variant = v; }
void* g_rn_stamps = nullptr;
extern "C" void rn_set_stamp_buffer(void* p) { g_rn_stamps = p; }

#define RN_CONV_CHECK_EP RN_CHECK_ARG(!ep || ((ep->partial || ep->bias) && !ep->bn_x), "rn_conv_fwd: the forward epilogue takes `partial` and/or `bias`");

namespace {


// NSTG LDS stages form a ring: at iteration `it` the DMA of tile it+NSTG-1 is issued while tile `it` is multiplied, so a
// tile has NSTG-2 full iterations to land (global latency under full-chip load is several thousand cycles: one
// iteration of cover is not enough).  Waits are COUNTED (`s_waitcnt vmcnt(k * PER)`, PER = DMA instructions a wave issues
// per tile, identical for all waves: the B tile is padded with out-of-range dummies) and the barrier is the raw
// `s_barrier`: `__syncthreads()` would make hipcc drain every DMA in flight.  All LDS lives in ONE __shared__ array (a
// second __shared__ object makes hipcc wait vmcnt(0) before every ds_read; cdna_hip_programming.md section 5).
// (the body is a device function over the argument record, the workgroup's index and the workgroups of the record: igemm_dma_classes_kernel below
// runs it over one record of several)
template <typename T, int BM, int BN, int WM, int WN, int CPRT, int NSTG>
__device__ __forceinline__ void igemm_dma_body(const IgemmArgs& a, const int wg, const int nwg_all) {
  constexpr int NW = WM * WN;                          // waves per workgroup
  constexpr int ES = (int)sizeof(T);
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int RPI = 64 / CPRT;                       // rows per DMA instruction (1 KiB)
  constexpr int AIT = BM / RPI, BIT = BN / RPI;        // DMA instructions per tile
  constexpr int AI = AIT / NW, BI = (BIT + NW - 1) / NW;   // per wave (B padded to a multiple of NW instructions)
  constexpr int PER = AI + BI;
  constexpr int KS = CPRT / 2;
  constexpr int NST = CPRT == 8 ? 2 : 1;               // distinct logical chunk columns a lane serves
  constexpr bool BDUMMY = NW * BI > BIT;               // padding DMAs (equal instruction count per wave) land in one shared dummy KiB
  constexpr int STAGE = (BM + BN) * CPRT + (BDUMMY ? 64 : 0);   // uint4 per stage
  static_assert((NW == 4 || NW == 8) && BM % (WM * 32) == 0 && BN % (WN * 32) == 0 && AIT % NW == 0 && BN % RPI == 0 && NSTG >= 2 && NSTG <= 4, "tile");
  __shared__ uint4 smem[NSTG * STAGE + TAP_INTS / 4];
  int* taps = reinterpret_cast<int*>(&smem[NSTG * STAGE]);     // fill_tap_tables

  preload_args(a);
  const int tid = threadIdx.x, lane = tid & 63;
  stamp(a.stamps, 0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: feeds M0 / SGPR operands
  const int nmt = (a.M + BM - 1) / BM;
  // tile order: the column tiles of one row tile are neighbours (they read the same input rows), then the next row tile
  // (which shares its halo rows); the XCD remap puts consecutive tiles on ONE XCD, so those re-reads hit its L2
  int bid = wg;
  if (a.xcd_remap) {
    const int nwg = nwg_all, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nnt_ = nwg_all / nmt;
  const int ntile = a.xcd_remap ? bid % nnt_ : bid / nmt, mt = a.xcd_remap ? bid / nnt_ : bid % nmt;
  const int m0 = mt * BM, n0 = ntile * BN;
  const int pq = a.Pc * a.Qc;

  const int n_first = m0 / pq;
  const size_t img_bytes = (size_t)a.Hs * a.Ws * a.Cs * ES;
  const size_t a_left = (size_t)(a.N - n_first) * img_bytes;
  const v4i32 ra_desc = make_desc(reinterpret_cast<const char*>(a.src) + (size_t)n_first * img_bytes, a_left);
  const size_t w_total = (size_t)a.Kd * a.wrs * a.Cs * ES;
  const v4i32 rb_desc = make_desc(a.wt, w_total);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);           // LDS byte address of the ring

  stamp(a.stamps, 10);
  fill_tap_tables<ES>(a, taps);
  __syncthreads();          // tap tables visible (no DMA in flight yet)
  stamp(a.stamps, 9);

  // ---- per-lane DMA roles ----
  const int lrow = lane / CPRT, p = lane % CPRT;
  unsigned abase[AI];
  unsigned long long amask[AI];
  TapGrid grid;
  load_tap_grid(a, taps, grid);
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + RPI * (wave * AI + i) + lrow;
    amask[i] = 0; abase[i] = 0;
    if (m < a.M) {
      if (a.dense_src) {                                 // 1x1, stride 1: no decode, no padding
        abase[i] = (unsigned)((size_t)(m - n_first * pq) * a.Cs * ES);
        amask[i] = 1;
      } else {
        int n, pp, q;
        decode_row(a, m, pq, n, pp, q);
        const int hb = pp * a.ss, wb = q * a.ss;
        abase[i] = (unsigned)((((size_t)(n - n_first) * a.Hs + hb) * a.Ws + wb) * a.Cs * ES);
        amask[i] = tap_mask(a, grid, hb, wb);
      }
    }
  }
  unsigned bbase[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int rn = RPI * (wave * BI + i) + lrow;
    const int k = n0 + rn;
    bbase[i] = (rn < BN && k < a.Kd) ? (unsigned)((size_t)k * a.wrs * a.Cs * ES) : OOB;     // padded rows: zeros, never read
  }
  TapWalk<CPRT> walk[NST];
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int sw = CPRT == 8 ? (((lane >> 4) + 4 * k) & 7) : ((lane >> 4) & 3);
    walk[k].init(a, taps, p ^ sw);
  }
  stamp(a.stamps, 7);

  auto dma_tile = [&](int stg) {
    if (a.probe_k == 2) return;
    const unsigned base = lds0 + (unsigned)(stg * STAGE * 16);
    const unsigned keep = m0_save();
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int j = wave * AI + i;
      bool kv; int tp; unsigned so, wo;
      pick_walk<CPRT, NST>(walk, j, kv, tp, so, wo);
      const unsigned off = (kv && ((amask[i] >> tp) & 1) && a.probe_k != 3) ? ((abase[i] + so) & a.probe_mask) : OOB;
      dma16(ra_desc, off, base + j * 1024);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int j = wave * BI + i;
      bool kv; int tp; unsigned so, wo;
      pick_walk<CPRT, NST>(walk, j, kv, tp, so, wo);
      const unsigned off = (kv && bbase[i] != OOB && a.probe_k != 3) ? ((bbase[i] + wo) & a.probe_mask) : OOB;
      dma16(rb_desc, off, base + (j < BIT ? BM * CPRT * 16 + j * 1024 : (BM + BN) * CPRT * 16));
    }
    m0_restore(keep);
#pragma unroll
    for (int k = 0; k < NST; ++k) walk[k].advance(a, taps);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int arow0 = wm * (BM / WM) + lr, brow0 = wn * (BN / WN) + lr;

  // prologue: EVERY ring slot is free, so tiles 0 .. NSTG-1 are issued before the wait for tile 0 (iteration 0 issues nothing).  Issuing
  // tile NSTG-1 at the head of iteration 0 instead, behind that wait, exposed its latency on every workgroup: 1 us of the 8.5 us a
  // 128 x 128 tile of a 1x1 convolution with C = 128 lives (two K steps; in-kernel stamps, WRN-50-2's 56 x 56 expansions)
#pragma unroll
  for (int t = 0; t < NSTG; ++t)
    if (t < a.nk) dma_tile(t);
  stamp(a.stamps, 8);
  {
    const int inflight = min(a.nk, NSTG) - 1;          // groups allowed to stay outstanding behind tile 0
    static_assert(3 * PER < 64, "vmcnt range");
    if (inflight >= 3) wait_vmcnt<3 * PER>(); else if (inflight == 2) wait_vmcnt<2 * PER>(); else if (inflight == 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  stamp(a.stamps, 1);
  int stg = 0;
  for (int it = 0; it < a.nk; ++it) {
    const uint4* cur_s = &smem[stg * STAGE];
    int nstg = stg + (NSTG - 1); if (nstg >= NSTG) nstg -= NSTG;
    if (it >= 1 && it + NSTG - 1 < a.nk) dma_tile(nstg);          // ring slot read last in iteration it-1: free since the barrier
    if (a.probe_k != 1) {
    uint4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = cur_s[swz_t<CPRT>(arow0 + 32 * i, lh)];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = cur_s[BM * CPRT + swz_t<CPRT>(brow0 + 32 * j, lh)];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < KS) {
        const int ch = 2 * (ks + 1) + lh;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[nxt][i] = cur_s[swz_t<CPRT>(arow0 + 32 * i, ch)];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[nxt][j] = cur_s[BM * CPRT + swz_t<CPRT>(brow0 + 32 * j, ch)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mfma<T>::run(fa[cur][i], fb[cur][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    // (spreading a tile's DMAs over the k-steps instead of issuing them at the head of the iteration was measured 2-6 % slower
  // with two ring stages: the late ones have no time to land)
  // tile it+1 must have landed (for every wave) before the next iteration reads it; later tiles stay in flight
    const int later = min(a.nk - 2 - it, NSTG - 2);    // DMA groups issued after tile it+1 that exist
    if (later >= 2) wait_vmcnt<2 * PER>(); else if (later == 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (++stg == NSTG) stg = 0;
  }

  // 32-column tiles (thin layers) stage the whole tile at once: their kernels are a chain of latencies, and every pass is two barriers
  constexpr int SR = CPRT == 4 ? 32 : (BN == 32 ? BM : 64);              // 64-byte K rows: the ring is half as large, so is the staged slab
  static_assert((size_t)SR * (BN + 4) * 4 + (size_t)(NW * 64 / (BN / Elem<T>::CE)) * 2 * BN * 4 <= sizeof(smem), "epilogue staging exceeds the ring");
  igemm_epilogue<T, BM, BN, WM, WN, TM, TN, NW * 64, SR>(a, acc, m0, n0, wave, lane, reinterpret_cast<float*>(&smem[0]));
}

template <typename T, int BM, int BN, int WM, int WN, int CPRT, int NSTG>
__global__ __launch_bounds__(WM * WN * 64, 2) void igemm_dma_kernel(const IgemmArgs a) {   // 2 waves per SIMD: <= 256 registers (two 4-wave workgroups per CU)
  igemm_dma_body<T, BM, BN, WM, WN, CPRT, NSTG>(a, (int)blockIdx.x, (int)gridDim.x);
}

// The parity classes of a stride-2 data gradient as ONE grid (they were four launches, each a fraction of a chip-filling grid with its own ramp and
// tail -- 250 TFLOP/s on WRN-28-10's 160 -> 320 and 320 -> 640 layers): the classes write disjoint pixels of dx and disjoint rows of the fused sums, so
// their tiles are one pool of independent workgroups.  Record i = the argument record rn_conv_dgrad builds for class i (its tap subset, its compute
// grid, its row offset into the sums) and owns the block range [first[i], first[i] + nwg[i]); ranges start on multiples of 8 so that blockIdx & 7 is
// still the XCD inside a record (the remap of the body), padding blocks exit.  Four records are 3.5 KiB of the 4 KiB kernel-argument segment.
struct IgemmClasses {
  int n;
  int first[4], nwg[4];
  IgemmArgs a[4];
};
static_assert(sizeof(IgemmClasses) <= 3840, "kernel-argument segment (4 KiB with the implicit arguments)");

template <typename T, int BM, int BN, int WM, int WN, int CPRT, int NSTG>
__global__ __launch_bounds__(WM * WN * 64, 2) void igemm_dma_classes_kernel(const IgemmClasses p) {
  int i = 0;
  while (i + 1 < p.n && (int)blockIdx.x >= p.first[i + 1]) ++i;            // workgroup-uniform
  const int wg = (int)blockIdx.x - p.first[i];
  if (wg >= p.nwg[i]) return;
  igemm_dma_body<T, BM, BN, WM, WN, CPRT, NSTG>(p.a[i], wg, p.nwg[i]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant: a workgroup = 4 CONSUMER waves (fragment ds_reads + MFMA, nothing else) + 4 LOADER waves
// (LDS-DMA of the next K tile, nothing else), one barrier per K tile.  Issuing a 64-address buffer_load costs the
// issuing wave ~100 cycles; in the homogeneous kernels every wave paid 9 of those per 20 MFMAs and both waves of a SIMD
// ran in phase.  Here each SIMD hosts consumers and loaders, so the matrix pipe runs under the DMA issue
// (MI355X_MICROARCH.md "Two waves per SIMD": matrix beside memory is the complementary pairing).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int CPRT, int NSTG>
__global__ __launch_bounds__(512, BM >= 256 ? 1 : 2) void igemm_ws_kernel(const IgemmArgs a) {
  constexpr int ES = (int)sizeof(T);
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int RPI = 64 / CPRT;
  constexpr int AIT = BM / RPI, BIT = BN / RPI;
  constexpr int AI = AIT / 4, BI = (BIT + 3) / 4;      // per LOADER wave
  constexpr int KS = CPRT / 2;
  constexpr int NST = CPRT == 8 ? 2 : 1;
  constexpr int STAGE = (BM + 4 * BI * RPI) * CPRT;
  constexpr int PER = AI + BI;                         // DMA instructions per loader wave per tile
  static_assert(WM * WN == 4 && BM % (WM * 32) == 0 && BN % (WN * 32) == 0 && AIT % 4 == 0 && BN % RPI == 0, "tile");
  __shared__ uint4 smem[NSTG * STAGE + TAP_INTS / 4];
  int* taps = reinterpret_cast<int*>(&smem[NSTG * STAGE]);

  preload_args(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int nmt = (a.M + BM - 1) / BM;
  // tile order: the column tiles of one row tile are neighbours (they read the same input rows), then the next row tile
  // (which shares its halo rows); the XCD remap puts consecutive tiles on ONE XCD, so those re-reads hit its L2
  int bid = blockIdx.x;
  if (a.xcd_remap) {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nnt_ = gridDim.x / nmt;
  const int ntile = a.xcd_remap ? bid % nnt_ : bid / nmt, mt = a.xcd_remap ? bid / nnt_ : bid % nmt;
  const int m0 = mt * BM, n0 = ntile * BN;
  const int pq = a.Pc * a.Qc;

  fill_tap_tables<ES>(a, taps);
  __syncthreads();

  f32x16 acc[TM][TN];
  if (loader) {
    const int lw = wave - 4;
    const int n_first = m0 / pq;
    const size_t img_bytes = (size_t)a.Hs * a.Ws * a.Cs * ES;
    const v4i32 ra_desc = make_desc(reinterpret_cast<const char*>(a.src) + (size_t)n_first * img_bytes, (size_t)(a.N - n_first) * img_bytes);
    const v4i32 rb_desc = make_desc(a.wt, (size_t)a.Kd * a.wrs * a.Cs * ES);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);
    const int lrow = lane / CPRT, p = lane % CPRT;
    unsigned abase[AI], bbase[BI];
    unsigned long long amask[AI];
    TapGrid grid;
    load_tap_grid(a, taps, grid);
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int m = m0 + RPI * (lw * AI + i) + lrow;
      amask[i] = 0; abase[i] = 0;
      if (m < a.M) {
        if (a.dense_src) {
          abase[i] = (unsigned)((size_t)(m - n_first * pq) * a.Cs * ES);
          amask[i] = 1;
        } else {
          int n, pp, q;
          decode_row(a, m, pq, n, pp, q);
          const int hb = pp * a.ss, wb = q * a.ss;
          abase[i] = (unsigned)((((size_t)(n - n_first) * a.Hs + hb) * a.Ws + wb) * a.Cs * ES);
          amask[i] = tap_mask(a, grid, hb, wb);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int rn = RPI * (lw * BI + i) + lrow;
      const int k = n0 + rn;
      bbase[i] = (rn < BN && k < a.Kd) ? (unsigned)((size_t)k * a.wrs * a.Cs * ES) : OOB;
    }
    TapWalk<CPRT> walk[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
      const int sw = CPRT == 8 ? (((lane >> 4) + 4 * k) & 7) : ((lane >> 4) & 3);
      walk[k].init(a, taps, p ^ sw);
    }
    auto dma_tile = [&](int stg) {
      const unsigned base = lds0 + (unsigned)(stg * STAGE * 16);
      const unsigned keep = m0_save();
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int j = lw * AI + i;
        bool kv; int tp; unsigned so, wo;
        pick_walk<CPRT, NST>(walk, j, kv, tp, so, wo);
        dma16(ra_desc, (kv && ((amask[i] >> tp) & 1)) ? abase[i] + so : OOB, base + j * 1024);
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int j = lw * BI + i;
        bool kv; int tp; unsigned so, wo;
        pick_walk<CPRT, NST>(walk, j, kv, tp, so, wo);
        dma16(rb_desc, (kv && bbase[i] != OOB) ? bbase[i] + wo : OOB, base + BM * CPRT * 16 + j * 1024);
      }
      m0_restore(keep);
#pragma unroll
      for (int k = 0; k < NST; ++k) walk[k].advance(a, taps);
    };
    // ring of NSTG stages: tiles it+1 .. it+NSTG-1 are in flight while the consumers multiply tile it; the whole ring is free at the
    // start, so tile NSTG-1 is issued with the others, before the wait for tile 0 (iteration 0 issues nothing)
#pragma unroll
    for (int t = 0; t < NSTG; ++t)
      if (t < a.nk) dma_tile(t);
    {
      const int behind = min(a.nk, NSTG) - 1;
      static_assert(3 * PER < 64, "vmcnt range");
      if (behind >= 3) wait_vmcnt<3 * PER>(); else if (behind == 2) wait_vmcnt<2 * PER>(); else if (behind == 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    int wstg = NSTG - 1;
    for (int it = 0; it < a.nk; ++it) {
      if (it >= 1 && it + NSTG - 1 < a.nk) dma_tile(wstg);            // the slot the consumers finished with in iteration it-1
      if (++wstg == NSTG) wstg = 0;
      const int later = min(a.nk - 2 - it, NSTG - 2);      // DMA groups issued after tile it+1
      if (later >= 2) wait_vmcnt<2 * PER>(); else if (later == 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int arow0 = wm * (BM / WM) + lr, brow0 = wn * (BN / WN) + lr;
    __builtin_amdgcn_s_barrier();                        // tile 0 landed
    int rstg = 0;
    for (int it = 0; it < a.nk; ++it) {
      const uint4* cur_s = &smem[rstg * STAGE];
      if (++rstg == NSTG) rstg = 0;
      uint4 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = cur_s[swz_t<CPRT>(arow0 + 32 * i, lh)];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = cur_s[BM * CPRT + swz_t<CPRT>(brow0 + 32 * j, lh)];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < KS) {
          const int ch = 2 * (ks + 1) + lh;
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[nxt][i] = cur_s[swz_t<CPRT>(arow0 + 32 * i, ch)];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[nxt][j] = cur_s[BM * CPRT + swz_t<CPRT>(brow0 + 32 * j, ch)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mfma<T>::run(fa[cur][i], fb[cur][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  constexpr int SR = BN == 32 ? BM : 64;               // as in igemm_dma_kernel
  static_assert((size_t)SR * (BN + 4) * 4 + (size_t)(512 / (BN / Elem<T>::CE)) * 2 * BN * 4 <= sizeof(smem), "epilogue staging exceeds the ring");
  igemm_epilogue<T, BM, BN, WM, WN, TM, TN, 512, SR>(a, acc, m0, n0, loader ? 0 : wave, lane, reinterpret_cast<float*>(&smem[0]), !loader);
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const IgemmArgs& a, hipStream_t s) {
  int nmt = cdiv(a.M, BM), nnt = cdiv(a.Kd, BN);
  // when the grid leaves at most one workgroup per CU (256 CUs), the wave-specialised kernel (4 consumer + 4 loader waves,
  // 3-stage DMA ring) keeps the matrix pipe fed (+21 % measured on WRN-28-10's 8x8 stage); with two or more workgroups per
  // CU the homogeneous 4-wave DMA kernel at two workgroups per CU is faster (measured).  rn_set_variant: 128 forces the
  // wave-specialised kernel, 512 forbids it (A/B in tools/conv_bench.py).
  const bool one_per_cu = nmt * nnt <= 256;
  const bool ws = (g_rn_variant & 128) || (one_per_cu && !(g_rn_variant & 512));
  rn_note_kernel("igemm_%s<%dx%d>", ws ? "ws" : "dma", BM, BN);
  if (rn_dry_run()) return 0;
  if (ws)
    hipLaunchKernelGGL((igemm_ws_kernel<T, BM, BN, WM, WN, 8, 3>), dim3(nmt * nnt), dim3(512), 0, s, a);
  else
    hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, WM, WN, 8, 2>), dim3(nmt * nnt), dim3(256), 0, s, a);
  RN_CHECK_LAUNCH("igemm");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-resident input patch for 3x3 stride-1 convolutions (forward and the stride-1 data gradient), 16-bit element types.
//
// Why: the implicit-GEMM kernels above are bound by the rate at which a CU can issue LDS-DMA instructions, not by the matrix
// pipe (round-2 timing probes, WRN-28-10 stage 1: K loop with the DMAs alone 72 us, with the MFMAs alone 54 us, both 84 us;
// a DMA whose 64 lanes are all out of range costs the same, so it is instruction issue, ~40 cycles per 1 KiB instruction and
// CU, not memory).  An im2col tile loads every input pixel once per tap: (BM + BN) * 128 B per 64-deep K step.  Here the
// input pixels an output tile needs (its rows plus a one-pixel halo, explicit zero pad columns) are loaded ONCE per 64-channel
// chunk into LDS and read by all nine taps through a per-tap pixel offset; only the weight tile of the (tap, chunk) step is
// streamed.  (A 256-pixel, 8-wave, one-workgroup-per-CU form with a double-buffered patch was built first and measured 0-20 %
// slower than the 128-pixel form below -- DESIGN.md 6a -- and is gone.)
//
// Tile: 128 consecutive output pixels = whole image rows (H*W >= 128) or whole images, at the proven occupancy of igemm_dma_kernel:
// 4 waves, TWO workgroups per CU (each hides the other's prologue, epilogue, DMA latency and patch reload).  The per-CU load path moves ~25 bytes per clock whatever the
// instruction mix (an LDS-DMA instruction costs ~5 cycles per 128-byte line it touches, out-of-range lanes included; 64-byte rows
// use half of every line and were measured slower), so the lever is BYTES per FLOP: the patch of a 64-channel chunk (208 pixels x
// 128 B = 26 KiB, single buffer: reloaded between chunks while the other workgroup computes) + one 20 KiB weight tile per
// (chunk, tap) step = 23 KiB per step instead of the im2col kernel's 36 KiB, at the same MFMA work per wave and barrier.
// MFMA shape: v_mfma_f32_16x16x32 (a wave's 32 x BN block = 2 x BN/16 tiles; one 64-channel step = two 32-channel k-steps) -- the chip
// holds a higher clock on it than on 32x32x16 in this loop (-2.5 / -7.5 % per launch, bit-identical results).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PATCH128_PP_MAX = 208;                     // 6x34 (W=32), 10x18 (W=16), 2 x 10x10 (W=8)
// FULLC: the channel count is a multiple of 64 (every chunk runs both k-steps: no tail branches in the hot loop)
// WN: the four waves as 2 x 2 (a wave = 64 rows x BN/2 columns: 9 fragment reads per 20 MFMAs; 4 x 1 -- 32 rows x BN columns, 12 reads -- measured
// equal on stage 1 and 2 % slower on stage 2)
template <typename T, int BN, bool FULLC, int WN = 2>
__global__ __launch_bounds__(256, 2) void igemm_patch128_kernel(const IgemmArgs a) {
  constexpr int BM = 128, NW = 4, ES = (int)sizeof(T);
  constexpr int WM = NW / WN, RT = BM / WM / 16, CT = BN / WN / 16;      // 16-row / 16-column MFMA tiles of a wave
  static_assert((BN / WN) % 16 == 0 && RT % 2 == 0, "wave tile");
  constexpr int CHK = 128 / ES;                          // channels per chunk (one 128-byte LDS row per pixel)
  constexpr int ASZ = PATCH128_PP_MAX * 8;               // uint4 in the patch buffer
  constexpr int BIT = BN / 8;                            // weight-tile DMA instructions per step (8 rows of 128 bytes each)
  constexpr int BI = (BIT + NW - 1) / NW;                // per wave
  constexpr int BSZ = BN * 8;                            // uint4 per ring stage
  constexpr int AI = (PATCH128_PP_MAX / 8 + NW - 1) / NW;    // patch DMAs per wave and chunk (26 instructions)
  static_assert(ES == 2 && BN % 32 == 0 && BIT % NW == 0, "patch128 tile");
  __shared__ uint4 smem[ASZ + 2 * BSZ + TAP_INTS / 4];
  int* taps = reinterpret_cast<int*>(&smem[ASZ + 2 * BSZ]);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  stamp(a.stamps, 0);
  const int nmt = (a.M + BM - 1) / BM;
  int bid = blockIdx.x;
  if (a.xcd_remap) {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nnt_ = gridDim.x / nmt;
  const int ntile = a.xcd_remap ? bid % nnt_ : bid / nmt, mt = a.xcd_remap ? bid / nnt_ : bid % nmt;
  const int m0 = mt * BM, n0 = ntile * BN;
  const int H = a.Hs, W = a.Ws, HW = H * W, W2 = W + 2;
  const bool multi = HW < BM;
  const int n_first = m0 / HW;
  const int h_first = multi ? 0 : (m0 - n_first * HW) / W;
  const int slab_rows = multi ? H + 2 : BM / W + 2;
  const int nimg = multi ? BM / HW : 1;
  const int PP = nimg * slab_rows * W2;
  const int nA = (PP + 7) >> 3;
  const int nchunk = (a.Cs + CHK - 1) / CHK;
  const int nstep = nchunk * 9;

  const size_t img_bytes = (size_t)HW * a.Cs * ES;
  const v4i32 ra_desc = make_desc(reinterpret_cast<const char*>(a.src) + (size_t)n_first * img_bytes, (size_t)(a.N - n_first) * img_bytes);
  const v4i32 rb_desc = make_desc(a.wt, (size_t)a.Kd * a.wrs * a.Cs * ES);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);

  fill_tap_tables<ES>(a, taps);
  __syncthreads();

  const int l8 = lane >> 3, sl = lane & 7;
  // the 16-byte chunk a lane copies: its slot `sl` XOR the row swizzle ((row >> 1) & 7).  Rows of one lane are 32 apart from one DMA
  // piece to the next (row = 8 (4 t + wave) + l8), so the swizzle -- and the chunk -- is the SAME for every piece: one register, not one per piece
  const int ch = sl ^ ((wave * 4 + (l8 >> 1)) & 7);
  // piece t of a lane is patch pixel pp = 8 (NW t + wave) + l8: its (image, slab row, column) is divided out ONCE and then walked
  // 8 NW pixels at a time (two runtime integer divisions per piece, 26 pieces: 2.5 us of ALU in front of every tile's first DMA --
  // in-kernel stamps, setup 4.1 us of a 33 us tile)
  unsigned aoff[AI];
  {
    int pp = wave * 8 + l8;
    int prow = pp / W2, pcol = pp - prow * W2;
    int img = prow / slab_rows, hr = prow - img * slab_rows;
#pragma unroll
    for (int t = 0; t < AI; ++t) {
      const int idx = t * NW + wave;
      aoff[t] = OOB;
      if (idx < nA && pp < PP) {
        const int h = h_first - 1 + hr, n = n_first + img;
        if (pcol >= 1 && pcol <= W && h >= 0 && h < H && n < a.N)
          aoff[t] = ((unsigned)((img * H + h) * W + (pcol - 1))) * (unsigned)(a.Cs * ES) + (unsigned)(ch * 16);    // < 4 GiB: check_geom
      }
      pp += NW * 8; pcol += NW * 8;
      while (pcol >= W2) {
        pcol -= W2;
        if (++hr == slab_rows) { hr = 0; ++img; }
      }
    }
  }
  unsigned boff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int j = i * NW + wave;
    const int rn = j * 8 + l8;
    const int k = n0 + rn;
    boff[i] = k < a.Kd ? (unsigned)((size_t)k * a.wrs * a.Cs * ES + ch * 16) : OOB;
  }
  const int l16 = lane & 15, lq = lane >> 4;             // row / column of a 16x16 tile, 8-channel group of the 32-channel k-step
  auto patch_pixel = [&](int m) {                        // tile row m -> its pixel in the patch
    if (m >= a.M) m = a.M - 1;
    const int n = m / HW, rem = m - n * HW;
    const int h = rem / W, w = rem - h * W;
    return ((n - n_first) * (multi ? slab_rows : 0) + (h - h_first) + 1) * W2 + w + 1;
  };
  const int wm = wave / WN, wn = wave % WN;
  int base_pp[RT];                                       // the wave's 16-row tiles
#pragma unroll
  for (int i = 0; i < RT; ++i) base_pp[i] = patch_pixel(m0 + wm * (BM / WM) + 16 * i + l16);
  const int bsw = (l16 >> 1) & 7;
  const int bcol0 = wn * (BN / WN);                      // first weight row (output column) of the wave; a multiple of 16: same swizzle

  f32x4 acc16[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;

  auto load_patch = [&](int chunk) {                     // the whole patch of `chunk` (AI instructions per wave)
#pragma unroll
    for (int t = 0; t < AI; ++t) {
      const int idx = t * NW + wave;
      if (idx < nA) {                                    // wave-uniform
        const bool ok = aoff[t] != OOB && chunk * 8 + ch < a.cpt;
        dma16(ra_desc, ok ? aoff[t] + (unsigned)(chunk * 128) : OOB, lds0 + (unsigned)(idx * 1024));
      }
    }
  };
  // per-tap weight byte offsets and patch pixel offsets, read ONCE (the tap index is a compile-time constant in the unrolled loop):
  // a table lookup per step put an LDS round trip in front of every step's first fragment read
  int woff_t[9], poff_t[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    woff_t[t] = __builtin_amdgcn_readfirstlane(taps[64 + t]);
    const int v = __builtin_amdgcn_readfirstlane(taps[128 + t]);
    poff_t[t] = (v >> 16) * W2 + (int)(short)(v & 0xFFFF);
  }
  auto load_weights = [&](int chunk, int woff, int stg) {   // the weight tile of a step into ring stage `stg`
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int j = i * NW + wave;
      const bool ok = boff[i] != OOB && chunk * 8 + ch < a.cpt;
      dma16(rb_desc, ok ? boff[i] + (unsigned)(woff + chunk * 128) : OOB, lds0 + (unsigned)((ASZ + stg * BSZ) * 16 + j * 1024));
    }
  };

  stamp(a.stamps, 7);
  {                                                      // both weight slots are free: step 1's tile is issued here too, behind step 0's, and
    const unsigned keep = m0_save();                     // may still be in flight when step 0 starts (step 0 then issues nothing)
    load_patch(0);
    load_weights(0, woff_t[0], 0);
    if (nstep > 1) load_weights(0, woff_t[1], 1);
    m0_restore(keep);
  }
  stamp(a.stamps, 8);
  if (nstep > 1) wait_vmcnt<BI>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  stamp(a.stamps, 1);

  // One step = (chunk, tap): two 32-channel k-steps of 2 x 2TN MFMAs per wave.  The weight fragments live in ONE buffer: fragment j of
  // the second k-step is read into the registers the first k-step's MFMAs on column j have just consumed.
  // (Deferring the last k-step's MFMAs behind the closing barrier, to cover the next step's first fragment reads, was measured +-0.)
  int step = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int kcount = FULLC ? 4 : min(4, (a.cpt - chunk * 8 + 1) >> 1);    // 16-channel k-steps of this chunk that hold data
    const bool full = FULLC || kcount == 4;
    const uint4* Ab = &smem[0];
#pragma unroll
    for (int i = 0; i < RT; ++i) asm volatile("" : "+v"(base_pp[i]));   // the per-tap fragment addresses repeat in every chunk: left alone, hipcc computes them all once and spills them
    {
#pragma unroll
      for (int t = 0; t < 9; ++t, ++step) {
        const uint4* Bb = &smem[ASZ + (step & 1) * BSZ];
        if (step + 1 < nstep && step > 0) {
          const unsigned keep = m0_save();
          load_weights(t == 8 ? chunk + 1 : chunk, woff_t[t == 8 ? 0 : t + 1], (step + 1) & 1);
          m0_restore(keep);
        }
        int pa[RT], sa[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const int pp = base_pp[i] + poff_t[t];
          pa[i] = pp * 8; sa[i] = (pp >> 1) & 7;
        }
        const bool two = full || 2 < kcount;             // the chunk's second 32 channels hold data
        uint4 ga0[RT], ga1[RT], gb[CT];
#pragma unroll
        for (int i = 0; i < RT; ++i) ga0[i] = Ab[pa[i] + (lq ^ sa[i])];
#pragma unroll
        for (int j = 0; j < CT; ++j) gb[j] = Bb[(bcol0 + l16 + 16 * j) * 8 + (lq ^ bsw)];
        if (two) {
#pragma unroll
          for (int i = 0; i < RT; ++i) ga1[i] = Ab[pa[i] + ((4 + lq) ^ sa[i])];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CT; ++j) {
#pragma unroll
          for (int i = 0; i < RT; ++i) Mfma16<T>::run(ga0[i], gb[j], acc16[i][j]);
          if (two) gb[j] = Bb[(bcol0 + l16 + 16 * j) * 8 + ((4 + lq) ^ bsw)];
          __builtin_amdgcn_sched_barrier(0);
        }
        if (two) {
#pragma unroll
          for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int i = 0; i < RT; ++i) Mfma16<T>::run(ga1[i], gb[j], acc16[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<0>();                                 // next step's weight tile landed (own DMAs), then everybody's
        __builtin_amdgcn_s_barrier();
      }
    }
    if (chunk < 2) stamp(a.stamps, 11 + 2 * chunk);     // diagnostic: end of the chunk's nine steps / patch reloaded
    if (chunk + 1 < nchunk) {                            // every wave is done with the patch: reload it for the next chunk
      const unsigned keep = m0_save();
      load_patch(chunk + 1);
      m0_restore(keep);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (chunk < 2) stamp(a.stamps, 12 + 2 * chunk);
    }
  }
  igemm_epilogue<T, BM, BN, WM, WN, RT / 2, CT, 256, 64, true>(a, acc16, m0, n0, wave, lane, reinterpret_cast<float*>(&smem[0]));
}

// geometry the LDS-patch kernel covers: 3x3 taps of a same-size stride-1 convolution (forward or data gradient), tiles of whole
// image rows / whole images, a patch that fits its LDS buffer, and a grid that keeps at least 3/4 of the CUs busy
static bool patch_ok(const IgemmArgs& a, int BN, int BM) {
  if (g_rn_variant & 2) return false;                    // A/B switch: 2 = never use the patch kernels
  if (a.nt != 9 || a.nth != 3 || a.ntw != 3 || a.ss != 1 || a.ds != 1 || a.Hs != a.Pc || a.Ws != a.Qc || a.Cs < 64) return false;
  for (int t = 0; t < 9; ++t)
    if (a.dh[t] < -1 || a.dh[t] > 1 || a.dw[t] < -1 || a.dw[t] > 1) return false;
  const int HW = a.Hs * a.Ws;
  int pp;
  if (HW >= BM) {
    if (HW % BM || BM % a.Ws) return false;
    pp = (BM / a.Ws + 2) * (a.Ws + 2);
  } else {
    if (BM % HW) return false;
    pp = (BM / HW) * (a.Hs + 2) * (a.Ws + 2);
  }
  if (pp > PATCH128_PP_MAX) return false;
  return (g_rn_variant & 16) || (long)cdiv(a.M, BM) * cdiv(a.Kd, BN) >= 192;     // 16: any grid (tests of small geometries)
}

template <typename T, int BN> int launch_patch128(const IgemmArgs& a, hipStream_t s) {
  rn_note_kernel("igemm_patch<128x%d>", BN);
  if (rn_dry_run()) return 0;
  const dim3 grid(cdiv(a.M, 128) * cdiv(a.Kd, BN));
  if (a.Cs % 64 == 0) hipLaunchKernelGGL((igemm_patch128_kernel<T, BN, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((igemm_patch128_kernel<T, BN, false>), grid, dim3(256), 0, s, a);
  RN_CHECK_LAUNCH("igemm_patch128");
  return 0;
}

// where the eight-phase kernel is the default (measured on the WRN-50-2 shapes at batch 256, forward and data gradient, per-op times inside the model:
// DESIGN.md section 6, round 3): every geometry it covers with a specialised epilogue, once the grid holds enough 256-row tiles (>= 192) to occupy the chip
// (one persistent workgroup per CU).  Parity classes of stride-2 data gradients (strided destination) stay on the 128-row kernels.
static bool igemm8_rule(const IgemmArgs& a) {
  if (a.Kd % 128) return false;
  const long tiles = (long)cdiv(a.M, 256) * (a.Kd / (a.Kd % 256 ? 128 : 256));
  return tiles >= 192 && rn_igemm8_fast(a);          // (160 tiles of 256 x 128 on WRN-28-10's 8 x 8 stage: 89 us vs 70 us for the 128 x 160 kernel)
}

template <typename T> int launch_igemm(const IgemmArgs& a, hipStream_t s) {
  if (a.M <= 0) return 0;
  const int K = a.Kd;
  // 256-row tiles on the eight-phase schedule (conv_igemm8.hip).  rn_set_variant: 1 << 22 = wherever the geometry allows, 1 << 27 = never.
  if constexpr (sizeof(T) == 2) {
    const bool force8 = (g_rn_variant & (1 << 22)) != 0, forbid8 = (g_rn_variant & (1 << 27)) != 0;
    if (!forbid8 && (force8 || igemm8_rule(a))) {
      const int e = rn_launch_igemm8(a, std::is_same<T, bf16_t>::value ? RN_BF16 : RN_F16, s);
      if (e >= 0) return e;
    }
  }
  // row-patch kernel (conv_igemm8r.hip): 3x3 stride 1 with 160 n output channels -- the WRN-28-10 family, which the tiles above do not divide.
  // rn_set_variant2: 1 = never, 2 = on any grid size.
  if constexpr (sizeof(T) == 2) {
    if (!(g_rn_variant2 & 1) && rn_igemm8r_ok(a) &&
        ((g_rn_variant2 & 2) || (long)cdiv(a.M, 256) * (K / 160) >= 192 || rn_igemm8r_split_ok(a)))  {
      const int e = rn_launch_igemm8r(a, std::is_same<T, bf16_t>::value ? RN_BF16 : RN_F16, s);
      if (e >= 0) return e;
    }
  }
  // LDS-patch kernels (3x3 stride 1, 16-bit types).  The 128-pixel kernel (two workgroups per CU, 23 KiB instead of 36 KiB of DMA per K
  // step) is the default wherever the grid keeps two workgroups on every CU (>= 512 tiles): measured in one process against the im2col
  // kernel on the WRN-28-10 shapes, forward / dgrad: stage 1 (1,024 tiles) 84.5 / 84.1 vs 88.2 / 89.0 us, stage 2 (512 tiles) 65.7 / 67.3
  // vs 71.5 / 74.6 us; on a 256-tile grid the wave-specialised im2col kernel wins (79 vs 69 us).  rn_set_variant: 2 = never,
  // 1 << 21 = wherever the geometry allows (with 16: any grid).
  if constexpr (sizeof(T) == 2) {
    const int bn = K % 160 == 0 ? 160 : (K % 128 == 0 ? 128 : 0);
    if (bn && patch_ok(a, bn, 128) && ((g_rn_variant & (1 << 21)) || (long)cdiv(a.M, 128) * cdiv(K, bn) >= 512))
      return bn == 160 ? launch_patch128<T, 160>(a, s) : launch_patch128<T, 128>(a, s);
  }
  // column tile: the widest of {160,128,96,64,32} that wastes no 32-column MFMA tile.  256-row tiles were measured and removed:
  // 4 consumer + 4 loader waves of 64 x BN, or 8 homogeneous waves with a 3-stage ring (one workgroup per CU): 5-12 % slower on
  // the WRN-28-10 shapes; 4 waves of 64 x BN with 64-byte K rows at two workgroups per CU (28 % fewer DMAs and 42 % fewer
  // fragment reads per FLOP): +5 % in isolation, +-0 in the model.
  if (K % 160 == 0) return launch_cfg<T, 128, 160, 4, 1>(a, s);
  if (K % 128 == 0) return launch_cfg<T, 128, 128, 2, 2>(a, s);
  if (K % 96 == 0) return launch_cfg<T, 128, 96, 4, 1>(a, s);
  if (K > 32) return launch_cfg<T, 128, 64, 2, 2>(a, s);
  // thin layers (K <= 32) on large maps: a tile's fixed cost (row setup, first stage, epilogue: ~8 us) dominates its two or
  // three K iterations, so 256-row tiles halve it per output row; still two workgroups per CU (74 KB of LDS each)
  if (!(g_rn_variant & 32) && (long)cdiv(a.M, 256) >= 512) {
    rn_note_kernel("igemm_dma<256x32>");
    if (rn_dry_run()) return 0;
    hipLaunchKernelGGL((igemm_dma_kernel<T, 256, 32, 4, 1, 8, 2>), dim3(cdiv(a.M, 256) * cdiv(K, 32)), dim3(256), 0, s, a);
    RN_CHECK_LAUNCH("igemm_thin");
    return 0;
  }
  return launch_cfg<T, 128, 32, 4, 1>(a, s);
}

// ---- the parity classes of a stride-2 data gradient as one grid (igemm_dma_classes_kernel) ----
// Taken when every class would run a 128-row igemm_dma / igemm_ws tile on its own (no eight-phase form, no thin 256-row form; the LDS-patch kernels
// never take a strided destination): the column tile depends on the output channels only, so the classes share it.  rn_set_variant 1 << 18 (or 128): never.
template <typename T, int BN, int WM, int WN>
int launch_classes_cfg(const IgemmClasses& p, int grid, hipStream_t s) {
  hipLaunchKernelGGL((igemm_dma_classes_kernel<T, 128, BN, WM, WN, 8, 2>), dim3(grid), dim3(256), 0, s, p);
  RN_CHECK_LAUNCH("igemm_classes");
  return 0;
}
template <typename T> bool classes_ok(const IgemmArgs* as, int n) {
  if (n < 2 || (g_rn_variant & ((1 << 18) | 128))) return false;        // (128 forces the wave-specialised kernel: one launch per class)
  for (int i = 0; i < n; ++i) {
    const IgemmArgs& a = as[i];
    if (a.M <= 0) return false;
    if constexpr (sizeof(T) == 2) {
      const bool force8 = (g_rn_variant & (1 << 22)) != 0, forbid8 = (g_rn_variant & (1 << 27)) != 0;
      if (!forbid8 && (force8 || igemm8_rule(a))) return false;
    }
    if (a.Kd <= 32 && !(g_rn_variant & 32) && (long)cdiv(a.M, 256) >= 512) return false;      // the thin 256-row form
  }
  return true;
}
template <typename T> int launch_classes(const IgemmArgs* as, int n, hipStream_t s) {
  const int K = as[0].Kd;
  const int bn = K % 160 == 0 ? 160 : (K % 128 == 0 ? 128 : (K % 96 == 0 ? 96 : (K > 32 ? 64 : 32)));
  IgemmClasses p{};
  p.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    p.a[i] = as[i];
    p.first[i] = blocks;
    p.nwg[i] = cdiv(as[i].M, 128) * cdiv(K, bn);
    blocks += (p.nwg[i] + 7) / 8 * 8;
    rn_note_kernel("igemm_dma<128x%d>", bn);
  }
  if (rn_dry_run()) return 0;
  if (bn == 160) return launch_classes_cfg<T, 160, 4, 1>(p, blocks, s);
  if (bn == 128) return launch_classes_cfg<T, 128, 2, 2>(p, blocks, s);
  if (bn == 96) return launch_classes_cfg<T, 96, 4, 1>(p, blocks, s);
  if (bn == 64) return launch_classes_cfg<T, 64, 2, 2>(p, blocks, s);
  return launch_classes_cfg<T, 32, 4, 1>(p, blocks, s);
}

int check_geom(const rn_conv_geom* g, int dtype, const char* who) {
  RN_CHECK_ARG(g != nullptr, "%s: null geometry", who);
  RN_CHECK_ARG(RN_DTYPE_OK(dtype), "%s: bad dtype %d", who, dtype);
  const int ce = dtype == RN_F32 ? 4 : 8;
  RN_CHECK_ARG(g->N > 0 && g->H > 0 && g->W > 0 && g->C > 0 && g->K > 0, "%s: non-positive shape", who);
  RN_CHECK_ARG(g->C % ce == 0 && g->K % ce == 0, "%s: C=%d and K=%d must be multiples of %d for this dtype", who, g->C, g->K, ce);
  RN_CHECK_ARG(g->R == g->S && g->R * g->S <= MAX_TAPS, "%s: kernel %dx%d unsupported", who, g->R, g->S);
  RN_CHECK_ARG(g->stride == 1 || g->stride == 2, "%s: stride %d unsupported", who, g->stride);
  RN_CHECK_ARG(g->P == (g->H + 2 * g->pad - g->R) / g->stride + 1 && g->Q == (g->W + 2 * g->pad - g->S) / g->stride + 1,
               "%s: inconsistent output size", who);
  RN_CHECK_ARG((long)g->N * g->H * g->W < (1L << 31) && (long)g->N * g->P * g->Q < (1L << 31), "%s: too many pixels", who);
  RN_CHECK_ARG((size_t)g->K * g->R * g->S * g->C * 4 < 0xFFFFFFE0ull, "%s: weights exceed the 4 GiB buffer-descriptor range", who);
  RN_CHECK_ARG(((size_t)g->H * g->W * g->C + (size_t)g->P * g->Q * g->K) * 4 * 130 < 0xFFFFFFE0ull, "%s: images too large for 32-bit tile offsets", who);
  return 0;
}

void fill_res(ResDesc& r, const void* res, int mode, int res_C, int dN, int dH, int dW, int dC) {
  r.ptr = res;
  r.mode = res ? mode : RN_RES_NONE;
  if (r.mode == RN_RES_SAME) { r.C = dC; r.H = dH; r.W = dW; }
  else if (r.mode == RN_RES_DOWN2PAD) { r.C = res_C; r.H = dH * 2; r.W = dW * 2; }
  else if (r.mode == RN_RES_UP2) { r.C = res_C; r.H = (dH + 1) / 2; r.W = (dW + 1) / 2; }
  else { r.C = r.H = r.W = 0; }
}

}  // namespace

static void fill_magic(IgemmArgs& a) {
  a.dense_src = (a.nt == 1 && a.dh[0] == 0 && a.dw[0] == 0 && a.ss == 1 && a.Hs == a.Pc && a.Ws == a.Qc) ? 1 : 0;
  const unsigned long long pq = (unsigned long long)a.Pc * a.Qc;
  a.magic_pq = pq <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / pq);
  a.magic_q = a.Qc <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned)a.Qc);
}

static void fill_ep(IgemmArgs& a, const rn_conv_epilogue* ep, int tile_base) {
  a.stats = ep ? ep->partial : nullptr;
  a.bn_x = ep ? ep->bn_x : nullptr;
  a.bn_mask = ep ? ep->bn_mask : nullptr;
  a.mask_from_x = (ep && ep->bn_mask && ep->mask_from_x) ? 1 : 0;
  a.bn_coef = ep ? ep->bn_coef : nullptr;
  a.gscale = ep ? ep->gscale : 1.f;
  a.tile_base = tile_base;
  a.bias = ep ? ep->bias : nullptr;
  a.probe_mask = (g_rn_variant & 64) ? 0x0000FFF0u : 0xFFFFFFFFu;
  a.stamps = reinterpret_cast<unsigned long long*>(g_rn_stamps);
  a.xcd_remap = (g_rn_variant & 8) ? 0 : 1;
  a.probe_k = (g_rn_variant & 1024) ? 1 : ((g_rn_variant & 2048) ? 2 : ((g_rn_variant & 4096) ? 3 : 0));
  a.probe_ep = (g_rn_variant & 8192) ? 1 : ((g_rn_variant & 16384) ? 2 : ((g_rn_variant & 32768) ? 3 : ((g_rn_variant & 65536) ? 4 : 0)));
}

extern "C" int rn_conv_stats_rows(const rn_conv_geom* g, int is_dgrad) {
  if (!g) return 0;
  if (!is_dgrad) return cdiv((long)g->N * g->P * g->Q, RN_CONV_STATS_ROWS);
  int rows = 0;
  for (int pa = 0; pa < g->stride; ++pa)
    for (int pb = 0; pb < g->stride; ++pb) {
      const long pc = (g->H - pa + g->stride - 1) / g->stride, qc = (g->W - pb + g->stride - 1) / g->stride;
      if (pc > 0 && qc > 0) rows += cdiv((long)g->N * pc * qc, RN_CONV_STATS_ROWS);
    }
  return rows;
}

extern "C" int rn_conv_fwd(const void* x, const void* w_fwd, void* y, const void* res, int res_mode, int res_C, int dtype,
                           const rn_conv_geom* g, const rn_conv_epilogue* ep, rn_stream s) {
  if (int e = check_geom(g, dtype, "rn_conv_fwd")) return e;
  RN_CHECK_ARG(x && w_fwd && y, "rn_conv_fwd: null pointer");
  IgemmArgs a{};
  a.src = x; a.wt = w_fwd; a.dst = y;
  fill_res(a.res, res, res_mode, res_C, g->N, g->P, g->Q, g->K);
  RN_CONV_CHECK_EP
  fill_ep(a, ep, 0);
  a.N = g->N; a.Hs = g->H; a.Ws = g->W; a.Cs = g->C;
  a.Pc = g->P; a.Qc = g->Q; a.M = g->N * g->P * g->Q;
  a.Hd = g->P; a.Wd = g->Q; a.Kd = g->K;
  a.ss = g->stride; a.ds = 1; a.oh = a.ow = 0;
  a.nt = g->R * g->S; a.wrs = g->R * g->S;
  a.nth = g->R; a.ntw = g->S;
  for (int r = 0; r < g->R; ++r)
    for (int t = 0; t < g->S; ++t) {
      int i = r * g->S + t;
      a.dh[i] = r - g->pad; a.dw[i] = t - g->pad; a.widx[i] = i;
    }
  const int ce = dtype == RN_F32 ? 4 : 8;
  a.cpt = g->C / ce;
  a.nk = cdiv((long)a.nt * a.cpt, CPR);
  a.accum = 0;
  fill_magic(a);
  RN_BY_DTYPE(dtype, return launch_igemm<T_>(a, as_stream(s)));
  return 1;
}

// ---- a parity class WITHOUT a tap (stride-2 1 x 1 data gradients: three of the four classes) ----------------------------------------
// Such a class has no product to add: its pixels of dx keep their value (accumulating launch), or take the residual / zero -- but a fused BatchNorm-backward
// reduction still has to see them.  As a convolution launch that was a tile machine with an empty K loop around three operand reads (WRN-50-2-B: 1.04 ms at
// 56 x 56 against 0.46 of traffic).  Here: one workgroup per statistics row block (RN_CONV_STATS_ROWS = 128 class pixels), thread = (16-byte channel chunk
// column, row lane), two rows of operands in flight; the same values, sums and partial-row slots as the convolution epilogue writes (igemm_shared.h), and
// NO store where nothing changes (accumulate without a residual).
template <typename T>
__global__ __launch_bounds__(256) void dgrad_notap_kernel(const IgemmClasses p) {
  constexpr int CE = Elem<T>::CE;
  __shared__ float red[2][256][CE + 1];
  int ci = 0;
  while (ci + 1 < p.n && (int)blockIdx.x >= p.first[ci + 1]) ++ci;         // workgroup-uniform: the classes of one data gradient as ONE grid
  const IgemmArgs& a = p.a[ci];
  const int blk = (int)blockIdx.x - p.first[ci];
  const int tid = threadIdx.x;
  const int CC = a.Kd / CE;
  const int cols = CC < 256 ? CC : 256;                     // chunk columns per pass
  const int lanes = 256 / cols;                             // row lanes per column
  const int rl = tid / cols, cl = tid - rl * cols;
  const int m0 = blk * RN_CONV_STATS_ROWS;
  const int pq = a.Pc * a.Qc;
  T* __restrict__ dst = reinterpret_cast<T*>(a.dst);
  const bool bn_bwd = a.stats != nullptr && a.bn_x != nullptr, want_stats = a.stats != nullptr;
  const bool res_same = a.res.mode == RN_RES_SAME, has_res = a.res.mode != RN_RES_NONE;
  const bool store = !(a.accum && !has_res);                // accumulate without a residual: the pixel keeps its value
  for (int cbase = 0; cbase < CC; cbase += cols) {
    const int cg = cbase + cl;
    const bool active = rl < lanes && cg < CC;
    const int k0 = cg * CE;
    float s0[CE], s1[CE], mean[CE], invstd[CE], sc[CE], sh[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { s0[e] = s1[e] = 0.f; mean[e] = invstd[e] = sc[e] = sh[e] = 0.f; }
    if (active && bn_bwd) {
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        sc[e] = a.bn_coef[k0 + e]; sh[e] = a.bn_coef[a.Kd + k0 + e];
        mean[e] = a.bn_coef[2 * a.Kd + k0 + e]; invstd[e] = a.bn_coef[3 * a.Kd + k0 + e];
      }
    }
    struct Row { bool ok; size_t off; int n, hd, wd; Chunk<T> cr, co, cx, cm; };
    auto fetch = [&](int r, Row& w) {
      const int m = m0 + r;
      w.ok = active && r < RN_CONV_STATS_ROWS && m < a.M;
      if (!w.ok) return;
      int pp, q;
      decode_row(a, m, pq, w.n, pp, q);
      w.hd = pp * a.ds + a.oh; w.wd = q * a.ds + a.ow;
      w.off = (((size_t)w.n * a.Hd + w.hd) * a.Wd + w.wd) * a.Kd + k0;
      if (res_same) w.cr = load_chunk<T>(reinterpret_cast<const T*>(a.res.ptr) + w.off);
      if (a.accum) w.co = load_chunk<T>(dst + w.off);
      if (bn_bwd) {
        w.cx = load_chunk<T>(reinterpret_cast<const T*>(a.bn_x) + w.off);
        if (a.bn_mask) w.cm = load_chunk<T>(reinterpret_cast<const T*>(a.bn_mask) + w.off);
      }
    };
    auto process = [&](const Row& w) {
      if (!w.ok) return;
      float v[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) v[e] = 0.f;
      if (res_same) {
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] += Elem<T>::to_f(w.cr.e[e]);
      } else if (has_res) {
        res_add_chunk<T>(a.res, w.n, w.hd, w.wd, k0, v);
      }
      if (a.accum) {
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] += Elem<T>::to_f(w.co.e[e]);
      }
      Chunk<T> st;
#pragma unroll
      for (int e = 0; e < CE; ++e) st.e[e] = Elem<T>::from_f(v[e]);
      if (store) store_chunk<T>(dst + w.off, st);
      if (want_stats) {
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          const float vs = Elem<T>::to_f(st.e[e]);
          if (!bn_bwd) { s0[e] += vs; s1[e] += vs * vs; continue; }
          float g = vs * a.gscale;
          const float xv = Elem<T>::to_f(w.cx.e[e]);
          if (a.bn_mask ? !(Elem<T>::to_f(w.cm.e[e]) > 0.f) : (a.mask_from_x && !(fmaf(xv, sc[e], sh[e]) > 0.f))) g = 0.f;
          s0[e] += g; s1[e] += g * ((xv - mean[e]) * invstd[e]);
        }
      }
    };
    for (int r = rl; r < RN_CONV_STATS_ROWS; r += 2 * lanes) {
      Row w0, w1;
      fetch(r, w0); fetch(r + lanes, w1);
      process(w0); process(w1);
    }
    if (want_stats) {                                       // (uniform: every thread of the workgroup takes the same branch)
#pragma unroll
      for (int e = 0; e < CE; ++e) { red[0][tid][e] = s0[e]; red[1][tid][e] = s1[e]; }
      __syncthreads();
      if (rl == 0 && cg < CC && m0 < a.M) {
        float t0[CE], t1[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) t0[e] = t1[e] = 0.f;
        for (int l = 0; l < lanes; ++l) {
#pragma unroll
          for (int e = 0; e < CE; ++e) { t0[e] += red[0][l * cols + cl][e]; t1[e] += red[1][l * cols + cl][e]; }
        }
        float* out = a.stats + ((size_t)(a.tile_base + blk) * 2) * a.Kd + k0;
#pragma unroll
        for (int e = 0; e < CE; ++e) { out[e] = t0[e]; out[a.Kd + e] = t1[e]; }
      }
      __syncthreads();
    }
  }
}

template <typename T> int launch_notap(const IgemmArgs* as, int n, hipStream_t s) {
  IgemmClasses p{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (as[i].M <= 0) continue;
    p.a[p.n] = as[i];
    p.first[p.n] = blocks;
    p.nwg[p.n] = cdiv(as[i].M, RN_CONV_STATS_ROWS);
    blocks += p.nwg[p.n];
    ++p.n;
  }
  if (!p.n) return 0;
  rn_note_kernel("dgrad_notap");
  if (rn_dry_run()) return 0;
  hipLaunchKernelGGL((dgrad_notap_kernel<T>), dim3(blocks), dim3(256), 0, s, p);
  RN_CHECK_LAUNCH("dgrad_notap");
  return 0;
}

extern "C" int rn_conv_dgrad(const void* dy, const void* w_dgrad, void* dx, const void* res, int res_mode, int res_C, int flags,
                             int dtype, const rn_conv_geom* g, const rn_conv_epilogue* ep, rn_stream s) {
  if (int e = check_geom(g, dtype, "rn_conv_dgrad")) return e;
  RN_CHECK_ARG(dy && w_dgrad && dx, "rn_conv_dgrad: null pointer");
  const int st = g->stride;
  const int ce = dtype == RN_F32 ? 4 : 8;
  RN_CHECK_ARG(!ep || (ep->partial && ep->bn_x && ep->bn_coef && !ep->bias), "rn_conv_dgrad: incomplete epilogue descriptor");
  int tile_base = 0;
  // one argument record per parity class (a, b) of the input grid: h = st*p' + a, w = st*q' + b; launched as one grid (launch_classes) or one by one
  IgemmArgs recs[4];
  int nrec = 0;
  for (int pa = 0; pa < st; ++pa)
    for (int pb = 0; pb < st; ++pb) {
      IgemmArgs& a = recs[nrec];
      a = IgemmArgs{};
      a.src = dy; a.wt = w_dgrad; a.dst = dx;
      fill_res(a.res, res, res_mode, res_C, g->N, g->H, g->W, g->C);
      a.N = g->N; a.Hs = g->P; a.Ws = g->Q; a.Cs = g->K;
      a.Pc = (g->H - pa + st - 1) / st; a.Qc = (g->W - pb + st - 1) / st;
      if (a.Pc <= 0 || a.Qc <= 0) continue;
      a.M = g->N * a.Pc * a.Qc;
      a.Hd = g->H; a.Wd = g->W; a.Kd = g->C;
      a.ss = 1; a.ds = st; a.oh = pa; a.ow = pb;
      a.wrs = g->R * g->S;
      int nt = 0;
      for (int r = 0; r < g->R; ++r) {
        if ((pa + g->pad - r) % st != 0) continue;           // (pa + pad - r) may be negative: C '%' keeps the sign, 0 stays 0
        for (int t = 0; t < g->S; ++t) {
          if ((pb + g->pad - t) % st != 0) continue;
          // floor division for possibly negative numerators that are exact multiples of st
          a.dh[nt] = (pa + g->pad - r) / st; a.dw[nt] = (pb + g->pad - t) / st; a.widx[nt] = r * g->S + t;
          ++nt;
        }
      }
      a.nt = nt;
      a.nth = 0; a.ntw = 0;
      for (int r = 0; r < g->R; ++r) if ((pa + g->pad - r) % st == 0) ++a.nth;
      for (int t = 0; t < g->S; ++t) if ((pb + g->pad - t) % st == 0) ++a.ntw;
      a.cpt = g->K / ce;
      a.nk = cdiv((long)nt * a.cpt, CPR);
      a.accum = (flags & RN_F_ACCUM) ? 1 : 0;
      fill_ep(a, ep, tile_base);
      fill_magic(a);
      tile_base += cdiv(a.M, RN_CONV_STATS_ROWS);
      if (nt == 0 && a.accum && a.res.mode == RN_RES_NONE && !ep) continue;   // nothing to add to this class (a fused reduction still has to see it)
      ++nrec;
    }
  int e = 0;
  // classes without a tap leave the convolution kernels (rn_set_variant2 1048576: they stay, A/B)
  if (!(g_rn_variant2 & 1048576)) {
    IgemmArgs bare[4];
    int keep = 0, nbare = 0;
    for (int i = 0; i < nrec; ++i) {
      if (recs[i].nt == 0) bare[nbare++] = recs[i];
      else { if (keep != i) recs[keep] = recs[i]; ++keep; }
    }
    if (nbare) RN_BY_DTYPE(dtype, e = launch_notap<T_>(bare, nbare, as_stream(s)));
    if (e) return e;
    nrec = keep;
  }
  bool merged = false;
  RN_BY_DTYPE(dtype, merged = classes_ok<T_>(recs, nrec));
  if (merged) {
    RN_BY_DTYPE(dtype, e = launch_classes<T_>(recs, nrec, as_stream(s)));
    return e;
  }
  for (int i = 0; i < nrec && !e; ++i) RN_BY_DTYPE(dtype, e = launch_igemm<T_>(recs[i], as_stream(s)));
  return e;
}
