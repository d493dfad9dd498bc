// Weight gradient of 3x3 stride-1 "same" convolutions with 160 n output channels -- the body of WRN-28-10 (residual_block.py:34-47; 22 of its 27
// convolutions) -- with ALL NINE TAPS of a 32-channel input slice in one output tile:
//
//   dw[k][t][c] = sum_{m=(n,p,q)} dy[m][k] * x[n, p+dh[t], q+dw[t], c]          (fp32, KRSC)
//
// Why.  A weight gradient streams BOTH operands (the reduction runs over the pixels), and on this chip the global -> LDS path, not the matrix pipe, bounds
// it: the 320 x 160 tile of conv_wgrad8r.hip moves 61 KB per 64-pixel K tile for 6.5 MFLOP (107 FLOP/B) and measured 2.5 us per K tile -- 26 GB/s of
// LDS-DMA per CU, 6 TB/s over the chip -- with the MFMAs a third of that (probes in DESIGN.md).  The nine taps of a 3x3 kernel read the SAME input pixels
// shifted by one row / column, so this kernel stages, per K tile of 64 output pixels (= 64 / W whole image rows), ONE input patch -- the rows above and
// below and one pad column on either side, 32 channels = 64 bytes per pixel, at most 8.5 KB -- and one dy tile (64 pixels x 160 channels, 20 KB), and
// takes all nine taps from the patch through shifted transposed reads: 29 KB per K tile for 5.9 MFLOP (203 FLOP/B), no tap left over (the tap pairs of
// the 320 x 160 tile leave a ninth tap with half a tile), 29 LDS-DMA instructions per K tile instead of 64.
// Tile = 288 rows (9 taps x 32 input channels) x 160 output channels = 18 x 10 MFMA tiles (v_mfma_f32_16x16x32); TWELVE waves (768 threads, three per SIMD,
// <= 168 registers) as 6 x 2, a wave = 3 x 5 tiles = 60 accumulator registers; a wave's three row tiles are fixed (tap, 16-channel half) pairs.
// LDS: three stages of [patch 9 KiB | dy 20 KiB]; rows are XOR-swizzled by bit 3 of the row index (the two 16-lane groups of a half-wave's transposed read
// are 8 rows apart: the swap of the 32-byte halves / pairs puts them on different bank groups), on the DMA source side.
// Schedule: a K tile = two k-steps of 32 pixels (16 transposed reads, 15 MFMAs per wave each), ONE barrier per K tile (SCHED 2 below; the first form of
// this round ran waves 0-3 one barrier ahead of waves 4-11 with two barriers per k-step); K tile kt+2 is issued during K tile kt (its stage was last read one
// K tile earlier) and waited for one K tile later: one counted vmcnt per K tile, raw s_barrier.
// Work: item = (pixel split, tile) over a table of layers, as conv_wgrad8r.hip; slabs summed by the fixed-order kernels of conv_wgrad.hip.
#include "igemm_shared.h"
#include <string.h>

extern int g_rn_variant2;

namespace {

constexpr int W9_MAX = 12;
struct W9Rec {
  const void* x;
  const void* dy;
  float* out;              // the gradient itself (splits == 1) or the slab region [splits][K][9][C]
  long slab_stride;        // K * 9 * C floats (0 when splits == 1)
  int N, H, W, C, K;
  int M, nk;               // pixels; K tiles of 64 pixels
  int splits, per;
  int nct, nkt;            // 32-channel slices, 160-channel output tiles; tiles = nct * nkt
  int lw, rows;            // W = 1 << lw; image rows per K tile = 64 >> lw
  unsigned magic_h;        // floor(2^32 / (H / rows)): K tile -> (image, row block)
  int accumulate;
};
struct W9Batch {
  int n;
  int first[W9_MAX + 1];
  W9Rec r[W9_MAX];
};
static_assert(sizeof(W9Batch) <= 3072, "kernel-argument segment");

template <typename T> struct Tr9;
template <> struct Tr9<bf16_t> {
  __device__ static inline uint2 rd(const char* p) {
    typedef __attribute__((address_space(3))) bf16x4* lp;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p)));
  }
};
template <> struct Tr9<f16_t> {
  __device__ static inline uint2 rd(const char* p) {
    typedef __fp16 h4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) h4* lp;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p)));
  }
};
__device__ inline void bar9() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ inline void wait_vm9(int n) {                     // wave-uniform count -> immediate
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 17: wait_vmcnt<17>(); break;
    case 18: wait_vmcnt<18>(); break;
    default: wait_vmcnt<19>(); break;
  }
}

constexpr int YB = 20480;         // bytes of a stage's dy tile; the patch region in front of it: 9 KiB (stride 1: up to 136 pixel rows of 64 B), 20 KiB (stride 2: 306)

// PROBE (diagnostic instantiations, rn_set_variant2 bits 8-10; wrong results, timing only): 1 = no LDS-DMA in the K loop, 2 = no MFMA, 3 = no fragment reads
// SCHED 2 (shipped; rn_set_variant2 8192 selects 0, the first form, for A/B): ONE barrier per K tile and no wave groups.  A wave reads a k-step's fragments,
// waits for them and multiplies; the SIMD's other two waves fill its read latency (three waves per SIMD drift apart by themselves); the LDS-DMA pieces of
// K tile kt + 2 go out behind the first five MFMAs of the second k-step (an LDS-DMA piece costs ~60 cycles of issue among MFMAs, 100-185 in a read
// segment).  The barrier at the end of a K tile orders both hazards: every wave has waited for its own pieces of K tile kt + 1 in front of it (RAW), and
// every wave's reads of K tile kt are retired in front of it, so its stage may be re-staged -- as K tile kt + 3 -- from the next K tile on (WAR).
// Measured in one process (tools/conv_bench.py wgrad, batch 128): first form 88.1 / 80.6 / 80.2 us on the 160 / 320 / 640-channel layers; pieces moved into
// the MFMA segment, barriers kept 81.0 / 74.6 / 73.8; this form 75.0 / 70.5 / 72.5 (round-3 kernel: 88.3 / 79.4 / 84.5).  A second register set that holds
// both k-steps' fragments from the head of the K tile (162 registers) lost 3-5 % against it; removed.  TWO K tiles per barrier on a ring of four stages: 77.1 / 70.1 / 69.3 us
// against 79.3 / 72.8 / 71.3 alone (+3 %) -- but as a loop body of its own hipcc kept 156 registers for it, and the step went 6.06 -> 6.48 ms: at most 136 registers per wave
// is what lets the chain's BatchNorm-backward kernels run BESIDE this kernel (tests/test_abi.py holds the budget).
// SCHED 5 (shipped since; rn_set_variant2 32768 selects 2): the same idea through THIS form's k-step body -- K tile kt in stage kt % 4, both tiles of the next pair issued
// during the pair's first K tile, vmcnt(0) + barrier behind every second K tile: 132 registers, 70.2 / 69.7 us against 74.1 / 73.6 (+5 %), the step 6.157 -> 6.059 ms.
// STR 2 (round 4, second session): the family's stride-2 3x3 layers (padding 1, output map = input map / 2).  The patch of a K tile is then the 2 rows + 1 input rows
// its output rows read, ALL columns (306 pixel rows for 16- and 8-wide output maps: 20 DMA pieces, a stage = 20 + 20 KiB), and a k index reads patch row
// (2 r + 1 + dh)(W + 2) + 2 q + 1 + dw; everything else is the stride-1 kernel.
template <typename T, int PROBE = 0, int SCHED = 5, int STR = 1>
__global__ __launch_bounds__(768, 3) void wgrad9_kernel(const W9Batch b) {
  constexpr int ES = 2;
  constexpr int XB = STR == 2 ? 20480 : 9216, STG = XB + YB;          // (shadow the stride-1 constants of the namespace)
  constexpr int NS = STR == 2 ? 4 : 3;                                 // DMA pieces per wave and K tile, at most
  __shared__ uint4 smem[(SCHED == 5 ? 4 : 3) * STG / 16];
  const char* lds = reinterpret_cast<const char*>(&smem[0]);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 1, wn = wave & 1;                   // 6 x 2 waves of 48 rows x 80 columns
  const int l16 = lane & 15, lq = lane >> 4;
  const int tq = (lane >> 2) & 3, tp = lane & 3;

  // every layer of a launch has ONE geometry (launcher): the lane roles are computed once
  const W9Rec& R0 = b.r[0];
  const int W = R0.W, H = R0.H, lw = R0.lw, RW = R0.rows, W2 = W + 2;      // W, H: the INPUT map; lw = log2 of the OUTPUT map's width; RW = output rows per K tile
  const int WQ = W / STR;                                    // output map width
  const int NPR = (STR * RW + 3 - STR) * W2;                 // pixel rows of the patch: RW + 2 input rows (stride 1), 2 RW + 1 (stride 2)
  const int NXI = (NPR * 4 + 63) >> 6;                       // DMA pieces of the patch; the dy tile has 20
  const int NDI = NXI + 20;
  const unsigned cb = (unsigned)(R0.C * ES), kb_ = (unsigned)(R0.K * ES);

  // ---- DMA roles: this wave's pieces are list entries wave, wave + 12, wave + 24 of [patch pieces | dy pieces] ----
  const int nw = (NDI - wave + 11) / 12;                     // 2 or 3 (stride 2: 3 or 4)
  int dkind[NS];                                             // 0 patch, 1 dy, -1 none
  int dconst[NS], dir[NS];                                   // lane constant of the source offset; patch: the lane's patch row (vertical range check), -1 = never valid
  unsigned ddst[NS];                                         // LDS byte offset of the piece inside a stage
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int e = wave + 12 * s;
    dkind[s] = e < NXI ? 0 : (e < NDI ? 1 : -1);
    dconst[s] = 0; dir[s] = -1; ddst[s] = 0;
    if (dkind[s] == 0) {
      const int f = 64 * e + lane, prow = f >> 2, cpos = f & 3;
      const int ir = prow / W2, cc = prow - ir * W2;
      const int logical = cpos ^ (2 * ((prow >> 3) & 1));
      ddst[s] = (unsigned)(e * 1024);
      if (prow < NPR && cc >= 1 && cc <= W) { dir[s] = ir; dconst[s] = ((ir - 1) * W + cc - 1) * (int)cb + logical * 16; }
    } else if (dkind[s] == 1) {
      const int f = 64 * (e - NXI) + lane, pix = f / 20, cpos = f - 20 * pix;
      const int logical = cpos ^ (2 * ((pix >> 3) & 1));
      ddst[s] = (unsigned)(XB + (e - NXI) * 1024);
      dir[s] = 0; dconst[s] = pix * (int)kb_ + logical * 16;
    }
  }

  // ---- fragment addresses (bytes inside a stage) ----
  // dy: k-index pixel row r = 32 ks + 8 lq + tq + 4 u, swizzle bit (r >> 3) & 1 = lq & 1; pair P = 5 wn + j stored at P ^ bit
  const int ybit = lq & 1;
  const int ylane = (8 * lq + tq) * 320 + 16 * (tp >> 1) + 8 * (tp & 1);
  const int yb_even = XB + ylane + 32 * ybit, yb_odd = XB + ylane - 32 * ybit;       // + 32 P (+ 1280 u + 10240 ks)
  // patch: the wave's row tile i is (tap, half) = ((3 wa + i) >> 1, (3 wa + i) & 1); the pixel of k index idx is patch row (idx / W + 1 + dh) (W + 2) + idx % W + 1 + dw
  int xa[3][2][2];                                           // [tile][ks][u]
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t16 = 3 * wa + i, tap = t16 >> 1, hf = t16 & 1;
    const int dh = tap / 3 - 1, dw = tap % 3 - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = 32 * ks + 8 * lq + tq + 4 * u;
        const int prow = (STR * (idx >> lw) + 1 + dh) * W2 + STR * (idx & (WQ - 1)) + 1 + dw;
        xa[i][ks][u] = prow * 64 + 32 * (hf ^ ((prow >> 3) & 1)) + 16 * (tp >> 1) + 8 * (tp & 1);
      }
  }

  // ---- the current item's scalars ----
  int rec = 0, kend = 0;
  int inb = 1;                                               // row blocks (K tiles) per image
  unsigned imagic = 0, xo = 0, yo = 0;                       // channel-slice / output-tile byte offsets
  int xw0 = 0, xw1 = 0, xw2 = 0, yw0 = 0, yw1 = 0, yw2 = 0;  // the item's buffer descriptors, as wave-uniform words
  auto descr = [&](int w0, int w1, int w2) {
    v4i32 d;
    d[0] = __builtin_amdgcn_readfirstlane(w0); d[1] = __builtin_amdgcn_readfirstlane(w1); d[2] = __builtin_amdgcn_readfirstlane(w2); d[3] = 0x00020000;
    return d;
  };
  // this wave's piece s of K tile g into the stage at stage_lds
  auto issue = [&](int s, int g, unsigned stage_lds) {
    if (dkind[s] < 0) return;                                // wave-uniform
    const bool live = g < kend;                              // wave-uniform: K tiles beyond the item read zeros
    if (dkind[s] == 0) {
      int n = (int)__umulhi((unsigned)g, imagic);            // K tile -> image n, first row p0 = (g - n * inb) * rows
      if (g - n * inb >= inb) ++n;
      const int p0 = STR * (g - n * inb) * RW;              // first INPUT row of the K tile's output rows
      const unsigned sbase = (unsigned)((n * H + p0) * W) * cb + xo;
      const bool ok = live && dir[s] >= 0 && (unsigned)(p0 + dir[s] - 1) < (unsigned)H;
      dma16(descr(xw0, xw1, xw2), ok ? sbase + (unsigned)dconst[s] : OOB, (unsigned)__builtin_amdgcn_readfirstlane((int)(stage_lds + ddst[s])));
    } else {
      const unsigned sbase = (unsigned)(64 * g) * kb_ + yo;
      dma16(descr(yw0, yw1, yw2), live ? sbase + (unsigned)dconst[s] : OOB, (unsigned)__builtin_amdgcn_readfirstlane((int)(stage_lds + ddst[s])));
    }
  };
  auto issue_tile = [&](int g, unsigned stage_lds) {
    const unsigned keep = m0_save();
#pragma unroll
    for (int s = 0; s < NS; ++s) issue(s, g, stage_lds);
    m0_restore(keep);
  };

  f32x4 acc[3][5];
  uint4 af[3], bfr[5];
  if constexpr (PROBE == 3) {
#pragma unroll
    for (int i = 0; i < 3; ++i) af[i] = make_uint4(lane, 1, 2, 3);
#pragma unroll
    for (int j = 0; j < 5; ++j) bfr[j] = make_uint4(3, 2, 1, lane);
  }
  auto rd2 = [&](int lo_off, int hi_off) {
    const uint2 lo = Tr9<T>::rd(lds + lo_off), hi = Tr9<T>::rd(lds + hi_off);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  // one phase: k-step ks of the K tile in the stage at byte offset sx; second phases also issue K tile gn into the stage at dst_lds and wait for the K tile after this one
  auto phase = [&](auto kstag, int sx, int gn, unsigned dst_lds, bool pair_end = true) {
    constexpr int ks = decltype(kstag)::value;
    if constexpr (PROBE != 3) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int P = 5 * wn + j;                            // (parity is wave-uniform: two address registers)
        const int base = (P & 1) ? yb_odd : yb_even;
        bfr[j] = rd2(sx + base + 32 * P + 10240 * ks, sx + base + 32 * P + 10240 * ks + 1280);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) af[i] = rd2(sx + xa[i][ks][0], sx + xa[i][ks][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ks == 1) {
      if constexpr (SCHED == 0) {
        if constexpr (PROBE != 1) issue_tile(gn, dst_lds);
        wait_vm9(PROBE == 1 ? 0 : nw);                       // everything older than this K tile's own pieces: K tile kt + 1 has landed
      }
    }
    if constexpr (SCHED != 2 && SCHED != 5) bar9();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#define W9_MF(i, j)                                                                                                   \
  do {                                                                                                                \
    if constexpr (PROBE != 2) Mfma16<T>::run(af[i], bfr[j], acc[i][j]);                                               \
    else asm volatile("" : "+v"(acc[i][j]) : "v"(af[i].x), "v"(af[i].w), "v"(bfr[j].x), "v"(bfr[j].w));               \
  } while (0)
#pragma unroll
    for (int j = 0; j < 5; ++j) W9_MF(0, j);
    if constexpr (SCHED == 5) {                              // (both tiles of the NEXT pair go out during the pair's first K tile: gn < 0 = nothing to issue)
      __builtin_amdgcn_sched_barrier(0);
      if (gn >= 0) issue_tile(gn, dst_lds);
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (ks == 1 && SCHED != 0 && PROBE != 1) {
      __builtin_amdgcn_sched_barrier(0);
      issue_tile(gn, dst_lds);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 1; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) W9_MF(i, j);
#undef W9_MF
    __builtin_amdgcn_s_setprio(0);
    if constexpr (SCHED != 2 && SCHED != 5) bar9();
    else if constexpr (ks == 1) {
      if constexpr (SCHED == 5) {                            // a barrier every SECOND K tile (ring of four stages; the next pair has landed behind vmcnt(0))
        if (pair_end) { wait_vm9(0); bar9(); }
      } else {
        wait_vm9(nw);                                        // K tile kt + 1 has landed; this K tile's own pieces (K tile kt + 2) stay in flight
        bar9();
      }
    }
  };
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;

  const int G = gridDim.x, nitems = b.first[b.n];
  int prec = -1, ptile = 0, psplit = 0;
  for (int it = 0;; ++it) {
    const int vb = it * G + (int)blockIdx.x;
    const bool more = vb < nitems;                           // wave-uniform
    int tile = 0, split = 0, kb = 0;
    if (more) {
      const int xcd = vb & 7, q = nitems >> 3, r = nitems & 7;
      const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
      rec = 0;
      while (rec + 1 < b.n && item >= b.first[rec + 1]) ++rec;
      const W9Rec& R = b.r[rec];
      const int local = item - b.first[rec];
      const int ntiles = R.nct * R.nkt;
      split = local / ntiles; tile = local - split * ntiles;
      kb = split * R.per; kend = min(R.nk, kb + R.per);
      const int ct = tile % R.nct, kt_ = tile / R.nct;
      xo = (unsigned)(ct * 32 * ES); yo = (unsigned)(kt_ * 160 * ES);
      inb = (R.H / STR) / R.rows; imagic = R.magic_h;
      { const v4i32 d = make_desc(R.x, (size_t)R.N * R.H * R.W * R.C * ES); xw0 = d[0]; xw1 = d[1]; xw2 = d[2]; }
      { const v4i32 d = make_desc(R.dy, (size_t)R.M * R.K * ES); yw0 = d[0]; yw1 = d[1]; yw2 = d[2]; }
      // prologue: K tiles kb and kb + 1 into stages 0 and 1 (every wave has left the previous K loop)
      issue_tile(kb, lds0); issue_tile(kb + 1, lds0 + STG);
    }
    bool stores_behind = false;
    if (prec >= 0) {        // acc[i][j][r]: tap / half of row tile 3 wa + i, input channel 32 ct + 16 half + 4 lq + r, output channel 160 kt + 80 wn + 16 j + l16
      const W9Rec& R = b.r[prec];
      const int ct = ptile % R.nct, kt_ = ptile / R.nct;
      float* out = R.out + (size_t)psplit * R.slab_stride;
      const int k0 = kt_ * 160 + 80 * wn + l16;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int t16 = 3 * wa + i, tap = t16 >> 1, hf = t16 & 1;
        const int c0 = ct * 32 + 16 * hf + 4 * lq;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          float* p = out + ((size_t)(k0 + 16 * j) * 9 + tap) * R.C + c0;
          float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          if (R.accumulate) { const float4 o = *reinterpret_cast<const float4*>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *reinterpret_cast<float4*>(p) = v;
        }
      }
      stores_behind = !R.accumulate;
    }
    if (!more) break;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // K tile kb has landed; kb + 1's pieces (and the previous item's 15 stores, which are younger) may fly
    if constexpr (SCHED == 5) { if (stores_behind) wait_vmcnt<15>(); else wait_vmcnt<0>(); } else wait_vm9(stores_behind ? nw + 15 : nw);
    bar9();
    if (SCHED == 0 && wave >= 4) bar9();                     // waves 4-11 run one barrier behind waves 0-3

    const int nseg_k = kend - kb;
    if constexpr (SCHED == 5) {
      // the prologue put K tiles kb, kb + 1 into stages 0, 1 and waited for both; K tile kt lives in stage kt % 4, K tile kt + 2 is issued during kt
      for (int kt = 0; kt < nseg_k; ++kt) {
        const int sc = (kt & 3) * STG;
        const bool even = (kt & 1) == 0;
        phase(K0{}, sc, even ? kb + kt + 2 : -1, lds0 + (unsigned)(((kt + 2) & 3) * STG));
        phase(K1{}, sc, even ? kb + kt + 3 : -1, lds0 + (unsigned)(((kt + 3) & 3) * STG), !even || kt + 1 == nseg_k);
      }
    } else {
      int s_cur = 0, s_nxt = STG, s_free = 2 * STG;          // stage of K tile kt, of kt + 1, and the one kt + 2 goes into (kt - 1's)
      for (int kt = 0; kt < nseg_k; ++kt) {
        phase(K0{}, s_cur, 0, 0u);
        phase(K1{}, s_cur, kb + kt + 2, lds0 + (unsigned)s_free);
        const int t = s_cur; s_cur = s_nxt; s_nxt = s_free; s_free = t;
      }
    }
    if (SCHED == 0 && wave < 4) bar9();
    prec = rec; ptile = tile; psplit = split;
  }
}

}  // namespace

// pixel splits of a launch of ntiles tiles (all layers of the launch together) over nk K tiles, and its modelled time in microseconds (calibrated on the
// one-barrier schedule, batch 128: ~1.6 us per K tile of a full chip, ~6 us of ramp per round; slabs written and re-read at ~4 TB/s)
static int w9_pick_splits(long ntiles, long nk, double n_floats, long cap = 128, double* cost_us = nullptr) {
  int best = 1;
  double best_cost = 1e30;
  long smax = nk / 4 < 128 ? (nk / 4 < 1 ? 1 : nk / 4) : 128;
  if (smax > cap) smax = cap < 1 ? 1 : cap;
  for (long S = 1; S <= smax; ++S) {
    const long per = (nk + S - 1) / S;
    if (per * (S - 1) >= nk) continue;                                       // an empty last split
    const long rounds = (ntiles * S + 255) / 256;
    const double cost = rounds * (per * 1.6 + 6.0) + (S > 1 ? S * n_floats * 8.0 / 4.0e6 : 0.0) + (S > 1 ? 6.0 : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = (int)S; }
  }
  if (cost_us) *cost_us = best_cost;
  return best;
}

static bool w9_geom_ok(const rn_conv_geom* g, int dtype) {
  if (dtype != RN_BF16 && dtype != RN_F16) return false;
  if (g_rn_variant2 & 16384) return false;                                     // A/B: never
  if (g->R != 3 || g->S != 3 || g->pad != 1 || (g->stride != 1 && g->stride != 2)) return false;
  if (g->P * g->stride != g->H || g->Q * g->stride != g->W) return false;     // (stride 2: even maps, output = input / 2)
  if (g->C % 32 || g->C < 32 || g->K % 160) return false;
  const int W = g->Q;                                                          // the OUTPUT map: a K tile = 64 / Q whole output rows
  if (W < 8 || W > (g->stride == 2 ? 16 : 32) || (W & (W - 1))) return false;  // the patch has to fit its LDS region (stride 1: 64-wide maps would not; stride 2: 32-wide outputs)
  if (g->stride == 2 && (g_rn_variant2 & 524288)) return false;                // A/B: the stride-2 form off (the 320 x 160 kernel takes those layers)
  if (g->P % (64 / W)) return false;
  if ((double)g->N * g->H * g->W * g->C * 2 >= 4.0e9 || (double)g->N * g->P * g->Q * g->K * 2 >= 4.0e9) return false;      // 32-bit DMA offsets
  return true;
}

// how many layers of this geometry the plan executor should collect for ONE launch (rn_conv_wgrad8r_batch): the count in 1..max_n with the lowest modelled
// time per layer.  The tiles of n layers share one grid of 256 persistent workgroups, so the count decides how well the items fill whole rounds: WRN-28-10's
// 640-channel layers (80 tiles each) fill 0.94 of one round in threes (no slabs at all), 0.62 of two rounds in fours, 0.73 of three rounds in sevens.
int rn_wgrad9_best_batch(const rn_conv_geom* g, int dtype, int max_n) {
  if (!w9_geom_ok(g, dtype) || max_n < 1) return 1;
  const long M = (long)g->N * g->P * g->Q, nk = M / 64;
  const long ntiles = (long)(g->C / 32) * (g->K / 160);
  const double nel = (double)g->K * 9 * g->C;
  int best = 1;
  double best_per_layer = 1e30;
  for (int n = 1; n <= max_n && n <= W9_MAX; ++n) {
    double c = 0;
    w9_pick_splits(ntiles * n, nk, nel * n, 128, &c);
    if (c / n < best_per_layer * 0.98) { best_per_layer = c / n; best = n; }      // a larger batch has to be worth 2 %: it delays the gradients
  }
  return best;
}

// 0: rn_conv_wgrad does not take this kernel for the geometry; S >= 1: it does, with S pixel splits
int rn_wgrad9_splits(const rn_conv_geom* g, int dtype) {
  if (!w9_geom_ok(g, dtype)) return 0;
  const long M = (long)g->N * g->P * g->Q, nk = M / 64;
  const long ntiles = (long)(g->C / 32) * (g->K / 160);
  if (ntiles * nk < 8L * 256 && !(g_rn_variant2 & 8)) return 0;               // too small to fill the chip (rn_set_variant2 8: any size, tests)
  return w9_pick_splits(ntiles, nk, (double)g->K * 9 * g->C);
}

static void w9_fill(W9Rec& r, const void* x, const void* dy, float* out, int splits, const rn_conv_geom* g, int accumulate) {
  r.x = x; r.dy = dy; r.out = out;
  r.N = g->N; r.H = g->H; r.W = g->W; r.C = g->C; r.K = g->K;
  r.M = g->N * g->P * g->Q; r.nk = r.M / 64;
  r.nct = g->C / 32; r.nkt = g->K / 160;
  r.lw = 0;
  while ((1 << r.lw) < g->Q) ++r.lw;
  r.rows = 64 >> r.lw;
  const unsigned nb = (unsigned)(g->P / r.rows);
  r.magic_h = nb <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / nb);
  r.splits = splits; r.per = (r.nk + splits - 1) / splits;
  r.slab_stride = splits > 1 ? (long)g->K * 9 * g->C : 0;
  r.accumulate = (splits == 1 && accumulate) ? 1 : 0;
}

template <typename T> static void w9_launch(const W9Batch& b, int grid, hipStream_t s, int stride) {
  if (stride == 2) { hipLaunchKernelGGL((wgrad9_kernel<T, 0, 2, 2>), dim3(grid), dim3(768), 0, s, b); return; }      // (three stages of 40 KiB: a fourth would fill the CU's LDS)
  if constexpr (std::is_same<T, f16_t>::value) {
    const int probe = (g_rn_variant2 >> 8) & 7;
    if (probe == 1) { hipLaunchKernelGGL((wgrad9_kernel<T, 1, 5>), dim3(grid), dim3(768), 0, s, b); return; }
    if (probe == 2) { hipLaunchKernelGGL((wgrad9_kernel<T, 2, 5>), dim3(grid), dim3(768), 0, s, b); return; }
    if (probe == 3) { hipLaunchKernelGGL((wgrad9_kernel<T, 3, 5>), dim3(grid), dim3(768), 0, s, b); return; }
  }
  if (g_rn_variant2 & 8192) { hipLaunchKernelGGL((wgrad9_kernel<T, 0, 0>), dim3(grid), dim3(768), 0, s, b); return; }
  if (g_rn_variant2 & 32768) { hipLaunchKernelGGL((wgrad9_kernel<T, 0, 2>), dim3(grid), dim3(768), 0, s, b); return; }
  hipLaunchKernelGGL((wgrad9_kernel<T>), dim3(grid), dim3(768), 0, s, b);
}

// out: the gradient itself (splits == 1) or the slab region [splits][K][9][C]
int rn_launch_wgrad9(const void* x, const void* dy, float* out, int splits, int dtype, const rn_conv_geom* g, int max_grid, hipStream_t s) {
  W9Batch b{};
  b.n = 1;
  w9_fill(b.r[0], x, dy, out, splits, g, 0);
  b.first[0] = 0; b.first[1] = b.r[0].nct * b.r[0].nkt * splits;
  rn_note_kernel("wgrad9<288x160>");
  if (rn_dry_run()) return 0;
  const int cap = max_grid > 0 && max_grid < 256 ? max_grid : 256;
  const int grid = b.first[1] < cap ? b.first[1] : cap;
  if (dtype == RN_BF16) w9_launch<bf16_t>(b, grid, s, g->stride); else w9_launch<f16_t>(b, grid, s, g->stride);
  RN_CHECK_LAUNCH("wgrad9");
  return 0;
}

int rn_wgrad_reduce_slabs(const float* ws, float* dw_krsc, long n, int splits, int accum, int eight_phase, hipStream_t s);      // conv_wgrad.hip

// n layers of ONE geometry as one launch (see rn_conv_wgrad8r_batch, which hands the geometries this kernel takes over to it)
// s_reduce / ev (both or neither): the slab sums go to s_reduce behind event ev, which is recorded on s behind the kernel
int rn_wgrad9_batch(const rn_wgrad8r_desc* descs, int n, int dtype, int max_grid, hipStream_t s, hipStream_t s_reduce, hipEvent_t ev) {
  static_assert(RN_WGRAD8R_BATCH_MAX == W9_MAX, "header and kernel disagree");
  const rn_conv_geom& g0 = descs[0].g;
  const long M = (long)g0.N * g0.P * g0.Q, nk = M / 64;
  const long ntiles = (long)(g0.C / 32) * (g0.K / 160);
  const size_t nel = (size_t)g0.K * 9 * g0.C;
  long ws_cap = 128;
  // records may share ONE workspace (the plan executor's side workspace): record i then takes the share-th region of it, share = the records before it
  // that name the same pointer, and the workspace has to hold the slabs of all of them
  int share[16], sharers[16];
  for (int i = 0; i < n; ++i) {
    share[i] = 0; sharers[i] = 0;
    for (int j = 0; j < n; ++j)
      if (descs[j].ws == descs[i].ws) { ++sharers[i]; if (j < i) ++share[i]; }
  }
  for (int i = 0; i < n; ++i) {
    const long c = descs[i].ws ? (long)(descs[i].ws_bytes / (nel * sizeof(float) * (size_t)sharers[i])) : 0;
    if (c < ws_cap) ws_cap = c;
  }
  const int splits = w9_pick_splits(ntiles * n, nk, (double)nel * n, ws_cap);
  W9Batch b{};
  b.n = n;
  int items = 0;
  for (int i = 0; i < n; ++i) {
    const rn_wgrad8r_desc& d = descs[i];
    RN_CHECK_ARG(d.x && d.dy && d.dw, "rn_conv_wgrad8r_batch: record %d: null pointer", i);
    RN_CHECK_ARG(memcmp(&d.g, &g0, sizeof(g0)) == 0, "rn_conv_wgrad8r_batch: record %d has another geometry", i);
    const bool direct = splits == 1 && !(d.flags & RN_F_ACCUM);
    RN_CHECK_ARG(direct || (d.ws && d.ws_bytes >= (size_t)sharers[i] * splits * nel * sizeof(float)), "rn_conv_wgrad8r_batch: record %d: workspace too small (%zu < %zu)", i, d.ws_bytes,
                 (size_t)sharers[i] * splits * nel * sizeof(float));
    w9_fill(b.r[i], d.x, d.dy, direct ? d.dw : reinterpret_cast<float*>(d.ws) + (size_t)share[i] * splits * nel, splits, &g0, 0);
    b.first[i] = items;
    items += b.r[i].nct * b.r[i].nkt * splits;
    rn_note_kernel("wgrad9<288x160>");
    if (!direct) rn_note_kernel("wgrad_reduce");
  }
  b.first[n] = items;
  if (rn_dry_run()) return 0;
  const int cap = max_grid > 0 && max_grid < 256 ? max_grid : 256;
  const int grid = items < cap ? items : cap;
  if (dtype == RN_BF16) w9_launch<bf16_t>(b, grid, s, g0.stride); else w9_launch<f16_t>(b, grid, s, g0.stride);
  RN_CHECK_LAUNCH("wgrad9 batch");
  if (s_reduce && ev) {
    if (hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(s_reduce, ev, 0) != hipSuccess) { rn_set_error("rn_conv_wgrad8r_batch: event record / wait failed"); return 2; }
    s = s_reduce;
  }
  for (int i = 0; i < n; ++i) {
    const rn_wgrad8r_desc& d = descs[i];
    if (splits == 1 && !(d.flags & RN_F_ACCUM)) continue;             // written in place
    if (int e = rn_wgrad_reduce_slabs(reinterpret_cast<const float*>(d.ws) + (size_t)share[i] * splits * nel, d.dw, (long)nel, splits, (d.flags & RN_F_ACCUM) ? 1 : 0, 1, s)) return e;
  }
  return 0;
}
