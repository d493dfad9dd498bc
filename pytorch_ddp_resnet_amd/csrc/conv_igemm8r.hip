// 3x3 stride-1 convolutions (forward and data gradient of residual_block.py:34-47, :67-99) whose channel counts are multiples of 32 / 160 --
// the WRN-28-10 family, C = K = 160 n (models_dir/wrn-28-10-dropout_cifar10/config.yaml:15) -- on the deep-pipelined schedule of conv_igemm8.hip,
// with the input staged as ROW PATCHES: 256 x 160 block tiles, 8 waves of 64 x 80 (4 x 5 v_mfma_f32_16x16x32 tiles = 80 accumulator registers),
// ONE persistent workgroup per CU.
//
// Why not the im2col walk of conv_igemm8.hip.  (1) 160 n channels are 2.5 n K tiles of 64: a K tile would straddle taps.  (2) With 80 columns per
// wave the kernel issues 1.3 LDS-DMA instructions per MFMA and wave where the 256 x 256 tile issues 1.0 -- and the DMA issue rate is what bounds
// these loops.  Both are answered by staging the input per KERNEL ROW instead of per tap:
//   * the reduction runs over ITEMS (i, hc) = (kernel row i, 32-channel half chunk hc), taken two at a time: a GROUP g holds items 2g and 2g+1, which
//     may belong to different kernel rows (C = 160: 15 items, 8 groups, the last half empty; C = 320: 30 items, 15 groups);
//   * per group the workgroup stages ONE A patch: for each of the tile's 256 / W image rows the W pixels of source row h + dh_i plus one pad
//     column on either side, 128 bytes per pixel = [64 B of item 2g | 64 B of item 2g+1]  ((256 / W)(W + 2) LDS rows: 272 for W = 32);
//   * the THREE column taps j of the group read that one patch through a row offset dw_j (the pad columns are the horizontal zero padding,
//     out-of-range rows are out-of-range DMA offsets that the hardware zero-fills): three K tiles of 2 x 32 channels per 34-40 KiB of input
//     DMA instead of 3 x 32 KiB -- 0.8 DMA instructions per MFMA and wave;
//   * the weights of K tile (g, j) are [64 B of (tap (i_2g, j), chunk hc_2g) | 64 B of (tap (i_2g+1, j), chunk hc_2g+1)] per output channel, read from
//     the ordinary [K][tap][C] copy: a lane's 16-byte piece belongs to one half for the kernel's lifetime, so the two halves cost one select per
//     lane and K tile, no re-packed weights.
// Schedule: a K tile = two phases of 20 MFMAs per wave (wave rows 0-31 x 80 columns with the B fragments of the K tile and the upper A half, then rows
// 32-63 with the same B fragments); per phase [fragment reads + LDS-DMA issue | barrier | MFMAs | barrier], the two wave groups (waves 0-3 / 4-7 = the
// two waves of every SIMD) one barrier apart so that one group's MFMAs cover the other's reads and DMA issue.  A patches double-buffered (the patch of
// group g+1 is issued piece by piece in phases 1-5 of group g), the weight tiles in a ring of three (tile kt+2 issued during tile kt), one counted
// s_waitcnt vmcnt per K tile, raw s_barrier (cdna_hip_programming.md section 5).
// Hazards (M_p / C_p = memory / MFMA segment of phase p):
//   RAW  a buffer is waited for (own pieces: counted vmcnt) in the M segment of the LAST phase before the one that reads it, in front of that
//        segment's closing barrier; it is read one phase later.
//   WAR  the weight stage of tile kt-1 is re-staged in phase 1 of tile kt: two phases after its reads (retired by the lgkmcnt(0) that opens the
//        reading phase's C segment).  The A buffer of group g-1 is re-staged in phase 1 of group g, ONE phase after its last reads (second phase of
//        the group's last K tile), which is why every second phase retires its reads (lgkmcnt(0)) in FRONT of its M segment's closing barrier.
// Split reduction (grids of <= 128 tiles: WRN-28-10's 640-channel stage at batch 128): every tile is TWO work items, the halves of its group range, on two
// CUs; each half writes its fp32 accumulators write-through into its slot of the stream-K workspace (rn_set_conv_workspace) and draws a ticket; the half
// that draws the second ticket adds the partner's part to its registers (two parts: the sum does not depend on who arrived last) and runs the epilogue.
// Nobody waits for anybody (conv_igemm8.hip's stream-K hand-off).
// Epilogue: the register epilogue of conv_igemm8.hip (transposed products; a 4 x 4 transpose over the wave's lane rows gives a lane 16 consecutive
// channels of one pixel) for 64 of a wave's 80 columns; the fifth column tile is transposed over the wave's four PIXEL tiles instead, which gives a
// lane 16 consecutive channels (columns 64..79) of pixel 16 * (lane row) + lane column.  Same fused operand sets, per template parameter.
#include "igemm_shared.h"

int g_rn_variant2 = 0;    // round-4 switches: 1 = never take the row-patch kernel, 2 = take it on any grid size (tests, conv_bench)
extern "C" void rn_set_variant2(int v) { g_rn_variant2 = v; }

namespace {

template <int N> __device__ inline void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void raw_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// wave-uniform count -> immediate
__device__ inline void wait_vm(int n) {
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    default: wait_vmcnt<5>(); break;
  }
}

__device__ inline void swap32(float& d, float& s) {       // lane rows 2,3 of d <-> rows 0,1 of s
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, d), __builtin_bit_cast(unsigned, s), false, false);
  d = __builtin_bit_cast(float, (unsigned)r[0]); s = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ inline void swap16(float& d, float& s) {       // lane rows 1,3 of d <-> rows 0,2 of s
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, d), __builtin_bit_cast(unsigned, s), false, false);
  d = __builtin_bit_cast(float, (unsigned)r[0]); s = __builtin_bit_cast(float, (unsigned)r[1]);
}
// in: x_c on lane row q = M[q][c]; out: x_s on lane row q = M[s][q]
__device__ inline void xpose4(float& x0, float& x1, float& x2, float& x3) {
  swap32(x0, x2); swap32(x1, x3);
  swap16(x0, x1); swap16(x2, x3);
}
__device__ inline float row_sum16(float x) {              // sum over the 16 lanes of a row, in every lane
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
  return x;
}
__device__ inline float rows_sum4(float x) {              // x uniform within a lane row: sum over the four rows, in every lane
  float y = x;
  swap32(x, y);                                           // x = [x0 x1 x0 x1], y = [x2 x3 x2 x3]
  x += y;
  y = x;
  swap16(x, y);                                           // x = [z0 z0 z2 z2], y = [z1 z1 z3 z3]
  return x + y;
}

enum { R8_PLAIN = 0, R8_RES = 1, R8_ACC = 2, R8_BNB = 4 };

// corner: 512 floats of LDS per wave.  [0, 80) mean, [80, 160) 1 / std of the wave's 80 channels (BatchNorm-backward forms); [320, 480) the sums a
// wave of the upper 64 rows of a 128-row statistics group hands to its partner
constexpr int CORNER = 512, CX = 320;

// The 16 pixels of an MFMA tile are INTERLEAVED over the lanes -- lanes l16 in {0-3, 12-15} hold the even pixels, {4-11} the odd ones -- so that the two
// 8-lane halves of a ds_read_b128 lane group ({0-3, 12-15 | 20-27}: logical chunks lq and lq + 1) read patch rows of DIFFERENT parity = different halves
// of the 256-byte bank window, whatever the alignment of the 16-row run (a kernel column shifts it by one row).  With pixel = l16 the two halves collide
// whenever the run does not start on a multiple of 4 rows (2 of 3 kernel columns): 21-30 % of the kernel's LDS cycles were conflicts
// (profiles/r04b_conv_bench_model_operands_pmc.txt).  PERM 0 (rn_set_variant2 4096, plain operand set only): pixel = l16, for A/B.
__device__ inline int pix16(int l16, bool perm) {
  if (!perm) return l16;
  return (l16 >= 4 && l16 < 12) ? 2 * (l16 - 4) + 1 : (l16 < 4 ? 2 * l16 : 2 * (l16 - 8));
}

template <typename T, int MODE, int PERM>
__device__ inline void epilogue8r(const IgemmArgs& a, f32x4 (&acc)[4][5], int mw, int kw, int lane, float* corner, bool upper, int partner_floats) {
  constexpr int CE = 8;
  constexpr bool C_RES = (MODE & R8_RES) != 0, C_ACC = (MODE & R8_ACC) != 0, C_BNB = (MODE & R8_BNB) != 0;
  const int l16 = lane & 15, lq = lane >> 4;
  T* __restrict__ dst = reinterpret_cast<T*>(a.dst);
  const bool want_stats = a.stats != nullptr;
  float sA0[16], sA1[16], sB0[16], sB1[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) sA0[e] = sA1[e] = sB0[e] = sB1[e] = 0.f;
  if (C_BNB) {
    if (lane < 20) *reinterpret_cast<float4*>(corner + 4 * lane) = *reinterpret_cast<const float4*>(a.bn_coef + 2 * a.Kd + kw + 4 * lane);
    else if (lane < 40) *reinterpret_cast<float4*>(corner + 80 + 4 * (lane - 20)) = *reinterpret_cast<const float4*>(a.bn_coef + 3 * a.Kd + kw + 4 * (lane - 20));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // same wave writes and reads: program order + the wait
  }
  // unit u < 4: pixel tile u, channels kw + 16 lq .. + 15; unit 4: pixel 16 lq + l16, channels kw + 64 .. + 15
  struct Ops { bool ok; size_t off; Chunk<T> cr[2], co[2], cx[2], cm[2]; };
  auto fetch = [&](int u, Ops& o) {
    const int m = mw + (u < 4 ? 16 * u : 16 * lq) + pix16(l16, PERM != 0);
    const int kc = kw + (u < 4 ? 16 * lq : 64);
    o.ok = m < a.M;
    o.off = (size_t)m * a.Kd + kc;
    if (!o.ok) return;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (C_RES) o.cr[c] = load_chunk<T>(reinterpret_cast<const T*>(a.res.ptr) + o.off + c * CE);
      if (C_ACC) o.co[c] = load_chunk<T>(dst + o.off + c * CE);
      if (C_BNB) {
        o.cx[c] = load_chunk<T>(reinterpret_cast<const T*>(a.bn_x) + o.off + c * CE);
        o.cm[c] = load_chunk<T>(reinterpret_cast<const T*>(a.bn_mask) + o.off + c * CE);
      }
    }
  };
  auto process = [&](int u, const Ops& o, float (&s0)[16], float (&s1)[16]) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x0, x1, x2, x3;
      if (u < 4) { x0 = acc[u][0][r]; x1 = acc[u][1][r]; x2 = acc[u][2][r]; x3 = acc[u][3][r]; }
      else { x0 = acc[0][4][r]; x1 = acc[1][4][r]; x2 = acc[2][4][r]; x3 = acc[3][4][r]; }
      xpose4(x0, x1, x2, x3);
      v[r] = x0; v[4 + r] = x1; v[8 + r] = x2; v[12 + r] = x3;
    }
    if (!o.ok) return;
    if (C_RES) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += Elem<T>::to_f(o.cr[e / CE].e[e % CE]);
    }
    if (C_ACC) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += Elem<T>::to_f(o.co[e / CE].e[e % CE]);
    }
    Chunk<T> st[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int e = 0; e < CE; ++e) st[c].e[e] = Elem<T>::from_f(v[c * CE + e]);
      store_chunk<T>(dst + o.off + c * CE, st[c]);
    }
    if (want_stats) {
      if (!C_BNB) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float vs = Elem<T>::to_f(st[e / CE].e[e % CE]); s0[e] += vs; s1[e] += vs * vs; }
      } else {
        const float* lm = corner + (u < 4 ? 16 * lq : 64);
#pragma unroll
        for (int e4 = 0; e4 < 16; e4 += 4) {
          const float4 mu = *reinterpret_cast<const float4*>(lm + e4);
          const float mean[4] = {mu.x, mu.y, mu.z, mu.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int e = e4 + q;
            float g = Elem<T>::to_f(st[e / CE].e[e % CE]) * a.gscale;
            const float xv = Elem<T>::to_f(o.cx[e / CE].e[e % CE]);
            if (!(Elem<T>::to_f(o.cm[e / CE].e[e % CE]) > 0.f)) g = 0.f;
            s0[e] += g; s1[e] += g * (xv - mean[q]);
          }
        }
      }
    }
  };
  Ops oa, ob;
  fetch(4, oa);
  fetch(0, ob);
  process(4, oa, sB0, sB1);
  fetch(1, oa);
  process(0, ob, sA0, sA1);
  fetch(2, ob);
  process(1, oa, sA0, sA1);
  fetch(3, oa);
  process(2, ob, sA0, sA1);
  process(3, oa, sA0, sA1);

  if (want_stats) {                              // one partial row per RN_CONV_STATS_ROWS = 128 output rows = two waves (wm even / odd)
    static_assert(RN_CONV_STATS_ROWS == 128, "a wave's 64 rows are half a partial-sum row");
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      sA0[e] = row_sum16(sA0[e]); sA1[e] = row_sum16(sA1[e]);
      sB0[e] = rows_sum4(row_sum16(sB0[e])); sB1[e] = rows_sum4(row_sum16(sB1[e]));
    }
    float* xc = corner + CX;                     // [0, 64) sum 0 of channels 16 lq + e, [64, 80) of channels 64 + e, [80, 160) the same for sum 1
    if (upper && l16 == 0) {
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        *reinterpret_cast<float4*>(xc + 16 * lq + e) = make_float4(sA0[e], sA0[e + 1], sA0[e + 2], sA0[e + 3]);
        *reinterpret_cast<float4*>(xc + 80 + 16 * lq + e) = make_float4(sA1[e], sA1[e + 1], sA1[e + 2], sA1[e + 3]);
      }
      if (lq == 0) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          *reinterpret_cast<float4*>(xc + 64 + e) = make_float4(sB0[e], sB0[e + 1], sB0[e + 2], sB0[e + 3]);
          *reinterpret_cast<float4*>(xc + 144 + e) = make_float4(sB1[e], sB1[e + 1], sB1[e + 2], sB1[e + 3]);
        }
      }
    }
    lds_barrier();                               // every wave of the workgroup is here (uniform control flow); waits for LDS traffic only
    if (!upper) {
      const float* other = xc + partner_floats;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        sA0[e] += other[16 * lq + e]; sA1[e] += other[80 + 16 * lq + e];
        sB0[e] += other[64 + e]; sB1[e] += other[144 + e];
      }
      if (C_BNB) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { sA1[e] *= corner[80 + 16 * lq + e]; sB1[e] *= corner[80 + 64 + e]; }
      }
      if (l16 == 0 && mw < a.M) {
        float* out = a.stats + ((size_t)(a.tile_base + mw / RN_CONV_STATS_ROWS) * 2) * a.Kd + kw;
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          *reinterpret_cast<float4*>(out + 16 * lq + e) = make_float4(sA0[e], sA0[e + 1], sA0[e + 2], sA0[e + 3]);
          *reinterpret_cast<float4*>(out + a.Kd + 16 * lq + e) = make_float4(sA1[e], sA1[e + 1], sA1[e + 2], sA1[e + 3]);
        }
        if (lq == 0) {
#pragma unroll
          for (int e = 0; e < 16; e += 4) {
            *reinterpret_cast<float4*>(out + 64 + e) = make_float4(sB0[e], sB0[e + 1], sB0[e + 2], sB0[e + 3]);
            *reinterpret_cast<float4*>(out + a.Kd + 64 + e) = make_float4(sB1[e], sB1[e + 1], sB1[e + 2], sB1[e + 3]);
          }
        }
      }
    }
    lds_barrier();                               // the exchange area and the coefficient corner are free again
  }
}

__host__ __device__ inline int r8_mode(const IgemmArgs& a) {     // -1: an operand set this kernel has no form for
  if (a.bias || a.ds != 1 || !(a.res.mode == RN_RES_NONE || a.res.mode == RN_RES_SAME)) return -1;
  const bool res = a.res.mode == RN_RES_SAME, acc = a.accum != 0;
  if (a.stats != nullptr && a.bn_x != nullptr) {
    if (!a.bn_mask || (res && acc)) return -1;
    return R8_BNB | (res ? R8_RES : 0) | (acc ? R8_ACC : 0);
  }
  if (acc) return -1;
  return res ? R8_RES : R8_PLAIN;
}

struct Items { int i0, h0, i1, h1; };          // the two (kernel row, half chunk) items of a group; i >= 3: past the end

// PROBE (diagnostic instantiations, rn_set_variant2 bits 4-6; wrong results, timing only): 1 = no LDS-DMA in the K loop, 2 = no MFMA, 3 = no fragment reads
// SPLIT: the two-halves form (a template parameter: its bookkeeping beside the widest operand sets tipped them into scratch)
template <typename T, int EPM, int PROBE = 0, int SPLIT = 0, int PERM = 1>
__global__ __launch_bounds__(512, 2) void igemm8r_kernel(const IgemmArgs a) {
  constexpr int BM = 256, BN = 160, ES = 2;
  constexpr int WN = 2;                                     // 4 x 2 waves of 64 x 80
  constexpr int WTM = 64, WTN = 80, RT = 4, CT = 5;
  constexpr int A_SZ = 2560;                                // uint4 per A patch buffer: up to 320 patch rows of 128 B (W = 8)
  constexpr int B_0 = 2 * A_SZ, B_SZ = BN * 8;              // three weight stages of 160 rows
  constexpr int C_0 = B_0 + 3 * B_SZ;                       // 8 corners of CORNER floats
  constexpr int NBQ = BN / 8;                               // 20 weight DMA pieces per K tile
  static_assert(sizeof(T) == ES, "16-bit element types");
  __shared__ uint4 smem[C_0 + 8 * CORNER / 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = a.Ws, H = a.Hs, W2 = W + 2, lw = a.w8_si;   // W = 1 << lw
  const int srows = BM >> lw;                               // image rows of a tile
  const int PP = srows * W2, nA = (PP + 7) >> 3;            // patch rows, DMA pieces of a patch
  const int nAw = (nA - wave + 7) >> 3;                     // this wave's pieces: q = wave + 8 t < nA  (4 or 5)
  const int nBw = wave < NBQ - 16 ? 3 : 2;                  // q = wave + 8 t < 20
  const int nhc = a.w8_cpc;                                 // half chunks per pixel: C / 32
  const int G = a.nk / 3;                                   // groups
  const int nnt = a.Kd / BN, nmt = (a.M + BM - 1) / BM, ntiles = nmt * nnt;
  constexpr int S = SPLIT ? 2 : 1;                          // 2: every tile is two work items (the halves of its group range)
  const int nitems = ntiles * S, gcut = (G + 1) >> 1;
  const int pq = H * W;
  const unsigned pixb = (unsigned)(a.Cs * ES);
  const size_t img_bytes = (size_t)pq * pixb;
  const v4i32 rb_desc = make_desc(a.wt, (size_t)a.Kd * a.wrs * a.Cs * ES);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);
  float* corner = reinterpret_cast<float*>(&smem[C_0]) + wave * CORNER;

  // the walk's constants in scalar registers: per kernel row the source byte shift, per tap the weight byte offset, per kernel column the patch row shift
  int dhv[3], dhoff[3], dwv[3], woff[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dhv[i] = __builtin_amdgcn_readfirstlane(a.dh[3 * i]);
    dhoff[i] = dhv[i] * W * (int)pixb;
    dwv[i] = __builtin_amdgcn_readfirstlane(a.dw[i]);
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) woff[t] = __builtin_amdgcn_readfirstlane(a.widx[t]) * (int)pixb;
  auto sel3 = [](int i, int v0, int v1, int v2) { return i == 0 ? v0 : (i == 1 ? v1 : v2); };
  auto items_next = [&](Items& it) {
    it.h0 += 2; if (it.h0 >= nhc) { it.h0 -= nhc; ++it.i0; }
    it.h1 += 2; if (it.h1 >= nhc) { it.h1 -= nhc; ++it.i1; }
  };

  // ---- lane roles that do not depend on the tile ----
  const int wm = wave / WN, wn = wave % WN;
  const int l16 = lane & 15, lq = lane >> 4;
  const int lrow = lane >> 3, p = lane & 7;
  // a DMA piece q covers LDS rows 8q .. 8q+7; lane = (row lrow, physical chunk p), which holds logical chunk p ^ ((row >> 1) & 7) = p ^ ((4q + lrow / 2) & 7):
  // the lane's pieces are q = wave + 8t, so bit 2 of its logical chunk -- the HALF of the K tile it copies -- is the same for all of them
  const int c3 = (p ^ (lrow >> 1)) & 3;
  const bool half = ((((p ^ (lrow >> 1)) >> 2) ^ wave) & 1) != 0;
  // fragment addresses (uint4 units): tile row -> patch row (+ the kernel column's shift); weights: row = column of the tile
  int fa[3][RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int ml = wm * WTM + 16 * i + pix16(l16, PERM != 0);
    const int pr0 = (ml >> lw) * W2 + (ml & (W - 1)) + 1;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pr = pr0 + dwv[j];
      fa[j][i] = pr * 8 + (lq ^ ((pr >> 1) & 7));
    }
  }
  const int fb0 = (wn * WTN + l16) * 8 + (lq ^ ((l16 >> 1) & 7));      // wn * 80 + 16 ct is a multiple of 16: the swizzle is that of l16

  // ---- per-tile DMA roles ----
  unsigned abase[5], amask = 0, bbase[3];
  int ra_w0 = 0, ra_w1 = 0, ra_w2 = 0;
  auto tile_roles = [&](int m0, int n0) {
    int n_first = (int)__umulhi((unsigned)m0, a.magic_pq);
    if (m0 - n_first * pq >= pq) ++n_first;
    {
      const v4i32 d = make_desc(reinterpret_cast<const char*>(a.src) + (size_t)n_first * img_bytes, (size_t)(a.N - n_first) * img_bytes);
      ra_w0 = d[0]; ra_w1 = d[1]; ra_w2 = d[2];
    }
    amask = 0;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int r = 8 * (wave + 8 * t) + lrow;                       // patch row
      const int s = (r * (int)a.w8_magic_ntw) >> 16, col = r - s * W2;          // r / (W + 2): exact for r < 1,024 (launcher)
      const int m = m0 + s * W + col - 1;
      unsigned base = 0, mk = 0;
      if (r < PP && col >= 1 && col <= W && m < a.M) {
        int n, hh, ww;
        decode_row(a, m, pq, n, hh, ww);
        base = (unsigned)(((n - n_first) * H + hh) * W + ww) * pixb + (unsigned)(c3 * 16);
#pragma unroll
        for (int i = 0; i < 3; ++i) mk |= ((unsigned)(hh + dhv[i]) < (unsigned)H ? 1u : 0u) << i;
      }
      abase[t] = base;
      amask |= mk << (4 * t);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int k = n0 + 8 * (wave + 8 * t) + lrow;                  // t = 2 exists for waves 0..3 only (issue guard)
      bbase[t] = (unsigned)k * (unsigned)a.wrs * pixb + (unsigned)(c3 * 16);
    }
  };
  auto a_desc = [&]() {
    v4i32 d;
    d[0] = __builtin_amdgcn_readfirstlane(ra_w0); d[1] = __builtin_amdgcn_readfirstlane(ra_w1);
    d[2] = __builtin_amdgcn_readfirstlane(ra_w2); d[3] = 0x00020000;
    return d;
  };
  // piece t of the A patch of a group: my_i = the lane's kernel row in that group (3: no item), sel = its source byte shift
  auto issue_a = [&](int t, unsigned buf_lds, int my_i, unsigned sel) {
    const v4i32 d = a_desc();
    const bool ok = (amask >> (4 * t + my_i)) & 1u;
    dma16(d, ok ? abase[t] + sel : OOB, buf_lds + (unsigned)((wave + 8 * t) * 1024));
  };
  auto issue_b = [&](int t, unsigned stage_lds, unsigned sel, bool ok) {
    dma16(rb_desc, ok ? bbase[t] + sel : OOB, stage_lds + (unsigned)((wave + 8 * t) * 1024));
  };
  // per-lane selects of a group's / K tile's two halves (the items are wave-uniform: pinned to scalar registers, so that a select between two of
  // them is a v_cndmask of two SGPRs -- left as struct members hipcc turned `half ? it.i1 : it.i0` into a dynamically indexed load from scratch)
  auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto lane_i = [&](const Items& it) { const int x0 = sgpr(it.i0), x1 = sgpr(it.i1); return half ? x1 : x0; };
  auto lane_src = [&](const Items& it) {
    const int lo = sgpr(sel3(it.i0, dhoff[0], dhoff[1], dhoff[2]) + it.h0 * 64), hi = sgpr(sel3(it.i1, dhoff[0], dhoff[1], dhoff[2]) + it.h1 * 64);
    return (unsigned)(half ? hi : lo);
  };
  auto lane_wt = [&](const Items& it, int j) {
    const int lo = sgpr(sel3(it.i0, woff[j], woff[3 + j], woff[6 + j]) + it.h0 * 64), hi = sgpr(sel3(it.i1, woff[j], woff[3 + j], woff[6 + j]) + it.h1 * 64);
    return (unsigned)(half ? hi : lo);
  };
  auto lane_ok = [&](const Items& it) { const int hv = sgpr(it.i1 < 3 ? 1 : 0); return (half ? hv : 1) != 0; };

  f32x4 acc[RT][CT];
  uint4 bf[CT][2], af[2][2];
  if constexpr (PROBE == 3) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bf[ct][0] = bf[ct][1] = make_uint4(lane, 1, 2, 3);
    af[0][0] = af[0][1] = af[1][0] = af[1][1] = make_uint4(3, 2, 1, lane);
  }
  // one K tile (g, J) = two phases.  F bit 0: the patch of group g+1 is issued during this group; bit 1: K tile kt+2 exists (its weights are issued);
  // bit 2: K tile kt+1 exists (its weights, and a next group's patch, are waited for)
  auto ktile = [&](auto jtag, auto ftag, int a_cur, unsigned a_nxt, int my_i_n, unsigned sel_a_n, unsigned sel_b, bool ok_b) {
    constexpr int J = decltype(jtag)::value, F = decltype(ftag)::value;
    constexpr bool IA = (F & 1) != 0 && PROBE != 1, IB = (F & 2) != 0 && PROBE != 1, NEXT = (F & 4) != 0;
    const uint4* SB = &smem[B_0 + J * B_SZ];               // K tile 3g + J lives in weight stage J
    const uint4* SA = &smem[a_cur];
    const unsigned b_nxt = lds0 + (unsigned)((B_0 + ((J + 2) % 3) * B_SZ) * 16);
    // ---- phase 1: all B fragments, A rows 0-31 of the wave ----
    if constexpr (PERM == 2 && (IA || IB)) {                 // experiment: the pieces in FRONT of the fragment reads
      const unsigned keep = m0_save();
      if constexpr (IA) { if (2 * J < 4 || nAw > 4) issue_a(2 * J, a_nxt, my_i_n, sel_a_n); }
      if constexpr (IB) issue_b(0, b_nxt, sel_b, ok_b);
      m0_restore(keep);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PROBE != 3) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { bf[ct][0] = SB[fb0 + ct * 128]; bf[ct][1] = SB[(fb0 ^ 4) + ct * 128]; }
#pragma unroll
      for (int i = 0; i < 2; ++i) { af[i][0] = SA[fa[J][i]]; af[i][1] = SA[fa[J][i] ^ 4]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PERM != 2 && (IA || IB)) {
      const unsigned keep = m0_save();
      if constexpr (IA) { if (2 * J < 4 || nAw > 4) issue_a(2 * J, a_nxt, my_i_n, sel_a_n); }
      if constexpr (IB) issue_b(0, b_nxt, sel_b, ok_b);
      m0_restore(keep);
    }
    raw_barrier();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if constexpr (PROBE != 2) Mfma16<T>::run(bf[ct][ks], af[i][ks], acc[i][ct]);
          else asm volatile("" : "+v"(acc[i][ct]) : "v"(bf[ct][ks].x), "v"(bf[ct][ks].w), "v"(af[i][ks].x), "v"(af[i][ks].w));
        }
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
    // ---- phase 2: A rows 32-63 of the wave, B from registers ----
    if constexpr (PERM == 2 && (IA || IB)) {
      const unsigned keep = m0_save();
      if constexpr (IA && J < 2) issue_a(2 * J + 1, a_nxt, my_i_n, sel_a_n);
      if constexpr (IB) {
        issue_b(1, b_nxt, sel_b, ok_b);
        if (nBw > 2) issue_b(2, b_nxt, sel_b, ok_b);
      }
      m0_restore(keep);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PROBE != 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { af[i][0] = SA[fa[J][2 + i]]; af[i][1] = SA[fa[J][2 + i] ^ 4]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PERM != 2 && (IA || IB)) {
      const unsigned keep = m0_save();
      if constexpr (IA && J < 2) issue_a(2 * J + 1, a_nxt, my_i_n, sel_a_n);
      if constexpr (IB) {
        issue_b(1, b_nxt, sel_b, ok_b);
        if (nBw > 2) issue_b(2, b_nxt, sel_b, ok_b);
      }
      m0_restore(keep);
    }
    // everything older than this K tile's own pieces has landed behind this wait: the weights of K tile kt+1, and -- J = 2 -- the last piece of
    // the next group's patch (issued in phase 1, in FRONT of this tile's weight pieces)
    if constexpr (NEXT) wait_vm((IB ? nBw : 0) + ((IA && J < 2) ? 2 : 0));
    wait_lgkm<0>();                                          // the A reads are retired in front of the barrier: the buffer may be re-staged next phase
    raw_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if constexpr (PROBE != 2) Mfma16<T>::run(bf[ct][ks], af[i][ks], acc[2 + i][ct]);
          else asm volatile("" : "+v"(acc[2 + i][ct]) : "v"(bf[ct][ks].x), "v"(bf[ct][ks].w), "v"(af[i][ks].x), "v"(af[i][ks].w));
        }
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using F7 = std::integral_constant<int, 7>; using F6 = std::integral_constant<int, 6>; using F4 = std::integral_constant<int, 4>;

  // ---- persistent walk over whole tiles: column tiles fastest, an XCD's workgroups (b, b + 8, ...) on a contiguous range of the order ----
  const int Gw = gridDim.x;
  int pm0 = -1, pn0 = 0, pitem = 0;
  for (int itn = 0;; ++itn) {
    const int vb = itn * Gw + blockIdx.x;
    const bool more = SPLIT ? itn == 0 : vb < nitems;       // wave-uniform; the split form's grid holds one work item per workgroup (launcher)
    const unsigned long long* stp = itn < 2 ? a.stamps : nullptr;      // diagnostic (rn_set_stamp_buffer): the workgroup's first two tiles, slots 6 itn + 0..5
    const int sb = 6 * itn;
    stamp(stp, sb);
    int m0 = 0, n0 = 0, item = 0, g_lo = 0, g_hi = G;
    Items cur{0, 0, 0, 1}, nxt{0, 0, 0, 0};
    if (more) {
      item = vb;
      if (a.xcd_remap) {
        const int xcd = vb & 7, q = nitems >> 3, r = nitems & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
      }
      int tile = item;
      if constexpr (S == 2) {
        tile = item >> 1;
        if (item & 1) g_lo = gcut; else g_hi = gcut;
        const int i0 = (2 * g_lo) / nhc, i1 = (2 * g_lo + 1) / nhc;
        cur = Items{i0, 2 * g_lo - i0 * nhc, i1, 2 * g_lo + 1 - i1 * nhc};
      }
      int mt = (int)__umulhi((unsigned)tile, (unsigned)a.w8_magic_nnt);
      if (tile - mt * nnt >= nnt) ++mt;
      m0 = mt * BM; n0 = (tile - mt * nnt) * BN;
      tile_roles(m0, n0);
      stamp(stp, sb + 1);
      // prologue: the patch of group 0 into A buffer 0, the weights of K tiles 0 and 1 into stages 0 and 1 (every wave has left the previous K loop)
      const int mi = lane_i(cur);
      const unsigned sa = lane_src(cur);
      const unsigned keep = m0_save();
#pragma unroll
      for (int t = 0; t < 5; ++t) if (t < 4 || nAw > 4) issue_a(t, lds0, mi, sa);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned sb = lane_wt(cur, j);
        const bool okb = lane_ok(cur);
        const unsigned stage = lds0 + (unsigned)((B_0 + j * B_SZ) * 16);
        issue_b(0, stage, sb, okb); issue_b(1, stage, sb, okb);
        if (nBw > 2) issue_b(2, stage, sb, okb);
      }
      m0_restore(keep);
      nxt = cur; items_next(nxt);
    }
    stamp(stp, sb + 2);
    // the previous tile leaves the registers while this one's first pieces are in flight
    bool stores_behind = false;
    if (pm0 >= 0) {
      bool finish = true;
      if constexpr (S == 2) {
        // ---- one half of a tile's reduction: publish it, draw a ticket; the second arriver adds the halves ----
        const int ptile = pitem >> 1;
        f32x4* slots = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.w8_ws) + 4096);
        constexpr size_t SLOT = (size_t)512 * 128 / 4;       // f32x4 per slot (conv_igemm8.hip's SK_SLOT_BYTES); register-major, thread-minor: coalesced 16-byte accesses
        {
          const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(slots), 0, 0x7FFFFFF0, 0x00020000);
          const unsigned mine_off = (unsigned)(((size_t)pitem * SLOT + tid) * 16);
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)           // write-through (sc1): the bytes leave this XCD's L2 at once, publishing needs no release fence
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, acc[i][j]), wsr, (int)(mine_off + (unsigned)((i * CT + j) * 512 * 16)), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains, then the workgroup meets, then ONE lane signals
        __syncthreads();
        int* cnt = reinterpret_cast<int*>(a.w8_ws) + ptile;
        volatile int* flagw = reinterpret_cast<volatile int*>(reinterpret_cast<float*>(&smem[C_0]) + 500);      // a word of wave 0's corner that no epilogue uses
        if (tid == 0) *flagw = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        finish = *flagw == 1;                                // wave-uniform
        if (finish) {
          if (tid == 0) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // both halves have arrived: the counter is clean for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __syncthreads();
          // the partner's part is added to the registers: fp32 addition commutes, so the sum of TWO parts does not depend on who arrived last
          const f32x4* so = slots + (size_t)(pitem ^ 1) * SLOT + tid;
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] += so[(size_t)(i * CT + j) * 512];
        } else {
          __syncthreads();                                   // same barrier count on both paths
        }
      }
      if (finish) {
        stores_behind = pm0 + BM <= a.M;
        epilogue8r<T, EPM, PERM>(a, acc, pm0 + wm * WTM, pn0 + wn * WTN, lane, corner, (wm & 1) != 0, WN * CORNER);
      }
    }
    stamp(stp, sb + 3);
    if (!more) break;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // the patch and K tile 0 have landed; K tile 1's pieces (and the previous tile's stores, which are younger than all of them) may stay in flight
    if (stores_behind) { if (nBw > 2) wait_vmcnt<3 + 10>(); else wait_vmcnt<2 + 10>(); }
    else wait_vm(nBw);
    raw_barrier();
    if (wave >= 4) raw_barrier();                           // the second wave group runs one barrier behind
    stamp(stp, sb + 4);

    int a_cur = 0;
    for (int g = g_lo; g + 1 < g_hi; ++g) {
      const unsigned a_nxt = lds0 + (unsigned)((a_cur ^ A_SZ) * 16);
      const int mi = lane_i(nxt);
      const unsigned sa = lane_src(nxt);
      ktile(I0{}, F7{}, a_cur, a_nxt, mi, sa, lane_wt(cur, 2), lane_ok(cur));        // issues the weights of K tile (g, 2)
      ktile(I1{}, F7{}, a_cur, a_nxt, mi, sa, lane_wt(nxt, 0), lane_ok(nxt));        // ... (g+1, 0)
      ktile(I2{}, F7{}, a_cur, a_nxt, mi, sa, lane_wt(nxt, 1), lane_ok(nxt));        // ... (g+1, 1)
      cur = nxt; items_next(nxt);
      a_cur ^= A_SZ;
    }
    ktile(I0{}, F6{}, a_cur, 0u, 0, 0u, lane_wt(cur, 2), lane_ok(cur));
    ktile(I1{}, F4{}, a_cur, 0u, 0, 0u, 0u, false);
    ktile(I2{}, std::integral_constant<int, 0>{}, a_cur, 0u, 0, 0u, 0u, false);
    if (wave < 4) raw_barrier();                            // the first group waits for the second: every wave has executed the same barriers
    stamp(stp, sb + 5);
    pm0 = m0; pn0 = n0; pitem = item;
  }
}

bool r8_geom_ok(const IgemmArgs& a) {
  if (a.nt != 9 || a.nth != 3 || a.ntw != 3 || a.ss != 1 || a.ds != 1 || a.Hs != a.Pc || a.Ws != a.Qc) return false;
  if (a.Cs < 64 || a.Cs % 32 || a.Kd % 160 || a.wrs != 9) return false;
  const int W = a.Ws;
  if (W < 8 || W > 256 || (W & (W - 1))) return false;             // a tile is 256 / W whole image rows; the patch fits 320 LDS rows
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int t = 3 * i + j;
      if (a.dh[t] != a.dh[3 * i] || a.dw[t] != a.dw[j] || a.dh[t] < -1 || a.dh[t] > 1 || a.dw[t] < -1 || a.dw[t] > 1) return false;
      if (a.widx[t] < 0 || a.widx[t] >= 9) return false;
    }
  if (a.dh[0] == a.dh[3] || a.dh[3] == a.dh[6] || a.dh[0] == a.dh[6]) return false;
  return true;
}

template <typename T> int launch8r(IgemmArgs& a, hipStream_t s) {
  const int epm = r8_mode(a);
  static const char* const EPN[] = {"plain", "res", "?", "?", "bnb", "bnb+res", "bnb+acc"};
  rn_note_kernel(rn_igemm8r_split_ok(a) ? "igemm8r<256x160/2:%s>" : "igemm8r<256x160:%s>", EPN[epm]);
  if (rn_dry_run()) return 0;
  const int W = a.Ws;
  int lw = 0;
  while ((1 << lw) < W) ++lw;
  a.w8_si = lw;
  a.w8_cpc = a.Cs / 32;
  a.nk = 3 * ((3 * a.w8_cpc + 1) / 2);
  a.w8_magic_ntw = (unsigned)((65536 + W + 1) / (W + 2));
  { const unsigned nnt = (unsigned)(a.Kd / 160); a.w8_magic_nnt = nnt <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / nnt); }
  const int ntiles = cdiv(a.M, 256) * (a.Kd / 160);
  a.w8_dp_tiles = 1;
  a.w8_ws = nullptr;
  if (rn_igemm8r_split_ok(a)) { a.w8_dp_tiles = 2; a.w8_ws = rn_sk_workspace(nullptr); }
  const int nitems = ntiles * a.w8_dp_tiles;
  const int grid = nitems < 256 ? nitems : 256;
  if constexpr (std::is_same<T, f16_t>::value) {
    const int probe = (g_rn_variant2 >> 4) & 7;
    if (probe && epm == R8_PLAIN) {
      if (probe == 1) hipLaunchKernelGGL((igemm8r_kernel<T, R8_PLAIN, 1>), dim3(grid), dim3(512), 0, s, a);
      else if (probe == 2) hipLaunchKernelGGL((igemm8r_kernel<T, R8_PLAIN, 2>), dim3(grid), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((igemm8r_kernel<T, R8_PLAIN, 3>), dim3(grid), dim3(512), 0, s, a);
      RN_CHECK_LAUNCH("igemm8r probe");
      return 0;
    }
  }
  if ((g_rn_variant2 & 2097152) && epm == R8_PLAIN && a.w8_dp_tiles == 1) {
    hipLaunchKernelGGL((igemm8r_kernel<T, R8_PLAIN, 0, 0, 2>), dim3(grid), dim3(512), 0, s, a);
    RN_CHECK_LAUNCH("igemm8r");
    return 0;
  }
  if ((g_rn_variant2 & 4096) && epm == R8_PLAIN && a.w8_dp_tiles == 1) {
    hipLaunchKernelGGL((igemm8r_kernel<T, R8_PLAIN, 0, 0, 0>), dim3(grid), dim3(512), 0, s, a);
    RN_CHECK_LAUNCH("igemm8r");
    return 0;
  }
  if (a.w8_dp_tiles == 2) {
    switch (epm) {
      case R8_PLAIN: hipLaunchKernelGGL((igemm8r_kernel<T, R8_PLAIN, 0, 1>), dim3(grid), dim3(512), 0, s, a); break;
      case R8_RES: hipLaunchKernelGGL((igemm8r_kernel<T, R8_RES, 0, 1>), dim3(grid), dim3(512), 0, s, a); break;
      case R8_BNB: hipLaunchKernelGGL((igemm8r_kernel<T, R8_BNB, 0, 1>), dim3(grid), dim3(512), 0, s, a); break;
      case R8_BNB | R8_RES: hipLaunchKernelGGL((igemm8r_kernel<T, R8_BNB | R8_RES, 0, 1>), dim3(grid), dim3(512), 0, s, a); break;
      default: hipLaunchKernelGGL((igemm8r_kernel<T, R8_BNB | R8_ACC, 0, 1>), dim3(grid), dim3(512), 0, s, a); break;
    }
    RN_CHECK_LAUNCH("igemm8r split");
    return 0;
  }
  switch (epm) {
    case R8_PLAIN: hipLaunchKernelGGL((igemm8r_kernel<T, R8_PLAIN>), dim3(grid), dim3(512), 0, s, a); break;
    case R8_RES: hipLaunchKernelGGL((igemm8r_kernel<T, R8_RES>), dim3(grid), dim3(512), 0, s, a); break;
    case R8_BNB: hipLaunchKernelGGL((igemm8r_kernel<T, R8_BNB>), dim3(grid), dim3(512), 0, s, a); break;
    case R8_BNB | R8_RES: hipLaunchKernelGGL((igemm8r_kernel<T, R8_BNB | R8_RES>), dim3(grid), dim3(512), 0, s, a); break;
    default: hipLaunchKernelGGL((igemm8r_kernel<T, R8_BNB | R8_ACC>), dim3(grid), dim3(512), 0, s, a); break;
  }
  RN_CHECK_LAUNCH("igemm8r");
  return 0;
}

}  // namespace

// 1: a geometry + operand set the row-patch kernel covers
int rn_igemm8r_ok(const IgemmArgs& a) {
  if (!r8_geom_ok(a) || r8_mode(a) < 0) return 0;
  const size_t img_bytes = (size_t)a.Hs * a.Ws * a.Cs * 2;
  const long pq = (long)a.Pc * a.Qc;
  if ((256 / pq + 3) * (double)img_bytes >= 4.0e9) return 0;         // per-tile source offsets are 32-bit
  if (((double)a.Kd + 256.0) * a.wrs * a.Cs * 2 >= 4.0e9) return 0;
  return 1;
}

// the split form: 2 x tiles work items fit one round of the chip, the reduction is long enough to cut, the workspace is there (rn_set_variant2 131072: never; 262144: on grids of < 96 tiles too)
int rn_igemm8r_split_ok(const IgemmArgs& a) {
  if (g_rn_variant2 & 131072) return 0;
  const long ntiles = (long)cdiv(a.M, 256) * (a.Kd / 160);
  const int groups = (3 * (a.Cs / 32) + 1) / 2;
  size_t slot = 0;
  if (2 * ntiles > 256 || groups < 8 || !rn_sk_workspace(&slot)) return 0;
  if (ntiles < 96 && !(g_rn_variant2 & 262144)) return 0;                      // (262144: at any size, tests)
  return slot >= (size_t)512 * 80 * 4 ? 1 : 0;
}

int rn_launch_igemm8r(const IgemmArgs& a_in, int dtype, hipStream_t s) {
  if (dtype != RN_BF16 && dtype != RN_F16) return -1;
  if (a_in.M <= 0 || !rn_igemm8r_ok(a_in)) return -1;
  IgemmArgs a = a_in;
  return dtype == RN_BF16 ? launch8r<bf16_t>(a, s) : launch8r<f16_t>(a, s);
}
