// Implicit-GEMM convolution on the EIGHT-PHASE schedule (gfx950), 16-bit element types: 256 x BN block tiles, 8 waves, one PERSISTENT workgroup per CU.
//
// Same contraction and tap-table view as conv_igemm.hip (forward conv and data gradient of residual_block.py:34-57, :129-159),
//   dst[n, p*ds+oh, q*ds+ow, k] (+)= sum_t sum_c src[n, p*ss+dh[t], q*ss+dw[t], c] * wt[k][widx[t]][c]   (+ res),
// but the K loop is the deep-pipelined structure of cdna_hip_programming.md section 5 ("256^2 8-phase template"), carried over to the
// implicit-GEMM operand walk:
//   * a K tile = one tap x 64 channels (128-byte rows); per K tile the A tile (256 pixels) and the B tile (BN output channels) are each
//     split in two HALF-TILES = what one quadrant row / column of every wave's output block reads;
//   * a wave owns 128 x 64 outputs as four quadrants; one PHASE = the fragment reads of one half-tile, the LDS-DMA of one half-tile two K
//     tiles ahead, and the MFMAs of one quadrant x 64 channels (v_mfma_f32_16x16x32), i.e. four phases per K tile; both LDS stages of every
//     half-tile live in ONE array (128 KiB);
//   * the two wave groups (waves 0-3 / 4-7, the two waves of each SIMD) run the same program ONE BARRIER apart, so one group's MFMA
//     segment covers the other's LDS reads and DMA issue;
//   * DMAs stay in flight across barriers: a counted s_waitcnt vmcnt once per K tile (three half-tiles remain outstanding), raw s_barrier.
// Hazard bookkeeping (M_i / C_i = memory / MFMA segment of phase i; group g runs M_i in barrier interval 2(i-1)+g, C_i one later):
//   RAW  the wait sits in M_4 of a K tile, in front of that segment's closing barrier; every group has passed it when interval 8k ends;
//        the data is first read in M_1 of the next K tile (interval 8k for group 0): one phase after the wait, never in the same phase.
//   WAR  a half-tile is restaged two phases after its last fragment read (the reads are retired by the lgkmcnt(0) of the reading phase's
//        C segment), or ONE phase after when a counted lgkmcnt in front of the reading phase's closing barrier retired them (the B reads
//        of phase 1 are issued first and retired by lgkmcnt(<A reads>)).
// The taps are walked with scalar adds (IgemmArgs.w8_*: a separable arithmetic progression, checked on the host), padding taps and row
// tails are out-of-range DMA offsets that the hardware zero-fills.
//
// MFMA orientation and epilogue.  The products are taken TRANSPOSED (weights as the A operand, pixels as B): a lane then holds, per 16 x 16 tile,
// FOUR CONSECUTIVE CHANNELS of one pixel.  A 4 x 4 transpose across the wave's four 16-lane rows (v_permlane32_swap + v_permlane16_swap, fp32)
// gives every lane 16 consecutive channels of its pixel = two 16-byte chunks, so the accumulators go to global memory straight from registers:
// no LDS staging, no workgroup barrier, every fused operand (bias, residual, accumulate, BatchNorm-backward x / mask) a 16-byte access in the
// same layout, the BatchNorm sums reduced over a lane row by DPP adds into ONE partial row per wave (a wave = 128 output rows = one row of
// the [rows][2][K] partial-sum buffer).
// Because the epilogue needs no LDS the workgroup is persistent: it walks its tiles, and the first K tiles of tile i+1 are already in flight
// (LDS-DMA) while the waves store tile i.
#include "igemm_shared.h"

static void* g_sk_ws = nullptr;
static size_t g_sk_ws_bytes = 0;
static int g_sk_ws_dev = -1;                              // the device the workspace lives on: launches on another device run without it (whole tiles)
static bool sk_ws_here() {
  int dev = -1;
  return g_sk_ws && hipGetDevice(&dev) == hipSuccess && dev == g_sk_ws_dev;
}

namespace {

constexpr size_t SK_CNT_BYTES = 4096;                     // stream-K workspace: 1,024 tile tickets, then 2 slots of fp32 accumulators per workgroup
constexpr int SK_GRID = 256;                              // persistent workgroups (one per CU)
constexpr size_t SK_SLOT_BYTES = (size_t)512 * 128 * 4;   // 256 x 256 tile: 128 accumulator registers of 512 threads

template <int N> __device__ inline void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void raw_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

struct Walk8 {          // scalar position of one K tile: tap bit, source / weight byte offsets
  int i, j, cc, t;
  unsigned src, wt;
};
// the walk's constants, held in scalar registers for the kernel's lifetime (the K loop must not re-read kernel arguments: a scalar load inside
// it would sit in the same counter as the fragment reads)
struct Walk8K { int cpc, ntw; unsigned dsj, dsi, dwj, dwi; };
__device__ inline void walk8_advance(const Walk8K& k, Walk8& w) {
  w.cc += 1; w.src += 128u; w.wt += 128u;
  if (w.cc == k.cpc) {
    w.cc = 0; w.j += 1; w.t += 1;
    w.src += k.dsj; w.wt += k.dwj;
    if (w.j == k.ntw) {
      w.j = 0; w.i += 1;
      w.src += k.dsi; w.wt += k.dwi;
    }
  }
}
__device__ inline void walk8_seek(const IgemmArgs& a, Walk8& w, int g) {      // K tile g = (tap g / cpc, chunk g % cpc)
  const int tap = g / a.w8_cpc;
  w.cc = g - tap * a.w8_cpc; w.t = tap;
  w.i = tap / a.ntw; w.j = tap - w.i * a.ntw;
  w.src = (unsigned)(a.w8_src0 + w.i * a.w8_si + w.j * a.w8_sj + w.cc * 128);
  w.wt = (unsigned)(a.w8_wt0 + w.i * a.w8_wi + w.j * a.w8_wj + w.cc * 128);
}

// ---- lane-row transposes (all 64 lanes active) ----
__device__ inline void swap32(float& d, float& s) {       // rows 2,3 of d <-> rows 0,1 of s
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, d), __builtin_bit_cast(unsigned, s), false, false);
  d = __builtin_bit_cast(float, (unsigned)r[0]); s = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ inline void swap16(float& d, float& s) {       // rows 1,3 of d <-> rows 0,2 of s
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, d), __builtin_bit_cast(unsigned, s), false, false);
  d = __builtin_bit_cast(float, (unsigned)r[0]); s = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ inline void swap32u(unsigned& d, unsigned& s) {
  const auto r = __builtin_amdgcn_permlane32_swap(d, s, false, false);
  d = r[0]; s = r[1];
}
__device__ inline void swap16u(unsigned& d, unsigned& s) {
  const auto r = __builtin_amdgcn_permlane16_swap(d, s, false, false);
  d = r[0]; s = r[1];
}
__device__ inline void xpose4u(unsigned& x0, unsigned& x1, unsigned& x2, unsigned& x3) {
  swap32u(x0, x2); swap32u(x1, x3);
  swap16u(x0, x1); swap16u(x2, x3);
}
// in: x_c on lane row q = M[q][c]; out: x_s on lane row q = M[s][q]
__device__ inline void xpose4(float& x0, float& x1, float& x2, float& x3) {
  swap32(x0, x2); swap32(x1, x3);
  swap16(x0, x1); swap16(x2, x3);
}
// sum over the 16 lanes of a row, result in every lane: xor 1, xor 2 (quad permutes), half-row mirror, row mirror
__device__ inline float row_sum16(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
  return x;
}

// ---- epilogue: a wave stores its 16 RT x 64 block from registers.  acc[pt][ct][r] = pixel mw + 16 pt + (lane & 15), channel kw + 16 ct + 4 (lane >> 4) + r ----
// Specialised per operand set (compile-time MODE, picked once per launch): with the flags at run time inside one unrolled body the register
// allocator has to cover the union of all operand sets next to the 128 accumulator registers and spills.
//   EP8_PLAIN        dense destination, no fused operand (forward convolutions; optional BatchNorm statistics)
//   EP8_RES          + identity residual (the block's last convolution)
//   EP8_BNB [+ RES | + ACC]   dense destination + BatchNorm-backward sums over (x, mask); with the shortcut gradient as identity residual, or
//                    accumulating into the destination (the three forms the block backward of residual_block.py:67-99, :173-215 lowers to)
//   EP8_GEN          everything else, flags at run time: strided destination (parity classes of a stride-2 data gradient), pad / subsample
//                    residuals, bias, sums without a mask (slow path: it spills)
enum { EP8_PLAIN = 0, EP8_RES = 1, EP8_ACC = 2, EP8_BNB = 4, EP8_GEN = 8, EP8_BIAS = 16, EP8_STRIDED = 32, EP8_MASKX = 64 };   // EP8_MASKX (with EP8_BNB): the ReLU mask is [x * scale + shift > 0] of the x the sums read anyway, not a second tensor   // EP8_STRIDED: destination pixel (2p + oh, 2q + ow): a parity class of a stride-2 data gradient     // EP8_BIAS: plain + per-channel bias (the stem convolution, resnet.py:69-75)

template <typename T, int RT, int MODE>
__device__ inline void epilogue8(const IgemmArgs& a, f32x4 (&acc)[RT][4], int mw, int kw, int lane, float* lds_mean, bool upper, int pair_floats) {
  constexpr int CE = 8;
  constexpr bool GEN = MODE == EP8_GEN, C_BIAS = MODE == EP8_BIAS, C_STRIDED = (MODE & EP8_STRIDED) != 0;
  constexpr bool C_RES = (MODE & EP8_RES) != 0, C_ACC = (MODE & EP8_ACC) != 0, C_BNB = (MODE & EP8_BNB) != 0, C_MASKX = (MODE & EP8_MASKX) != 0;
  constexpr int D = MODE == EP8_RES ? 2 : 1;     // pixel tiles of operand loads in flight ahead of the one being processed (registers decide)
  const int l16 = lane & 15, lq = lane >> 4;
  const int kc = kw + 16 * lq;                   // after the transpose this lane owns channels kc .. kc + 15 of its pixel
  T* __restrict__ dst = reinterpret_cast<T*>(a.dst);
  const bool dense = !C_STRIDED && (!GEN || ((a.ds == 1) && (a.res.mode == RN_RES_NONE || a.res.mode == RN_RES_SAME)));
  const bool want_stats = a.stats != nullptr;
  const bool bn_bwd = C_BNB || (GEN && want_stats && a.bn_x != nullptr);
  const bool has_mask = (C_BNB && !C_MASKX) || (GEN && a.bn_mask != nullptr);
  const bool res_same = C_RES || (GEN && a.res.mode == RN_RES_SAME);
  const bool accum = C_ACC || (GEN && a.accum);
  const int pq = a.Pc * a.Qc;
  float s0[16], s1[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) s0[e] = s1[e] = 0.f;

  auto transposed = [&](int pt, float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x0 = acc[pt][0][r], x1 = acc[pt][1][r], x2 = acc[pt][2][r], x3 = acc[pt][3][r];
      xpose4(x0, x1, x2, x3);
      v[r] = x0; v[4 + r] = x1; v[8 + r] = x2; v[12 + r] = x3;
    }
  };
  auto round_store = [&](size_t off, const float (&v)[16], Chunk<T> (&st)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int e = 0; e < CE; ++e) st[c].e[e] = Elem<T>::from_f(v[c * CE + e]);
      store_chunk<T>(dst + off + c * CE, st[c]);
    }
  };

  if constexpr (MODE == EP8_PLAIN) {
    // nothing is added to the accumulators: round FIRST, then transpose the packed pairs -- half the lane exchanges
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
      Chunk<T> st[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned x[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          union { T e[2]; unsigned u; } pk;
          pk.e[0] = Elem<T>::from_f(acc[pt][ct][2 * h]); pk.e[1] = Elem<T>::from_f(acc[pt][ct][2 * h + 1]);
          x[ct] = pk.u;
        }
        xpose4u(x[0], x[1], x[2], x[3]);           // x[s] = channels 16 lq + 4 s + 2 h + {0, 1}
        st[0].u.x = h == 0 ? x[0] : st[0].u.x; st[0].u.y = h == 1 ? x[0] : st[0].u.y;
        st[0].u.z = h == 0 ? x[1] : st[0].u.z; st[0].u.w = h == 1 ? x[1] : st[0].u.w;
        st[1].u.x = h == 0 ? x[2] : st[1].u.x; st[1].u.y = h == 1 ? x[2] : st[1].u.y;
        st[1].u.z = h == 0 ? x[3] : st[1].u.z; st[1].u.w = h == 1 ? x[3] : st[1].u.w;
      }
      const int m = mw + 16 * pt + l16;
      if (m < a.M) {
        T* o = dst + (size_t)m * a.Kd + kc;
        store_chunk<T>(o, st[0]);
        store_chunk<T>(o + CE, st[1]);
        if (want_stats) {
#pragma unroll
          for (int e = 0; e < 16; ++e) { const float vs = Elem<T>::to_f(st[e / CE].e[e % CE]); s0[e] += vs; s1[e] += vs * vs; }
        }
      }
    }
  } else {
    // BatchNorm-backward second sum: sum g * xhat = invstd * sum g * (x - mean): only the mean is held per channel, invstd multiplies the row sum
    // The means of this lane row's 16 channels are parked in a wave-private corner of LDS (64 floats per wave) and re-read per pixel tile: sixteen
    // registers the BatchNorm-backward specialisations do not have (LDS reads do not touch vmcnt; a scratch reload would wait for the stores in flight).
    // The 1 / std of the same channels wait next to them (floats 64..127 of the corner) until the sums are scaled: a vector load issued behind the tile's
    // stores would have to wait for every one of them (vmcnt counts in issue order).
    float* lm = lds_mean + 16 * lq;
    if (bn_bwd) {
      if (l16 < 4) *reinterpret_cast<float4*>(lm + 4 * l16) = *reinterpret_cast<const float4*>(a.bn_coef + 2 * a.Kd + kc + 4 * l16);
      else if (l16 < 8) *reinterpret_cast<float4*>(lm + 64 + 4 * (l16 - 4)) = *reinterpret_cast<const float4*>(a.bn_coef + 3 * a.Kd + kc + 4 * (l16 - 4));
      else if (C_MASKX && l16 < 12) *reinterpret_cast<float4*>(lm + 128 + 4 * (l16 - 8)) = *reinterpret_cast<const float4*>(a.bn_coef + kc + 4 * (l16 - 8));            // scale
      else if (C_MASKX) *reinterpret_cast<float4*>(lm + 192 + 4 * (l16 - 12)) = *reinterpret_cast<const float4*>(a.bn_coef + a.Kd + kc + 4 * (l16 - 12));      // shift
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // same wave writes and reads: program order + the wait
    }
    struct Ops { bool ok; size_t off; int n, hd, wd; Chunk<T> cr[2], co[2], cx[2], cm[2]; };
    auto fetch = [&](int pt, Ops& o) {
      const int m = mw + 16 * pt + l16;
      o.ok = m < a.M;
      o.n = o.hd = o.wd = 0; o.off = 0;
      if (!o.ok) return;
      size_t pix;
      if (dense) {
        pix = (size_t)m;
      } else {
        int pp, q;
        decode_row(a, m, pq, o.n, pp, q);
        o.hd = pp * a.ds + a.oh;
        o.wd = q * a.ds + a.ow;
        pix = ((size_t)o.n * a.Hd + o.hd) * a.Wd + o.wd;
      }
      o.off = pix * a.Kd + kc;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (res_same) o.cr[c] = load_chunk<T>(reinterpret_cast<const T*>(a.res.ptr) + o.off + c * CE);
        if (accum) o.co[c] = load_chunk<T>(dst + o.off + c * CE);
        if (bn_bwd) {
          o.cx[c] = load_chunk<T>(reinterpret_cast<const T*>(a.bn_x) + o.off + c * CE);
          if (has_mask) o.cm[c] = load_chunk<T>(reinterpret_cast<const T*>(a.bn_mask) + o.off + c * CE);
        }
      }
    };
    Ops ops[RT];
#pragma unroll
    for (int pt = 0; pt < D && pt < RT; ++pt) fetch(pt, ops[pt]);
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
      if (pt + D < RT) fetch(pt + D, ops[pt + D]);
      float v[16];
      transposed(pt, v);
      const Ops& o = ops[pt];
      if (o.ok) {
        if (C_BIAS || (GEN && a.bias)) {
#pragma unroll
          for (int e4 = 0; e4 < 16; e4 += 4) {
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + kc + e4);
            v[e4] += bv.x; v[e4 + 1] += bv.y; v[e4 + 2] += bv.z; v[e4 + 3] += bv.w;
          }
        }
        if (res_same) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += Elem<T>::to_f(o.cr[e / CE].e[e % CE]);
        } else if (GEN && a.res.mode != RN_RES_NONE) {
          res_add_chunk<T>(a.res, o.n, o.hd, o.wd, kc, v);
          res_add_chunk<T>(a.res, o.n, o.hd, o.wd, kc + CE, v + CE);
        }
        if (accum) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] += Elem<T>::to_f(o.co[e / CE].e[e % CE]);
        }
        Chunk<T> st[2];
        round_store(o.off, v, st);
        if (want_stats) {
          if (!bn_bwd) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float vs = Elem<T>::to_f(st[e / CE].e[e % CE]); s0[e] += vs; s1[e] += vs * vs; }
          } else {
#pragma unroll
            for (int e4 = 0; e4 < 16; e4 += 4) {
              const float4 mu = *reinterpret_cast<const float4*>(lm + e4);
              const float mean[4] = {mu.x, mu.y, mu.z, mu.w};
              float sc4[4] = {0.f, 0.f, 0.f, 0.f}, sh4[4] = {0.f, 0.f, 0.f, 0.f};
              if (C_MASKX) {
                const float4 a4 = *reinterpret_cast<const float4*>(lm + 128 + e4), b4 = *reinterpret_cast<const float4*>(lm + 192 + e4);
                sc4[0] = a4.x; sc4[1] = a4.y; sc4[2] = a4.z; sc4[3] = a4.w; sh4[0] = b4.x; sh4[1] = b4.y; sh4[2] = b4.z; sh4[3] = b4.w;
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int e = e4 + u;
                float g = Elem<T>::to_f(st[e / CE].e[e % CE]) * a.gscale;
                const float xv = Elem<T>::to_f(o.cx[e / CE].e[e % CE]);
                if (has_mask && !(Elem<T>::to_f(o.cm[e / CE].e[e % CE]) > 0.f)) g = 0.f;
                if (C_MASKX && !(fmaf(xv, sc4[u], sh4[u]) > 0.f)) g = 0.f;      // the test of bn_bwd's recomputed mask (bn.hip use_mask == 2)
                s0[e] += g; s1[e] += g * (xv - mean[u]);
              }
            }
          }
        }
      }
    }
  }
  if (want_stats) {                              // one partial row per RN_CONV_STATS_ROWS = 128 output rows
    static_assert(16 * RT == RN_CONV_STATS_ROWS || 32 * RT == RN_CONV_STATS_ROWS, "a wave's block is one partial-sum row, or half of one");
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0[e] = row_sum16(s0[e]); s1[e] = row_sum16(s1[e]); }
    float istd[16];
    if (MODE != EP8_PLAIN && bn_bwd) {           // parked at the top of the epilogue; read before the corner is reused below
#pragma unroll
      for (int e4 = 0; e4 < 16; e4 += 4) {
        const float4 iv = *reinterpret_cast<const float4*>(lds_mean + 64 + 16 * lq + e4);
        istd[e4] = iv.x; istd[e4 + 1] = iv.y; istd[e4 + 2] = iv.z; istd[e4 + 3] = iv.w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (32 * RT == RN_CONV_STATS_ROWS) {
      // 64-row waves (BN = 128): the wave of the row's upper half hands its sums to its partner (the wave WN below it) through its LDS corner; every
      // wave of the workgroup is here (uniform control flow), so a workgroup barrier orders the exchange; it waits for LDS traffic only
      if (upper && l16 == 0) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          *reinterpret_cast<float4*>(lds_mean + 32 * lq + e) = make_float4(s0[e], s0[e + 1], s0[e + 2], s0[e + 3]);
          *reinterpret_cast<float4*>(lds_mean + 32 * lq + 16 + e) = make_float4(s1[e], s1[e + 1], s1[e + 2], s1[e + 3]);
        }
      }
      lds_barrier();
      if (!upper) {
        const float* other = lds_mean + pair_floats + 32 * lq;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s0[e] += other[e]; s1[e] += other[16 + e]; }
      }
      lds_barrier();                             // the corner is free again (the next tile's means)
    }
    if (MODE != EP8_PLAIN && bn_bwd) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s1[e] *= istd[e];
    }
    if (l16 == 0 && !upper && mw < a.M) {
      float* out = a.stats + ((size_t)(a.tile_base + mw / RN_CONV_STATS_ROWS) * 2) * a.Kd + kc;
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        *reinterpret_cast<float4*>(out + e) = make_float4(s0[e], s0[e + 1], s0[e + 2], s0[e + 3]);
        *reinterpret_cast<float4*>(out + a.Kd + e) = make_float4(s1[e], s1[e + 1], s1[e + 2], s1[e + 3]);
      }
    }
  }
}

// the launch's epilogue specialisation (host and device agree: the launcher's rule reads it too)
__host__ __device__ inline int ep8_mode(const IgemmArgs& a) {
  const bool bnb = a.stats != nullptr && a.bn_x != nullptr && a.bn_mask != nullptr;
  if (a.bias && a.ds == 1 && a.res.mode == RN_RES_NONE && !a.accum && !a.bn_x) return EP8_BIAS;      // a biased forward convolution (+ statistics): the stems
  if (a.ds == 2 && a.res.mode == RN_RES_NONE && !a.bias && bnb)        // a parity class of a stride-2 data gradient with the BatchNorm-backward sums
    return EP8_BNB | EP8_STRIDED | (a.accum ? EP8_ACC : (a.mask_from_x ? EP8_MASKX : 0));
  const bool dense = (a.ds == 1) && (a.res.mode == RN_RES_NONE || a.res.mode == RN_RES_SAME) && !a.bias;
  if (!dense) return EP8_GEN;
  const bool res = a.res.mode == RN_RES_SAME, acc = a.accum != 0;
  if (a.stats != nullptr && a.bn_x != nullptr) {
    if (!a.bn_mask || (res && acc)) return EP8_GEN;
    return EP8_BNB | (res ? EP8_RES : 0) | (acc ? EP8_ACC : 0) | (!res && !acc && a.mask_from_x ? EP8_MASKX : 0);
  }
  if (acc) return EP8_GEN;
  return res ? EP8_RES : EP8_PLAIN;
}

// TC ("tap chunks"): the source has ONE 16-byte chunk per pixel (the 7x7 stem: 3 image channels zero-padded to 8), so a K tile is eight TAPS x one chunk
// instead of one tap x eight chunks: the 16-byte column a lane copies selects the tap (tap = 8 g + chunk), walked per lane
template <typename T, int BN, int EPM, bool TC = false>
__global__ __launch_bounds__(512, 2) void igemm8_kernel(const IgemmArgs a) {
  constexpr int BM = 256, ES = 2;
  constexpr int WM = BN == 256 ? 2 : 4, WN = 8 / WM;        // 2 x 4 waves of 128 x 64 (BN = 256), 4 x 2 waves of 64 x 64 (BN = 128)
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int RT = WTM / 16, CT = WTN / 16;               // 16 x 16 accumulator tiles of a wave: pixel tiles x channel tiles
  constexpr int QR = RT / 2, QC = CT / 2;                   // ... of a quadrant
  constexpr int HR = WTM / 2, HC = WTN / 2;                 // rows / columns of a wave's quadrant
  constexpr int AI = 2, BI = BN / 128;                      // DMA instructions per wave and half-tile (A: 128 rows, B: BN/2 rows)
  constexpr int STG = 4096;                                 // uint4 per stage: 64 KiB (A0 | A1 | B0 | B1), power of two: the stage toggles by XOR
  constexpr int A_H = 1024, B_0 = 2048, B_H = BN * 4;       // uint4 offsets: second A half, B, second B half
  static_assert(sizeof(T) == ES && (BN == 256 || BN == 128) && CT == 4 && B_0 + 2 * B_H <= STG, "tile");
  __shared__ uint4 smem[2 * STG + TAP_INTS / 4 + 8 * 64 + 1];     // stages | tap tables | 256 floats per wave (epilogue8) | one ticket word
  int* taps = reinterpret_cast<int*>(&smem[2 * STG]);

  // (no blanket preload of the kernel arguments: this kernel is persistent and its epilogues read many of them -- held in SGPRs across the whole
  // kernel they spill into VGPR lanes, which the K loop cannot afford; only what the K loop reads is pinned)
  Walk8K wkk;
  wkk.cpc = TC ? 0x7FFFFFFF : a.w8_cpc; wkk.ntw = a.ntw;          // TC: cc is the K-tile index itself
  wkk.dsj = (unsigned)(a.w8_sj - a.w8_cpc * 128); wkk.dwj = (unsigned)(a.w8_wj - a.w8_cpc * 128);
  wkk.dsi = (unsigned)(a.w8_si - a.ntw * a.w8_sj); wkk.dwi = (unsigned)(a.w8_wi - a.ntw * a.w8_wj);
  asm volatile("" : "+s"(wkk.cpc), "+s"(wkk.ntw), "+s"(wkk.dsj), "+s"(wkk.dwj), "+s"(wkk.dsi), "+s"(wkk.dwi));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nmt = (a.M + BM - 1) / BM, nnt = a.Kd / BN, ntiles = nmt * nnt;
  const int pq = a.Pc * a.Qc;
  const unsigned pixb = a.w8_pixb ? (unsigned)a.w8_pixb : (unsigned)(a.Cs * ES);         // bytes of one source pixel (row-segment form: Cs is the K tile's 64, a pixel is less)
  const size_t img_bytes = (size_t)a.Hs * a.Ws * pixb;
  const v4i32 rb_desc = make_desc(a.wt, (size_t)a.Kd * a.wrs * a.Cs * ES);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);
  const int nk = a.nk;
  float* lds_mean = reinterpret_cast<float*>(&smem[2 * STG + TAP_INTS / 4 + wave * 64]);     // 256 floats per wave (epilogue8: mean, 1 / std, scale, shift of the lane rows' channels)
  // the launch's epilogue specialisation (wave-uniform, from the kernel arguments)

  fill_tap_tables<ES>(a, taps);
  __syncthreads();
  TapGrid grid;
  load_tap_grid(a, taps, grid);
#pragma unroll
  for (int i = 0; i < MAX_GRID; ++i) {                      // wave-uniform: keep the tap grid in scalar registers for the kernel's lifetime
    grid.dh[i] = __builtin_amdgcn_readfirstlane(grid.dh[i]);
    grid.dw[i] = __builtin_amdgcn_readfirstlane(grid.dw[i]);
  }

  // ---- fragment addresses (uint4 units inside a stage): 16 consecutive LDS rows at chunk (4 ks + lq) ^ ((row >> 1) & 7); the row bases are multiples of 16
  const int wm = wave / WN, wn = wave % WN;
  const int l16 = lane & 15, lq = lane >> 4;
  const int sw = (l16 >> 1) & 7;
  const int fa0 = (wm * HR + l16) * 8 + (lq ^ sw), fa1 = (wm * HR + l16) * 8 + ((4 + lq) ^ sw);
  const int fb0 = B_0 + (wn * HC + l16) * 8 + (lq ^ sw), fb1 = B_0 + (wn * HC + l16) * 8 + ((4 + lq) ^ sw);
  const int lrow = lane >> 3, p = lane & 7;

  // ---- per-tile DMA roles.  One instruction = 8 LDS rows x 128 B; instruction q of a half-tile covers its LDS rows 8q .. 8q+7; lane = (row lrow,
  // physical chunk p), which holds logical chunk p ^ ((row >> 1) & 7) (the XOR goes on the SOURCE address, the destination is lane-linear).
  // A half h, LDS row r  <->  tile row (r / HR) * WTM + h * HR + r % HR;  B half h, LDS row r  <->  tile column (r / HC) * WTN + h * HC + r % HC.
  unsigned abase[2 * AI];
  unsigned amask[AI];                 // two 16-bit tap masks per register: half 0 low, half 1 high
  unsigned ahw[TC ? 2 * AI : 1];      // TC: (hb << 16) | wb of the lane's four pixels
  const int cj0 = p ^ ((lrow >> 1) & 7);      // the logical chunk of this lane's DMA pieces: cj0 for piece 0, cj0 ^ 4 for piece 1 (row 8 q + lrow: (row >> 1) & 7 = 4 q + lrow / 2)
  unsigned bbase[BI];
  int ra_w0 = 0, ra_w1 = 0, ra_w2 = 0;   // the source descriptor of the current tile (based at its first image), as three wave-uniform words
  const unsigned bhalf = (unsigned)((size_t)HC * a.wrs * a.Cs * ES);                      // B half 1 = the columns HC further
  const unsigned rowb = (unsigned)(a.Cs * ES);                       // bytes of one tap's channels in the weights
  auto tile_roles = [&](int m0, int n0) {
    int n_first = (int)__umulhi((unsigned)m0, a.magic_pq);           // m0 / pq (magic multiply + one correction, as decode_row)
    if (m0 - n_first * pq >= pq) ++n_first;
    {
      const v4i32 d = make_desc(reinterpret_cast<const char*>(a.src) + (size_t)n_first * img_bytes, (size_t)(a.N - n_first) * img_bytes);
      ra_w0 = d[0]; ra_w1 = d[1]; ra_w2 = d[2];
    }
#pragma unroll
    for (int jj = 0; jj < AI; ++jj) amask[jj] = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int jj = 0; jj < AI; ++jj) {
        const int r = 8 * (wave * AI + jj) + lrow;
        const int c = p ^ ((r >> 1) & 7);
        const int m = m0 + (r / HR) * WTM + h * HR + (r % HR);
        unsigned base = 0, mk = 0;
        if constexpr (TC) {                                          // base = the pixel itself, mask register = (hb << 16) | wb for the per-tap range checks
          mk = 0x7FFF0000u;                                          // a row beyond M: every tap out of range
          if (m < a.M) {
            int n, pp, q;
            decode_row(a, m, pq, n, pp, q);
            const int hb = pp * a.ss, wb = q * a.ss;
            base = (unsigned)(((n - n_first) * a.Hs + hb) * a.Ws + wb) * pixb;
            mk = ((unsigned)hb << 16) | (unsigned)wb;
          }
          abase[h * AI + jj] = base;
          ahw[h * AI + jj] = mk;
          continue;
        }
        if (m < a.M) {                                               // offsets fit 32 bits (launcher)
          if (a.dense_src) {                                         // 1x1, stride 1: row m reads pixel m, no padding
            base = (unsigned)(m - n_first * pq) * pixb + (unsigned)(c * 16);
            mk = 1u;
          } else {
            int n, pp, q;
            decode_row(a, m, pq, n, pp, q);
            const int hb = pp * a.ss, wb = q * a.ss;
            base = (unsigned)(((n - n_first) * a.Hs + hb) * a.Ws + wb) * pixb + (unsigned)(c * 16);
            mk = (unsigned)tap_mask(a, grid, hb, wb) & 0xFFFFu;
          }
        }
        abase[h * AI + jj] = base;
        amask[jj] |= mk << (16 * h);
      }
#pragma unroll
    for (int jj = 0; jj < BI; ++jj) {
      const int r = 8 * (wave * BI + jj) + lrow;
      const int c = p ^ ((r >> 1) & 7);
      const int k = n0 + (r / HC) * WTN + (r % HC);
      bbase[jj] = (unsigned)k * (unsigned)a.wrs * rowb + (TC ? 0u : (unsigned)(c * 16));          // Kd % BN == 0 (launcher): every column exists
    }
  };
  auto issue_a = [&](int h, unsigned stage_lds, const Walk8& w) {
    v4i32 ra_desc;                                          // rebuilt from readfirstlane'd words: provably wave-uniform for the "s" operand
    ra_desc[0] = __builtin_amdgcn_readfirstlane(ra_w0); ra_desc[1] = __builtin_amdgcn_readfirstlane(ra_w1);
    ra_desc[2] = __builtin_amdgcn_readfirstlane(ra_w2); ra_desc[3] = 0x00020000;
    const unsigned keep = m0_save();
#pragma unroll
    for (int jj = 0; jj < AI; ++jj) {
      if constexpr (TC) {                                   // K tile g = w.cc: this piece's tap = 8 g + chunk -> (i, j) of the tap grid -> shifted pixel, range-checked
        const int t = 8 * w.cc + (cj0 ^ (4 * jj));
        const int ti = (t * (int)a.w8_magic_ntw) >> 16, tj = t - ti * a.ntw;
        const int dh = a.dh[0] + ti, dw = a.dw[0] + tj;
        const unsigned hw = ahw[h * AI + jj];
        const bool ok = t < a.nt && (unsigned)((int)(hw >> 16) + dh) < (unsigned)a.Hs && (unsigned)((int)(hw & 0xFFFFu) + dw) < (unsigned)a.Ws;
        dma16(ra_desc, ok ? abase[h * AI + jj] + (unsigned)((dh * a.Ws + dw) * 16) : OOB, stage_lds + (unsigned)((h * A_H) * 16 + (wave * AI + jj) * 1024));
      } else {
        const bool ok = (amask[jj] >> (w.t + 16 * h)) & 1u;
        dma16(ra_desc, ok ? abase[h * AI + jj] + w.src : OOB, stage_lds + (unsigned)((h * A_H) * 16 + (wave * AI + jj) * 1024));
      }
    }
    m0_restore(keep);
  };
  auto issue_b = [&](int h, unsigned stage_lds, const Walk8& w) {
    const unsigned keep = m0_save();
#pragma unroll
    for (int jj = 0; jj < BI; ++jj) {
      if constexpr (TC) {                                   // weights [K][taps][8 channels]: tap 8 g + chunk of this row, zeros beyond the last tap
        const int t = 8 * w.cc + (cj0 ^ (4 * (jj & 1)));
        dma16(rb_desc, t < a.nt ? bbase[jj] + (h ? bhalf : 0u) + (unsigned)(t * 16) : OOB, stage_lds + (unsigned)((B_0 + h * B_H) * 16 + (wave * BI + jj) * 1024));
      } else {
        dma16(rb_desc, bbase[jj] + (h ? bhalf : 0u) + w.wt, stage_lds + (unsigned)((B_0 + h * B_H) * 16 + (wave * BI + jj) * 1024));
      }
    }
    m0_restore(keep);
  };

  f32x4 acc[RT][CT];
  uint4 af[QR][2], b0[QC][2], b1[QC][2];
  // one K tile = four phases.  MODE 2: K tiles kt+1 and kt+2 exist (steady state); 1: kt+1 is the last; 0: kt is the last
  auto ktile = [&](auto mode_tag, int sx, const Walk8& prev, const Walk8& cur) {
    constexpr int MODE = decltype(mode_tag)::value;
    const uint4* S = &smem[sx];                           // this K tile's stage
    const unsigned mine = lds0 + (unsigned)(sx * 16), other = lds0 + (unsigned)((sx ^ STG) * 16);
    // ---- phase 1: B0, A0 -> quadrant (0, 0); stage A1 of K tile kt+1 ----
#pragma unroll
    for (int j = 0; j < QC; ++j) { b0[j][0] = S[fb0 + j * 128]; b0[j][1] = S[fb1 + j * 128]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < QR; ++i) { af[i][0] = S[fa0 + i * 128]; af[i][1] = S[fa1 + i * 128]; }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE >= 1) issue_a(1, other, prev);
    wait_lgkm<2 * QR>();                                  // the B0 reads (issued first) are back: B0 may be restaged in phase 2
    raw_barrier();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(b0[j][ks], af[i][ks], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
    // ---- phase 2: B1 -> quadrant (0, 1); stage B0 of K tile kt+2 ----
#pragma unroll
    for (int j = 0; j < QC; ++j) { b1[j][0] = S[fb0 + B_H + j * 128]; b1[j][1] = S[fb1 + B_H + j * 128]; }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 2) issue_b(0, mine, cur);
    raw_barrier();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(b1[j][ks], af[i][ks], acc[i][QC + j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
    // ---- phase 3: A1 -> quadrant (1, 1); stage A0 of K tile kt+2 ----
#pragma unroll
    for (int i = 0; i < QR; ++i) { af[i][0] = S[fa0 + A_H + i * 128]; af[i][1] = S[fa1 + A_H + i * 128]; }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 2) issue_a(0, mine, cur);
    raw_barrier();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(b1[j][ks], af[i][ks], acc[QR + i][QC + j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
    // ---- phase 4: quadrant (1, 0) from registers; stage B1 of K tile kt+2; K tile kt+1 has landed behind this wait ----
    if constexpr (MODE == 2) { issue_b(1, mine, cur); wait_vmcnt<AI + 2 * BI>(); }
    else if constexpr (MODE == 1) wait_vmcnt<0>();
    raw_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < QR; ++i)
#pragma unroll
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(b0[j][ks], af[i][ks], acc[QR + i][j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
  };

  // ---- persistent walk.  Work = whole tiles (data-parallel part) and then a range of K-TILE UNITS (stream-K part, Osama et al. 2023):
  //  * tiles [0, dp_tiles): workgroup b takes tiles b, b + G, ... (dp_tiles is a multiple of G).  Workgroups b and b + 8 share an XCD (round-robin
  //    dispatch); the tile order is column tiles fastest and each XCD class takes a contiguous eighth of it, its workgroups interleaved -- the tiles
  //    in flight on one XCD at one time are neighbours and read the same input rows / weights out of that XCD's L2;
  //  * tiles [dp_tiles, ntiles): their sk_tiles * nk K tiles are cut into G equal contiguous ranges, one per workgroup, so the last partial round of a
  //    grid that is no multiple of the CU count costs ~(1 + fraction) / 2 rounds instead of one (196 tiles on 256 CUs: 77 % -> ~100 % occupancy).
  //    A tile cut between workgroups is summed through memory: every part writes its fp32 accumulators to its slot, releases, and draws a
  //    ticket (agent-scope atomic); whoever draws the LAST ticket acquires, adds the parts in slice order (fixed order: bitwise reproducible, own
  //    part re-read like the others) and runs the epilogue.  Nobody waits for anybody: no spin, no residency assumption.
  const int G = gridDim.x;                                  // a multiple of 8, or ntiles (launcher)
  const int dp_tiles = a.w8_dp_tiles;
  const unsigned U = (unsigned)(ntiles - dp_tiles) * (unsigned)nk;             // stream-K units; U * G < 2^31 (launcher)
  const int wv = (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);    // XCD-contiguous index: neighbours in the unit order share an L2
  const unsigned su0 = U ? (unsigned)wv * U / (unsigned)G : 0u, su1 = U ? (unsigned)(wv + 1) * U / (unsigned)G : 0u;
  unsigned su = su0;
  int it = 0;
  int pm0 = -1, pn0 = 0, pt_sk = -1, pseg = 0;              // the segment whose accumulators are still in registers (pt_sk >= 0: a partial stream-K tile)
  for (;;) {
    const int vb = it * G + blockIdx.x;
    int tile = 0, kb = 0, ke = nk, t_sk = -1, seg = 0;
    bool more = true;                                       // wave-uniform
    if (vb < dp_tiles) {
      tile = vb;
      if (a.xcd_remap) {
        const int xcd = vb & 7, q = dp_tiles >> 3, r = dp_tiles & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
      }
      ++it;
    } else if (su < su1) {
      t_sk = (int)(su / (unsigned)nk);
      kb = (int)(su - (unsigned)t_sk * (unsigned)nk);
      ke = min(nk, kb + (int)(su1 - su));
      seg = su == su0 ? 0 : 1;
      tile = dp_tiles + t_sk;
      su += (unsigned)(ke - kb);
      if (kb == 0 && ke == nk) t_sk = -1;                   // the whole tile is this workgroup's: nothing to sum
    } else {
      more = false;
    }
    const int nseg = ke - kb;                               // K tiles of this segment
    int m0 = 0, n0 = 0;
    Walk8 wk;
    const unsigned long long* stp = it == 2 && vb < dp_tiles ? a.stamps : nullptr;      // diagnostic (rn_set_stamp_buffer): the workgroup's SECOND tile
    stamp(stp, 0);
    if (more) {
      int mt = (int)__umulhi((unsigned)tile, (unsigned)a.w8_magic_nnt);      // tile / nnt
      if (tile - mt * nnt >= nnt) ++mt;
      m0 = mt * BM; n0 = (tile - mt * nnt) * BN;
      tile_roles(m0, n0);
      stamp(stp, 1);
      // prologue: the segment's first K tile whole, then B0 / A0 / B1 of its second (the state every K tile's phase 1 starts from).  Every wave has
      // left the previous segment's K loop (its closing rendezvous), so both stages are free.
      if constexpr (TC) { wk.i = wk.j = wk.t = 0; wk.src = wk.wt = 0u; wk.cc = kb; } else walk8_seek(a, wk, kb);
      issue_b(0, lds0, wk); issue_a(0, lds0, wk); issue_b(1, lds0, wk); issue_a(1, lds0, wk);
      walk8_advance(wkk, wk);
      if (nseg > 1) { issue_b(0, lds0 + STG * 16, wk); issue_a(0, lds0 + STG * 16, wk); issue_b(1, lds0 + STG * 16, wk); }
    }
    stamp(stp, 2);
    // the previous segment leaves the registers while this one's first K tiles are in flight (ONE call site: the epilogue is large)
    bool stores_behind = false;
    if (pm0 >= 0) {
      bool finish = true;
      if (pt_sk >= 0) {
        // ---- a part of a cut tile: publish it, draw a ticket; the last arriver sums the parts ----
        const unsigned uf = (unsigned)pt_sk * (unsigned)nk;
        const int w_first = (int)(((uf + 1u) * (unsigned)G - 1u) / U), w_last = (int)(((uf + (unsigned)nk) * (unsigned)G - 1u) / U);   // owners of the tile's first / last unit
        const int parts = w_last - w_first + 1;
        f32x4* slots = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.w8_ws) + SK_CNT_BYTES);
        constexpr size_t SLOT = (size_t)RT * CT * 512;       // f32x4 per slot: register-major, thread-minor (every access 16 bytes per lane, coalesced)
        // the part goes out WRITE-THROUGH (sc1 buffer stores: the bytes leave this XCD's L2 at once), so publishing it needs no release fence -- a
        // fence here writes back the whole L2 (256 KiB of fresh lines per workgroup, 32 workgroups per XCD: tens of microseconds, measured)
        {
          const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(slots), 0, 0x7FFFFFF0, 0x00020000);
          const unsigned mine_off = (unsigned)(((size_t)(wv * 2 + pseg) * SLOT + tid) * 16);
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, acc[i][j]), wsr, (int)(mine_off + (unsigned)((i * CT + j) * 512 * 16)), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains, then the workgroup meets, then ONE lane releases and signals
        __syncthreads();
        int* cnt = reinterpret_cast<int*>(a.w8_ws) + pt_sk;
        volatile int* flagw = reinterpret_cast<volatile int*>(&smem[2 * STG + TAP_INTS / 4 + 8 * 64]);
        if (tid == 0) {
          *flagw = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        finish = *flagw == parts - 1;                        // wave-uniform: every thread reads the same LDS word
        if (finish) {
          if (tid == 0) {
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every part has arrived: the counter is clean for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __syncthreads();
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
          for (int sl = 0; sl < parts; ++sl) {               // slice order: the sum does not depend on who arrived last
            const int wq = w_first + sl;
            const int sq = (int)(((unsigned)wq * U / (unsigned)G) / (unsigned)nk) == pt_sk ? 0 : 1;       // that workgroup's first segment, or its last
            const f32x4* src = slots + (size_t)(wq * 2 + sq) * SLOT + tid;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
              for (int j = 0; j < CT; ++j) acc[i][j] += src[(size_t)(i * CT + j) * 512];
          }
        } else {
          __syncthreads();                                   // same barrier count on both paths
        }
      }
      if (finish) {
        stores_behind = pm0 + BM <= a.M;                     // a whole tile: every lane stores its 2 x RT chunks of the tile, whatever else the epilogue does
        const int mw = pm0 + wm * WTM, kw = pn0 + wn * WTN;
        // (the epilogue specialisation is a kernel template parameter: one copy per kernel; 64-row waves pair up for the statistics row)
        epilogue8<T, RT, EPM>(a, acc, mw, kw, lane, lds_mean, RT == 4 && (wm & 1), WN * 256);
      }
    }
    stamp(stp, 3);
    if (!more) break;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // vmcnt counts in issue order, and the epilogue's stores are YOUNGER than this segment's first loads: the wait names them too, or the first K tile
    // would start only once the previous tile has drained to memory.  Counted: the 2 x RT tile stores of a whole tile (a lower bound -- the sums' stores
    // are not counted, which only makes the wait stricter); a ragged tile or a stream-K part that stored nothing waits as before.
    if (nseg > 1) {
      if (stores_behind && !a.w8_drain) wait_vmcnt<AI + 2 * BI + 2 * RT>(); else wait_vmcnt<AI + 2 * BI>();
    } else {
      wait_vmcnt<0>();
    }
    raw_barrier();
    if (wave >= 4) raw_barrier();                           // the second wave group runs one barrier behind
    stamp(stp, 4);

    int sx = 0;
    Walk8 prev = wk;                                        // the segment's second K tile
    for (int kt = 0; kt + 2 < nseg; ++kt) {
      walk8_advance(wkk, wk);                               // K tile kt+2
      ktile(std::integral_constant<int, 2>{}, sx, prev, wk);
      prev = wk;
      sx ^= STG;
    }
    if (nseg > 1) {
      ktile(std::integral_constant<int, 1>{}, sx, prev, wk);
      sx ^= STG;
    }
    ktile(std::integral_constant<int, 0>{}, sx, prev, wk);
    if (wave < 4) raw_barrier();                            // the first group waits for the second: every wave has executed the same barriers
    stamp(stp, 5);
    pm0 = m0; pn0 = n0; pt_sk = t_sk; pseg = seg;
  }
}

// the taps must form a separable arithmetic progression: dh[i * ntw + j] = dh0 + i * ddh, dw = dw0 + j * ddw, widx = widx0 + i * dwi + j * dwj
bool fill_walk8(IgemmArgs& a) {
  if (a.nt < 1 || a.nt > 16 || a.nth * a.ntw != a.nt) return false;
  const int ddh = a.nth > 1 ? a.dh[a.ntw] - a.dh[0] : 0, ddw = a.ntw > 1 ? a.dw[1] - a.dw[0] : 0;
  const int dwi = a.nth > 1 ? a.widx[a.ntw] - a.widx[0] : 0, dwj = a.ntw > 1 ? a.widx[1] - a.widx[0] : 0;
  for (int i = 0; i < a.nth; ++i)
    for (int j = 0; j < a.ntw; ++j) {
      const int t = i * a.ntw + j;
      if (a.dh[t] != a.dh[0] + i * ddh || a.dw[t] != a.dw[0] + j * ddw || a.widx[t] != a.widx[0] + i * dwi + j * dwj) return false;
    }
  const long row = (long)a.Cs * 2, prow = a.w8_pixb ? (long)a.w8_pixb : row;     // a tap's bytes in the weights / a pixel's in the source
  a.w8_src0 = (int)(((long)a.dh[0] * a.Ws + a.dw[0]) * prow);
  a.w8_si = (int)((long)ddh * a.Ws * prow);
  a.w8_sj = (int)((long)ddw * prow);
  a.w8_wt0 = (int)((long)a.widx[0] * row);
  a.w8_wi = (int)((long)dwi * row);
  a.w8_wj = (int)((long)dwj * row);
  a.w8_cpc = a.Cs / 64;
  return true;
}

template <typename T, int BN> int launch8(IgemmArgs& a, hipStream_t s) {
  if (!fill_walk8(a)) return -1;
  a.nk = a.nt * a.w8_cpc;
  { const unsigned nnt = (unsigned)(a.Kd / BN); a.w8_magic_nnt = nnt <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / nnt); }
  static const char* const EPN[] = {"plain", "res", "?", "?", "bnb", "bnb+res", "bnb+acc", "?", "gen"};
  const int epm = ep8_mode(a);
  rn_note_kernel("igemm8<256x%d:%s%s%s%s>", BN, epm == EP8_BIAS ? "bias" : EPN[epm & 15], (epm & EP8_MASKX) ? "/xmask" : "", (epm & EP8_STRIDED) ? "/s2" : "",
                 a.w8_pixb ? "/rows" : "");
  if (rn_dry_run()) return 0;
  const int ntiles = cdiv(a.M, 256) * (a.Kd / BN);
  // stream-K (tiles cut into K-tile unit ranges, cut tiles summed through the workspace) only where whole tiles cannot occupy the chip: grids of at most
  // half the CUs.  On larger grids the fp32 parts (256 KiB written + re-read per cut) cost what the saved tail round buys -- measured on the WRN-50-2
  // shapes at batch 256: 3x3 512@14 237 vs 230 us, 1024@7 231 vs 219, 1x1 2048 -> 512 164 vs 111.  rn_set_variant 1 << 28: never;
  // 1 << 31: wherever the grid is no multiple of the CU count (tests of the mixed data-parallel + stream-K walk).
  int grid = ntiles < SK_GRID ? ntiles : SK_GRID;
  a.w8_dp_tiles = ntiles;
  a.w8_drain = (g_rn_variant & (1 << 26)) ? 1 : 0;
  const bool ws_here = sk_ws_here();
  a.w8_ws = ws_here ? g_sk_ws : nullptr;
  const int rem = ntiles % SK_GRID, full = ntiles / SK_GRID;
  const bool sk_forced = (g_rn_variant & (1u << 31)) != 0;
  if (rem != 0 && ws_here && g_sk_ws_bytes >= SK_CNT_BYTES + 2 * SK_GRID * SK_SLOT_BYTES && !(g_rn_variant & (1 << 28)) && (sk_forced || 2 * ntiles <= SK_GRID)) {
    const int sk_tiles = full >= 1 ? SK_GRID + rem : rem;
    const long units = (long)sk_tiles * a.nk;
    if (units >= 4L * SK_GRID && units * SK_GRID < (1L << 31) && sk_tiles <= (int)(SK_CNT_BYTES / 4)) {
      a.w8_dp_tiles = ntiles - sk_tiles;
      grid = SK_GRID;
    }
  }
  switch (ep8_mode(a)) {
    case EP8_PLAIN: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_PLAIN>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_RES: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_RES>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_BIAS: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_BIAS>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_BNB: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_BNB>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_BNB | EP8_MASKX: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_BNB | EP8_MASKX>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_BNB | EP8_MASKX | EP8_STRIDED: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_BNB | EP8_MASKX | EP8_STRIDED>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_BNB | EP8_RES: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_BNB | EP8_RES>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_BNB | EP8_ACC: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_BNB | EP8_ACC>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_BNB | EP8_STRIDED: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_BNB | EP8_STRIDED>), dim3(grid), dim3(512), 0, s, a); break;
    case EP8_BNB | EP8_ACC | EP8_STRIDED: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_BNB | EP8_ACC | EP8_STRIDED>), dim3(grid), dim3(512), 0, s, a); break;
    default: hipLaunchKernelGGL((igemm8_kernel<T, BN, EP8_GEN>), dim3(grid), dim3(512), 0, s, a); break;
  }
  RN_CHECK_LAUNCH("igemm8");
  return 0;
}

}  // namespace

extern "C" size_t rn_conv_workspace_bytes(void) { return SK_CNT_BYTES + 2 * SK_GRID * SK_SLOT_BYTES; }
// the stream-K workspace of the eight-phase convolution kernels: device memory of rn_conv_workspace_bytes() bytes whose first 4 KiB are ZERO (the tile
// tickets; the kernels leave them zero), owned by the caller, used by one stream at a time (the plan executor launches every convolution forward /
// data gradient on its launch stream).  NULL (the default): no stream-K, whole tiles only.
extern "C" int rn_set_conv_workspace(void* p, size_t bytes) {
  g_sk_ws = p; g_sk_ws_bytes = p ? bytes : 0;
  g_sk_ws_dev = -1;
  if (p) {                                               // (one process-wide pointer -- at least never hand it to a launch on another device)
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess) g_sk_ws_dev = at.device;
    else { (void)hipGetLastError(); (void)hipGetDevice(&g_sk_ws_dev); }
  }
  return 0;
}
// the same workspace for the split reductions of conv_igemm8r.hip (tickets in the first 4 KiB, one slot of SK_SLOT_BYTES per work item): NULL when it is
// not set or smaller than rn_conv_workspace_bytes()
void* rn_sk_workspace(size_t* slot_bytes) {
  if (slot_bytes) *slot_bytes = SK_SLOT_BYTES;
  return (sk_ws_here() && g_sk_ws_bytes >= SK_CNT_BYTES + 2 * SK_GRID * SK_SLOT_BYTES) ? g_sk_ws : nullptr;
}

// geometry the eight-phase kernel covers: 16-bit elements, channel count a multiple of 64 (a K tile never straddles a tap), output channels a
// multiple of the column tile, 1..16 taps in a separable progression, 32-bit tile offsets; the grid rule (enough tiles for the chip) is the caller's
// the 7x7 (or any <= 64-tap) stem: one 16-byte chunk per source pixel, forward convolution with unit tap steps, bias and / or statistics, nothing else fused
static bool stem_capable(const IgemmArgs& a) {
  if (a.Cs != 8 || a.Kd % 256 || a.nt < 1 || a.nt > 64 || a.nth * a.ntw != a.nt || a.ntw < 1) return false;
  if (a.ds != 1 || a.res.mode != RN_RES_NONE || a.accum || a.bn_x) return false;
  for (int i = 0; i < a.nth; ++i)
    for (int j = 0; j < a.ntw; ++j) {
      const int t = i * a.ntw + j;
      if (a.dh[t] != a.dh[0] + i || a.dw[t] != a.dw[0] + j || a.widx[t] != t) return false;
    }
  return true;
}

// row-segment form: fewer than 64 channels per pixel, but the ntw taps of a kernel row are consecutive pixels and ntw * Cs = 64 -- a kernel row IS one K tile
// of 128 contiguous bytes per output pixel (the space-to-depth ImageNet stem: 16 channels, 4 x 4 taps).  Forward only, every tap in range for every output
// (a VALID convolution: the copy is not checked per pixel), weights [K][taps][Cs] = [K][nth][64].
static bool rowseg_capable(const IgemmArgs& a) {
  if (a.Cs <= 0 || a.Cs % 8 || a.Cs * a.ntw != 64 || a.nth < 1 || a.nth * a.ntw != a.nt || a.nt > 16 || a.Kd % 128) return false;
  if (a.ds != 1 || a.res.mode != RN_RES_NONE || a.accum || a.bn_x || a.wrs != a.nt) return false;
  for (int i = 0; i < a.nth; ++i)
    for (int j = 0; j < a.ntw; ++j) {
      const int t = i * a.ntw + j;
      if (a.dh[t] != a.dh[i * a.ntw] || a.dw[t] != a.dw[i * a.ntw] + j || a.widx[t] != t) return false;
      if (a.dh[t] < 0 || a.dw[t] < 0 || (a.Pc - 1) * a.ss + a.dh[t] >= a.Hs || (a.Qc - 1) * a.ss + a.dw[t] >= a.Ws) return false;
    }
  return true;
}
static void rowseg_rewrite(IgemmArgs& a) {               // nth taps of 64 'channels'; the source keeps its own pixel size
  a.w8_pixb = a.Cs * 2;
  for (int i = 0; i < a.nth; ++i) { a.dh[i] = a.dh[i * a.ntw]; a.dw[i] = a.dw[i * a.ntw]; a.widx[i] = i; }
  a.nt = a.nth; a.ntw = 1; a.wrs = a.nth; a.Cs = 64;
}

// 1: the geometry is one the eight-phase kernel covers with a specialised (spill-free) epilogue; 0: not covered, or only by its general epilogue
int rn_igemm8_fast(const IgemmArgs& a) {
  if (a.Cs == 8) return stem_capable(a) ? 1 : 0;
  if (a.Cs % 64) return (rowseg_capable(a) && ep8_mode(a) != EP8_GEN) ? 1 : 0;
  return (a.Cs % 64 == 0 && a.Kd % 128 == 0 && ep8_mode(a) != EP8_GEN) ? 1 : 0;
}

template <typename T> static int launch8_stem(IgemmArgs& a, hipStream_t s) {
  a.nk = (a.nt + 7) / 8;
  a.w8_cpc = 1;
  a.w8_magic_ntw = (unsigned)((65536 + a.ntw - 1) / a.ntw);
  { const unsigned nnt = (unsigned)(a.Kd / 256); a.w8_magic_nnt = nnt <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / nnt); }
  rn_note_kernel("igemm8<256x256:stem%s>", a.bias ? "+bias" : "");
  if (rn_dry_run()) return 0;
  const int ntiles = cdiv(a.M, 256) * (a.Kd / 256);
  const int grid = ntiles < SK_GRID ? ntiles : SK_GRID;
  a.w8_dp_tiles = ntiles;
  a.w8_drain = (g_rn_variant & (1 << 26)) ? 1 : 0;
  a.w8_ws = nullptr;
  if (a.bias) hipLaunchKernelGGL((igemm8_kernel<T, 256, EP8_BIAS, true>), dim3(grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((igemm8_kernel<T, 256, EP8_PLAIN, true>), dim3(grid), dim3(512), 0, s, a);
  RN_CHECK_LAUNCH("igemm8 stem");
  return 0;
}

int rn_launch_igemm8(const IgemmArgs& a_in, int dtype, hipStream_t s) {
  if (dtype != RN_BF16 && dtype != RN_F16) return -1;
  if (a_in.M <= 0) return -1;
  const size_t img_bytes = (size_t)a_in.Hs * a_in.Ws * a_in.Cs * 2;
  const long pq = (long)a_in.Pc * a_in.Qc;
  if ((256 / pq + 3) * (double)img_bytes >= 4.0e9) return -1;                          // per-tile source offsets are 32-bit (descriptor based at the tile's first image)
  if (((double)a_in.Kd + 256.0) * a_in.wrs * a_in.Cs * 2 >= 4.0e9) return -1;
  IgemmArgs a = a_in;
  a.w8_pixb = 0;
  if (a.Cs == 8) {
    if (!stem_capable(a)) return -1;
    return dtype == RN_BF16 ? launch8_stem<bf16_t>(a, s) : launch8_stem<f16_t>(a, s);
  }
  if (a.Cs % 64 && rowseg_capable(a)) rowseg_rewrite(a);
  if (a.Cs % 64 || a.Kd % 128) return -1;
  if (a.Kd % 256 == 0) return dtype == RN_BF16 ? launch8<bf16_t, 256>(a, s) : launch8<f16_t, 256>(a, s);
  return dtype == RN_BF16 ? launch8<bf16_t, 128>(a, s) : launch8<f16_t, 128>(a, s);           // column tiles of 128: K = 128, 384, 640, ...
}
