// Weight gradient for layers whose channel counts are multiples of 160 -- the WRN-28-10 family (residual_block.py:34-47 with 160 n channels,
// models_dir/wrn-28-10-dropout_cifar10/config.yaml:15) -- on the deep-pipelined schedule of conv_wgrad8.hip (gfx950, 16-bit element types):
//
//   dw[k][t][c] = sum_{m=(n,p,q)} dy[m][k] * x[n, p*stride+dh[t], q*stride+dw[t], c]          (fp32, KRSC)
//
// Output tile = 320 input-channel rows x 160 output channels: the rows are TWO 160-channel segments of the (tap, channel slice) list -- two taps of a
// 160-channel layer, the two slices of one tap of a 320-channel layer -- so the dy tile of a K tile is read once for both (the im2col column trick of
// the stems); 8 waves of 80 x 80 (5 x 5 v_mfma_f32_16x16x32 tiles = 100 accumulator registers), one persistent workgroup per CU.
// LDS image: one 1 KiB ROW PER PIXEL = [160 channels of segment 0 | 160 of segment 1 | 160 of dy | 64 B pad] = 64 chunks of 16 bytes, so ONE LDS-DMA
// instruction stages one pixel of all three operands (a pixel's tap shifts and range checks are wave-uniform: scalar arithmetic, no per-lane decode);
// the instruction reads three different places -- two tensors -- so it is a global_load_lds (64-bit per-lane addresses) and padding / out-of-range
// pixels read a page of zeros instead of an out-of-range offset.  Chunks are XOR-swizzled over the whole row by the pixel (2 * ((pix & 3) + 4 * ((pix >> 3) & 1)),
// on the source side): the eight pixel rows one half-wave's transposed read (ds_read_b64_tr_b16) touches land on eight distinct 32-byte bank groups.
// Schedule: a K tile = 64 pixels = two phases (one 32-pixel k-step each: 20 transposed reads, 25 MFMAs per wave); per phase
// [fragment reads + 4 DMAs | barrier | MFMAs | barrier], the two wave groups one barrier apart.  Two stages of 64 KiB; a phase re-stages the 32 pixel
// rows the PREVIOUS phase read (which therefore retires its reads -- lgkmcnt(0) -- in front of its closing barrier) with the pixels three phases ahead,
// so every phase issues 4 DMAs per wave and waits for vmcnt(8): uniform, no prologue / tail variants (K tiles beyond an item's range read the zero page).
// Work: item = (pixel split, tile) of a table of up to W8R_MAX layers (one launch for the weight gradients of several layers: the tiles of one
// 160-channel layer are five, which would mean ~51 pixel splits -- 47 MB of slabs -- per layer); split-major inside a layer so that XCD-mates read the
// same pixels; slabs in the gradient's own layout, summed by the fixed-order kernels of conv_wgrad.hip.
#include "igemm_shared.h"
#include <string.h>

__device__ uint4 g_w8r_zero[64];       // 1 KiB of zeros: the source of padding / out-of-range pixels and of the pad chunks

namespace {

constexpr int W8R_MAX = 12;
struct W8rRec {
  const void* x;
  const void* dy;
  float* out;              // the gradient itself (splits == 1) or the slab region [splits][K][RS][C]
  long slab_stride;        // K * RS * C floats (0 when splits == 1)
  int N, H, W, C, P, Q, K;
  int stride, pad, S, RS;
  int M, nk;               // output pixels; K tiles of 64 pixels
  int splits, per;         // pixel splits, K tiles per split
  int csl, nseg;           // 160-channel slices per tap; segments = RS * csl
  int nau, nbt;            // A units (pairs of segments), B tiles (K / 160); tiles = nau * nbt
  unsigned magic_pq, magic_q;
  int accumulate;          // splits == 1: dw += (load-add-store)
};
struct W8rBatch {
  int n;
  int first[W8R_MAX + 1];  // item prefix sums
  W8rRec r[W8R_MAX];
};
static_assert(sizeof(W8rBatch) <= 3072, "kernel-argument segment");

template <typename T> struct Tr16r;
template <> struct Tr16r<bf16_t> {
  __device__ static inline uint2 rd(const char* p) {
    typedef __attribute__((address_space(3))) bf16x4* lp;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p)));
  }
};
template <> struct Tr16r<f16_t> {
  __device__ static inline uint2 rd(const char* p) {
    typedef __fp16 h4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) h4* lp;
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p)));
  }
};

__device__ inline void rbar() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
// one LDS-DMA wave instruction with per-lane 64-bit source addresses: LDS[m0 + lane * 16 .. + 16) = *addr (invisible to hipcc's waitcnt pass)
__device__ inline void gdma16(unsigned long long addr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(addr), "s"(lds_addr) : "memory");
}

// PROBE (diagnostic instantiations, rn_set_variant2 bits 8-10; wrong results, timing only): 1 = no LDS-DMA in the K loop, 2 = no MFMA, 3 = no fragment reads,
// 4 = no pixel decode in the K loop (the prologue's addresses again), 5 = DMAs issued but every lane reads the zero page
template <typename T, int PROBE = 0>
__global__ __launch_bounds__(512, 2) void wgrad8r_kernel(const W8rBatch b) {
  constexpr int ES = 2;
  constexpr int STGB = 65536, ROWB = 1024;
  __shared__ uint4 smem[2 * STGB / 16];
  const char* lds = reinterpret_cast<const char*>(&smem[0]);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                   // 4 x 2 waves of 80 rows x 80 columns
  const int l16 = lane & 15, lq = lane >> 4;
  const unsigned long long zero_page = (unsigned long long)(size_t)&g_w8r_zero[0];

  // ---- DMA roles.  Wave w stages pixels 4w + j and 32 + 4w + j (j = 0..3) of a K tile, one instruction per pixel, lane = physical chunk; the lane's
  // logical chunk for pixel j is lane ^ f, f = 2 (j + 4 ((w >> 1) & 1)); logical chunks [0,20) segment 0, [20,40) segment 1, [40,60) dy, [60,64) pad
  // (the lane's segment as three all-ones / zero masks per pixel slot: the 64-bit source address is picked with and / or -- written as a conditional
  // expression hipcc compiled the pick into divergent branches, 19 exec-mask flips and 130 moves per K tile)
  unsigned mk0[4], mk1[4], mk2[4], loff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane ^ (2 * (j + 4 * ((wave >> 1) & 1)));
    const int sg = c / 20;
    mk0[j] = sg == 0 ? 0xFFFFFFFFu : 0u; mk1[j] = sg == 1 ? 0xFFFFFFFFu : 0u; mk2[j] = sg == 2 ? 0xFFFFFFFFu : 0u;
    loff[j] = sg == 3 ? 0u : (unsigned)((c - 20 * sg) * 16);
  }
  const unsigned zlo = (unsigned)(zero_page & 0xFFFFFFFFull), zhi = (unsigned)(zero_page >> 32);

  // ---- fragment addresses (bytes inside a stage).  Transposed read: lane t = 4q + p of a 16-lane group supplies pixel row q, 8 bytes p of the 32 ----
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  const int frow = (8 * lq + tq) * ROWB + 8 * (tp & 1);
  const int ff = 2 * (tq + 4 * (lq & 1));                    // f(pixel) of the pixels this lane addresses (8 lq + tq, + 4, + 32)
  int fa[5], fb[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int pa = 10 * (wm >> 1) + 5 * (wm & 1) + i, pb = 20 + 5 * wn + i;       // the operand tile's pair of chunks
    fa[i] = frow + ((((2 * pa) ^ ff) | (tp >> 1)) << 4);
    fb[i] = frow + ((((2 * pb) ^ ff) | (tp >> 1)) << 4);
  }

  // ---- the current item's scalars (copied out of its layer record ONCE per item: a scalar load of a kernel argument inside the K loop would share the
  // lgkmcnt counter with the fragment reads) ----
  int rec = 0;
  int dh0 = 0, dw0 = 0, dh1 = 0, dw1 = 0, seg1_ok = 0;
  unsigned xo0 = 0, xo1 = 0, yo = 0;                         // byte offsets of the segments' channel slices / the dy channel tile
  int kend = 0;                                              // the item's K tiles are [.., kend)
  int iM = 0, iQ = 0, ipq = 1, iH = 0, iW = 0, istride = 1;
  unsigned imagic_pq = 0, imagic_q = 0, icrow = 0, ikrow = 0;
  unsigned long long ixb = 0, iyb = 0;
  // The source addresses of the wave's pixel rows, decoded with the LANES running over pixels (a scalar decode per pixel -- magic divisions, range checks,
  // 64-bit address arithmetic: ~90 SALU instructions -- made a phase 800 instructions long, 8 waves on one scalar unit: 5 us per K tile).  Lane l & 7 < 4 decodes
  // pixel 32 + 4 wave + (l & 3) of K tile g1 (its second half), lane l & 7 >= 4 pixel 4 wave + (l & 3) of K tile g2 (its first half); out-of-range pixels,
  // padding taps and K tiles beyond the item read the zero page.
  struct Addr3 { unsigned long long a0, a1, ay; };
  auto decode8 = [&](int g1, int g2) {
    const int l = lane & 7;
    const bool first_half = l >= 4;
    const int g = first_half ? g2 : g1;
    const int m = 64 * g + (first_half ? 0 : 32) + 4 * wave + (l & 3);
    Addr3 r{zero_page, zero_page, zero_page};
    const bool ok = g < kend && m < iM;
    int n = (int)__umulhi((unsigned)m, imagic_pq);
    int rem = m - n * ipq;
    if (rem >= ipq) { ++n; rem -= ipq; }
    int p = (int)__umulhi((unsigned)rem, imagic_q);
    int q = rem - p * iQ;
    if (q >= iQ) { ++p; q -= iQ; }
    const int h0 = p * istride + dh0, w0 = q * istride + dw0;
    if (ok && (unsigned)h0 < (unsigned)iH && (unsigned)w0 < (unsigned)iW) r.a0 = ixb + (unsigned long long)((unsigned)((n * iH + h0) * iW + w0)) * icrow + xo0;
    const int h1 = p * istride + dh1, w1 = q * istride + dw1;
    if (ok && seg1_ok && (unsigned)h1 < (unsigned)iH && (unsigned)w1 < (unsigned)iW) r.a1 = ixb + (unsigned long long)((unsigned)((n * iH + h1) * iW + w1)) * icrow + xo1;
    if (ok) r.ay = iyb + (unsigned long long)(unsigned)m * ikrow + yo;
    return r;
  };
  auto rl = [](unsigned v, int src) { return (unsigned)__builtin_amdgcn_readlane((int)v, src); };
  // the wave's four pixel rows of half h of a stage: the addresses sit in lanes lb .. lb + 3 of `ad`
  auto stage_half = [&](const Addr3& ad, int lb, int h, unsigned stage_lds) {
    const unsigned keep = m0_save();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // the pixel's three addresses minus the zero page's, broadcast from lane lb + j: (x - z) & mask summed over the three segments, + z = the lane's source
      const unsigned d0l = rl((unsigned)(ad.a0 & 0xFFFFFFFFull), lb + j) ^ zlo, d0h = rl((unsigned)(ad.a0 >> 32), lb + j) ^ zhi;
      const unsigned d1l = rl((unsigned)(ad.a1 & 0xFFFFFFFFull), lb + j) ^ zlo, d1h = rl((unsigned)(ad.a1 >> 32), lb + j) ^ zhi;
      const unsigned dyl = rl((unsigned)(ad.ay & 0xFFFFFFFFull), lb + j) ^ zlo, dyh = rl((unsigned)(ad.ay >> 32), lb + j) ^ zhi;
      unsigned lo = zlo ^ ((d0l & mk0[j]) | (d1l & mk1[j]) | (dyl & mk2[j])), hi = zhi ^ ((d0h & mk0[j]) | (d1h & mk1[j]) | (dyh & mk2[j]));
      if constexpr (PROBE == 5) { lo = zlo; hi = zhi; asm volatile("" : "+v"(lo), "+v"(hi) : "v"(d0l), "v"(d1h), "v"(dyl)); }
      gdma16((((unsigned long long)hi << 32) | lo) + loff[j], stage_lds + (unsigned)((32 * h + 4 * wave + j) * ROWB));      // (the zero page is 1 KiB: any chunk offset stays inside it)
    }
    m0_restore(keep);
  };

  f32x4 acc[5][5];
  uint4 af[5], bfr[5];
  if constexpr (PROBE == 3) {
#pragma unroll
    for (int i = 0; i < 5; ++i) { af[i] = make_uint4(lane, 1, 2, 3); bfr[i] = make_uint4(3, 2, 1, lane); }
  }
  auto rd = [&](int off) {                                   // one MFMA operand: pixels 8 lq .. 8 lq + 7 of a 32-pixel k-step, two transposed reads
    const uint2 lo = Tr16r<T>::rd(lds + off), hi = Tr16r<T>::rd(lds + off + 4 * ROWB);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  // one phase: k-step ks of the K tile in stage sx; meanwhile the 32 pixel rows the PREVIOUS phase read are re-staged with K tile gn, half hn
  auto phase = [&](int sx, int ks, const Addr3& ad, int lb, int hn, unsigned dst_lds) {
    if constexpr (PROBE != 3) {
#pragma unroll
      for (int j = 0; j < 5; ++j) bfr[j] = rd(sx + fb[j] + ks * 32 * ROWB);
#pragma unroll
      for (int i = 0; i < 5; ++i) af[i] = rd(sx + fa[i] + ks * 32 * ROWB);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PROBE != 1) stage_half(ad, lb, hn, dst_lds);
    wait_vmcnt<8>();                                         // all but this phase's and the previous phase's pieces: the half-tile the NEXT phase reads has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this phase's reads are retired in front of the barrier: the next phase re-stages these rows
    rbar();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if constexpr (PROBE != 2) Mfma16<T>::run(af[i], bfr[j], acc[i][j]);
        else asm volatile("" : "+v"(acc[i][j]) : "v"(af[i].x), "v"(af[i].w), "v"(bfr[j].x), "v"(bfr[j].w));
      }
    __builtin_amdgcn_s_setprio(0);
    rbar();
  };

  const int G = gridDim.x, nitems = b.first[b.n];
  int prec = -1, ptile = 0, psplit = 0;                      // the item whose accumulators are still in registers
  for (int it = 0;; ++it) {
    const int vb = it * G + (int)blockIdx.x;
    const bool more = vb < nitems;                           // wave-uniform
    int tile = 0, split = 0, kb = 0;
    Addr3 pkeep{zero_page, zero_page, zero_page};
    if (more) {
      const int xcd = vb & 7, q = nitems >> 3, r = nitems & 7;
      const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
      rec = 0;
      while (rec + 1 < b.n && item >= b.first[rec + 1]) ++rec;
      const W8rRec& R = b.r[rec];
      const int local = item - b.first[rec];
      const int ntiles = R.nau * R.nbt;
      split = local / ntiles; tile = local - split * ntiles;
      kb = split * R.per; kend = min(R.nk, kb + R.per);
      const int au = tile % R.nau, bt = tile / R.nau;
      const int s0 = 2 * au, s1 = 2 * au + 1;
      const int t0 = s0 / R.csl, t1 = s1 / R.csl;
      seg1_ok = s1 < R.nseg ? 1 : 0;
      dh0 = t0 / R.S - R.pad; dw0 = t0 % R.S - R.pad;
      dh1 = t1 / R.S - R.pad; dw1 = t1 % R.S - R.pad;
      xo0 = (unsigned)((s0 - t0 * R.csl) * 160 * ES); xo1 = (unsigned)((s1 - t1 * R.csl) * 160 * ES);
      yo = (unsigned)(bt * 160 * ES);
      iM = R.M; iQ = R.Q; ipq = R.P * R.Q; iH = R.H; iW = R.W; istride = R.stride;
      imagic_pq = R.magic_pq; imagic_q = R.magic_q; icrow = (unsigned)(R.C * ES); ikrow = (unsigned)(R.K * ES);
      ixb = (unsigned long long)(size_t)R.x; iyb = (unsigned long long)(size_t)R.dy;
      // prologue: K tile kb whole into stage 0, the first half of K tile kb + 1 into stage 1 (every wave has left the previous K loop)
      const Addr3 p0 = decode8(kb, kb), p1 = decode8(kb, kb + 1);
      pkeep = p1;
      stage_half(p0, 4, 0, lds0); stage_half(p0, 0, 1, lds0); stage_half(p1, 4, 0, lds0 + STGB);
    }
    bool stores_behind = false;
    if (prec >= 0) {        // acc[i][j][r]: row 80 wm + 16 i + 4 lq + r of the 320, output channel 160 bt + 80 wn + 16 j + l16
      const W8rRec& R = b.r[prec];
      const int au = ptile % R.nau, bt = ptile / R.nau;
      const int sg = 2 * au + (wm >> 1);
      if (sg < R.nseg) {
        const int t = sg / R.csl, c0 = (sg - t * R.csl) * 160 + 80 * (wm & 1) + 4 * lq, k0 = bt * 160 + 80 * wn + l16;
        float* out = R.out + (size_t)psplit * R.slab_stride;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          float* row = out + ((size_t)(k0 + 16 * j) * R.RS + t) * R.C + c0;
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (R.accumulate) { const float4 o = *reinterpret_cast<const float4*>(row + 16 * i); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(row + 16 * i) = v;
          }
        }
        stores_behind = !R.accumulate;
      }
    }
    if (!more) break;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // the first half of K tile kb has landed; its second half and the first half of kb + 1 (and the previous item's 25 stores, which are younger) may fly
    if (stores_behind) wait_vmcnt<8 + 25>(); else wait_vmcnt<8>();
    rbar();
    if (wave >= 4) rbar();                                   // the second wave group runs one barrier behind

    int sx = 0;
    const int nseg_k = kend - kb;
    for (int kt = 0; kt < nseg_k; ++kt) {
      const int g = kb + kt;
      const Addr3 ad = PROBE == 4 ? pkeep : decode8(g + 1, g + 2);
      phase(sx, 0, ad, 0, 1, lds0 + (unsigned)(sx ^ STGB));          // re-stage the other stage's pixels 32-63 (read by the previous phase) with K tile g + 1
      phase(sx, 1, ad, 4, 0, lds0 + (unsigned)sx);                   // re-stage this stage's pixels 0-31 with K tile g + 2
      sx ^= STGB;
    }
    if (wave < 4) rbar();
    prec = rec; ptile = tile; psplit = split;
  }
}

}  // namespace

// pixel splits of ONE layer's launch: one resident round of 256 persistent workgroups, or two when that balances better; cost = rounds x K tiles per
// split (~1.3 us each) + slab traffic (written once, read once)
static int w8r_pick_splits(long ntiles, long nk, double n_floats, long cap = 128) {
  int best = 1;
  double best_cost = 1e30;
  long smax = nk / 4 < 128 ? (nk / 4 < 1 ? 1 : nk / 4) : 128;
  if (smax > cap) smax = cap < 1 ? 1 : cap;
  for (long S = 1; S <= smax; ++S) {
    const long per = (nk + S - 1) / S;
    if (per * (S - 1) >= nk) continue;                                       // an empty last split
    const long rounds = (ntiles * S + 255) / 256;
    const double cost = rounds * (per * 1.3 + 8.0) + (S > 1 ? S * n_floats * 8.0 / 4.0e6 : 0.0) + (S > 1 ? 6.0 : 0.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = (int)S; }
  }
  return best;
}

static bool w8r_geom_ok(const rn_conv_geom* g, int dtype) {
  if (dtype != RN_BF16 && dtype != RN_F16) return false;
  if (g_rn_variant2 & 4) return false;                                         // A/B: never
  if (g->C % 160 || g->K % 160 || g->R != g->S) return false;
  if (g->R * g->S * (g->C / 160) < 2) return false;                            // a single segment would leave half of every tile empty (1x1 with 160 channels)
  return true;
}

// 0: rn_conv_wgrad does not take this kernel for the geometry; S >= 1: it does, with S pixel splits (S > 1: slabs + the reduction kernels)
int rn_wgrad8r_splits(const rn_conv_geom* g, int dtype) {
  if (!w8r_geom_ok(g, dtype)) return 0;
  const long M = (long)g->N * g->P * g->Q, nk = (M + 63) / 64;
  const long nseg = (long)g->R * g->S * (g->C / 160), ntiles = ((nseg + 1) / 2) * (g->K / 160);
  if (ntiles * nk < 8L * 256 && !(g_rn_variant2 & 8)) return 0;               // too small to fill the chip (rn_set_variant2 8: any size, tests)
  return w8r_pick_splits(ntiles, nk, (double)g->K * g->R * g->S * g->C);
}

static void w8r_fill(W8rRec& r, const void* x, const void* dy, float* out, int splits, const rn_conv_geom* g, int accumulate) {
  r.x = x; r.dy = dy; r.out = out;
  r.N = g->N; r.H = g->H; r.W = g->W; r.C = g->C; r.P = g->P; r.Q = g->Q; r.K = g->K;
  r.stride = g->stride; r.pad = g->pad; r.S = g->S; r.RS = g->R * g->S;
  r.M = g->N * g->P * g->Q; r.nk = (r.M + 63) / 64;
  r.csl = g->C / 160; r.nseg = r.RS * r.csl; r.nau = (r.nseg + 1) / 2; r.nbt = g->K / 160;
  const unsigned long long pq = (unsigned long long)g->P * g->Q;
  r.magic_pq = pq <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / pq);
  r.magic_q = g->Q <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned)g->Q);
  r.splits = splits; r.per = (r.nk + splits - 1) / splits;
  r.slab_stride = splits > 1 ? (long)g->K * r.RS * g->C : 0;
  r.accumulate = (splits == 1 && accumulate) ? 1 : 0;
}

// out: the gradient itself (splits == 1) or the slab region [splits][K][RS][C]
int rn_launch_wgrad8r(const void* x, const void* dy, float* out, int splits, int accumulate, int dtype, const rn_conv_geom* g, int max_grid, hipStream_t s) {
  W8rBatch b{};
  b.n = 1;
  w8r_fill(b.r[0], x, dy, out, splits, g, accumulate);
  b.first[0] = 0; b.first[1] = b.r[0].nau * b.r[0].nbt * splits;
  rn_note_kernel("wgrad8r<320x160>");
  if (rn_dry_run()) return 0;
  const int cap = max_grid > 0 && max_grid < 256 ? max_grid : 256;
  const int grid = b.first[1] < cap ? b.first[1] : cap;
  const int probe = (g_rn_variant2 >> 8) & 7;
  if (dtype == RN_BF16) hipLaunchKernelGGL((wgrad8r_kernel<bf16_t>), dim3(grid), dim3(512), 0, s, b);
  else if (probe == 1) hipLaunchKernelGGL((wgrad8r_kernel<f16_t, 1>), dim3(grid), dim3(512), 0, s, b);
  else if (probe == 2) hipLaunchKernelGGL((wgrad8r_kernel<f16_t, 2>), dim3(grid), dim3(512), 0, s, b);
  else if (probe == 3) hipLaunchKernelGGL((wgrad8r_kernel<f16_t, 3>), dim3(grid), dim3(512), 0, s, b);
  else if (probe == 4) hipLaunchKernelGGL((wgrad8r_kernel<f16_t, 4>), dim3(grid), dim3(512), 0, s, b);
  else if (probe == 5) hipLaunchKernelGGL((wgrad8r_kernel<f16_t, 5>), dim3(grid), dim3(512), 0, s, b);
  else hipLaunchKernelGGL((wgrad8r_kernel<f16_t>), dim3(grid), dim3(512), 0, s, b);
  RN_CHECK_LAUNCH("wgrad8r");
  return 0;
}

int rn_wgrad_reduce_slabs(const float* ws, float* dw_krsc, long n, int splits, int accum, int eight_phase, hipStream_t s);      // conv_wgrad.hip

int rn_wgrad9_splits(const rn_conv_geom* g, int dtype);                                                                      // conv_wgrad9.hip
int rn_wgrad9_batch(const rn_wgrad8r_desc* descs, int n, int dtype, int max_grid, hipStream_t s, hipStream_t s_reduce, hipEvent_t ev);

extern "C" int rn_conv_wgrad8r_ok(const rn_conv_geom* g, int dtype) { return g && (rn_wgrad9_splits(g, dtype) > 0 || rn_wgrad8r_splits(g, dtype) > 0) ? 1 : 0; }

// The weight gradients of n layers of ONE geometry as one launch (the plan executor queues the forked weight gradients of a residual stage: the tiles of
// n layers share the chip, so a layer is cut into ~1/n of the pixel splits a launch of its own takes -- 1/n of the slab traffic -- and the launch's ramp,
// tail and tile quantisation are paid once).  Every record's slabs go to its own workspace and are summed into its dw (+= with RN_F_ACCUM) right behind.
int rn_wgrad9_best_batch(const rn_conv_geom* g, int dtype, int max_n);      // conv_wgrad9.hip
// the number of layers of geometry g worth collecting for one launch (<= max_n)
extern "C" int rn_conv_wgrad8r_best_batch(const rn_conv_geom* g, int dtype, int max_n) {
  if (!g || max_n < 1 || !rn_conv_wgrad8r_ok(g, dtype)) return 1;
  if (max_n > RN_WGRAD8R_BATCH_MAX) max_n = RN_WGRAD8R_BATCH_MAX;
  if (rn_wgrad9_splits(g, dtype) > 0) return rn_wgrad9_best_batch(g, dtype, max_n);
  return max_n < 4 ? max_n : 4;
}

static int w8r_batch_impl(const rn_wgrad8r_desc* descs, int n, int dtype, int max_grid, rn_stream s, hipStream_t s_reduce, hipEvent_t ev);
extern "C" int rn_conv_wgrad8r_batch(const rn_wgrad8r_desc* descs, int n, int dtype, int max_grid, rn_stream s) {
  return w8r_batch_impl(descs, n, dtype, max_grid, s, nullptr, nullptr);
}
// the same with the slab sums on a stream of their own: `ev` (a hipEvent_t of the caller) is recorded on s behind the kernel, s_reduce waits for it and runs
// the sums -- they are light HBM-bound launches that fit beside the NEXT weight gradient's (and a data gradient's) workgroups on a CU.  The caller orders the
// workspace's re-use (the next launch into the same region waits for these sums) and folds s_reduce back before anything consumes the gradients.
extern "C" int rn_conv_wgrad8r_batch2(const rn_wgrad8r_desc* descs, int n, int dtype, int max_grid, rn_stream s, rn_stream s_reduce, void* ev) {
  RN_CHECK_ARG(s_reduce && ev, "rn_conv_wgrad8r_batch2: null reduce stream / event");
  return w8r_batch_impl(descs, n, dtype, max_grid, s, as_stream(s_reduce), reinterpret_cast<hipEvent_t>(ev));
}
static int w8r_batch_impl(const rn_wgrad8r_desc* descs, int n, int dtype, int max_grid, rn_stream s, hipStream_t s_reduce, hipEvent_t ev) {
  static_assert(RN_WGRAD8R_BATCH_MAX == W8R_MAX, "header and kernel disagree");
  RN_CHECK_ARG(descs && n > 0 && n <= RN_WGRAD8R_BATCH_MAX, "rn_conv_wgrad8r_batch: n=%d out of range (1..%d)", n, RN_WGRAD8R_BATCH_MAX);
  const rn_conv_geom& g0 = descs[0].g;
  RN_CHECK_ARG(rn_conv_wgrad8r_ok(&g0, dtype), "rn_conv_wgrad8r_batch: the geometry is not one the 320 x 160 / 288 x 160 kernels take");
  if (rn_wgrad9_splits(&g0, dtype) > 0) return rn_wgrad9_batch(descs, n, dtype, max_grid, as_stream(s), s_reduce, ev);      // 3x3 stride 1: the nine-tap kernel
  const long M = (long)g0.N * g0.P * g0.Q, nk = (M + 63) / 64;
  const long nseg = (long)g0.R * g0.S * (g0.C / 160), ntiles = ((nseg + 1) / 2) * (g0.K / 160);
  const size_t nel = (size_t)g0.K * g0.R * g0.S * g0.C;
  long ws_cap = 128;                                     // slabs every record's workspace holds
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
  const int splits = w8r_pick_splits(ntiles * n, nk, (double)nel * n, ws_cap);
  W8rBatch b{};
  b.n = n;
  int items = 0;
  for (int i = 0; i < n; ++i) {
    const rn_wgrad8r_desc& d = descs[i];
    RN_CHECK_ARG(d.x && d.dy && d.dw, "rn_conv_wgrad8r_batch: record %d: null pointer", i);
    RN_CHECK_ARG(memcmp(&d.g, &g0, sizeof(g0)) == 0, "rn_conv_wgrad8r_batch: record %d has another geometry", i);
    const bool direct = splits == 1 && !(d.flags & RN_F_ACCUM);
    RN_CHECK_ARG(direct || (d.ws && d.ws_bytes >= (size_t)sharers[i] * splits * nel * sizeof(float)), "rn_conv_wgrad8r_batch: record %d: workspace too small (%zu < %zu)", i, d.ws_bytes,
                 (size_t)sharers[i] * splits * nel * sizeof(float));
    w8r_fill(b.r[i], d.x, d.dy, direct ? d.dw : reinterpret_cast<float*>(d.ws) + (size_t)share[i] * splits * nel, splits, &g0, 0);
    b.first[i] = items;
    items += b.r[i].nau * b.r[i].nbt * splits;
    rn_note_kernel("wgrad8r<320x160>");
    if (!direct) rn_note_kernel("wgrad_reduce");
  }
  b.first[n] = items;
  if (rn_dry_run()) return 0;
  const int cap = max_grid > 0 && max_grid < 256 ? max_grid : 256;
  const int grid = items < cap ? items : cap;
  if (dtype == RN_BF16) hipLaunchKernelGGL((wgrad8r_kernel<bf16_t>), dim3(grid), dim3(512), 0, as_stream(s), b);
  else hipLaunchKernelGGL((wgrad8r_kernel<f16_t>), dim3(grid), dim3(512), 0, as_stream(s), b);
  RN_CHECK_LAUNCH("wgrad8r batch");
  hipStream_t sr = as_stream(s);
  if (s_reduce && ev) {
    if (hipEventRecord(ev, as_stream(s)) != hipSuccess || hipStreamWaitEvent(s_reduce, ev, 0) != hipSuccess) { rn_set_error("rn_conv_wgrad8r_batch: event record / wait failed"); return 2; }
    sr = s_reduce;
  }
  for (int i = 0; i < n; ++i) {
    const rn_wgrad8r_desc& d = descs[i];
    if (splits == 1 && !(d.flags & RN_F_ACCUM)) continue;             // written in place
    if (int e = rn_wgrad_reduce_slabs(reinterpret_cast<const float*>(d.ws) + (size_t)share[i] * splits * nel, d.dw, (long)nel, splits, (d.flags & RN_F_ACCUM) ? 1 : 0, 1, sr)) return e;
  }
  return 0;
}
