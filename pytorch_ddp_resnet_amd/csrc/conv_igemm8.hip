// Implicit-GEMM convolution on the EIGHT-PHASE schedule (gfx950), 16-bit element types: 256 x BN block tiles, 8 waves, one workgroup per CU.
//
// Same contraction and tap-table view as conv_igemm.hip (forward conv and data gradient of residual_block.py:34-57, :129-159),
//   dst[n, p*ds+oh, q*ds+ow, k] (+)= sum_t sum_c src[n, p*ss+dh[t], q*ss+dw[t], c] * wt[k][widx[t]][c]   (+ res),
// but the K loop is the deep-pipelined structure of cdna_hip_programming.md section 5 ("256^2 8-phase template"), carried over to the
// implicit-GEMM operand walk:
//   * a K tile = one tap x 64 channels (128-byte rows); per K tile the A tile (256 pixels) and the B tile (BN output channels) are each
//     split in two HALF-TILES = what one quadrant row / column of every wave's output block reads;
//   * a wave owns 128 x 64 outputs (BN = 256; 64 x 64 at BN = 128) as four quadrants; one PHASE = the fragment reads of one half-tile,
//     the LDS-DMA of one half-tile two K tiles ahead, and the MFMAs of one quadrant x 64 channels (v_mfma_f32_16x16x32), i.e. four
//     phases per K tile; both LDS stages of every half-tile live in ONE array (128 KiB);
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
// tails are out-of-range DMA offsets that the hardware zero-fills.  Epilogue: igemm_epilogue of igemm_shared.h (16x16 accumulator tiles).
#include "igemm_shared.h"

namespace {

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
__device__ inline void walk8_advance(const IgemmArgs& a, Walk8& w) {
  w.cc += 1; w.src += 128u; w.wt += 128u;
  if (w.cc == a.w8_cpc) {
    w.cc = 0; w.j += 1; w.t += 1;
    w.src += (unsigned)(a.w8_sj - a.w8_cpc * 128); w.wt += (unsigned)(a.w8_wj - a.w8_cpc * 128);
    if (w.j == a.ntw) {
      w.j = 0; w.i += 1;
      w.src += (unsigned)(a.w8_si - a.ntw * a.w8_sj); w.wt += (unsigned)(a.w8_wi - a.ntw * a.w8_wj);
    }
  }
}

template <typename T, int BN>
__global__ __launch_bounds__(512, 2) void igemm8_kernel(const IgemmArgs a) {
  constexpr int BM = 256, ES = 2;
  constexpr int WM = BN == 256 ? 2 : 4, WN = 8 / WM;        // 2 x 4 waves of 128 x 64, or 4 x 2 waves of 64 x 64
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int RT = WTM / 16, CT = WTN / 16;               // 16 x 16 accumulator tiles of a wave
  constexpr int QR = RT / 2, QC = CT / 2;                   // ... of a quadrant
  constexpr int HR = WTM / 2, HC = WTN / 2;                 // rows / columns of a wave's quadrant
  constexpr int AI = 2, BI = BN / 128;                      // DMA instructions per wave and half-tile (A: 128 rows, B: BN/2 rows)
  constexpr int STG = 4096;                                 // uint4 per stage: 64 KiB (A0 | A1 | B0 | B1), power of two: the stage toggles by XOR
  constexpr int A_H = 1024, B_0 = 2048, B_H = BN * 4;       // uint4 offsets: second A half, B, second B half
  static_assert(sizeof(T) == ES && (BN == 256 || BN == 128) && B_0 + 2 * B_H <= STG, "tile");
  __shared__ uint4 smem[2 * STG + TAP_INTS / 4];
  int* taps = reinterpret_cast<int*>(&smem[2 * STG]);

  preload_args(a);
  asm volatile("" ::"s"(a.w8_src0), "s"(a.w8_si), "s"(a.w8_sj), "s"(a.w8_wt0), "s"(a.w8_wi), "s"(a.w8_wj), "s"(a.w8_cpc));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nmt = (a.M + BM - 1) / BM;
  int bid = blockIdx.x;
  if (a.xcd_remap) {
    const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nnt_ = gridDim.x / nmt;
  const int ntile = a.xcd_remap ? bid % nnt_ : bid / nmt, mt = a.xcd_remap ? bid / nnt_ : bid % nmt;
  const int m0 = mt * BM, n0 = ntile * BN;
  const int pq = a.Pc * a.Qc;
  const int n_first = m0 / pq;
  const size_t img_bytes = (size_t)a.Hs * a.Ws * a.Cs * ES;
  const v4i32 ra_desc = make_desc(reinterpret_cast<const char*>(a.src) + (size_t)n_first * img_bytes, (size_t)(a.N - n_first) * img_bytes);
  const v4i32 rb_desc = make_desc(a.wt, (size_t)a.Kd * a.wrs * a.Cs * ES);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)(&smem[0]);

  fill_tap_tables<ES>(a, taps);
  __syncthreads();

  // ---- DMA roles.  One instruction = 8 LDS rows x 128 B; instruction q of a half-tile covers its LDS rows 8q .. 8q+7; lane = (row lrow, physical
  // chunk p), which holds logical chunk p ^ ((row >> 1) & 7) (the XOR goes on the SOURCE address, the destination is lane-linear).
  // A half h, LDS row r  <->  tile row (r / HR) * WTM + h * HR + r % HR;  B half h, LDS row r  <->  tile column (r / HC) * WTN + h * HC + r % HC.
  const int lrow = lane >> 3, p = lane & 7;
  unsigned abase[2 * AI];
  unsigned amask[AI];                 // two 16-bit tap masks per register: half 0 low, half 1 high
  {
    TapGrid grid;
    load_tap_grid(a, taps, grid);
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
        if (m < a.M) {
          int n, pp, q;
          decode_row(a, m, pq, n, pp, q);
          const int hb = pp * a.ss, wb = q * a.ss;
          base = (unsigned)((((size_t)(n - n_first) * a.Hs + hb) * a.Ws + wb) * a.Cs * ES) + (unsigned)(c * 16);
          mk = (unsigned)tap_mask(a, grid, hb, wb) & 0xFFFFu;
        }
        abase[h * AI + jj] = base;
        amask[jj] |= mk << (16 * h);
      }
  }
  unsigned bbase[BI];
#pragma unroll
  for (int jj = 0; jj < BI; ++jj) {
    const int r = 8 * (wave * BI + jj) + lrow;
    const int c = p ^ ((r >> 1) & 7);
    const int k = n0 + (r / HC) * WTN + (r % HC);
    bbase[jj] = (unsigned)((size_t)k * a.wrs * a.Cs * ES) + (unsigned)(c * 16);          // Kd % BN == 0 (launcher): every column exists
  }
  const unsigned bhalf = (unsigned)((size_t)HC * a.wrs * a.Cs * ES);                      // B half 1 = the columns HC further

  auto issue_a = [&](int h, unsigned stage_lds, const Walk8& w) {
    const unsigned keep = m0_save();
#pragma unroll
    for (int jj = 0; jj < AI; ++jj) {
      const bool ok = (amask[jj] >> (w.t + 16 * h)) & 1u;
      dma16(ra_desc, ok ? abase[h * AI + jj] + w.src : OOB, stage_lds + (unsigned)((h * A_H) * 16 + (wave * AI + jj) * 1024));
    }
    m0_restore(keep);
  };
  auto issue_b = [&](int h, unsigned stage_lds, const Walk8& w) {
    const unsigned keep = m0_save();
#pragma unroll
    for (int jj = 0; jj < BI; ++jj)
      dma16(rb_desc, bbase[jj] + (h ? bhalf : 0u) + w.wt, stage_lds + (unsigned)((B_0 + h * B_H) * 16 + (wave * BI + jj) * 1024));
    m0_restore(keep);
  };

  // ---- fragment addresses (uint4 units inside a stage): 16 consecutive LDS rows at chunk (4 ks + lq) ^ ((row >> 1) & 7); the row bases are multiples of 16
  const int wm = wave / WN, wn = wave % WN;
  const int l16 = lane & 15, lq = lane >> 4;
  const int sw = (l16 >> 1) & 7;
  const int fa0 = (wm * HR + l16) * 8 + (lq ^ sw), fa1 = (wm * HR + l16) * 8 + ((4 + lq) ^ sw);
  const int fb0 = B_0 + (wn * HC + l16) * 8 + (lq ^ sw), fb1 = B_0 + (wn * HC + l16) * 8 + ((4 + lq) ^ sw);

  f32x4 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: K tile 0 whole, then B0 / A0 / B1 of K tile 1 (the state every K tile's phase 1 starts from) ----
  Walk8 wk;
  wk.i = 0; wk.j = 0; wk.cc = 0; wk.t = 0; wk.src = (unsigned)a.w8_src0; wk.wt = (unsigned)a.w8_wt0;
  const int nk = a.nk;
  issue_b(0, lds0, wk); issue_a(0, lds0, wk); issue_b(1, lds0, wk); issue_a(1, lds0, wk);
  walk8_advance(a, wk);
  if (nk > 1) {
    issue_b(0, lds0 + STG * 16, wk); issue_a(0, lds0 + STG * 16, wk); issue_b(1, lds0 + STG * 16, wk);
    wait_vmcnt<AI + 2 * BI>();
  } else {
    wait_vmcnt<0>();
  }
  raw_barrier();
  if (wave >= 4) raw_barrier();                           // the second wave group runs one barrier behind

  uint4 af[QR][2], b0[QC][2], b1[QC][2];
  // MODE 2: K tiles kt+1 and kt+2 exist (steady state); 1: kt+1 is the last; 0: kt is the last
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
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(af[i][ks], b0[j][ks], acc[i][j]);
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
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(af[i][ks], b1[j][ks], acc[i][QC + j]);
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
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(af[i][ks], b1[j][ks], acc[QR + i][QC + j]);
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
        for (int j = 0; j < QC; ++j) Mfma16<T>::run(af[i][ks], b0[j][ks], acc[QR + i][j]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
  };

  int sx = 0;
  Walk8 prev = wk;                                        // K tile 1
  for (int kt = 0; kt + 2 < nk; ++kt) {
    walk8_advance(a, wk);                                 // K tile kt+2
    ktile(std::integral_constant<int, 2>{}, sx, prev, wk);
    prev = wk;
    sx ^= STG;
  }
  if (nk > 1) {
    ktile(std::integral_constant<int, 1>{}, sx, prev, wk);
    sx ^= STG;
  }
  ktile(std::integral_constant<int, 0>{}, sx, prev, wk);
  if (wave < 4) raw_barrier();                            // the first group waits for the second: every wave has executed the same barriers

  igemm_epilogue<T, BM, BN, WM, WN, WTM / 32, CT, 512, 64, true>(a, acc, m0, n0, wave, lane, reinterpret_cast<float*>(&smem[0]));
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
  const long row = (long)a.Cs * 2;
  a.w8_src0 = (int)(((long)a.dh[0] * a.Ws + a.dw[0]) * row);
  a.w8_si = (int)((long)ddh * a.Ws * row);
  a.w8_sj = (int)((long)ddw * row);
  a.w8_wt0 = (int)((long)a.widx[0] * row);
  a.w8_wi = (int)((long)dwi * row);
  a.w8_wj = (int)((long)dwj * row);
  a.w8_cpc = a.Cs / 64;
  return true;
}

template <typename T> int launch8(IgemmArgs& a, hipStream_t s) {
  const int BN = 256;
  if (!fill_walk8(a)) return -1;
  a.nk = a.nt * a.w8_cpc;
  rn_note_kernel("igemm8<256x%d>", BN);
  if (rn_dry_run()) return 0;
  const dim3 grid(cdiv(a.M, 256) * (a.Kd / BN));
  hipLaunchKernelGGL((igemm8_kernel<T, 256>), grid, dim3(512), 0, s, a);
  RN_CHECK_LAUNCH("igemm8");
  return 0;
}

}  // namespace

// geometry the eight-phase kernel covers: 16-bit elements, channel count a multiple of 64 (a K tile never straddles a tap), output channels a
// multiple of the column tile, 1..16 taps in a separable progression, 32-bit tile offsets; the grid rule (enough tiles for the chip) is the caller's
int rn_launch_igemm8(const IgemmArgs& a_in, int dtype, hipStream_t s) {
  if (dtype != RN_BF16 && dtype != RN_F16) return -1;
  if (a_in.Cs % 64 || a_in.Kd % 256 || a_in.M <= 0) return -1;
  const size_t img_bytes = (size_t)a_in.Hs * a_in.Ws * a_in.Cs * 2;
  const long pq = (long)a_in.Pc * a_in.Qc;
  if ((256 / pq + 3) * (double)img_bytes >= 4.0e9) return -1;                          // per-tile source offsets are 32-bit (descriptor based at the tile's first image)
  if (((double)a_in.Kd + 256.0) * a_in.wrs * a_in.Cs * 2 >= 4.0e9) return -1;
  IgemmArgs a = a_in;
  if (dtype == RN_BF16) return launch8<bf16_t>(a, s);
  return launch8<f16_t>(a, s);
}
