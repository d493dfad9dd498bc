// Shared device/host helpers for librn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/rn_hip.h"


typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void rn_set_error(const char* fmt, ...);
// kernel log (rn_kernel_log / rn_conv_kernel_names): every convolution launcher reports the instantiation it picks; in a
// dry run (rn_dry_run() true) it reports and returns without launching, so tests can ask which kernel a geometry selects
void rn_note_kernel(const char* fmt, ...);
bool rn_dry_run();

#define RN_CHECK_ARG(cond, ...)              \
  do {                                       \
    if (!(cond)) {                           \
      rn_set_error(__VA_ARGS__);             \
      return 1;                              \
    }                                        \
  } while (0)

#define RN_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      rn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

// ---- element traits: 16-byte chunks are the unit of every global / LDS access ----
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int CE = 4;  // elements per 16-byte chunk
  __device__ static inline float to_f(float v) { return v; }
  __device__ static inline float from_f(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int CE = 8;
  __device__ static inline float to_f(bf16_t v) { return (float)v; }
  __device__ static inline bf16_t from_f(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
};

// fp16 storage (the reference's own GPU arithmetic: autocast + GradScaler, script.py:63, training.py:95-110): 11 significant bits
// instead of bf16's 8, same MFMA rate; the narrow exponent is handled by dynamic loss scaling (rn_softmax_ce / rn_amp_*).
// Conversions are v_cvt_f16_f32 (RNE); overflow becomes inf, which the unscale pass detects.
template <> struct Elem<f16_t> {
  static constexpr int CE = 8;
  __device__ static inline float to_f(f16_t v) { return (float)v; }
  __device__ static inline f16_t from_f(float v) { return (f16_t)v; }
};

#define RN_DTYPE_OK(d) ((d) == RN_F32 || (d) == RN_BF16 || (d) == RN_F16)
// runs the statement(s) once with T_ bound to the element type of `dtype`
#define RN_BY_DTYPE(dtype, ...)                                  \
  do {                                                           \
    if ((dtype) == RN_F32) { typedef float T_; __VA_ARGS__; }    \
    else if ((dtype) == RN_BF16) { typedef bf16_t T_; __VA_ARGS__; } \
    else { typedef f16_t T_; __VA_ARGS__; }                      \
  } while (0)

// a 16-byte chunk viewed as CE elements of T
template <typename T> struct Chunk {
  union {
    uint4 u;
    T e[Elem<T>::CE];
  };
  __device__ Chunk() {}
};

template <typename T> __device__ inline Chunk<T> load_chunk(const T* p) {
  Chunk<T> c;
  c.u = *reinterpret_cast<const uint4*>(p);
  return c;
}
template <typename T> __device__ inline void store_chunk(T* p, const Chunk<T>& c) {
  *reinterpret_cast<uint4*>(p) = c.u;
}

// ---- dropout: counter-based keep mask, recomputed in the backward (tests/np_interp.py keep_mask is the spec) ----
__host__ __device__ inline uint32_t rn_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint32_t rn_drop_key(uint32_t site, uint64_t step_seed) {
  return rn_lowbias32((site * 0x9E3779B9u) ^ (uint32_t)(step_seed & 0xFFFFFFFFu) ^ ((uint32_t)(step_seed >> 32) * 0x85EBCA6Bu));
}
__host__ __device__ inline uint32_t rn_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
// keep element idx iff its 16-bit half of the hash of the element PAIR idx >> 1 is >= the threshold's upper 16 bits: one hash per two elements (the mask of a
// chunk of 8 costs 4 hashes; per element the hash was ~40 % of bn_apply's issue slots on a dropout network).  p is quantised to 2^-16.
// tests/np_interp.py:keep_mask is the specification the kernels are tested against.
__device__ inline bool rn_keep(uint32_t key, uint32_t idx, uint32_t thr) {
  const uint32_t h = rn_lowbias32((idx >> 1) ^ key);
  return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= (thr >> 16);
}
// the keep bits of CE consecutive elements from an EVEN index base (a 16-byte chunk: base is a multiple of CE): bit e = element base + e
template <int CE>
__device__ inline uint32_t rn_keep_chunk(uint32_t key, uint32_t base, uint32_t thr) {
  static_assert(CE % 2 == 0, "pairs");
  const uint32_t t16 = thr >> 16;
  uint32_t bits = 0;
#pragma unroll
  for (int q = 0; q < CE / 2; ++q) {
    const uint32_t h = rn_lowbias32(((base >> 1) + (uint32_t)q) ^ key);
    bits |= ((h & 0xFFFFu) >= t16 ? 1u : 0u) << (2 * q);
    bits |= ((h >> 16) >= t16 ? 1u : 0u) << (2 * q + 1);
  }
  return bits;
}

// ---- residual / merge operand (RN_RES_*): value seen at destination (n,h,w,c) of a [N,H,W,C] tensor ----
struct ResDesc {
  const void* ptr;
  int mode;
  int C;      // channels of the residual tensor
  int H, W;   // spatial size of the residual tensor
};

template <typename T>
__device__ inline float res_load1(const ResDesc& r, int n, int h, int w, int c) {
  const T* p = reinterpret_cast<const T*>(r.ptr);
  if (r.mode == RN_RES_SAME) {
    return Elem<T>::to_f(p[(((size_t)n * r.H + h) * r.W + w) * r.C + c]);
  } else if (r.mode == RN_RES_DOWN2PAD) {
    if (c >= r.C) return 0.f;
    return Elem<T>::to_f(p[(((size_t)n * r.H + 2 * h) * r.W + 2 * w) * r.C + c]);
  } else {  // RN_RES_UP2
    if ((h | w) & 1) return 0.f;
    return Elem<T>::to_f(p[(((size_t)n * r.H + (h >> 1)) * r.W + (w >> 1)) * r.C + c]);
  }
}

// chunk version: c is a multiple of CE; whole chunk is in or out (res C is a multiple of CE)
template <typename T>
__device__ inline void res_add_chunk(const ResDesc& r, int n, int h, int w, int c, float* v) {
  constexpr int CE = Elem<T>::CE;
  const T* p = reinterpret_cast<const T*>(r.ptr);
  size_t off;
  if (r.mode == RN_RES_SAME) {
    off = (((size_t)n * r.H + h) * r.W + w) * r.C + c;
  } else if (r.mode == RN_RES_DOWN2PAD) {
    if (c >= r.C) return;
    off = (((size_t)n * r.H + 2 * h) * r.W + 2 * w) * r.C + c;
  } else {
    if ((h | w) & 1) return;
    off = (((size_t)n * r.H + (h >> 1)) * r.W + (w >> 1)) * r.C + c;
  }
  Chunk<T> ch = load_chunk<T>(p + off);
#pragma unroll
  for (int i = 0; i < CE; ++i) v[i] += Elem<T>::to_f(ch.e[i]);
}

static inline hipStream_t as_stream(rn_stream s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
