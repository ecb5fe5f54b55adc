// Shared device/host helpers for libhamt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include "../../include/hamt.h"

#define HAMT_WAVE 64

void hamt_set_error(const char* fmt, ...);
void hamt_set_last_kernel(const char* fmt, ...);

#define HAMT_CHECK_ARG(cond, ...)                      \
  do {                                                 \
    if (!(cond)) {                                     \
      hamt_set_error(__VA_ARGS__);                     \
      return HAMT_ERR_ARG;                             \
    }                                                  \
  } while (0)

#define HAMT_CHECK_LAUNCH(name)                                                        \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      hamt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
      return HAMT_ERR_LAUNCH;                                                          \
    }                                                                                  \
  } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef unsigned short bf16_t;  // raw bf16 bits in memory

// ---- bf16 <-> f32 (round to nearest even; NaN preserved): gfx950's v_cvt_pk_bf16_f32, one instruction per PAIR
typedef __attribute__((ext_vector_type(2))) __bf16 hamt_bf2;
typedef __attribute__((ext_vector_type(2))) float hamt_f2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const hamt_f2 v = {lo, hi};
  const hamt_bf2 b = __builtin_convertvector(v, hamt_bf2);
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.0f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// ---- IEEE half <-> f32 (HAMT_F16: the 2-byte format of a dense layer's output in front of a LayerNorm -- 10 mantissa bits, a
// rounding 8 x finer than bf16's at the same bytes; round to nearest even, v_cvt_f16_f32)
typedef __attribute__((ext_vector_type(2))) _Float16 hamt_h2;
// Values beyond half's range SATURATE at +-65504 (one v_med3_f32 each) instead of becoming inf: the consumer is a LayerNorm, which an
// inf turns into a row of NaN and a saturated outlier does not.  (HAMT's dense outputs are O(1 .. 100); the clamp is insurance.)
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  lo = __builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f);
  hi = __builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f);
  const hamt_h2 b = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ void unpack_h2(uint32_t w, float& lo, float& hi) {
  const hamt_h2 b = __builtin_bit_cast(hamt_h2, w);
  lo = (float)b[0]; hi = (float)b[1];
}
__device__ __forceinline__ bf16_t f2h(float f) { return (bf16_t)(pack_h2(f, 0.0f) & 0xffffu); }
__device__ __forceinline__ float h2f(bf16_t h) { float a, b; unpack_h2((uint32_t)h, a, b); return a; }

// ---- counter-based dropout RNG: 24-bit uniform from (seed, epoch, call_id, element index)
__device__ __forceinline__ uint32_t hamt_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
struct RngKey { uint32_t k0, k1; };
__device__ __forceinline__ RngKey rng_key(const uint64_t* rng, uint32_t call_id) {
  RngKey k;
  uint64_t seed = rng ? rng[0] : 0x9E3779B97F4A7C15ull, epoch = rng ? rng[1] : 0ull;
  k.k0 = hamt_mix32((uint32_t)seed ^ hamt_mix32((uint32_t)epoch + 0x9e3779b9u) ^ (call_id * 0x85ebca6bu));
  k.k1 = hamt_mix32((uint32_t)(seed >> 32) ^ (uint32_t)(epoch >> 32) ^ (call_id + 0xc2b2ae35u));
  return k;
}
// keep-probability test for element idx; returns the multiplicative factor (0 or 1/(1-p))
__device__ __forceinline__ float drop_scale(RngKey k, uint64_t idx, float p, float inv_keep) {
  uint32_t x = hamt_mix32((uint32_t)idx ^ k.k0);
  x = hamt_mix32(x + (uint32_t)(idx >> 32) * 0x9e3779b9u + k.k1);
  float u = (float)(x >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}

// Four keep factors at once for 4 consecutive elements of one row (short-sequence attention): one 2-round hash of
// (row, group-of-4) gives 4 x 16-bit uniforms.  `rowh` = hamt_mix32(row_id ^ k.k0) is computed once per row.
__device__ __forceinline__ void drop_scale4(RngKey k, uint32_t rowh, uint32_t grp, float p, float inv_keep, float (&f)[4]) {
  const uint32_t x = hamt_mix32(rowh + grp * 0x9e3779b9u + k.k1);
  const uint32_t y = hamt_mix32(x ^ 0x85ebca6bu);
  const uint32_t thr = (uint32_t)(p * 65536.0f);           // keep iff u16 >= thr
  f[0] = (x & 0xffffu) >= thr ? inv_keep : 0.0f;
  f[1] = (x >> 16) >= thr ? inv_keep : 0.0f;
  f[2] = (y & 0xffffu) >= thr ? inv_keep : 0.0f;
  f[3] = (y >> 16) >= thr ? inv_keep : 0.0f;
}

// ---- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// gelu(x) and gelu'(x) from one exp: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)
__device__ __forceinline__ void gelu_and_grad(float x, float& g, float& dg) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float e = __expf(-u * u);                                   // exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * u);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_u = 1.0f - poly * e;                              // erf(|x|/sqrt2)
  const float phi = 0.5f * (1.0f + copysignf(erf_u, x));            // Phi(x)
  g = x * phi;
  dg = phi + x * 0.39894228040143267794f * e;
}

// ---- HAMT_U8G: gelu'(x) in [-0.129, 1.129] as one byte, value = 0.005 q - 0.13 (hamt.h)
__device__ __forceinline__ uint32_t g8_pack4(const float* v) {
  uint32_t r = 0;       // v_cvt_pk_u8_f32: round to nearest, clamp to [0, 255], insert into byte `sel`
  r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[0], 200.0f, 26.0f), 0, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[1], 200.0f, 26.0f), 1, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[2], 200.0f, 26.0f), 2, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[3], 200.0f, 26.0f), 3, r);
  return r;
}
__device__ __forceinline__ void g8_unpack4(uint32_t w, float* o) {
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = __builtin_fmaf((float)((w >> (8 * j)) & 0xffu), 0.005f, -0.13f);
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
