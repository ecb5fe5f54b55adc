// attn_small, bf16-MFMA version (HAMT_PREC_BF16): the cross-modal / self attention core of HAMT on
// v_mfma_f32_16x16x32_bf16 with fp32 softmax statistics.
//
//   S = Q K^T * scale + mask[b,key];  P = softmax(S);  P~ = dropout(P);  O = P~ V        (d_head = 64)
//
// Forward: one workgroup per (batch, head, 64-query tile), wave w owns query rows [16w, 16w+16).  K/V are walked in
// 64-key tiles staged as bf16 in LDS (row stride 72: ds_read_b64_tr_b16 quads conflict free, ds_read_b128 2-way).
// Every product is issued "swapped" so that a lane always owns ONE query row:
//     S^T = K Q^T   (A = K rows from LDS, B = Q rows held in registers)  -> lane (q = lane&15) holds 16 keys
//     O^T = V^T P^T (A = V through the hardware transpose read, B = P straight from the softmax registers)
// so the online-softmax rescale is a per-lane scalar, P never goes through LDS, and the 64 keys of a row need only two
// cross-lane butterflies (xor 16, 32).  The k-order of the P V product is defined by the S^T accumulator layout
// (keys 4g..4g+3 of two adjacent 16-key blocks) and the transpose reads of V fetch exactly those rows.
// Backward (flash-style recompute from the saved log-sum-exp): one workgroup per (batch, head), key tiles outer, query
// tiles inner; P~ and dS go through LDS once (bf16, [key][q]) because dK/dV reduce over queries while dQ reduces over
// keys; all three gradients are again produced transposed so each lane stores 4 consecutive head-dim elements.
//
// Replaces BertSelfAttention.forward core (vilmodel.py:101-126) and BertOutAttention.forward core (:327-348).
#include "common.h"

namespace {

constexpr int T64 = 64;    // tile edge (queries / keys) == d_head
constexpr int AST = 72;    // LDS row stride in bf16 elements (144 B)
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct Attn16Args {
  hamt_attn_desc d;
  const void *q, *k, *v, *o, *d_o;
  const float* mask;
  void *out, *dq, *dk, *dv;
  float* lse;
  const uint64_t* rng;
  // packed layouts (hamt_attn_varlen_*): sequence b owns rows [cu_q[b], cu_q[b + 1]) of q / o / d_o / dq and / or rows [cu_k[b], cu_k[b + 1])
  // of k / v / dk / dv instead of a fixed stride (nullptr: fixed stride b * Sq resp. b * Sk).  Query sequences b >= n_pairs have no keys
  // (filler sequences of a bucketed packed batch): their outputs / query gradients are zeros.
  const int* cu_q;
  const int* cu_k;
  int n_pairs;
  const int* pair;   // optional [B]: the key-side index of query sequence b (the key sample at a fixed stride, or the entry of cu_k), < 0 = a
                     // filler; nullptr: b itself, fillers = the sequences behind n_pairs
};

template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&f)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&f)[8]) {
  const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&f)[8]) {
  const uint4 x = *(const uint4*)p;
  const uint32_t u[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(u[j] << 16); f[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}
template <typename T> __device__ __forceinline__ void st4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void st4<float>(float* p, float a, float b, float c, float d) { *(float4*)p = make_float4(a, b, c, d); }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float a, float b, float c, float d) { *(uint2*)p = make_uint2(pack_bf2(a, b), pack_bf2(c, d)); }
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 x = *(const uint2*)p;
  return make_float4(__uint_as_float(x.x << 16), __uint_as_float(x.x & 0xffff0000u), __uint_as_float(x.y << 16), __uint_as_float(x.y & 0xffff0000u));
}

// Stage rows [r0, r0+64) x 64 columns (one head) into lds[64][AST] as bf16; rows >= rlim are zero.
// thread t: row t>>2, columns (t&3)*16 .. +15.  `keep` returns the 16 fp32 values (for the delta computation).
template <typename T>
__device__ __forceinline__ void stage16(const T* base, int ld, int r0, int rlim, bf16_t* lds, int t, float (&keep)[16]) {
  const int r = t >> 2, c = (t & 3) * 16;
  float a[8], b[8];
  if (r0 + r < rlim) {
    ld8<T>(base + (size_t)(r0 + r) * ld + c, a);
    ld8<T>(base + (size_t)(r0 + r) * ld + c + 8, b);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = 0.f; b[i] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { keep[i] = a[i]; keep[8 + i] = b[i]; }
  *(uint4*)(lds + r * AST + c) = pack8(a);
  *(uint4*)(lds + r * AST + c + 8) = pack8(b);
}

// row fragment: 8 consecutive k (columns) of row r: A[i=r][k] or B[k][j=r] for K-contiguous operands
__device__ __forceinline__ bf16x8 rfrag(const bf16_t* lds, int r, int k8) {
  union { uint4 u; bf16x8 v; } f;
  f.u = *(const uint4*)(lds + r * AST + k8);
  return f.v;
}
// transposed fragment: column (c16 + lane&15), rows ra..ra+3 and rb..rb+3 (the lane group's 8 k values)
__device__ __forceinline__ bf16x8 tfrag(const bf16_t* lds, int ra, int rb, int c16, int lane) {
  union { bf16x8 v; s16x4 h[2]; } f;
  const int i = lane & 15, col = c16 + (i & 3) * 4, dr = i >> 2;
  f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (ra + dr) * AST + col));
  f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (rb + dr) * AST + col));
  return f.v;
}
__device__ __forceinline__ bf16x8 pack_frag(const float (&lo)[4], const float (&hi)[4]) {
  union { uint4 u; bf16x8 v; } f;
  f.u = make_uint4(pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3]));
  return f.v;
}
__device__ __forceinline__ float xg_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float xg_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// =================================================================================================
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void attn16_fwd_kernel(Attn16Args a) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[T64 * AST], Vs[T64 * AST];
  const hamt_attn_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * T64, h = blockIdx.y, b = blockIdx.z;
  const TI* Q = (const TI*)a.q + (size_t)b * d.Sq * d.ldq + h * 64;
  const TI* K = (const TI*)a.k + (size_t)b * d.Sk * d.ldk + h * 64;
  const TI* V = (const TI*)a.v + (size_t)b * d.Sk * d.ldv + h * 64;
  const bool active = q0 + 16 * w < d.Sq;
  const int qrow = q0 + 16 * w + l15;                  // the ONE query row this lane owns
  const bool qok = qrow < d.Sq;
  // Q fragments for both 32-wide k-steps, straight from global (B operand: lane (j = q, g) holds Q[q][32s+8g..+7])
  bf16x8 qf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float f[8];
    if (active && qok) ld8<TI>(Q + (size_t)qrow * d.ldq + 32 * s + 8 * g, f);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = 0.f;
    }
    union { uint4 u; bf16x8 v; } c;
    c.u = pack8(f);
    qf[s] = c.v;
  }
  const RngKey key = rng_key(a.rng, d.call_id);
  const float inv_keep = d.p_drop > 0.f ? 1.0f / (1.0f - d.p_drop) : 1.0f;
  const uint32_t rowh = hamt_mix32((uint32_t)((b * d.heads + h) * d.Sq + qrow) ^ key.k0);   // same mask stream as the single-pass kernels
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 of[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) of[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float keep[16];

  for (int k0 = 0; k0 < d.Sk; k0 += T64) {
    __syncthreads();
    stage16<TI>(K, d.ldk, k0, d.Sk, Ks, t, keep);
    stage16<TI>(V, d.ldv, k0, d.Sk, Vs, t, keep);
    __syncthreads();
    const int nkb = (min(T64, d.Sk - k0) + 15) >> 4;
    if (active) {
      f32x4 sf[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        sf[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kb < nkb) {
#pragma unroll
          for (int s = 0; s < 2; ++s) sf[kb] = MFMA16(rfrag(Ks, 16 * kb + l15, 32 * s + 8 * g), qf[s], sf[kb]);
        }
      }
      // sf[kb][r] = S[q = qrow][key = k0 + 16kb + 4g + r]
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = k0 + 16 * kb + 4 * g + r;
          const bool kval = kb < nkb && kk < d.Sk;
          const float mk = (kval && a.mask) ? a.mask[(size_t)b * d.Sk + kk] : 0.f;
          sf[kb][r] = kval ? sf[kb][r] * d.scale + mk : -INFINITY;
          mx = fmaxf(mx, sf[kb][r]);
        }
      const float mn = fmaxf(m_run, xg_max(mx));
      const float alpha = __expf(m_run - mn);
      m_run = mn;
      float rs = 0.f;
      float p[4][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        float ds4[4] = {1.f, 1.f, 1.f, 1.f};
        if (d.p_drop > 0.f) drop_scale4(key, rowh, (uint32_t)((k0 + 16 * kb + 4 * g) >> 2), d.p_drop, inv_keep, ds4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = (kb < nkb) ? expf(sf[kb][r] - mn) : 0.f;
          rs += e;
          p[kb][r] = e * ds4[r];
        }
      }
      l_run = l_run * alpha + xg_sum(rs);
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) of[db][r] *= alpha;
      // O^T += V^T P^T : k-step s covers key blocks 2s, 2s+1; this lane group's k = keys 4g..4g+3 of each block
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (2 * s < nkb) {
          const bf16x8 pf = pack_frag(p[2 * s], p[2 * s + 1]);
#pragma unroll
          for (int db = 0; db < 4; ++db)
            of[db] = MFMA16(tfrag(Vs, 32 * s + 4 * g, 32 * s + 16 + 4 * g, 16 * db, lane), pf, of[db]);
        }
      }
    }
  }
  if (active && qok) {
    TO* O = (TO*)a.out + (size_t)b * d.Sq * d.ldo + h * 64 + (size_t)qrow * d.ldo;
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int db = 0; db < 4; ++db) st4<TO>(O + 16 * db + 4 * g, of[db][0] * inv, of[db][1] * inv, of[db][2] * inv, of[db][3] * inv);
    if (g == 0) a.lse[((size_t)b * d.heads + h) * d.Sq + qrow] = m_run + logf(l_run);
  }
}

// =================================================================================================
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void attn16_bwd_kernel(Attn16Args a) {
  __shared__ __attribute__((aligned(16))) bf16_t Qs[T64 * AST], Ks[T64 * AST], Vs[T64 * AST], dOs[T64 * AST], Pt[T64 * AST], dSt[T64 * AST];
  __shared__ float dpart[4][T64], lse_s[T64], delta_s[T64];
  const hamt_attn_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  const TI* Q = (const TI*)a.q + (size_t)b * d.Sq * d.ldq + h * 64;
  const TI* K = (const TI*)a.k + (size_t)b * d.Sk * d.ldk + h * 64;
  const TI* V = (const TI*)a.v + (size_t)b * d.Sk * d.ldv + h * 64;
  const TO* O = (const TO*)a.o + (size_t)b * d.Sq * d.ldo + h * 64;
  const TO* dO = (const TO*)a.d_o + (size_t)b * d.Sq * d.ldo + h * 64;
  TI* dQ = (TI*)a.dq + (size_t)b * d.Sq * d.ldq + h * 64;
  TI* dK = (TI*)a.dk + (size_t)b * d.Sk * d.ldk + h * 64;
  TI* dV = (TI*)a.dv + (size_t)b * d.Sk * d.ldv + h * 64;
  const RngKey key = rng_key(a.rng, d.call_id);
  const float inv_keep = d.p_drop > 0.f ? 1.0f / (1.0f - d.p_drop) : 1.0f;
  float keep[16], okeep[16];

  for (int k0 = 0; k0 < d.Sk; k0 += T64) {
    const int nkb = (min(T64, d.Sk - k0) + 15) >> 4;
    const bool kact = k0 + 16 * w < d.Sk;            // this wave owns key rows [16w, 16w+16) for dK / dV
    f32x4 dkf[4], dvf[4];                            // transposed accumulators: [d = 16db+4g+r][key = 16w + l15]
#pragma unroll
    for (int i = 0; i < 4; ++i) { dkf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();
    stage16<TI>(K, d.ldk, k0, d.Sk, Ks, t, keep);
    stage16<TI>(V, d.ldv, k0, d.Sk, Vs, t, keep);
    for (int q0 = 0; q0 < d.Sq; q0 += T64) {
      const int nqb = (min(T64, d.Sq - q0) + 15) >> 4;
      const bool qact = q0 + 16 * w < d.Sq;
      __syncthreads();
      stage16<TI>(Q, d.ldq, q0, d.Sq, Qs, t, keep);
      stage16<TO>(dO, d.ldo, q0, d.Sq, dOs, t, keep);
      {  // delta = rowsum(dO * O) (fp32, from the un-rounded values) and the saved lse
        const int r = t >> 2, c = (t & 3) * 16;
        float acc = 0.f;
        if (q0 + r < d.Sq) {
          float a8[8], b8[8];
          ld8<TO>(O + (size_t)(q0 + r) * d.ldo + c, a8);
          ld8<TO>(O + (size_t)(q0 + r) * d.ldo + c + 8, b8);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc += keep[i] * a8[i] + keep[8 + i] * b8[i];
        }
        dpart[t & 3][r] = acc;
        if (t < T64) lse_s[t] = (q0 + t < d.Sq) ? a.lse[((size_t)b * d.heads + h) * d.Sq + q0 + t] : 0.f;
      }
      __syncthreads();
      if (t < T64) delta_s[t] = (dpart[0][t] + dpart[1][t]) + (dpart[2][t] + dpart[3][t]);
      __syncthreads();
      if (qact) {   // phase A: this wave's 16 queries x the tile's keys; lane owns query ql = 16w + l15
        const int ql = 16 * w + l15, qq = q0 + ql;
        bf16x8 qf[2], df[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) { qf[s] = rfrag(Qs, ql, 32 * s + 8 * g); df[s] = rfrag(dOs, ql, 32 * s + 8 * g); }
        const float lse_q = lse_s[ql], delta_q = delta_s[ql];
        const uint32_t rowh = hamt_mix32((uint32_t)((b * d.heads + h) * d.Sq + qq) ^ key.k0);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          if (kb >= nkb && kb < ((nkb + 1) & ~1)) {   // the 32-wide k-step of dQ also reads this (all-padding) key block
#pragma unroll
            for (int r = 0; r < 4; ++r) { Pt[(16 * kb + 4 * g + r) * AST + ql] = 0; dSt[(16 * kb + 4 * g + r) * AST + ql] = 0; }
          }
          if (kb < nkb) {
            f32x4 sf = (f32x4){0.f, 0.f, 0.f, 0.f}, dpf = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              sf = MFMA16(rfrag(Ks, 16 * kb + l15, 32 * s + 8 * g), qf[s], sf);      // S^T[key][q]
              dpf = MFMA16(rfrag(Vs, 16 * kb + l15, 32 * s + 8 * g), df[s], dpf);    // dP^T[key][q] = V dO^T
            }
            float ds4[4] = {1.f, 1.f, 1.f, 1.f};
            if (d.p_drop > 0.f) drop_scale4(key, rowh, (uint32_t)((k0 + 16 * kb + 4 * g) >> 2), d.p_drop, inv_keep, ds4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kl = 16 * kb + 4 * g + r, kk = k0 + kl;
              float pd = 0.f, ds = 0.f;
              if (kk < d.Sk && qq < d.Sq) {
                const float mk = a.mask ? a.mask[(size_t)b * d.Sk + kk] : 0.f;
                const float p = expf(sf[r] * d.scale + mk - lse_q);
                const float dsc = ds4[r];
                pd = p * dsc;
                ds = p * (dpf[r] * dsc - delta_q) * d.scale;
              }
              Pt[kl * AST + ql] = f2bf(pd);
              dSt[kl * AST + ql] = f2bf(ds);
            }
          }
        }
      }
      else if (w < ((nqb + 1) & ~1)) {   // idle query block inside the last 32-wide k-step of dK/dV: zero columns
        const int ql = 16 * w + l15;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          if (kb < ((nkb + 1) & ~1)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { Pt[(16 * kb + 4 * g + r) * AST + ql] = 0; dSt[(16 * kb + 4 * g + r) * AST + ql] = 0; }
          }
      }
      __syncthreads();
      if (kact) {   // dV^T[d][key] += dO^T P~ ; dK^T[d][key] += Q^T dS   (reduction over the tile's queries)
        for (int s = 0; 2 * s < nqb; ++s) {
          const bf16x8 pb = rfrag(Pt, 16 * w + l15, 32 * s + 8 * g), sb = rfrag(dSt, 16 * w + l15, 32 * s + 8 * g);
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            dvf[db] = MFMA16(tfrag(dOs, 32 * s + 8 * g, 32 * s + 8 * g + 4, 16 * db, lane), pb, dvf[db]);
            dkf[db] = MFMA16(tfrag(Qs, 32 * s + 8 * g, 32 * s + 8 * g + 4, 16 * db, lane), sb, dkf[db]);
          }
        }
      }
      if (qact) {   // dQ^T[d][q] (+)= K^T dS^T  (reduction over the tile's keys)
        f32x4 dqf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dqf[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; 2 * s < nkb; ++s) {
          const bf16x8 sb = tfrag(dSt, 32 * s + 8 * g, 32 * s + 8 * g + 4, 16 * w, lane);   // B[k = key][j = q]
#pragma unroll
          for (int db = 0; db < 4; ++db) dqf[db] = MFMA16(tfrag(Ks, 32 * s + 8 * g, 32 * s + 8 * g + 4, 16 * db, lane), sb, dqf[db]);
        }
        const int qq = q0 + 16 * w + l15;
        if (qq < d.Sq) {
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            TI* p = dQ + (size_t)qq * d.ldq + 16 * db + 4 * g;
            float4 v = make_float4(dqf[db][0], dqf[db][1], dqf[db][2], dqf[db][3]);
            if (k0 != 0) { const float4 o = ld4<TI>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            st4<TI>(p, v.x, v.y, v.z, v.w);
          }
        }
      }
    }
    if (kact) {
      const int kk = k0 + 16 * w + l15;
      if (kk < d.Sk) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          st4<TI>(dK + (size_t)kk * d.ldk + 16 * db + 4 * g, dkf[db][0], dkf[db][1], dkf[db][2], dkf[db][3]);
          st4<TI>(dV + (size_t)kk * d.ldv + 16 * db + 4 * g, dvf[db][0], dvf[db][1], dvf[db][2], dvf[db][3]);
        }
      }
    }
  }
}

// =================================================================================================
// Short-sequence kernels (Sq, Sk <= 128: every attention of HAMT -- 80 instruction tokens, <= 37 views, a handful of
// history steps).  One workgroup per (batch, head) with one wave per 16 query rows (and, in the backward, per 16 key
// rows): K/V (and Q/dO) are staged ONCE, the softmax is a single pass over all keys held in registers, and work is
// counted in 16-row blocks instead of 64-row tiles -- for 80 tokens 5x5 blocks instead of the 8x8 the tiled kernels
// above touch, and no second staging of K/V for the 16-row remainder tile.  Same products, same lane ownership, same
// dropout indexing as the tiled kernels (they remain the path for longer sequences).
__device__ __forceinline__ bf16x8 zero_frag() { union { uint4 u; bf16x8 v; } f; f.u = make_uint4(0, 0, 0, 0); return f.v; }
// transposed fragment with a row limit: a 4-row group (ra.., rb..) outside [0, rlim) reads as zeros.  Every lane issues
// the transpose read (it is a cross-lane operation: the address is clamped instead of branching around it) and the
// out-of-range groups are zeroed afterwards.
__device__ __forceinline__ bf16x8 tfrag_lim(const bf16_t* lds, int stride, int ra, int rb, int rlim, int c16, int lane) {
  union { bf16x8 v; s16x4 h[2]; } f;
  const int i = lane & 15, col = c16 + (i & 3) * 4, dr = i >> 2;
  const bool oka = ra < rlim, okb = rb < rlim;
  f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + ((oka ? ra : 0) + dr) * stride + col));
  f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + ((okb ? rb : 0) + dr) * stride + col));
  const s16x4 z = {0, 0, 0, 0};
  if (!oka) f.h[0] = z;
  if (!okb) f.h[1] = z;
  return f.v;
}
__device__ __forceinline__ bf16x8 rfrag_lim(const bf16_t* lds, int stride, int r, int k8, int klim) {
  union { uint4 u; bf16x8 v; } f;
  f.u = make_uint4(0, 0, 0, 0);
  if (k8 < klim) f.u = *(const uint4*)(lds + r * stride + k8);
  return f.v;
}
// Staging in two phases so that every global load of a matrix (or of several matrices) is in flight before the first
// LDS store waits for one: raw 8-element pieces first (8 threads per row, IT pieces per thread), conversion + store after.
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { uint4 u; };
template <> struct Raw8<float> { float4 a, b; };
__device__ __forceinline__ void raw_load(const bf16_t* p, Raw8<bf16_t>& r) { r.u = *(const uint4*)p; }
__device__ __forceinline__ void raw_load(const float* p, Raw8<float>& r) { r.a = ((const float4*)p)[0]; r.b = ((const float4*)p)[1]; }
__device__ __forceinline__ void raw_zero(Raw8<bf16_t>& r) { r.u = make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void raw_zero(Raw8<float>& r) { r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; }
__device__ __forceinline__ uint4 raw_bf16(const Raw8<bf16_t>& r) { return r.u; }
__device__ __forceinline__ uint4 raw_bf16(const Raw8<float>& r) {
  return make_uint4(pack_bf2(r.a.x, r.a.y), pack_bf2(r.a.z, r.a.w), pack_bf2(r.b.x, r.b.y), pack_bf2(r.b.z, r.b.w));
}
__device__ __forceinline__ void raw_f32(const Raw8<bf16_t>& r, float (&f)[8]) {
  const uint32_t u[4] = {r.u.x, r.u.y, r.u.z, r.u.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(u[j] << 16); f[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u); }
}
__device__ __forceinline__ void raw_f32(const Raw8<float>& r, float (&f)[8]) {
  f[0] = r.a.x; f[1] = r.a.y; f[2] = r.a.z; f[3] = r.a.w; f[4] = r.b.x; f[5] = r.b.y; f[6] = r.b.z; f[7] = r.b.w;
}
// rows [0, rows16) of one head: piece i of thread t is row (t >> 3) + i * (nt >> 3), columns (t & 7) * 8 .. +7
template <typename T, int IT>
__device__ __forceinline__ void rows_load(const T* base, int ld, int rlim, int t, int nt, Raw8<T> (&raw)[IT]) {
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int r = (t >> 3) + i * (nt >> 3);
    // unconditional load from a clamped row, zeroed afterwards: a branch around the load would make hipcc wait for each
    // piece separately (cdna_hip_programming.md, "register or load" trap) instead of keeping all of them in flight
    raw_load(base + (size_t)(r < rlim ? r : rlim - 1) * ld + (t & 7) * 8, raw[i]);
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int r = (t >> 3) + i * (nt >> 3);
    if (r >= rlim) raw_zero(raw[i]);
  }
}
template <typename T, int IT>
__device__ __forceinline__ void rows_store(const Raw8<T> (&raw)[IT], int rows16, bf16_t* lds, int t, int nt) {
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int r = (t >> 3) + i * (nt >> 3);
    if (r < rows16) *(uint4*)(lds + r * AST + (t & 7) * 8) = raw_bf16(raw[i]);
  }
}

// bf16 MFMA fragment of 8 consecutive head-dim elements of one global row (Q): no conversion for bf16 sources
template <typename T> __device__ __forceinline__ bf16x8 gfrag(const T* p) {
  union { uint4 u; bf16x8 v; } c;
  if constexpr (sizeof(T) == 2) c.u = *(const uint4*)p;
  else { float f[8]; ld8<T>(p, f); c.u = pack8(f); }
  return c.v;
}
// additive key mask of one (batch) row into LDS: mask value for keys < Sk (0 without a mask), -inf for the padding keys,
// so that the score update is ONE fma per element and padding keys get probability exactly 0
__device__ __forceinline__ void stage_mask(const float* mask, int b, int Sk, int sk16, float* mk_s, int t, int nt) {
  for (int i = t; i < sk16; i += nt) mk_s[i] = i < Sk ? (mask ? mask[(size_t)b * Sk + i] : 0.f) : -INFINITY;
}

// KB: number of 16-key blocks staged and processed (even, <= 16) >= ceil(Sk / 16); IT: staging pieces per thread and matrix.
// One workgroup per (head, batch row, chunk of 128 queries): any Sq, Sk <= 256.
template <typename TI, typename TO, int KB, int IT>
__global__ __launch_bounds__(512) void attn_s128_fwd_kernel(Attn16Args a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  const hamt_attn_desc& d = a.d;
  const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  // rows and lengths of this sample: fixed stride (padded batch), or its own row range of a packed batch (every key is real)
  int Sq = d.Sq, Sk = d.Sk;
  size_t qrow0 = (size_t)b * d.Sq, krow0 = (size_t)b * d.Sk;
  if (a.cu_q) { const int c0 = a.cu_q[b]; Sq = a.cu_q[b + 1] - c0; qrow0 = (size_t)c0; }
  if (Sq <= 0) return;                                 // (whole workgroup: an empty slot of a bucketed batch)
  const int kb_ = a.pair ? a.pair[b] : b;              // the key side of this query sequence
  const bool filler = a.pair ? kb_ < 0 : ((a.cu_q || a.cu_k) && b >= a.n_pairs);
  if (!filler) {
    krow0 = (size_t)kb_ * d.Sk;
    if (a.cu_k) { const int c0 = a.cu_k[kb_]; Sk = a.cu_k[kb_ + 1] - c0; krow0 = (size_t)c0; }
  }
  if (filler || Sk <= 0) {   // a filler query sequence (or an empty key sequence): no keys -> zero output rows, lse 0
    TO* O0 = (TO*)a.out + qrow0 * d.ldo + h * 64;
    for (int i = t; i < Sq * 8; i += nt) st4<TO>(O0 + (size_t)(i >> 3) * d.ldo + (i & 7) * 8, 0.f, 0.f, 0.f, 0.f), st4<TO>(O0 + (size_t)(i >> 3) * d.ldo + (i & 7) * 8 + 4, 0.f, 0.f, 0.f, 0.f);
    for (int i = t; i < Sq; i += nt) a.lse[((size_t)b * d.heads + h) * d.Sq + i] = 0.f;
    return;
  }
  constexpr int SKP = KB * 16;                         // staged key rows: zeros (K, V) and -inf (mask) beyond Sk, so the
  bf16_t* Ks = sm;                                     // whole kernel is branch-free in the key dimension
  bf16_t* Vs = sm + SKP * AST;
  float* mk_s = (float*)(Vs + SKP * AST);
  const TI* Q = (const TI*)a.q + qrow0 * d.ldq + h * 64;
  const TI* K = (const TI*)a.k + krow0 * d.ldk + h * 64;
  const TI* V = (const TI*)a.v + krow0 * d.ldv + h * 64;
  Raw8<TI> rk[IT], rv[IT];                             // the launcher guarantees 8 * SKP <= IT * nt
  rows_load<TI, IT>(K, d.ldk, Sk, t, nt, rk);
  rows_load<TI, IT>(V, d.ldv, Sk, t, nt, rv);
  const int q0 = 128 * blockIdx.z;
  const int qrow = q0 + 16 * w + l15;                  // the ONE query row this lane owns
  const bool qok = qrow < Sq;
  bf16x8 qf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)   // clamped, not branched (rows >= Sq are never stored)
    qf[s] = gfrag<TI>(Q + (size_t)(qok ? qrow : Sq - 1) * d.ldq + 32 * s + 8 * g);
  stage_mask(a.cu_k ? nullptr : a.mask, kb_, Sk, SKP, mk_s, t, nt);
  rows_store<TI, IT>(rk, SKP, Ks, t, nt);
  rows_store<TI, IT>(rv, SKP, Vs, t, nt);
  __syncthreads();
  if (q0 + 16 * w >= Sq || w >= 8) return;             // waves that only helped staging (no barrier below)
  const RngKey key = rng_key(a.rng, d.call_id);
  const float inv_keep = d.p_drop > 0.f ? 1.0f / (1.0f - d.p_drop) : 1.0f;
  const uint32_t rowh = hamt_mix32((uint32_t)((b * d.heads + h) * d.Sq + qrow) ^ key.k0);
  f32x4 sf[KB];
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {                    // sf[kb][r] = S[q = qrow][key = 16kb + 4g + r]
    sf[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) sf[kb] = MFMA16(rfrag(Ks, 16 * kb + l15, 32 * s + 8 * g), qf[s], sf[kb]);
    const float4 mk = *(const float4*)(mk_s + 16 * kb + 4 * g);
    sf[kb][0] = fmaf(sf[kb][0], d.scale, mk.x); sf[kb][1] = fmaf(sf[kb][1], d.scale, mk.y);
    sf[kb][2] = fmaf(sf[kb][2], d.scale, mk.z); sf[kb][3] = fmaf(sf[kb][3], d.scale, mk.w);
    mx = fmaxf(fmaxf(mx, fmaxf(sf[kb][0], sf[kb][1])), fmaxf(sf[kb][2], sf[kb][3]));
  }
  const float mn = xg_max(mx);
  float rs = 0.f;
  f32x4 of[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) of[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // O^T = V^T P^T : k-step s covers key blocks 2s, 2s+1; this lane group's k = keys 4g..4g+3 of each block
#pragma unroll
  for (int s = 0; s < KB / 2; ++s) {
    float p[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[e][r] = __expf(sf[2 * s + e][r] - mn); rs += p[e][r]; }   // exp(-inf) = 0: padding keys
      if (d.p_drop > 0.f) {
        float ds4[4];
        drop_scale4(key, rowh, (uint32_t)(4 * (2 * s + e) + g), d.p_drop, inv_keep, ds4);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[e][r] *= ds4[r];
      }
    }
    const bf16x8 pf = pack_frag(p[0], p[1]);
#pragma unroll
    for (int db = 0; db < 4; ++db) of[db] = MFMA16(tfrag(Vs, 32 * s + 4 * g, 32 * s + 16 + 4 * g, 16 * db, lane), pf, of[db]);
  }
  const float l_run = xg_sum(rs);
  if (qok) {
    TO* O = (TO*)a.out + qrow0 * d.ldo + h * 64 + (size_t)qrow * d.ldo;
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int db = 0; db < 4; ++db) st4<TO>(O + 16 * db + 4 * g, of[db][0] * inv, of[db][1] * inv, of[db][2] * inv, of[db][3] * inv);
    if (g == 0) a.lse[((size_t)b * d.heads + h) * d.Sq + qrow] = mn + __logf(l_run);
  }
}

template <typename TI, typename TO, int NB>   // NB: compile-time bound on the 16-row blocks of queries and of keys
__global__ __launch_bounds__(512) void attn_s128_bwd_kernel(Attn16Args a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  const hamt_attn_desc& d = a.d;
  const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  int Sq = d.Sq, Sk = d.Sk;                            // (as in the forward kernel: fixed stride, or this sample's rows of a packed batch)
  size_t qrow0 = (size_t)b * d.Sq, krow0 = (size_t)b * d.Sk;
  if (a.cu_q) { const int c0 = a.cu_q[b]; Sq = a.cu_q[b + 1] - c0; qrow0 = (size_t)c0; }
  if (Sq <= 0) return;
  const int kb_ = a.pair ? a.pair[b] : b;              // the key side of this query sequence
  const bool filler = a.pair ? kb_ < 0 : ((a.cu_q || a.cu_k) && b >= a.n_pairs);
  if (!filler) {
    krow0 = (size_t)kb_ * d.Sk;
    if (a.cu_k) { const int c0 = a.cu_k[kb_]; Sk = a.cu_k[kb_ + 1] - c0; krow0 = (size_t)c0; }
  }
  if (filler || Sk <= 0) {   // a filler query sequence: zero query gradients, no keys to give gradients to
    TI* dQ0 = (TI*)a.dq + qrow0 * d.ldq + h * 64;
    for (int i = t; i < Sq * 8; i += nt) { st4<TI>(dQ0 + (size_t)(i >> 3) * d.ldq + (i & 7) * 8, 0.f, 0.f, 0.f, 0.f); st4<TI>(dQ0 + (size_t)(i >> 3) * d.ldq + (i & 7) * 8 + 4, 0.f, 0.f, 0.f, 0.f); }
    return;
  }
  const int nqb = (Sq + 15) >> 4, nkb = (Sk + 15) >> 4, sq16 = nqb * 16, sk16 = nkb * 16, PST = sq16 + 8;
  bf16_t* Qs = sm;
  bf16_t* dOs = Qs + sq16 * AST;
  bf16_t* Ks = dOs + sq16 * AST;
  bf16_t* Vs = Ks + sk16 * AST;
  bf16_t* Pt = Vs + sk16 * AST;                        // [key][q] (row stride PST): P~ and dS of the whole head
  bf16_t* dSt = Pt + sk16 * PST;
  float* lse_s = (float*)(dSt + sk16 * PST);
  float* delta_s = lse_s + sq16;
  float* mk_s = delta_s + sq16;
  const TI* Q = (const TI*)a.q + qrow0 * d.ldq + h * 64;
  const TI* K = (const TI*)a.k + krow0 * d.ldk + h * 64;
  const TI* V = (const TI*)a.v + krow0 * d.ldv + h * 64;
  const TO* O = (const TO*)a.o + qrow0 * d.ldo + h * 64;
  const TO* dO = (const TO*)a.d_o + qrow0 * d.ldo + h * 64;
  {  // all five matrices in flight at once (nt = 64 * max(nqb, nkb) => 2 pieces per thread and matrix)
    Raw8<TI> rq[2], rk[2], rv[2];
    Raw8<TO> rdo[2], ro[2];
    rows_load<TI, 2>(Q, d.ldq, Sq, t, nt, rq);
    rows_load<TO, 2>(dO, d.ldo, Sq, t, nt, rdo);
    rows_load<TO, 2>(O, d.ldo, Sq, t, nt, ro);
    rows_load<TI, 2>(K, d.ldk, Sk, t, nt, rk);
    rows_load<TI, 2>(V, d.ldv, Sk, t, nt, rv);
    stage_mask(a.cu_k ? nullptr : a.mask, kb_, Sk, sk16, mk_s, t, nt);
    for (int i = t; i < sk16 * PST / 8; i += nt) { ((uint4*)Pt)[i] = make_uint4(0, 0, 0, 0); ((uint4*)dSt)[i] = make_uint4(0, 0, 0, 0); }
    rows_store<TI, 2>(rq, sq16, Qs, t, nt);
    rows_store<TO, 2>(rdo, sq16, dOs, t, nt);
    // delta = rowsum(dO * O) in fp32 from the un-rounded values (8 threads per row) and the saved lse
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (t >> 3) + i * (nt >> 3);
      float f[8], o8[8], acc = 0.f;
      raw_f32(rdo[i], f);
      raw_f32(ro[i], o8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += f[j] * o8[j];
      acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
      if ((t & 7) == 0 && r < sq16) { delta_s[r] = acc; lse_s[r] = r < Sq ? a.lse[((size_t)b * d.heads + h) * d.Sq + r] : 0.f; }
    }
    rows_store<TI, 2>(rk, sk16, Ks, t, nt);
    rows_store<TI, 2>(rv, sk16, Vs, t, nt);
  }
  __syncthreads();
  const RngKey key = rng_key(a.rng, d.call_id);
  const float inv_keep = d.p_drop > 0.f ? 1.0f / (1.0f - d.p_drop) : 1.0f;
  if (w < nqb) {   // phase A: this wave's 16 queries x all keys; lane owns query ql = 16w + l15
    const int ql = 16 * w + l15;
    bf16x8 qf[2], df[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { qf[s] = rfrag(Qs, ql, 32 * s + 8 * g); df[s] = rfrag(dOs, ql, 32 * s + 8 * g); }
    const float lse_q = lse_s[ql], delta_q = delta_s[ql];
    const uint32_t rowh = hamt_mix32((uint32_t)((b * d.heads + h) * d.Sq + ql) ^ key.k0);
    const bool qv = ql < Sq;                         // padding queries keep their (zero) columns; the MFMAs need every lane
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      if (kb < nkb) {
        f32x4 sf = (f32x4){0.f, 0.f, 0.f, 0.f}, dpf = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          sf = MFMA16(rfrag(Ks, 16 * kb + l15, 32 * s + 8 * g), qf[s], sf);      // S^T[key][q]
          dpf = MFMA16(rfrag(Vs, 16 * kb + l15, 32 * s + 8 * g), df[s], dpf);    // dP^T[key][q] = V dO^T
        }
        const float4 mk = *(const float4*)(mk_s + 16 * kb + 4 * g);
        const float mk4[4] = {mk.x, mk.y, mk.z, mk.w};
        float ds4[4] = {1.f, 1.f, 1.f, 1.f};
        if (d.p_drop > 0.f) drop_scale4(key, rowh, (uint32_t)(4 * kb + g), d.p_drop, inv_keep, ds4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = 16 * kb + 4 * g + r;
          const float pr = __expf(fmaf(sf[r], d.scale, mk4[r]) - lse_q);          // 0 for the padding keys (mask = -inf)
          if (qv) {
            Pt[kk * PST + ql] = f2bf(pr * ds4[r]);
            dSt[kk * PST + ql] = f2bf(pr * (dpf[r] * ds4[r] - delta_q) * d.scale);
          }
        }
      }
    }
  }
  __syncthreads();
  if (w < nkb) {   // dV^T[d][key] = dO^T P~ ; dK^T[d][key] = Q^T dS   (reduction over the queries), key = 16w + l15
    f32x4 dkf[4], dvf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dkf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < NB / 2; ++s) {
      if (2 * s < nqb) {
        const bf16x8 pb = rfrag_lim(Pt, PST, 16 * w + l15, 32 * s + 8 * g, sq16), sb = rfrag_lim(dSt, PST, 16 * w + l15, 32 * s + 8 * g, sq16);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          dvf[db] = MFMA16(tfrag_lim(dOs, AST, 32 * s + 8 * g, 32 * s + 8 * g + 4, sq16, 16 * db, lane), pb, dvf[db]);
          dkf[db] = MFMA16(tfrag_lim(Qs, AST, 32 * s + 8 * g, 32 * s + 8 * g + 4, sq16, 16 * db, lane), sb, dkf[db]);
        }
      }
    }
    const int kk = 16 * w + l15;
    if (kk < Sk) {
      TI* dK = (TI*)a.dk + krow0 * d.ldk + h * 64 + (size_t)kk * d.ldk;
      TI* dV = (TI*)a.dv + krow0 * d.ldv + h * 64 + (size_t)kk * d.ldv;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        st4<TI>(dK + 16 * db + 4 * g, dkf[db][0], dkf[db][1], dkf[db][2], dkf[db][3]);
        st4<TI>(dV + 16 * db + 4 * g, dvf[db][0], dvf[db][1], dvf[db][2], dvf[db][3]);
      }
    }
  }
  if (w < nqb) {   // dQ^T[d][q] = K^T dS^T  (reduction over the keys), q = 16w + l15
    f32x4 dqf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dqf[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NB / 2; ++s) {
      if (2 * s < nkb) {
        const bf16x8 sb = tfrag_lim(dSt, PST, 32 * s + 8 * g, 32 * s + 8 * g + 4, sk16, 16 * w, lane);   // B[k = key][j = q]
#pragma unroll
        for (int db = 0; db < 4; ++db) dqf[db] = MFMA16(tfrag_lim(Ks, AST, 32 * s + 8 * g, 32 * s + 8 * g + 4, sk16, 16 * db, lane), sb, dqf[db]);
      }
    }
    const int qq = 16 * w + l15;
    if (qq < Sq) {
      TI* dQ = (TI*)a.dq + qrow0 * d.ldq + h * 64 + (size_t)qq * d.ldq;
#pragma unroll
      for (int db = 0; db < 4; ++db) st4<TI>(dQ + 16 * db + 4 * g, dqf[db][0], dqf[db][1], dqf[db][2], dqf[db][3]);
    }
  }
}

// Single-pass backward for sequences of 129 .. 224 tokens (ViT-B/16: S = 197, vision_transformer.py:154-178): one workgroup per
// head with one wave per 16 rows, NOTHING but the four operands in LDS (130 KB at 224 tokens: the P~ / dS arrays of the kernel
// above would not fit next to them).  The probabilities are computed twice, once per ownership:
//   phase A, wave = 16 queries:  S^T, dP^T [key][q] -> dS^T stays in registers as the B operand of dQ^T = K^T dS^T (the forward
//            kernel's O^T = V^T P^T idiom: the MFMA output layout IS the operand layout when the reduction index is permuted the
//            same way on both sides);
//   phase B, wave = 16 keys:     S, dP [q][key] again (4 MFMAs per 16 x 16 block, exp from the saved lse: no row reduction to
//            redo) -> P~ and dS in registers as the B operands of dV^T = dO^T P~ and dK^T = Q^T dS.
// No barrier between the phases, no LDS traffic but fragment reads.  Same dropout stream as every other attention kernel: phase B
// redoes the hash of (query row, group of 4 keys) and picks its key's 16 bits.  (For <= 128 tokens this form was measured against
// the kernel above and lost: profiles/r03_attn_bwd_recompute.txt.)
template <typename TI, typename TO>
__global__ __launch_bounds__(1024) void attn_wide_bwd_kernel(Attn16Args a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  const hamt_attn_desc& d = a.d;
  const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  const int Sq = d.Sq, Sk = d.Sk;
  const size_t qrow0 = (size_t)b * d.Sq, krow0 = (size_t)b * d.Sk;
  const int nqb = (Sq + 15) >> 4, nkb = (Sk + 15) >> 4, sq16 = nqb * 16, sk16 = nkb * 16;
  bf16_t* Qs = sm;
  bf16_t* dOs = Qs + sq16 * AST;
  bf16_t* Ks = dOs + sq16 * AST;
  bf16_t* Vs = Ks + sk16 * AST;
  float* lse_s = (float*)(Vs + sk16 * AST);
  float* delta_s = lse_s + sq16;
  float* mk_s = delta_s + sq16;
  uint32_t* rowh_s = (uint32_t*)(mk_s + sk16);         // per query: the row half of the dropout hash
  const TI* Q = (const TI*)a.q + qrow0 * d.ldq + h * 64;
  const TI* K = (const TI*)a.k + krow0 * d.ldk + h * 64;
  const TI* V = (const TI*)a.v + krow0 * d.ldv + h * 64;
  const TO* O = (const TO*)a.o + qrow0 * d.ldo + h * 64;
  const TO* dO = (const TO*)a.d_o + qrow0 * d.ldo + h * 64;
  const RngKey key = rng_key(a.rng, d.call_id);
  {  // all five matrices in flight at once (nt = 64 * max(nqb, nkb) => 2 pieces per thread and matrix)
    Raw8<TI> rq[2], rk[2], rv[2];
    Raw8<TO> rdo[2], ro[2];
    rows_load<TI, 2>(Q, d.ldq, Sq, t, nt, rq);
    rows_load<TO, 2>(dO, d.ldo, Sq, t, nt, rdo);
    rows_load<TO, 2>(O, d.ldo, Sq, t, nt, ro);
    rows_load<TI, 2>(K, d.ldk, Sk, t, nt, rk);
    rows_load<TI, 2>(V, d.ldv, Sk, t, nt, rv);
    stage_mask(a.mask, b, Sk, sk16, mk_s, t, nt);
    for (int i = t; i < sq16; i += nt) rowh_s[i] = hamt_mix32((uint32_t)((b * d.heads + h) * d.Sq + i) ^ key.k0);
    rows_store<TI, 2>(rq, sq16, Qs, t, nt);
    rows_store<TO, 2>(rdo, sq16, dOs, t, nt);
    // delta = rowsum(dO * O) in fp32 from the un-rounded values (8 threads per row) and the saved lse
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (t >> 3) + i * (nt >> 3);
      float f[8], o8[8], acc = 0.f;
      raw_f32(rdo[i], f);
      raw_f32(ro[i], o8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += f[j] * o8[j];
      acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
      if ((t & 7) == 0 && r < sq16) { delta_s[r] = acc; lse_s[r] = r < Sq ? a.lse[((size_t)b * d.heads + h) * d.Sq + r] : 0.f; }
    }
    rows_store<TI, 2>(rk, sk16, Ks, t, nt);
    rows_store<TI, 2>(rv, sk16, Vs, t, nt);
  }
  __syncthreads();
  const float inv_keep = d.p_drop > 0.f ? 1.0f / (1.0f - d.p_drop) : 1.0f;
  if (w < nqb) {   // phase A: this wave's 16 queries x all keys; lane owns query ql = 16w + l15
    const int ql = 16 * w + l15;
    bf16x8 qf[2], df[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { qf[s] = rfrag(Qs, ql, 32 * s + 8 * g); df[s] = rfrag(dOs, ql, 32 * s + 8 * g); }
    const float lse_q = lse_s[ql], delta_q = delta_s[ql];
    const uint32_t rowh = rowh_s[ql];
    const bool qv = ql < Sq;                         // (a padding query has Q = dO = 0: its dS is 0 by itself; pinned anyway)
    // dQ^T[d][q] = K^T dS^T, two key blocks at a time: k-step s covers key blocks 2s, 2s+1; this lane group's k = keys 4g..4g+3 of each
    f32x4 dqf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dqf[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int s = 0; 2 * s < nkb; ++s) {
      float ds[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kb = 2 * s + e;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[e][r] = 0.f;
        if (kb < nkb) {
          f32x4 sf = (f32x4){0.f, 0.f, 0.f, 0.f}, dpf = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            sf = MFMA16(rfrag(Ks, 16 * kb + l15, 32 * s2 + 8 * g), qf[s2], sf);      // S^T[key][q]
            dpf = MFMA16(rfrag(Vs, 16 * kb + l15, 32 * s2 + 8 * g), df[s2], dpf);    // dP^T[key][q] = V dO^T
          }
          const float4 mk = *(const float4*)(mk_s + 16 * kb + 4 * g);
          const float mk4[4] = {mk.x, mk.y, mk.z, mk.w};
          float ds4[4] = {1.f, 1.f, 1.f, 1.f};
          if (d.p_drop > 0.f) drop_scale4(key, rowh, (uint32_t)(4 * kb + g), d.p_drop, inv_keep, ds4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __expf(fmaf(sf[r], d.scale, mk4[r]) - lse_q);          // 0 for the padding keys (mask = -inf)
            ds[e][r] = qv ? pr * (dpf[r] * ds4[r] - delta_q) * d.scale : 0.f;
          }
        }
      }
      const bf16x8 sb = pack_frag(ds[0], ds[1]);
#pragma unroll
      for (int db = 0; db < 4; ++db) dqf[db] = MFMA16(tfrag_lim(Ks, AST, 32 * s + 4 * g, 32 * s + 16 + 4 * g, sk16, 16 * db, lane), sb, dqf[db]);
    }
    if (qv) {
      TI* dQ = (TI*)a.dq + qrow0 * d.ldq + h * 64 + (size_t)ql * d.ldq;
#pragma unroll
      for (int db = 0; db < 4; ++db) st4<TI>(dQ + 16 * db + 4 * g, dqf[db][0], dqf[db][1], dqf[db][2], dqf[db][3]);
    }
  }
  if (w < nkb) {   // phase B: this wave's 16 keys x all queries; lane owns key kl = 16w + l15
    const int kl = 16 * w + l15;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { kf[s] = rfrag(Ks, kl, 32 * s + 8 * g); vf[s] = rfrag(Vs, kl, 32 * s + 8 * g); }
    const float mk = mk_s[kl];
    const uint32_t grp_h = (uint32_t)(kl >> 2) * 0x9e3779b9u + key.k1, thr = (uint32_t)(d.p_drop * 65536.0f);
    const bool second = (kl & 2) != 0, high = (kl & 1) != 0;   // which 16 bits of the group's two hash words are this key's
    f32x4 dkf[4], dvf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dkf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int s = 0; 2 * s < nqb; ++s) {
      float pt[2][4], dsv[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int qb = 2 * s + e;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pt[e][r] = 0.f; dsv[e][r] = 0.f; }
        if (qb < nqb) {
          f32x4 sf = (f32x4){0.f, 0.f, 0.f, 0.f}, dpf = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            sf = MFMA16(rfrag(Qs, 16 * qb + l15, 32 * s2 + 8 * g), kf[s2], sf);      // S[q][key]
            dpf = MFMA16(rfrag(dOs, 16 * qb + l15, 32 * s2 + 8 * g), vf[s2], dpf);   // dP[q][key] = dO V^T
          }
          const float4 lse4 = *(const float4*)(lse_s + 16 * qb + 4 * g), del4 = *(const float4*)(delta_s + 16 * qb + 4 * g);
          const uint4 rh4 = *(const uint4*)(rowh_s + 16 * qb + 4 * g);
          const float lq[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dq_[4] = {del4.x, del4.y, del4.z, del4.w};
          const uint32_t rh[4] = {rh4.x, rh4.y, rh4.z, rh4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __expf(fmaf(sf[r], d.scale, mk) - lq[r]);
            float keep = 1.f;
            if (d.p_drop > 0.f) {                  // drop_scale4's element (kl & 3) of group kl >> 2 of row q
              uint32_t x = hamt_mix32(rh[r] + grp_h);
              if (second) x = hamt_mix32(x ^ 0x85ebca6bu);
              keep = (high ? (x >> 16) : (x & 0xffffu)) >= thr ? inv_keep : 0.0f;
            }
            const bool qv = 16 * qb + 4 * g + r < Sq;
            pt[e][r] = qv ? pr * keep : 0.f;
            dsv[e][r] = qv ? pr * (dpf[r] * keep - dq_[r]) * d.scale : 0.f;
          }
        }
      }
      const bf16x8 pb = pack_frag(pt[0], pt[1]), sb = pack_frag(dsv[0], dsv[1]);
#pragma unroll
      for (int db = 0; db < 4; ++db) {   // reduction over the queries 32s + {4g.., 16 + 4g..}
        dvf[db] = MFMA16(tfrag_lim(dOs, AST, 32 * s + 4 * g, 32 * s + 16 + 4 * g, sq16, 16 * db, lane), pb, dvf[db]);
        dkf[db] = MFMA16(tfrag_lim(Qs, AST, 32 * s + 4 * g, 32 * s + 16 + 4 * g, sq16, 16 * db, lane), sb, dkf[db]);
      }
    }
    if (kl < Sk) {
      TI* dK = (TI*)a.dk + krow0 * d.ldk + h * 64 + (size_t)kl * d.ldk;
      TI* dV = (TI*)a.dv + krow0 * d.ldv + h * 64 + (size_t)kl * d.ldv;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        st4<TI>(dK + 16 * db + 4 * g, dkf[db][0], dkf[db][1], dkf[db][2], dkf[db][3]);
        st4<TI>(dV + 16 * db + 4 * g, dvf[db][0], dvf[db][1], dvf[db][2], dvf[db][3]);
      }
    }
  }
}

template <typename TI, typename TO>
void launch_wide_bwd(const Attn16Args& a, hipStream_t s) {
  const hamt_attn_desc& d = a.d;
  const int nqb = (d.Sq + 15) / 16, nkb = (d.Sk + 15) / 16, nb = nqb > nkb ? nqb : nkb;
  const int sq16 = nqb * 16, sk16 = nkb * 16;
  const size_t lds = (size_t)2 * (sq16 + sk16) * AST * sizeof(bf16_t) + (size_t)(3 * sq16 + sk16) * sizeof(float);
  static bool raised = false;                      // > 64 KiB of dynamic LDS needs the opt-in once per kernel
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_wide_bwd_kernel<TI, TO>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = true;
  }
  hipLaunchKernelGGL((attn_wide_bwd_kernel<TI, TO>), dim3(d.heads, d.B), dim3(64 * nb), lds, s, a);
}

template <typename TI, typename TO, int KB, int IT>
void launch_s128_fwd_kb(const Attn16Args& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  static bool raised = false;                      // > 64 KiB of dynamic LDS needs the opt-in once per kernel
  if (lds > 64 * 1024 && !raised) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_s128_fwd_kernel<TI, TO, KB, IT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = true;
  }
  hipLaunchKernelGGL((attn_s128_fwd_kernel<TI, TO, KB, IT>), grid, block, lds, s, a);
}

template <typename TI, typename TO>
void launch_s128_fwd(const Attn16Args& a, hipStream_t s) {
  const hamt_attn_desc& d = a.d;
  const int nqb = (d.Sq + 15) / 16, nkb = (d.Sk + 15) / 16;
  const int kb = nkb <= 2 ? 2 : (nkb + 1) & ~1;
  const int it = kb > 8 ? 4 : 2;
  const int need = 2 * kb / it;                        // waves that make 8 * 16 kb staging pieces = it per thread
  const int nqw = nqb < 8 ? nqb : 8;                   // one wave per 16 queries of the chunk
  const int nw = nqw > need ? nqw : need;
  const dim3 grid(d.heads, d.B, (d.Sq + 127) / 128), block(64 * nw);
  const size_t lds = (size_t)2 * kb * 16 * AST * sizeof(bf16_t) + (size_t)kb * 16 * sizeof(float);
  switch (kb) {
    case 2: launch_s128_fwd_kb<TI, TO, 2, 2>(a, grid, block, lds, s); break;
    case 4: launch_s128_fwd_kb<TI, TO, 4, 2>(a, grid, block, lds, s); break;
    case 6: launch_s128_fwd_kb<TI, TO, 6, 2>(a, grid, block, lds, s); break;
    case 8: launch_s128_fwd_kb<TI, TO, 8, 2>(a, grid, block, lds, s); break;
    case 10: launch_s128_fwd_kb<TI, TO, 10, 4>(a, grid, block, lds, s); break;
    case 12: launch_s128_fwd_kb<TI, TO, 12, 4>(a, grid, block, lds, s); break;
    case 14: launch_s128_fwd_kb<TI, TO, 14, 4>(a, grid, block, lds, s); break;
    default: launch_s128_fwd_kb<TI, TO, 16, 4>(a, grid, block, lds, s); break;
  }
}

template <typename TI, typename TO, int NB>
void launch_s128_bwd_nb(const Attn16Args& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  static bool raised = false;                      // > 64 KiB of dynamic LDS needs the opt-in once per kernel
  if (!raised) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_s128_bwd_kernel<TI, TO, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = true;
  }
  hipLaunchKernelGGL((attn_s128_bwd_kernel<TI, TO, NB>), grid, block, lds, s, a);
}

template <typename TI, typename TO>
void launch_s128_bwd(const Attn16Args& a, hipStream_t s) {
  const hamt_attn_desc& d = a.d;
  const int nqb = (d.Sq + 15) / 16, nkb = (d.Sk + 15) / 16, nb = nqb > nkb ? nqb : nkb;
  const int sq16 = nqb * 16, sk16 = nkb * 16;
  const dim3 grid(d.heads, d.B), block(64 * nb);
  const size_t lds = ((size_t)2 * (sq16 + sk16) * AST + (size_t)2 * sk16 * (sq16 + 8)) * sizeof(bf16_t) + (size_t)(2 * sq16 + sk16) * sizeof(float);
  if (nb <= 2) launch_s128_bwd_nb<TI, TO, 2>(a, grid, block, lds, s);
  else if (nb <= 4) launch_s128_bwd_nb<TI, TO, 4>(a, grid, block, lds, s);
  else if (nb <= 6) launch_s128_bwd_nb<TI, TO, 6>(a, grid, block, lds, s);
  else launch_s128_bwd_nb<TI, TO, 8>(a, grid, block, lds, s);
}

// single-pass forward: every key of a head in LDS and one score row in registers (any Sq, in chunks of 128 queries);
// backward: the whole head's P~ / dS in LDS.  Both draw the same dropout masks as the tiled kernels, so the forward of
// one family and the backward of the other form a valid pair (ViT: S = 197).
bool use_s128_fwd(const hamt_attn_desc* d) {
  static const bool off = getenv("HAMT_NO_ATTN_S128") != nullptr;
  static const bool off_wide = getenv("HAMT_NO_ATTN_WIDE") != nullptr;
  return !off && d->Sk <= (off_wide ? 128 : 256) && (d->Sq <= 128 || !off_wide);
}
bool use_s128_bwd(const hamt_attn_desc* d) {
  static const bool off = getenv("HAMT_NO_ATTN_S128") != nullptr;
  return !off && d->Sq <= 128 && d->Sk <= 128;
}
bool use_wide_bwd(const hamt_attn_desc* d) {     // 129 .. 224 tokens on a side (ViT-B/16: 197): the operands of a head fit in LDS
  static const bool off = getenv("HAMT_NO_ATTN_WIDE_BWD") != nullptr;
  return !off && d->Sq <= 224 && d->Sk <= 224;
}

}  // namespace

void hamt_attn16_fwd_launch(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const float* mask, void* o,
                            float* lse, const uint64_t* rng, hipStream_t s, const int* cu_q, const int* cu_k, int n_pairs, const int* pair) {
  Attn16Args a{*d, q, k, v, nullptr, nullptr, mask, o, nullptr, nullptr, nullptr, lse, rng, cu_q, cu_k, n_pairs, pair};
  const int* cu = cu_q ? cu_q : cu_k;
  dim3 grid((d->Sq + T64 - 1) / T64, d->heads, d->B), block(256);
  const bool ib = d->dtype_qkv == HAMT_BF16, ob = d->dtype_o == HAMT_BF16;
  if (use_s128_fwd(d) || cu) {      // (the packed form exists in the single-pass kernels only)
    if (!ib && !ob) launch_s128_fwd<float, float>(a, s);
    else if (ib && ob) launch_s128_fwd<bf16_t, bf16_t>(a, s);
    else if (ib) launch_s128_fwd<bf16_t, float>(a, s);
    else launch_s128_fwd<float, bf16_t>(a, s);
    return;
  }
  if (!ib && !ob) hipLaunchKernelGGL((attn16_fwd_kernel<float, float>), grid, block, 0, s, a);
  else if (ib && ob) hipLaunchKernelGGL((attn16_fwd_kernel<bf16_t, bf16_t>), grid, block, 0, s, a);
  else if (ib) hipLaunchKernelGGL((attn16_fwd_kernel<bf16_t, float>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn16_fwd_kernel<float, bf16_t>), grid, block, 0, s, a);
}

void hamt_attn16_bwd_launch(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const float* mask, const void* o,
                            const void* d_o, const float* lse, void* dq, void* dk, void* dv, const uint64_t* rng, hipStream_t s,
                            const int* cu_q, const int* cu_k, int n_pairs, const int* pair) {
  Attn16Args a{*d, q, k, v, o, d_o, mask, nullptr, dq, dk, dv, const_cast<float*>(lse), rng, cu_q, cu_k, n_pairs, pair};
  const int* cu = cu_q ? cu_q : cu_k;
  dim3 grid(d->heads, d->B), block(256);
  const bool ib = d->dtype_qkv == HAMT_BF16, ob = d->dtype_o == HAMT_BF16;
  if (use_s128_bwd(d) || cu) {
    if (!ib && !ob) launch_s128_bwd<float, float>(a, s);
    else if (ib && ob) launch_s128_bwd<bf16_t, bf16_t>(a, s);
    else if (ib) launch_s128_bwd<bf16_t, float>(a, s);
    else launch_s128_bwd<float, bf16_t>(a, s);
    return;
  }
  if (use_wide_bwd(d)) {
    if (!ib && !ob) launch_wide_bwd<float, float>(a, s);
    else if (ib && ob) launch_wide_bwd<bf16_t, bf16_t>(a, s);
    else if (ib) launch_wide_bwd<bf16_t, float>(a, s);
    else launch_wide_bwd<float, bf16_t>(a, s);
    return;
  }
  if (!ib && !ob) hipLaunchKernelGGL((attn16_bwd_kernel<float, float>), grid, block, 0, s, a);
  else if (ib && ob) hipLaunchKernelGGL((attn16_bwd_kernel<bf16_t, bf16_t>), grid, block, 0, s, a);
  else if (ib) hipLaunchKernelGGL((attn16_bwd_kernel<bf16_t, float>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn16_bwd_kernel<float, bf16_t>), grid, block, 0, s, a);
}
