// attn_small: fused multi-head softmax attention for the short sequences of HAMT (36..250 tokens).
//
//   S = Q K^T * scale + mask[b,key];  P = softmax(S);  P~ = dropout(P);  O = P~ V
//
// Forward: one workgroup per (batch, head, 64-query tile); K/V walked in 64-key tiles with an online
// softmax (running max / sum per query row held in registers), everything staged through LDS as fp32,
// contractions on v_mfma_f32_16x16x4_f32 (exact fp32 products; operands are single floats per lane so
// any LDS orientation can feed A or B -- no transposed copies).  LDS rows are padded to 66 floats: a
// "row-per-lane" fragment read (A of Q K^T, P V; B of Q K^T) is conflict free, a "column-per-lane"
// read is 2-way.  Row statistics use 16-lane butterfly shuffles (the C/D layout of the 16x16 MFMA puts
// one score row on 16 lanes x 4 fragments).
// Backward: flash-style recomputation of P from the saved log-sum-exp; one workgroup per (batch, head)
// loops key tiles (outer) x query tiles (inner); dK/dV accumulate in registers, dQ through memory.
//
// Replaces BertSelfAttention.forward core (vilmodel.py:101-126) and BertOutAttention.forward core
// (vilmodel.py:327-348), including transpose_for_scores / permute / contiguous (pointer arithmetic).
#include "common.h"

namespace {

constexpr int AT = 64;   // tile edge (queries, keys); d_head is 64 too
constexpr int ALD = 66;  // LDS row stride in floats

struct AttnArgs {
  hamt_attn_desc d;
  const void *q, *k, *v, *o, *d_o;
  const float* mask;
  void *out, *dq, *dk, *dv;
  float* lse;
  const uint64_t* rng;
};

template <typename T> __device__ __forceinline__ void ld16(const T* p, float (&f)[16]);
template <> __device__ __forceinline__ void ld16<float>(const float* p, float (&f)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { float4 x = ((const float4*)p)[i]; f[4 * i] = x.x; f[4 * i + 1] = x.y; f[4 * i + 2] = x.z; f[4 * i + 3] = x.w; }
}
template <> __device__ __forceinline__ void ld16<bf16_t>(const bf16_t* p, float (&f)[16]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint4 x = ((const uint4*)p)[i];
    uint32_t u[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[8 * i + 2 * j] = __uint_as_float(u[j] << 16); f[8 * i + 2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u); }
  }
}
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }

// Stage rows [r0, r0+64) x 64 columns (one head) of a [rows, ld] matrix into lds[64][ALD]; rows >= rlim -> 0.
// thread t: row t%64, columns (t/64)*16 .. +15.  Optionally returns this thread's 16 values.
template <typename T>
__device__ __forceinline__ void stage_tile(const T* base, int ld, int r0, int rlim, float* lds, int t, float (&f)[16]) {
  const int r = t & 63, c = (t >> 6) * 16;
  if (r0 + r < rlim) ld16<T>(base + (size_t)(r0 + r) * ld + c, f);
  else {
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) *(float2*)(lds + r * ALD + c + 2 * i) = make_float2(f[2 * i], f[2 * i + 1]);
}

__device__ __forceinline__ float grp16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float grp16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// =================================================================================================
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Qs[AT * ALD], Ks[AT * ALD], Vs[AT * ALD], Ps[AT * ALD];
  const hamt_attn_desc& d = a.d;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * AT, h = blockIdx.y, b = blockIdx.z;
  const TI* Q = (const TI*)a.q + (size_t)b * d.Sq * d.ldq + h * 64;
  const TI* K = (const TI*)a.k + (size_t)b * d.Sk * d.ldk + h * 64;
  const TI* V = (const TI*)a.v + (size_t)b * d.Sk * d.ldv + h * 64;
  float tmp[16];
  stage_tile<TI>(Q, d.ldq, q0, d.Sq, Qs, t, tmp);
  const bool active = q0 + 16 * w < d.Sq;  // wave-uniform: this wave owns query rows [16w, 16w+16) of the tile
  const RngKey key = rng_key(a.rng, d.call_id);
  const float inv_keep = d.p_drop > 0.f ? 1.0f / (1.0f - d.p_drop) : 1.0f;
  float m_run[4], l_run[4];
  f32x4 of[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
  for (int i = 0; i < 4; ++i) of[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < d.Sk; k0 += AT) {
    __syncthreads();  // previous tile's readers are done (also orders the Q staging on the first pass)
    stage_tile<TI>(K, d.ldk, k0, d.Sk, Ks, t, tmp);
    stage_tile<TI>(V, d.ldv, k0, d.Sk, Vs, t, tmp);
    __syncthreads();
    const int nkb = (min(AT, d.Sk - k0) + 15) >> 4;  // 16-key blocks holding at least one real key
    if (active) {
      f32x4 sf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sf[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const float qa = Qs[(16 * w + l15) * ALD + 4 * s + g];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          if (kb < nkb) sf[kb] = MFMA4(qa, Ks[(16 * kb + l15) * ALD + 4 * s + g], sf[kb]);
      }
      // online softmax; element sf[kb][r] is (query 16w+4g+r, key k0+16kb+l15)
      float mx[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) mx[r] = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int kk = k0 + 16 * kb + l15;
        const bool kval = kb < nkb && kk < d.Sk;
        const float mk = (kval && a.mask) ? a.mask[(size_t)b * d.Sk + kk] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sf[kb][r] = kval ? sf[kb][r] * d.scale + mk : -INFINITY;
          mx[r] = fmaxf(mx[r], sf[kb][r]);
        }
      }
      float alpha[4], rs[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float mn = fmaxf(m_run[r], grp16_max(mx[r]));
        alpha[r] = __expf(m_run[r] - mn);  // exp(-inf) = 0 on the first tile
        m_run[r] = mn;
        rs[r] = 0.f;
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int kk = k0 + 16 * kb + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = (kb < nkb) ? expf(sf[kb][r] - m_run[r]) : 0.f;
          rs[r] += p;
          if (d.p_drop > 0.f) {
            const int qq = q0 + 16 * w + 4 * g + r;
            p *= drop_scale(key, ((uint64_t)(b * d.heads + h) * d.Sq + qq) * d.Sk + kk, d.p_drop, inv_keep);
          }
          if (kb < nkb) Ps[(16 * w + 4 * g + r) * ALD + 16 * kb + l15] = p;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        l_run[r] = l_run[r] * alpha[r] + grp16_sum(rs[r]);
#pragma unroll
        for (int db = 0; db < 4; ++db) of[db][r] *= alpha[r];
      }
    }
    __syncthreads();  // P visible (each wave only reads its own rows, the barrier is the simple fence)
    if (active) {
      for (int ks = 0; ks < nkb * 4; ++ks) {
        const float pa = Ps[(16 * w + l15) * ALD + 4 * ks + g];
#pragma unroll
        for (int db = 0; db < 4; ++db) of[db] = MFMA4(pa, Vs[(4 * ks + g) * ALD + 16 * db + l15], of[db]);
      }
    }
  }
  if (active) {
    TO* O = (TO*)a.out + (size_t)b * d.Sq * d.ldo + h * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 16 * w + 4 * g + r;
      if (qq < d.Sq) {
        const float inv = 1.0f / l_run[r];
#pragma unroll
        for (int db = 0; db < 4; ++db) st1<TO>(O + (size_t)qq * d.ldo + 16 * db + l15, of[db][r] * inv);
        if (l15 == 0) a.lse[((size_t)b * d.heads + h) * d.Sq + qq] = m_run[r] + logf(l_run[r]);
      }
    }
  }
}

// =================================================================================================
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Qs[AT * ALD], Ks[AT * ALD], Vs[AT * ALD], dOs[AT * ALD], PS[AT * ALD], DS[AT * ALD];
  __shared__ float dpart[4][AT], lse_s[AT], delta_s[AT];
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
  float tmp[16], tmp2[16];

  for (int k0 = 0; k0 < d.Sk; k0 += AT) {
    const int nkb = (min(AT, d.Sk - k0) + 15) >> 4;
    const bool kact = k0 + 16 * w < d.Sk;  // this wave owns key rows [16w, 16w+16) of the tile for dK/dV
    f32x4 dkf[4], dvf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dkf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();
    stage_tile<TI>(K, d.ldk, k0, d.Sk, Ks, t, tmp);
    stage_tile<TI>(V, d.ldv, k0, d.Sk, Vs, t, tmp);
    for (int q0 = 0; q0 < d.Sq; q0 += AT) {
      const int nqb = (min(AT, d.Sq - q0) + 15) >> 4;
      const bool qact = q0 + 16 * w < d.Sq;
      __syncthreads();  // previous (k,q) tile fully consumed
      stage_tile<TI>(Q, d.ldq, q0, d.Sq, Qs, t, tmp);
      stage_tile<TO>(dO, d.ldo, q0, d.Sq, dOs, t, tmp);
      {  // delta = rowsum(dO * O), lse
        const int r = t & 63, c = (t >> 6) * 16;
        float acc = 0.f;
        if (q0 + r < d.Sq) {
          ld16<TO>(O + (size_t)(q0 + r) * d.ldo + c, tmp2);
#pragma unroll
          for (int i = 0; i < 16; ++i) acc += tmp[i] * tmp2[i];
        }
        dpart[t >> 6][r] = acc;
        if (t < AT) lse_s[t] = (q0 + t < d.Sq) ? a.lse[((size_t)b * d.heads + h) * d.Sq + q0 + t] : 0.f;
      }
      __syncthreads();
      if (t < AT) delta_s[t] = dpart[0][t] + dpart[1][t] + dpart[2][t] + dpart[3][t];
      __syncthreads();
      if (qact) {
        f32x4 sf[4], dpf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { sf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dpf[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
          const float qa = Qs[(16 * w + l15) * ALD + 4 * s + g], da = dOs[(16 * w + l15) * ALD + 4 * s + g];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
            if (kb < nkb) {
              sf[kb] = MFMA4(qa, Ks[(16 * kb + l15) * ALD + 4 * s + g], sf[kb]);
              dpf[kb] = MFMA4(da, Vs[(16 * kb + l15) * ALD + 4 * s + g], dpf[kb]);
            }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const int kk = k0 + 16 * kb + l15;
          const bool kval = kb < nkb && kk < d.Sk;
          const float mk = (kval && a.mask) ? a.mask[(size_t)b * d.Sk + kk] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ql = 16 * w + 4 * g + r, qq = q0 + ql;
            float p = 0.f, ds = 0.f, pd = 0.f;
            if (kval && qq < d.Sq) {
              p = expf(sf[kb][r] * d.scale + mk - lse_s[ql]);
              float dsc = 1.0f;
              if (d.p_drop > 0.f) dsc = drop_scale(key, ((uint64_t)(b * d.heads + h) * d.Sq + qq) * d.Sk + kk, d.p_drop, inv_keep);
              pd = p * dsc;
              ds = p * (dpf[kb][r] * dsc - delta_s[ql]) * d.scale;
            }
            if (kb < nkb) { PS[ql * ALD + 16 * kb + l15] = pd; DS[ql * ALD + 16 * kb + l15] = ds; }
          }
        }
      }
      __syncthreads();
      if (kact) {  // dV += P~^T dO ; dK += dS^T Q   (rows = keys 16w.., reduction over the tile's queries)
        for (int qs = 0; qs < nqb * 4; ++qs) {
          const float pa = PS[(4 * qs + g) * ALD + 16 * w + l15], sa = DS[(4 * qs + g) * ALD + 16 * w + l15];
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            dvf[db] = MFMA4(pa, dOs[(4 * qs + g) * ALD + 16 * db + l15], dvf[db]);
            dkf[db] = MFMA4(sa, Qs[(4 * qs + g) * ALD + 16 * db + l15], dkf[db]);
          }
        }
      }
      if (qact) {  // dQ (+)= dS K  (rows = queries 16w.., reduction over the tile's keys)
        f32x4 dqf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dqf[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nkb * 4; ++ks) {
          const float sa = DS[(16 * w + l15) * ALD + 4 * ks + g];
#pragma unroll
          for (int db = 0; db < 4; ++db) dqf[db] = MFMA4(sa, Ks[(4 * ks + g) * ALD + 16 * db + l15], dqf[db]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = q0 + 16 * w + 4 * g + r;
          if (qq < d.Sq) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
              TI* p = dQ + (size_t)qq * d.ldq + 16 * db + l15;
              st1<TI>(p, k0 == 0 ? dqf[db][r] : ld1<TI>(p) + dqf[db][r]);
            }
          }
        }
      }
    }
    if (kact) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = k0 + 16 * w + 4 * g + r;
        if (kk < d.Sk) {
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            st1<TI>(dK + (size_t)kk * d.ldk + 16 * db + l15, dkf[db][r]);
            st1<TI>(dV + (size_t)kk * d.ldv + 16 * db + l15, dvf[db][r]);
          }
        }
      }
    }
  }
}

int check_desc(const hamt_attn_desc* d, const char* who) {
  HAMT_CHECK_ARG(d, "%s: null desc", who);
  HAMT_CHECK_ARG(d->d_head == 64, "%s: d_head=%d unsupported (64 only)", who, d->d_head);
  HAMT_CHECK_ARG(d->B >= 0 && d->heads > 0 && d->Sq > 0 && d->Sk > 0, "%s: bad sizes", who);
  const int es = d->dtype_qkv == HAMT_BF16 ? 2 : 4, eo = d->dtype_o == HAMT_BF16 ? 2 : 4;
  HAMT_CHECK_ARG((d->ldq * es) % 16 == 0 && (d->ldk * es) % 16 == 0 && (d->ldv * es) % 16 == 0 && (d->ldo * eo) % 16 == 0,
                 "%s: rows must be 16-byte aligned", who);
  HAMT_CHECK_ARG(d->p_drop >= 0.f && d->p_drop < 1.f, "%s: bad p_drop", who);
  return HAMT_OK;
}

}  // namespace

void hamt_attn16_fwd_launch(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const float* mask, void* o,
                            float* lse, const uint64_t* rng, hipStream_t s, const int* cu_q = nullptr, const int* cu_k = nullptr,
                            int n_pairs = 0, const int* pair = nullptr);
void hamt_attn16_bwd_launch(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const float* mask, const void* o,
                            const void* d_o, const float* lse, void* dq, void* dk, void* dv, const uint64_t* rng, hipStream_t s,
                            const int* cu_q = nullptr, const int* cu_k = nullptr, int n_pairs = 0, const int* pair = nullptr);

// Packed ("varlen") self-attention: the B sequences lie back to back, sample b owns rows [cu[b], cu[b + 1]) of q / k / v / o (every
// one of its keys is real: no mask), at most d->Sq = d->Sk <= 128 rows each; lse is [B, heads, d->Sq].  What BertSelfAttention
// (vilmodel.py:96-129) computes for the REAL tokens of a padded batch, without the padded rows.
static int check_varlen(const hamt_attn_desc* d, const int* cu, const char* who) {
  int rc = check_desc(d, who);
  if (rc) return rc;
  HAMT_CHECK_ARG(cu != nullptr, "%s: cu_seqlens is null", who);
  HAMT_CHECK_ARG(d->prec == HAMT_PREC_BF16 && d->Sq == d->Sk && d->Sq <= 128, "%s: packed attention is built for the bf16 path, self-attention, <= 128 tokens per sequence (Sq %d, Sk %d)", who, d->Sq, d->Sk);
  return HAMT_OK;
}
extern "C" int hamt_attn_varlen_fwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const int* cu_seqlens,
                                    void* o, float* lse, const uint64_t* rng, void* stream) {
  int rc = check_varlen(d, cu_seqlens, "hamt_attn_varlen_fwd");
  if (rc) return rc;
  HAMT_CHECK_ARG(q && k && v && o && lse, "hamt_attn_varlen_fwd: null pointer");
  if (d->B == 0) return HAMT_OK;
  hamt_attn16_fwd_launch(d, q, k, v, nullptr, o, lse, rng, as_stream(stream), cu_seqlens, cu_seqlens, d->B);
  HAMT_CHECK_LAUNCH("hamt_attn_varlen_fwd");
  return HAMT_OK;
}
extern "C" int hamt_attn_varlen_bwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const int* cu_seqlens,
                                    const void* o, const void* d_o, const float* lse, void* dq, void* dk, void* dv,
                                    const uint64_t* rng, void* stream) {
  int rc = check_varlen(d, cu_seqlens, "hamt_attn_varlen_bwd");
  if (rc) return rc;
  HAMT_CHECK_ARG(q && k && v && o && d_o && lse && dq && dk && dv, "hamt_attn_varlen_bwd: null pointer");
  if (d->B == 0) return HAMT_OK;
  hamt_attn16_bwd_launch(d, q, k, v, nullptr, o, d_o, lse, dq, dk, dv, rng, as_stream(stream), cu_seqlens, cu_seqlens, d->B);
  HAMT_CHECK_LAUNCH("hamt_attn_varlen_bwd");
  return HAMT_OK;
}

// Cross attention with ONE side packed (the x-layers of a ragged batch, vilmodel.py:351-360: the instruction tokens lie back to back,
// the visual stream keeps its fixed stride): cu_q != NULL: query sequence b = rows [cu_q[b], cu_q[b + 1]) of q / o / d_o / dq, its keys
// rows [b * Sk, (b + 1) * Sk) of k / v under add_mask [B, Sk]; query sequences b >= n_pairs (fillers of a bucketed row count) have
// no keys: zero output / zero dq.  cu_k != NULL: queries at fixed stride Sq, sample b's keys = rows [cu_k[b], cu_k[b + 1]), all real.
// `pair` (optional, [B] on the device) replaces "b pairs with b": the key sample (cu_q form) or the entry of cu_k (cu_k form) of query
// sequence b, negative = a filler -- packed copies of a batch whose fillers lie between the copies (ITM's replicated text).
static int check_varlen_cross(const hamt_attn_desc* d, const int* cu_q, const int* cu_k, int n_pairs, const int* pair, const char* who) {
  int rc = check_desc(d, who);
  if (rc) return rc;
  HAMT_CHECK_ARG((cu_q != nullptr) != (cu_k != nullptr), "%s: exactly one of cu_q / cu_k must be given", who);
  HAMT_CHECK_ARG(d->prec == HAMT_PREC_BF16 && d->Sq <= 128 && d->Sk <= 128, "%s: packed attention is built for the bf16 path, <= 128 tokens per sequence (Sq %d, Sk %d)", who, d->Sq, d->Sk);
  HAMT_CHECK_ARG(pair != nullptr || (n_pairs >= 0 && n_pairs <= d->B && (cu_k == nullptr || n_pairs == d->B)), "%s: n_pairs = %d outside [0, B = %d] (packed keys: every query sample has keys)", who, n_pairs, d->B);
  return HAMT_OK;
}
extern "C" int hamt_attn_varlen_cross_fwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const int* cu_q,
                                          const int* cu_k, int n_pairs, const int* pair, const float* add_mask, void* o, float* lse,
                                          const uint64_t* rng, void* stream) {
  int rc = check_varlen_cross(d, cu_q, cu_k, n_pairs, pair, "hamt_attn_varlen_cross_fwd");
  if (rc) return rc;
  HAMT_CHECK_ARG(q && k && v && o && lse, "hamt_attn_varlen_cross_fwd: null pointer");
  if (d->B == 0) return HAMT_OK;
  hamt_attn16_fwd_launch(d, q, k, v, cu_k ? nullptr : add_mask, o, lse, rng, as_stream(stream), cu_q, cu_k, n_pairs, pair);
  HAMT_CHECK_LAUNCH("hamt_attn_varlen_cross_fwd");
  return HAMT_OK;
}
extern "C" int hamt_attn_varlen_cross_bwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v, const int* cu_q,
                                          const int* cu_k, int n_pairs, const int* pair, const float* add_mask, const void* o, const void* d_o,
                                          const float* lse, void* dq, void* dk, void* dv, const uint64_t* rng, void* stream) {
  int rc = check_varlen_cross(d, cu_q, cu_k, n_pairs, pair, "hamt_attn_varlen_cross_bwd");
  if (rc) return rc;
  HAMT_CHECK_ARG(q && k && v && o && d_o && lse && dq && dk && dv, "hamt_attn_varlen_cross_bwd: null pointer");
  if (d->B == 0) return HAMT_OK;
  hamt_attn16_bwd_launch(d, q, k, v, cu_k ? nullptr : add_mask, o, d_o, lse, dq, dk, dv, rng, as_stream(stream), cu_q, cu_k, n_pairs, pair);
  HAMT_CHECK_LAUNCH("hamt_attn_varlen_cross_bwd");
  return HAMT_OK;
}

extern "C" int hamt_attn_small_fwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v,
                                   const float* add_mask, void* o, float* lse, const uint64_t* rng, void* stream) {
  int rc = check_desc(d, "hamt_attn_small_fwd");
  if (rc) return rc;
  HAMT_CHECK_ARG(q && k && v && o && lse, "hamt_attn_small_fwd: null pointer");
  if (d->B == 0) return HAMT_OK;
  if (d->prec == HAMT_PREC_BF16) {
    hamt_attn16_fwd_launch(d, q, k, v, add_mask, o, lse, rng, as_stream(stream));
    HAMT_CHECK_LAUNCH("hamt_attn_small_fwd(bf16)");
    return HAMT_OK;
  }
  AttnArgs a{*d, q, k, v, nullptr, nullptr, add_mask, o, nullptr, nullptr, nullptr, lse, rng};
  dim3 grid((d->Sq + AT - 1) / AT, d->heads, d->B), block(256);
  hipStream_t s = as_stream(stream);
  const bool ib = d->dtype_qkv == HAMT_BF16, ob = d->dtype_o == HAMT_BF16;
  if (!ib && !ob) hipLaunchKernelGGL((attn_fwd_kernel<float, float>), grid, block, 0, s, a);
  else if (ib && ob) hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, bf16_t>), grid, block, 0, s, a);
  else if (ib) hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, float>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<float, bf16_t>), grid, block, 0, s, a);
  HAMT_CHECK_LAUNCH("hamt_attn_small_fwd");
  return HAMT_OK;
}

extern "C" int hamt_attn_small_bwd(const hamt_attn_desc* d, const void* q, const void* k, const void* v,
                                   const float* add_mask, const void* o, const void* d_o, const float* lse,
                                   float* delta, void* dq, void* dk, void* dv, const uint64_t* rng, void* stream) {
  (void)delta;
  int rc = check_desc(d, "hamt_attn_small_bwd");
  if (rc) return rc;
  HAMT_CHECK_ARG(q && k && v && o && d_o && lse && dq && dk && dv, "hamt_attn_small_bwd: null pointer");
  if (d->B == 0) return HAMT_OK;
  if (d->prec == HAMT_PREC_BF16) {
    hamt_attn16_bwd_launch(d, q, k, v, add_mask, o, d_o, lse, dq, dk, dv, rng, as_stream(stream));
    HAMT_CHECK_LAUNCH("hamt_attn_small_bwd(bf16)");
    return HAMT_OK;
  }
  AttnArgs a{*d, q, k, v, o, d_o, add_mask, nullptr, dq, dk, dv, const_cast<float*>(lse), rng};
  dim3 grid(d->heads, d->B), block(256);
  hipStream_t s = as_stream(stream);
  const bool ib = d->dtype_qkv == HAMT_BF16, ob = d->dtype_o == HAMT_BF16;
  if (!ib && !ob) hipLaunchKernelGGL((attn_bwd_kernel<float, float>), grid, block, 0, s, a);
  else if (ib && ob) hipLaunchKernelGGL((attn_bwd_kernel<bf16_t, bf16_t>), grid, block, 0, s, a);
  else if (ib) hipLaunchKernelGGL((attn_bwd_kernel<bf16_t, float>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn_bwd_kernel<float, bf16_t>), grid, block, 0, s, a);
  HAMT_CHECK_LAUNCH("hamt_attn_small_bwd");
  return HAMT_OK;
}
