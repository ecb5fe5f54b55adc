// LayerNorm family for gfx950:  y = dropout_post( LN( dropout_pre(x) + residual ) ), fp32 statistics.
// One 64-lane wave per row (H <= 1024 kept in registers as float4s), 4 rows per 256-thread block,
// wave-shuffle reductions, 16-byte coalesced loads/stores.  HBM/L2-bound: 3 reads + 2 writes per element.
//
// Replaces BertSelfOutput/BertOutput's dropout+add+LayerNorm (vilmodel.py:139-143, 181-185), the
// LayerNorm+dropout tails of BertEmbeddings / ImageEmbeddings / HistoryEmbeddings (vilmodel.py:67-68,
// 503-504, 545-546, 570-571) and the LN(+Dropout) inside the prediction heads (pretrain_cmt.py:18-19).
#include "common.h"

void hamt_reduce_partials(int R, int N, const float* ws, float* out, int accumulate, hipStream_t s);

namespace {

// keep factors of the 4 consecutive elements (row, c .. c+3), c % 4 == 0: the 4-wide mask stream of common.h (one hash
// pair per 4 elements instead of two hashes per element -- the LayerNorm-backward kernel spent 15 % of its time hashing)
__device__ __forceinline__ void ln_keep4(RngKey k, int row, int c, float p, float inv_keep, float (&f)[4]) {
  drop_scale4(k, hamt_mix32((uint32_t)row ^ k.k0), (uint32_t)(c >> 2), p, inv_keep, f);
}


constexpr uint32_t POST_SALT = 0x5bd1e995u;

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(hamt_ln_desc d, const void* __restrict__ xv,
                                                     const float* __restrict__ res, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ zv,
                                                     float* __restrict__ y, bf16_t* __restrict__ y16,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     const uint64_t* __restrict__ rng) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  const int H = d.H;
  if (row >= d.M) {   // rows [M, Mpad16) of the bf16 image are zero (they are reduction padding for the fast GEMMs)
    if (y16 && row < d.Mpad16)
      for (int c = lane * 4; c < H; c += 256) *(uint2*)(y16 + (size_t)row * H + c) = make_uint2(0u, 0u);
    return;
  }
  const RngKey kpre = rng_key(rng, d.call_id), kpost = rng_key(rng, d.call_id ^ POST_SALT);
  const float ik_pre = d.p_pre > 0.f ? 1.0f / (1.0f - d.p_pre) : 1.0f, ik_post = d.p_post > 0.f ? 1.0f / (1.0f - d.p_post) : 1.0f;
  float4 v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const size_t o = (size_t)row * H + c;
      float4 a;
      if (d.io16 & HAMT_LN_X_F16) {       // the dense layer in front wrote IEEE half (HAMT_F16)
        const uint2 u = *(const uint2*)((const bf16_t*)xv + o);
        unpack_h2(u.x, a.x, a.y); unpack_h2(u.y, a.z, a.w);
      } else if (d.io16 & HAMT_LN_X_BF16) {      // the dense layer in front wrote bf16 (what autocast does with a linear's output)
        const uint2 u = *(const uint2*)((const bf16_t*)xv + o);
        a = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
      } else a = *(const float4*)((const float*)xv + o);
      if (d.p_pre > 0.f) {
        float f_[4]; ln_keep4(kpre, row, c, d.p_pre, ik_pre, f_);
        a.x *= f_[0]; a.y *= f_[1]; a.z *= f_[2]; a.w *= f_[3];
      }
      if (res) { float4 r = *(const float4*)(res + o); a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
      v[i] = a;
      sum += a.x + a.y + a.z + a.w;
    } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, e = v[i].w - mean;
      sq += a * a + b * b + cc * cc + e * e;
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)H + d.eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const size_t o = (size_t)row * H + c;
      if (zv) {
        if (d.io16 & HAMT_LN_Z_F16) *(uint2*)((bf16_t*)zv + o) = make_uint2(pack_h2(v[i].x, v[i].y), pack_h2(v[i].z, v[i].w));
        else if (d.io16 & HAMT_LN_Z_BF16) *(uint2*)((bf16_t*)zv + o) = make_uint2(pack_bf2(v[i].x, v[i].y), pack_bf2(v[i].z, v[i].w));
        else *(float4*)((float*)zv + o) = v[i];
      }
      const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
      float4 r;
      r.x = (v[i].x - mean) * rstd * g.x + b.x; r.y = (v[i].y - mean) * rstd * g.y + b.y;
      r.z = (v[i].z - mean) * rstd * g.z + b.z; r.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (d.p_post > 0.f) {
        float f_[4]; ln_keep4(kpost, row, c, d.p_post, ik_post, f_);
        r.x *= f_[0]; r.y *= f_[1]; r.z *= f_[2]; r.w *= f_[3];
      }
      if (y) *(float4*)(y + o) = r;
      if (y16) *(uint2*)(y16 + o) = make_uint2(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w));
    }
  }
  if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

template <int NV, int NWV>
__global__ __launch_bounds__(64 * NWV) void ln_bwd_kernel(hamt_ln_desc d, const float* __restrict__ dy,
                                                     const void* __restrict__ zv, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, const float* __restrict__ gamma,
                                                     float* __restrict__ dz, float* __restrict__ dx, bf16_t* __restrict__ dx16,
                                                     float* __restrict__ ws, const uint64_t* __restrict__ rng,
                                                     const float* __restrict__ add) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int H = d.H;
  if (dx16 && blockIdx.x == 0)      // zero the reduction-padding rows [M, Mpad16) of the bf16 gradient image
    for (int row = d.M + w; row < d.Mpad16; row += NWV)
      for (int c = lane * 4; c < H; c += 256) *(uint2*)(dx16 + (size_t)row * H + c) = make_uint2(0u, 0u);
  const RngKey kpre = rng_key(rng, d.call_id), kpost = rng_key(rng, d.call_id ^ POST_SALT);
  const float ik_pre = d.p_pre > 0.f ? 1.0f / (1.0f - d.p_pre) : 1.0f, ik_post = d.p_post > 0.f ? 1.0f / (1.0f - d.p_post) : 1.0f;
  float4 gam[NV], dg[NV], db[NV], dxs[NV];   // dxs: column sums of dx = bias gradient of the dense layer that produced x
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    gam[i] = c < H ? *(const float4*)(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    dxs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = blockIdx.x * NWV + w; row < d.M; row += gridDim.x * NWV) {
    const float mean = mean_i[row], rstd = rstd_i[row];
    float4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        const size_t o = (size_t)row * H + c;
        float4 a = *(const float4*)(dy + o);
        if (d.p_post > 0.f) {
          float f_[4]; ln_keep4(kpost, row, c, d.p_post, ik_post, f_);
          a.x *= f_[0]; a.y *= f_[1]; a.z *= f_[2]; a.w *= f_[3];
        }
        float4 zz;
        if (d.io16 & HAMT_LN_Z_F16) {
          const uint2 u = *(const uint2*)((const bf16_t*)zv + o);
          unpack_h2(u.x, zz.x, zz.y); unpack_h2(u.y, zz.z, zz.w);
        } else if (d.io16 & HAMT_LN_Z_BF16) {
          const uint2 u = *(const uint2*)((const bf16_t*)zv + o);
          zz = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
        } else zz = *(const float4*)((const float*)zv + o);
        float4 h;
        h.x = (zz.x - mean) * rstd; h.y = (zz.y - mean) * rstd; h.z = (zz.z - mean) * rstd; h.w = (zz.w - mean) * rstd;
        dg[i].x += a.x * h.x; dg[i].y += a.y * h.y; dg[i].z += a.z * h.z; dg[i].w += a.w * h.w;
        db[i].x += a.x; db[i].y += a.y; db[i].z += a.z; db[i].w += a.w;
        a.x *= gam[i].x; a.y *= gam[i].y; a.z *= gam[i].z; a.w *= gam[i].w;
        s1 += a.x + a.y + a.z + a.w;
        s2 += a.x * h.x + a.y * h.y + a.z * h.z + a.w * h.w;
        g[i] = a; xh[i] = h;
      }
    }
    const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        const size_t o = (size_t)row * H + c;
        float4 r;
        r.x = rstd * (g[i].x - c1 - xh[i].x * c2); r.y = rstd * (g[i].y - c1 - xh[i].y * c2);
        r.z = rstd * (g[i].z - c1 - xh[i].z * c2); r.w = rstd * (g[i].w - c1 - xh[i].w * c2);
        if (add) {   // pre-LN block: the residual path's gradient rides along (dz_out = LN-backward + add); dx / dx16 /
          const float4 q = *(const float4*)(add + o);     // dxsum below stay the pure LayerNorm-input gradient
          *(float4*)(dz + o) = make_float4(r.x + q.x, r.y + q.y, r.z + q.z, r.w + q.w);
        } else *(float4*)(dz + o) = r;
        if (d.p_pre > 0.f) {
          float f_[4]; ln_keep4(kpre, row, c, d.p_pre, ik_pre, f_);
          r.x *= f_[0]; r.y *= f_[1]; r.z *= f_[2]; r.w *= f_[3];
        }
        if (dx) *(float4*)(dx + o) = r;
        if (dx16) *(uint2*)(dx16 + o) = make_uint2(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w));
        dxs[i].x += r.x; dxs[i].y += r.y; dxs[i].z += r.z; dxs[i].w += r.w;
      }
    }
  }
  // block partials: ws[block][0][H] = dgamma, ws[block][1][H] = dbeta, ws[block][2][H] = sum dx
  // (NWV waves per block keep enough rows in flight to cover the HBM latency -- 4 waves per CU ran at 2.1 TB/s -- while the
  // number of partials, one per block, stays <= 256: first a tree over the upper waves, then the 4-wave stage)
  __shared__ float4 red[NWV > 4 ? NWV / 2 : 4][3][NV * 64];
#pragma unroll
  for (int half = NWV / 2; half >= 4; half >>= 1) {
    if (w >= half && w < 2 * half) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { red[w - half][0][i * 64 + lane] = dg[i]; red[w - half][1][i * 64 + lane] = db[i]; red[w - half][2][i * 64 + lane] = dxs[i]; }
    }
    __syncthreads();
    if (w < half) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 a = red[w][0][i * 64 + lane], b = red[w][1][i * 64 + lane], c = red[w][2][i * 64 + lane];
        dg[i].x += a.x; dg[i].y += a.y; dg[i].z += a.z; dg[i].w += a.w;
        db[i].x += b.x; db[i].y += b.y; db[i].z += b.z; db[i].w += b.w;
        dxs[i].x += c.x; dxs[i].y += c.y; dxs[i].z += c.z; dxs[i].w += c.w;
      }
    }
    __syncthreads();
  }
  if (w < 4) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { red[w][0][i * 64 + lane] = dg[i]; red[w][1][i * 64 + lane] = db[i]; red[w][2][i * 64 + lane] = dxs[i]; }
  }
  __syncthreads();
  if (w < 3) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        float4 a = red[0][w][i * 64 + lane], b = red[1][w][i * 64 + lane], cc = red[2][w][i * 64 + lane], e = red[3][w][i * 64 + lane];
        *(float4*)(ws + ((size_t)blockIdx.x * 3 + w) * H + c) = make_float4(a.x + b.x + cc.x + e.x, a.y + b.y + cc.y + e.y, a.z + b.z + cc.z + e.z, a.w + b.w + cc.w + e.w);
      }
    }
  }
}

// ws[block][3][H] -> dgamma[H], dbeta[H], dxsum[H] (each accumulated).  block = 64 columns x 16 partial-row phases.
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(int nb, int H, const float* __restrict__ ws, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dxsum) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int which = c / H, col = c % H;
  float s = 0.f;
  if (c < 3 * H)
    for (int b = ph; b < nb; b += 16) s += ws[((size_t)b * 3 + which) * H + col];
  __shared__ float red[16][64];
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && c < 3 * H) {
    float* o = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
    if (o) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
      o[col] = t;
    }
  }
}

// Same reduction for H % 64 == 0 with 16-byte loads: block = 64 columns (16 lanes x float4) x 16 phases; every thread keeps
// its 16 (nb = 256) loads in flight before the first add.
__global__ __launch_bounds__(256) void ln_bwd_reduce4_kernel(int nb, int H, const float* __restrict__ ws, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dxsum) {
  const int l16 = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + l16 * 4;            // < 3 * H by construction
  const int which = c / H, col = c - which * H;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = ph; b0 < nb; b0 += 64) {              // 4 independent loads per trip
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + 16 * u;
      v[u] = b < nb ? *(const float4*)(ws + ((size_t)b * 3 + which) * H + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  __shared__ float4 red[16][16];
  red[ph][l16] = s;
  __syncthreads();
  if (ph == 0) {
    float* o = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
    if (o) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float4 r = red[i][l16]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
      *(float4*)(o + col) = t;
    }
  }
}

// Grouped form of the reduction: entry e owns blocks [e * bpe, (e + 1) * bpe) (bpe = 3 * Hmax / 64; blocks beyond an entry's
// own 3 * H / 64 exit).  The entry table is written into device memory by tiny kernels from kernarg chunks, like the
// grouped weight-gradient launch, so the call is capturable and `descs` need not outlive it.
struct LnRedEntry { const float* ws; float* dgamma; float* dbeta; float* dxsum; int nb, H, atomic, pad1; };   // atomic: bit i = output i is ADDED (atomically)
constexpr int LNRED_CHUNK = 64;                  // entries per kernarg chunk (64 x 48 B)
struct LnRedChunk { LnRedEntry e[LNRED_CHUNK]; };
__global__ void ln_red_table_write_kernel(LnRedChunk c, LnRedEntry* tab, int off, int cnt) {
  if ((int)threadIdx.x < cnt) tab[off + threadIdx.x] = c.e[threadIdx.x];
}
__global__ __launch_bounds__(256) void ln_bwd_reduce_grouped_kernel(const LnRedEntry* __restrict__ tab, int bpe) {
  const LnRedEntry q = tab[blockIdx.x / bpe];
  const int blk = blockIdx.x % bpe;
  const int H = q.H, nb = q.nb;
  if (blk * 64 >= 3 * H) return;
  const int l16 = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int c = blk * 64 + l16 * 4;
  const int which = c / H, col = c - which * H;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = ph; b0 < nb; b0 += 64) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + 16 * u;
      v[u] = b < nb ? *(const float4*)(q.ws + ((size_t)b * 3 + which) * H + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  __shared__ float4 red[16][16];
  red[ph][l16] = s;
  __syncthreads();
  if (ph == 0) {
    float* o = which == 0 ? q.dgamma : (which == 1 ? q.dbeta : q.dxsum);
    if (o) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float4 r = red[i][l16]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
      if ((q.atomic >> which) & 1) {     // the output is a gradient-arena slot shared with other uses of the parameter (zero at the start of a step)
        atomicAdd(o + col, t.x); atomicAdd(o + col + 1, t.y); atomicAdd(o + col + 2, t.z); atomicAdd(o + col + 3, t.w);
      } else *(float4*)(o + col) = t;
    }
  }
}

}  // namespace

extern "C" int hamt_ln_fwd(const hamt_ln_desc* d, const void* x, const float* residual, const float* gamma,
                           const float* beta, void* z, float* y, void* y16, float* mean, float* rstd,
                           const uint64_t* rng, void* stream) {
  HAMT_CHECK_ARG(d && x && gamma && beta && (y || y16) && mean && rstd, "hamt_ln_fwd: null pointer");
  HAMT_CHECK_ARG(d->H % 4 == 0 && d->H >= 4 && d->H <= 1024, "hamt_ln_fwd: H=%d unsupported (need H%%4==0, H<=1024)", d->H);
  HAMT_CHECK_ARG(d->p_pre >= 0.f && d->p_pre < 1.f && d->p_post >= 0.f && d->p_post < 1.f, "hamt_ln_fwd: bad dropout p");
  HAMT_CHECK_ARG((d->io16 & (HAMT_LN_X_BF16 | HAMT_LN_X_F16)) != (HAMT_LN_X_BF16 | HAMT_LN_X_F16) && (d->io16 & (HAMT_LN_Z_BF16 | HAMT_LN_Z_F16)) != (HAMT_LN_Z_BF16 | HAMT_LN_Z_F16),
                 "hamt_ln_fwd: io16 = %d names two formats for one tensor", d->io16);
  if (d->M == 0) return HAMT_OK;
  const int nv = (d->H + 255) / 256;
  const int rows = (y16 && d->Mpad16 > d->M) ? d->Mpad16 : d->M;
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = as_stream(stream);
#define LAUNCH(NV) hipLaunchKernelGGL((ln_fwd_kernel<NV>), grid, block, 0, s, *d, x, residual, gamma, beta, z, y, (bf16_t*)y16, mean, rstd, rng)
  switch (nv) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
  HAMT_CHECK_LAUNCH("hamt_ln_fwd");
  return HAMT_OK;
}

// waves per block and number of blocks (= partials) of the backward kernel for M rows (measured, H = 768, incl. the reduce
// pass: M = 11520 42 -> 31 us, 5120 24 -> 22, 2368 17.6 -> 14.1 with 16 waves; 384 rows are fastest with 4)
// (round 6: the cap at 256 blocks leaves a quarter of the waves of a 5 120-row call with a second row; 320 / 720 blocks -- one row per wave --
// measured SLOWER, tools/ln_bench.py: 17.1 -> 20.0 us at 5 120 rows, 24.5 -> 34.2 at 11 520, the B = 64 step 9.17 -> 9.28 ms: every block
// ends with a 16-wave LDS tree and 3 H floats of partials, and that tail grows with the block count faster than the row loop shrinks)
static void ln_bwd_geometry(int M, int* nwv_out, int* nb_out) {
  static const int force_w = getenv("HAMT_LN_BWD_WAVES") ? atoi(getenv("HAMT_LN_BWD_WAVES")) : 0;
  const int nwv = force_w ? force_w : (M >= 2048 ? 16 : (M >= 1024 ? 8 : 4));
  int nb = (M + nwv - 1) / nwv;
  if (nb > 256) nb = 256;
  *nwv_out = nwv;
  *nb_out = nb;
}
static void ln_bwd_reduce_launch(int nb, int H, const float* ws, float* dgamma, float* dbeta, float* dxsum, hipStream_t s) {
  const bool al = H % 64 == 0 && (((uintptr_t)dgamma | (uintptr_t)dbeta | (uintptr_t)dxsum | (uintptr_t)ws) % 16) == 0;
  if (al) hipLaunchKernelGGL(ln_bwd_reduce4_kernel, dim3(3 * H / 64), dim3(256), 0, s, nb, H, ws, dgamma, dbeta, dxsum);
  else hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((3 * H + 63) / 64), dim3(1024), 0, s, nb, H, ws, dgamma, dbeta, dxsum);
}

static int ln_bwd_impl(const hamt_ln_desc* d, const float* dy, const void* z, const float* mean,
                       const float* rstd, const float* gamma, float* dz, float* dx, void* dx16, float* dgamma,
                       float* dbeta, float* dxsum, float* ws, const uint64_t* rng, const float* add, void* stream) {
  HAMT_CHECK_ARG(d && dy && z && mean && rstd && gamma && dz && ws, "hamt_ln_bwd: null pointer");
  HAMT_CHECK_ARG(d->H % 4 == 0 && d->H >= 4 && d->H <= 1024, "hamt_ln_bwd: H=%d unsupported", d->H);
  HAMT_CHECK_ARG(!(d->p_pre > 0.f) || dx || dx16, "hamt_ln_bwd: p_pre > 0 needs dx or dx16");
  if (d->M == 0) return HAMT_OK;
  const int nv = (d->H + 255) / 256;
  int nwv, nb;
  ln_bwd_geometry(d->M, &nwv, &nb);
  hipStream_t s = as_stream(stream);
  float* dxx = d->p_pre > 0.f ? dx : nullptr;
#define LAUNCH2(NV, W) hipLaunchKernelGGL((ln_bwd_kernel<NV, W>), dim3(nb), dim3(64 * W), 0, s, *d, dy, z, mean, rstd, gamma, dz, dxx, (bf16_t*)dx16, ws, rng, add)
#define LAUNCH(NV) { if (nwv == 16) LAUNCH2(NV, 16); else if (nwv == 8) LAUNCH2(NV, 8); else LAUNCH2(NV, 4); }
  switch (nv) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
#undef LAUNCH2
  if (dgamma || dbeta || dxsum) ln_bwd_reduce_launch(nb, d->H, ws, dgamma, dbeta, dxsum, s);
  HAMT_CHECK_LAUNCH("hamt_ln_bwd");
  return HAMT_OK;
}

extern "C" int hamt_ln_bwd(const hamt_ln_desc* d, const float* dy, const void* z, const float* mean,
                           const float* rstd, const float* gamma, float* dz, float* dx, void* dx16, float* dgamma,
                           float* dbeta, float* dxsum, float* ws, const uint64_t* rng, void* stream) {
  return ln_bwd_impl(d, dy, z, mean, rstd, gamma, dz, dx, dx16, dgamma, dbeta, dxsum, ws, rng, nullptr, stream);
}

extern "C" int hamt_ln_bwd_add(const hamt_ln_desc* d, const float* dy, const void* z, const float* mean,
                               const float* rstd, const float* gamma, const float* add, float* dz, float* dgamma,
                               float* dbeta, float* ws, void* stream) {
  HAMT_CHECK_ARG(add && d && !(d->p_pre > 0.f) && !(d->p_post > 0.f), "hamt_ln_bwd_add: needs `add`, and no dropout inside the LayerNorm");
  return ln_bwd_impl(d, dy, z, mean, rstd, gamma, dz, nullptr, nullptr, dgamma, dbeta, nullptr, ws, nullptr, add, stream);
}

extern "C" int hamt_ln_bwd_reduce(int M, int H, const float* ws, float* dgamma, float* dbeta, float* dxsum, void* stream) {
  HAMT_CHECK_ARG(ws && M >= 0 && H % 4 == 0 && H >= 4 && H <= 1024, "hamt_ln_bwd_reduce: bad argument");
  if (M == 0 || !(dgamma || dbeta || dxsum)) return HAMT_OK;
  int nwv, nb;
  ln_bwd_geometry(M, &nwv, &nb);
  ln_bwd_reduce_launch(nb, H, ws, dgamma, dbeta, dxsum, as_stream(stream));
  HAMT_CHECK_LAUNCH("hamt_ln_bwd_reduce");
  return HAMT_OK;
}

extern "C" int hamt_ln_bwd_reduce_grouped(int n, const hamt_ln_reduce_desc* descs, void* table, size_t table_bytes, void* stream) {
  HAMT_CHECK_ARG(n >= 0 && (n == 0 || (descs && table)), "hamt_ln_bwd_reduce_grouped: bad argument");
  if (n == 0) return HAMT_OK;
  HAMT_CHECK_ARG((size_t)n * sizeof(LnRedEntry) <= table_bytes, "hamt_ln_bwd_reduce_grouped: table too small (%zu bytes needed)", (size_t)n * sizeof(LnRedEntry));
  hipStream_t s = as_stream(stream);
  LnRedEntry* tab = (LnRedEntry*)table;
  int hmax = 0;
  for (int i = 0; i < n; ++i) {
    const hamt_ln_reduce_desc& d = descs[i];
    HAMT_CHECK_ARG(d.ws && d.M > 0 && d.H % 64 == 0 && d.H <= 1024, "hamt_ln_bwd_reduce_grouped: entry %d: needs ws, M > 0, H %% 64 == 0 (H = %d)", i, d.H);
    HAMT_CHECK_ARG((((uintptr_t)d.ws | (uintptr_t)d.dgamma | (uintptr_t)d.dbeta | (uintptr_t)d.dxsum) % 16) == 0, "hamt_ln_bwd_reduce_grouped: entry %d: pointers must be 16-byte aligned", i);
    hmax = d.H > hmax ? d.H : hmax;
  }
  for (int b0 = 0; b0 < n; b0 += LNRED_CHUNK) {
    LnRedChunk ch;
    const int cnt = n - b0 < LNRED_CHUNK ? n - b0 : LNRED_CHUNK;
    for (int i = 0; i < cnt; ++i) {
      const hamt_ln_reduce_desc& d = descs[b0 + i];
      int nwv, nb;
      ln_bwd_geometry(d.M, &nwv, &nb);
      ch.e[i] = LnRedEntry{d.ws, d.dgamma, d.dbeta, d.dxsum, nb, d.H, d.atomic, 0};
    }
    hipLaunchKernelGGL(ln_red_table_write_kernel, dim3(1), dim3(64), 0, s, ch, tab, b0, cnt);
  }
  const int bpe = 3 * hmax / 64;
  hipLaunchKernelGGL(ln_bwd_reduce_grouped_kernel, dim3(n * bpe), dim3(256), 0, s, tab, bpe);
  HAMT_CHECK_LAUNCH("hamt_ln_bwd_reduce_grouped");
  return HAMT_OK;
}
