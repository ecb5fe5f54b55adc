// The two-stream visual embedding of gfx950:   e = LN_img(x1) + LN_ang(ang W_ang^T + b_ang)
// where x1 = img W_img^T + b_img is the dense layer's output (bf16 in bf16 mode, what a linear returns under autocast; fp32 in
// fp32 mode).  ONE launch instead of four (the K = 4 angle projection, two LayerNorms, the sum): the angle projection is 4 FMAs
// per element and never exists in memory, the two normalised streams are added in registers, the bf16 image of the sum the
// panorama encoder's first GEMM reads is written alongside.  Backward is one launch + a small reduction: dx1 (the dense layer's
// output gradient, as the padded bf16 image its weight-gradient GEMM reads, or fp32), the angle projection recomputed, and the
// column sums that are the gradients of both LayerNorms' parameters and of the angle projection -- per-block partials, summed
// into the parameters' gradient slots by the second kernel.
//
// Replaces ImageEmbeddings.forward's / HistoryEmbeddings.forward's
//     img_layer_norm(img_linear(img)) + ang_layer_norm(ang_linear(ang))            (vilmodel.py:498-500, 549-551, 557-558;
//                                                                                    finetune vilmodel_cmt.py:575-578, 585-586)
// One 64-lane wave per row, H <= 1024 kept in registers as float4s, fp32 statistics; angle_feat_size is 4 everywhere in the
// reference (r2r_model_config.json / vlnbert_init.py) and the only size built.
#include "common.h"

namespace {

constexpr int VE_A = 4;            // angle features
constexpr int VE_V = 4 + VE_A;     // partial vectors per block: dgamma_img, dbeta (both), dgamma_ang, db_ang, dW_ang[:, 0..3]

__device__ __forceinline__ float4 ve_load_x(const void* xv, size_t o, int bf16) {
  if (bf16) {
    const uint2 u = *(const uint2*)((const bf16_t*)xv + o);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  }
  return *(const float4*)((const float*)xv + o);
}

// a[j] = b[c + j] + sum_k ang[k] * W[(c + j) * 4 + k], j = 0..3  (W is nn.Linear's [H, 4]: one float4 per output column)
__device__ __forceinline__ float4 ve_angle(const float* __restrict__ W, const float* __restrict__ b, int c, const float4 ang) {
  const float4 w0 = *(const float4*)(W + (size_t)c * 4), w1 = *(const float4*)(W + (size_t)c * 4 + 4);
  const float4 w2 = *(const float4*)(W + (size_t)c * 4 + 8), w3 = *(const float4*)(W + (size_t)c * 4 + 12);
  const float4 bb = *(const float4*)(b + c);
  float4 a;
  a.x = bb.x + ang.x * w0.x + ang.y * w0.y + ang.z * w0.z + ang.w * w0.w;
  a.y = bb.y + ang.x * w1.x + ang.y * w1.y + ang.z * w1.z + ang.w * w1.w;
  a.z = bb.z + ang.x * w2.x + ang.y * w2.y + ang.z * w2.z + ang.w * w2.w;
  a.w = bb.w + ang.x * w3.x + ang.y * w3.y + ang.z * w3.z + ang.w * w3.w;
  return a;
}

__device__ __forceinline__ void wave_sum2(float& a, float& b) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
}

template <int NV>
__global__ __launch_bounds__(256) void vis_embed_fwd_kernel(hamt_vis_embed_desc d, const void* __restrict__ x1, const float* __restrict__ ang,
                                                            const float* __restrict__ W2, const float* __restrict__ b2,
                                                            const float* __restrict__ g1, const float* __restrict__ be1,
                                                            const float* __restrict__ g2, const float* __restrict__ be2,
                                                            float* __restrict__ y, bf16_t* __restrict__ y16, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  const int H = d.H;
  if (row >= d.M) {   // rows [M, Mpad16) of the bf16 image are zero (reduction padding of the fast GEMMs)
    if (y16 && row < d.Mpad16)
      for (int c = lane * 4; c < H; c += 256) *(uint2*)(y16 + (size_t)row * H + c) = make_uint2(0u, 0u);
    return;
  }
  const float4 av = *(const float4*)(ang + (size_t)row * d.ld_ang);
  float4 v[NV], a[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      v[i] = ve_load_x(x1, (size_t)row * H + c, d.x_bf16);
      a[i] = ve_angle(W2, b2, c, av);
      s1 += v[i].x + v[i].y + v[i].z + v[i].w;
      s2 += a[i].x + a[i].y + a[i].z + a[i].w;
    } else { v[i] = make_float4(0.f, 0.f, 0.f, 0.f); a[i] = v[i]; }
  }
  wave_sum2(s1, s2);
  const float m1 = s1 / (float)H, m2 = s2 / (float)H;
  float q1 = 0.f, q2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      float e0 = v[i].x - m1, e1 = v[i].y - m1, e2 = v[i].z - m1, e3 = v[i].w - m1;
      q1 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      e0 = a[i].x - m2; e1 = a[i].y - m2; e2 = a[i].z - m2; e3 = a[i].w - m2;
      q2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
  }
  wave_sum2(q1, q2);
  const float r1 = rsqrtf(q1 / (float)H + d.eps1), r2 = rsqrtf(q2 / (float)H + d.eps2);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const size_t o = (size_t)row * H + c;
      const float4 ga = *(const float4*)(g1 + c), ba = *(const float4*)(be1 + c), gb = *(const float4*)(g2 + c), bb = *(const float4*)(be2 + c);
      float4 r;
      r.x = ((v[i].x - m1) * r1 * ga.x + ba.x) + ((a[i].x - m2) * r2 * gb.x + bb.x);
      r.y = ((v[i].y - m1) * r1 * ga.y + ba.y) + ((a[i].y - m2) * r2 * gb.y + bb.y);
      r.z = ((v[i].z - m1) * r1 * ga.z + ba.z) + ((a[i].z - m2) * r2 * gb.z + bb.z);
      r.w = ((v[i].w - m1) * r1 * ga.w + ba.w) + ((a[i].w - m2) * r2 * gb.w + bb.w);
      *(float4*)(y + o) = r;
      if (y16) *(uint2*)(y16 + o) = make_uint2(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w));
    }
  }
  if (lane == 0) {
    stats[row] = m1; stats[(size_t)d.M + row] = r1; stats[2 * (size_t)d.M + row] = m2; stats[3 * (size_t)d.M + row] = r2;
  }
}

#define VE_ACC(dst, p, q) { dst.x += p.x * q.x; dst.y += p.y * q.y; dst.z += p.z * q.z; dst.w += p.w * q.w; }

template <int NV, int NWV>
__global__ __launch_bounds__(64 * NWV) void vis_embed_bwd_kernel(hamt_vis_embed_desc d, const float* __restrict__ dy, const void* __restrict__ x1,
                                                                 const float* __restrict__ ang, const float* __restrict__ W2,
                                                                 const float* __restrict__ b2, const float* __restrict__ g1,
                                                                 const float* __restrict__ g2, const float* __restrict__ stats,
                                                                 float* __restrict__ dx, bf16_t* __restrict__ dx16, float* __restrict__ ws) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int H = d.H;
  if (dx16 && blockIdx.x == 0)      // zero the reduction-padding rows [M, Mpad16) of the bf16 gradient image
    for (int row = d.M + w; row < d.Mpad16; row += NWV)
      for (int c = lane * 4; c < H; c += 256) *(uint2*)(dx16 + (size_t)row * H + c) = make_uint2(0u, 0u);
  float4 acc[VE_V][NV];
#pragma unroll
  for (int v = 0; v < VE_V; ++v)
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[v][i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int row = blockIdx.x * NWV + w; row < d.M; row += gridDim.x * NWV) {
    const float m1 = stats[row], r1 = stats[(size_t)d.M + row], m2 = stats[2 * (size_t)d.M + row], r2 = stats[3 * (size_t)d.M + row];
    const float4 av = *(const float4*)(ang + (size_t)row * d.ld_ang);
    float4 ga[NV], gb[NV], h1[NV], h2[NV];
    float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        const size_t o = (size_t)row * H + c;
        const float4 g = *(const float4*)(dy + o);
        const float4 xv = ve_load_x(x1, o, d.x_bf16), aa = ve_angle(W2, b2, c, av);
        float4 p, q;
        p.x = (xv.x - m1) * r1; p.y = (xv.y - m1) * r1; p.z = (xv.z - m1) * r1; p.w = (xv.w - m1) * r1;
        q.x = (aa.x - m2) * r2; q.y = (aa.y - m2) * r2; q.z = (aa.z - m2) * r2; q.w = (aa.w - m2) * r2;
        VE_ACC(acc[0][i], g, p)
        acc[1][i].x += g.x; acc[1][i].y += g.y; acc[1][i].z += g.z; acc[1][i].w += g.w;
        VE_ACC(acc[2][i], g, q)
        const float4 k1 = *(const float4*)(g1 + c), k2 = *(const float4*)(g2 + c);
        float4 u, z;
        u.x = g.x * k1.x; u.y = g.y * k1.y; u.z = g.z * k1.z; u.w = g.w * k1.w;
        z.x = g.x * k2.x; z.y = g.y * k2.y; z.z = g.z * k2.z; z.w = g.w * k2.w;
        s1 += u.x + u.y + u.z + u.w;
        t1 += u.x * p.x + u.y * p.y + u.z * p.z + u.w * p.w;
        s2 += z.x + z.y + z.z + z.w;
        t2 += z.x * q.x + z.y * q.y + z.z * q.z + z.w * q.w;
        ga[i] = u; gb[i] = z; h1[i] = p; h2[i] = q;
      }
    }
    wave_sum2(s1, t1);
    wave_sum2(s2, t2);
    const float c1 = s1 / (float)H, e1 = t1 / (float)H, c2 = s2 / (float)H, e2 = t2 / (float)H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        const size_t o = (size_t)row * H + c;
        float4 r, z;
        r.x = r1 * (ga[i].x - c1 - h1[i].x * e1); r.y = r1 * (ga[i].y - c1 - h1[i].y * e1);
        r.z = r1 * (ga[i].z - c1 - h1[i].z * e1); r.w = r1 * (ga[i].w - c1 - h1[i].w * e1);
        if (dx) *(float4*)(dx + o) = r;
        if (dx16) *(uint2*)(dx16 + o) = make_uint2(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w));
        z.x = r2 * (gb[i].x - c2 - h2[i].x * e2); z.y = r2 * (gb[i].y - c2 - h2[i].y * e2);
        z.z = r2 * (gb[i].z - c2 - h2[i].z * e2); z.w = r2 * (gb[i].w - c2 - h2[i].w * e2);
        acc[3][i].x += z.x; acc[3][i].y += z.y; acc[3][i].z += z.z; acc[3][i].w += z.w;
        acc[4][i].x += z.x * av.x; acc[4][i].y += z.y * av.x; acc[4][i].z += z.z * av.x; acc[4][i].w += z.w * av.x;
        acc[5][i].x += z.x * av.y; acc[5][i].y += z.y * av.y; acc[5][i].z += z.z * av.y; acc[5][i].w += z.w * av.y;
        acc[6][i].x += z.x * av.z; acc[6][i].y += z.y * av.z; acc[6][i].z += z.z * av.z; acc[6][i].w += z.w * av.z;
        acc[7][i].x += z.x * av.w; acc[7][i].y += z.y * av.w; acc[7][i].z += z.z * av.w; acc[7][i].w += z.w * av.w;
      }
    }
  }
  // block partials ws[block][v][H], one vector at a time through LDS (NWV x NV*64 float4)
  __shared__ float4 red[NWV][NV * 64];
#pragma unroll
  for (int v = 0; v < VE_V; ++v) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[w][i * 64 + lane] = acc[v][i];
    __syncthreads();
    for (int e = threadIdx.x; e < NV * 64; e += 64 * NWV) {
      const int c = e * 4;
      if (c < H) {
        float4 t = red[0][e];
#pragma unroll
        for (int k = 1; k < NWV; ++k) { const float4 r = red[k][e]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
        *(float4*)(ws + ((size_t)blockIdx.x * VE_V + v) * H + c) = t;
      }
    }
    __syncthreads();
  }
}

struct VeOut { float* p[6]; };    // dgamma_img, dbeta_img, dgamma_ang, dbeta_ang, db_ang [H each], dW_ang [H][4]

// ws[nb][VE_V][H] -> the six gradients (ADDED to what is there).  block = 64 columns (16 float4 lanes) x 16 partial-row phases.
__global__ __launch_bounds__(256) void vis_embed_reduce_kernel(int nb, int H, const float* __restrict__ ws, VeOut out) {
  const int l16 = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int per = H / 64;                          // blocks per vector
  const int v = blockIdx.x / per, col = (blockIdx.x % per) * 64 + l16 * 4;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = ph; b0 < nb; b0 += 64) {           // 4 independent loads per trip
    float4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + 16 * u;
      q[u] = b < nb ? *(const float4*)(ws + ((size_t)b * VE_V + v) * H + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { t.x += q[u].x; t.y += q[u].y; t.z += q[u].z; t.w += q[u].w; }
  }
  __shared__ float4 red[16][16];
  red[ph][l16] = t;
  __syncthreads();
  if (ph == 0) {
    t = red[0][l16];
#pragma unroll
    for (int i = 1; i < 16; ++i) { const float4 r = red[i][l16]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
    const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = col + k;
      if (v == 0) out.p[0][c] += e[k];
      else if (v == 1) { if (out.p[1]) out.p[1][c] += e[k]; if (out.p[3]) out.p[3][c] += e[k]; }
      else if (v == 2) out.p[2][c] += e[k];
      else if (v == 3) out.p[4][c] += e[k];
      else out.p[5][(size_t)c * VE_A + (v - 4)] += e[k];
    }
  }
}

void ve_geometry(int M, int* nb) {      // 8-wave blocks, >= 2 rows per wave, <= 256 partials
  int b = (M + 15) / 16;
  *nb = b < 1 ? 1 : (b > 256 ? 256 : b);
}

}  // namespace

size_t hamt_vis_embed_ws_bytes(int M, int H) {     // (hamt_workspace_bytes: HAMT_WS_VIS_EMBED_BWD)
  int nb;
  ve_geometry(M, &nb);
  return (size_t)nb * VE_V * H * 4;
}

extern "C" int hamt_vis_embed_fwd(const hamt_vis_embed_desc* d, const void* x1, const float* ang, const float* w_ang, const float* b_ang,
                                  const float* gamma_img, const float* beta_img, const float* gamma_ang, const float* beta_ang,
                                  float* y, void* y16, float* stats, void* stream) {
  HAMT_CHECK_ARG(d && x1 && ang && w_ang && b_ang && gamma_img && beta_img && gamma_ang && beta_ang && y && stats, "hamt_vis_embed_fwd: null pointer");
  HAMT_CHECK_ARG(d->H % 64 == 0 && d->H >= 64 && d->H <= 1024, "hamt_vis_embed_fwd: H=%d unsupported (need H%%64==0, H<=1024)", d->H);
  HAMT_CHECK_ARG(d->A == VE_A && d->ld_ang >= VE_A && d->ld_ang % 4 == 0, "hamt_vis_embed_fwd: angle features %d (ld %d): only 4 is built", d->A, d->ld_ang);
  if (d->M == 0) return HAMT_OK;
  const int nv = (d->H + 255) / 256;
  const int rows = (y16 && d->Mpad16 > d->M) ? d->Mpad16 : d->M;
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(NV) hipLaunchKernelGGL((vis_embed_fwd_kernel<NV>), grid, block, 0, s, *d, x1, ang, w_ang, b_ang, gamma_img, beta_img, gamma_ang, beta_ang, y, (bf16_t*)y16, stats)
  switch (nv) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
  HAMT_CHECK_LAUNCH("hamt_vis_embed_fwd");
  return HAMT_OK;
}

extern "C" int hamt_vis_embed_bwd(const hamt_vis_embed_desc* d, const float* dy, const void* x1, const float* ang, const float* w_ang,
                                  const float* b_ang, const float* gamma_img, const float* gamma_ang, const float* stats, float* dx, void* dx16,
                                  float* dgamma_img, float* dbeta_img, float* dgamma_ang, float* dbeta_ang, float* db_ang, float* dw_ang,
                                  float* ws, void* stream) {
  HAMT_CHECK_ARG(d && dy && x1 && ang && w_ang && b_ang && gamma_img && gamma_ang && stats && ws, "hamt_vis_embed_bwd: null pointer");
  HAMT_CHECK_ARG(dgamma_img && dgamma_ang && db_ang && dw_ang && (dbeta_img || dbeta_ang), "hamt_vis_embed_bwd: null gradient pointer");
  HAMT_CHECK_ARG(dx || dx16, "hamt_vis_embed_bwd: neither dx nor dx16");
  HAMT_CHECK_ARG(d->H % 64 == 0 && d->H >= 64 && d->H <= 1024, "hamt_vis_embed_bwd: H=%d unsupported (need H%%64==0, H<=1024)", d->H);
  HAMT_CHECK_ARG(d->A == VE_A && d->ld_ang >= VE_A && d->ld_ang % 4 == 0, "hamt_vis_embed_bwd: angle features %d (ld %d): only 4 is built", d->A, d->ld_ang);
  if (d->M == 0) return HAMT_OK;
  int nb;
  ve_geometry(d->M, &nb);
  const int nv = (d->H + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(NV) hipLaunchKernelGGL((vis_embed_bwd_kernel<NV, 8>), dim3(nb), dim3(512), 0, s, *d, dy, x1, ang, w_ang, b_ang, gamma_img, gamma_ang, stats, dx, (bf16_t*)dx16, ws)
  switch (nv) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
#undef LAUNCH
  HAMT_CHECK_LAUNCH("hamt_vis_embed_bwd");
  VeOut o{{dgamma_img, dbeta_img, dgamma_ang, dbeta_ang, db_ang, dw_ang}};
  hipLaunchKernelGGL(vis_embed_reduce_kernel, dim3(VE_V * (d->H / 64)), dim3(256), 0, s, nb, d->H, ws, o);
  HAMT_CHECK_LAUNCH("hamt_vis_embed_bwd (reduce)");
  return HAMT_OK;
}
