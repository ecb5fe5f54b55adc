// Optimiser-side kernels over flat fp32 arenas (pure HBM streaming, 16-byte accesses):
//   hamt_sumsq       global L2 norm of the gradient arena (torch clip_grad_norm_, main_r2r.py:271-273)
//   hamt_adamw_flat  the reference's HF AdamW (optim/adamw.py:85-110) fused with the clip scaling, the
//                    bf16 shadow refresh used by the MFMA GEMMs, and optimizer.zero_grad (main_r2r.py:280)
// Traffic per parameter: read p,g,m,v (16 B) + write p,m,v (12 B) + g zero (4 B) + bf16 shadow (2 B) = 34 B.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(size_t n, const float* __restrict__ g, float* __restrict__ ws) {
  float s = 0.f;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = ((const float4*)g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(int nb, const float* __restrict__ ws, float* __restrict__ out, int accumulate) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += ws[i];
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float t = (red[0] + red[1]) + (red[2] + red[3]); *out = accumulate ? *out + t : t; }
}
__device__ __forceinline__ float clip_coef(const float* gnorm_sq, float max_norm) {
  if (!gnorm_sq || max_norm <= 0.f) return 1.0f;
  return fminf(1.0f, max_norm / (sqrtf(*gnorm_sq) + 1e-6f));
}
__device__ __forceinline__ void adamw_one(float& p, float& g, float& m, float& v, float coef, float lr, float step, float b1,
                                          float b2, float eps, float wd) {
  const float gg = g * coef;
  m = m * b1 + (1.0f - b1) * gg;                    // exp_avg.mul_(b1).add_(grad, alpha=1-b1)           adamw.py:89
  v = v * b2 + (1.0f - b2) * gg * gg;               // exp_avg_sq.mul_(b2).addcmul_(grad, grad, 1-b2)    adamw.py:90
  p = p - step * (m / (sqrtf(v) + eps));            // denom = sqrt(v)+eps; p.addcdiv_(m, denom, -step)  adamw.py:91-99
  if (wd > 0.f) p = p - lr * wd * p;                // decoupled decay AFTER the update                   adamw.py:109-110
}
__global__ __launch_bounds__(256) void adamw_kernel(size_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ p16, const float* __restrict__ hyper,
                                                    const float* __restrict__ gnorm_sq, float b1, float b2, float eps, float wd,
                                                    int zero_grad) {
  const float lr = hyper[0], step = hyper[1], coef = clip_coef(gnorm_sq, hyper[2]);
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 P = ((float4*)p)[i], G = ((float4*)g)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
    adamw_one(P.x, G.x, M.x, V.x, coef, lr, step, b1, b2, eps, wd);
    adamw_one(P.y, G.y, M.y, V.y, coef, lr, step, b1, b2, eps, wd);
    adamw_one(P.z, G.z, M.z, V.z, coef, lr, step, b1, b2, eps, wd);
    adamw_one(P.w, G.w, M.w, V.w, coef, lr, step, b1, b2, eps, wd);
    ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V;
    if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p16) ((uint2*)p16)[i] = make_uint2(pack_bf2(P.x, P.y), pack_bf2(P.z, P.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    float P = p[i], G = g[i], M = m[i], V = v[i];
    adamw_one(P, G, M, V, coef, lr, step, b1, b2, eps, wd);
    p[i] = P; m[i] = M; v[i] = V;
    if (zero_grad) g[i] = 0.f;
    if (p16) p16[i] = f2bf(P);
  }
}
// One launch for the whole arena: per-parameter hyper-parameters come from a device table
// (ends[i] = exclusive end offset of parameter i, hyp[i] = {lr, step_size, weight_decay, active}); parameters with
// active == 0 (grad is None this step -- the reference's `continue`, adamw.py:70-71) are left untouched.  active == 2: updated
// like 1, but the gradient slot is NOT zeroed (its producer overwrites it: the grouped weight-gradient launch stores, it does
// not accumulate) -- 30 instead of 34 bytes per parameter; hamt_sumsq_table skips inactive slots, whatever they hold.
// Offsets are multiples of 8, so a 4-element chunk never straddles two parameters.
// The blocks SWEEP the arena together: block b takes the 256 * U float4 chunks b, b + gridDim, b + 2 gridDim, ... -- at any moment the whole
// grid works inside one window of gridDim * 16 KiB of each of the five arrays (2 048 blocks: 32 MiB), U float4 of each of p, g, m, v in flight
// per thread, stores non-temporal (nothing re-reads p, m, v before the next step; the bf16 shadow is re-read by the next forward and keeps the
// default policy).  Rounds 2-4 gave every block ONE contiguous range of the arena (2 048 places spread over all 700 MB of each array, the
// hyper-parameters of its one or two parameters in registers): that form's time moved 945 -> 1 140 us from PROCESS to process (and so from
// box to box: VERDICT r4 weak 11) while a plain 1-GiB copy held 5.35 - 5.46 TB/s in every one of them -- ten thousand concurrently streamed
// regions depend on how the driver happened to back the arenas (translation reach), a narrow moving window does not: 968 - 983 us in every
// process (tools/r5_adamw2.sh; 1 024 blocks the same, 4 096 blocks 1 016 - 1 024 us).
template <int U>
__global__ __launch_bounds__(256) void adamw_table_kernel(size_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                                float* __restrict__ v, bf16_t* __restrict__ p16, const int* __restrict__ ends,
                                                                const float4* __restrict__ hyp, int nparams, const float* __restrict__ gnorm_sq,
                                                                float max_norm, float b1, float b2, float eps, int zero_grad, size_t first4) {
  const float coef = clip_coef(gnorm_sq, max_norm);
  const size_t n4 = n >> 2;
  constexpr size_t CH = 256 * U;
  typedef float f4 __attribute__((ext_vector_type(4)));
  int pi = -1;
  for (size_t lo = (size_t)blockIdx.x * CH; lo < n4; lo += (size_t)gridDim.x * CH) {
    const size_t hi = min(n4, lo + CH);
    const long e0 = (long)(lo + first4) * 4;
    if (pi < 0) {   // binary search once: first parameter whose end is beyond this block's first element
      int a = 0, b = nparams - 1;
      while (a < b) { const int mid = (a + b) >> 1; if ((long)ends[mid] > e0) b = mid; else a = mid + 1; }
      pi = a;
    } else {
      while (pi < nparams - 1 && (long)ends[pi] <= e0) ++pi;       // (the chunks of a block only move forward)
    }
    int q = pi;
    for (size_t seg = lo; seg < hi; ++q) {
      const size_t pend = q < nparams - 1 ? min(hi, ((size_t)ends[q] >> 2) - first4) : hi;
      const float4 h = hyp[q];
      if (h.w != 0.f) {
        const bool zg = zero_grad && h.w == 1.f;
        const size_t i = seg + threadIdx.x;
        f4 P[U], G[U], M[U], V[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const size_t j = i + (size_t)u * 256;
          if (j < pend) { P[u] = ((const f4*)p)[j]; G[u] = ((const f4*)g)[j]; M[u] = ((const f4*)m)[j]; V[u] = ((const f4*)v)[j]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const size_t j = i + (size_t)u * 256;
          if (j < pend) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { float pp = P[u][c], gg = G[u][c], mm = M[u][c], vv = V[u][c]; adamw_one(pp, gg, mm, vv, coef, h.x, h.y, b1, b2, eps, h.z); P[u][c] = pp; M[u][c] = mm; V[u][c] = vv; }
            __builtin_nontemporal_store(P[u], (f4*)p + j);
            __builtin_nontemporal_store(M[u], (f4*)m + j);
            __builtin_nontemporal_store(V[u], (f4*)v + j);
            if (zg) __builtin_nontemporal_store((f4){0.f, 0.f, 0.f, 0.f}, (f4*)g + j);
            if (p16) ((uint2*)p16)[j] = make_uint2(pack_bf2(P[u][0], P[u][1]), pack_bf2(P[u][2], P[u][3]));
          }
        }
      }
      seg = pend;
    }
  }
}
// sum(g^2) over the ACTIVE parameters of an arena range (same segment walk as adamw_table_kernel): slots of inactive
// parameters are not read (they may hold a stale gradient, see active == 2 above).  One partial per block, summed in block
// order by sumsq_final_kernel: deterministic.
__global__ __launch_bounds__(256) void sumsq_table_partial_kernel(size_t n, const float* __restrict__ g, const int* __restrict__ ends,
                                                                  const float4* __restrict__ hyp, int nparams, size_t first4,
                                                                  float* __restrict__ ws) {
  // (the blocks sweep the range together, 16-KiB chunks b, b + gridDim, ...: see adamw_table_kernel; a block's partial is the sum over ITS
  // chunks in order, the partials are summed in block order: a fixed partition for a given grid, deterministic)
  const size_t n4 = n >> 2;
  constexpr size_t CH = 256 * 4;
  float s = 0.f;
  int pi = -1;
  for (size_t lo = (size_t)blockIdx.x * CH; lo < n4; lo += (size_t)gridDim.x * CH) {
    const size_t hi = min(n4, lo + CH);
    const long e0 = (long)(lo + first4) * 4;
    if (pi < 0) {
      int a = 0, b = nparams - 1;
      while (a < b) { const int mid = (a + b) >> 1; if ((long)ends[mid] > e0) b = mid; else a = mid + 1; }
      pi = a;
    } else {
      while (pi < nparams - 1 && (long)ends[pi] <= e0) ++pi;
    }
    int q = pi;
    for (size_t seg = lo; seg < hi; ++q) {
      const size_t pend = q < nparams - 1 ? min(hi, ((size_t)ends[q] >> 2) - first4) : hi;
      const float act = hyp[q].w;
      if (act != 0.f && act != 3.f) {      // 3: accounted for by the weight-gradient tiles (hamt_wgrad_desc.ss)
        const size_t i = seg + threadIdx.x;
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + (size_t)u * 256 < pend ? ((const float4*)g)[i + (size_t)u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
      }
      seg = pend;
    }
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// The same sum with the ACTIVE elements spread evenly over the blocks.  When most of a range is skipped (the GEMM-weight region once the
// weight-gradient tiles carry their own sums: what is left are the few weights written twice in a pass, ~6 % of the range in a handful
// of places) the uniform element ranges above leave ~60 of 1024 blocks with all the reading: 42 us for 38 MB.  Here every block builds
// the prefix sums of the active lengths (nparams <= SS_NP_MAX entries, LDS), takes an equal share of the active float4s and maps it
// back to arena offsets.  Same determinism: a fixed partition for a given table, partials summed in block order.
constexpr int SS_NP_MAX = 2048;
__global__ __launch_bounds__(256) void sumsq_table_balanced_kernel(size_t n, const float* __restrict__ g, const int* __restrict__ ends,
                                                                   const float4* __restrict__ hyp, int nparams, size_t first4,
                                                                   float* __restrict__ ws) {
  __shared__ unsigned pref[SS_NP_MAX + 1];      // pref[i] = active float4s of parameters [0, i) inside the range
  __shared__ unsigned tsum[256];
  const size_t n4 = n >> 2, r_lo = first4, r_hi = first4 + n4;
  const int t = threadIdx.x, per = (nparams + 255) / 256;
  unsigned loc = 0;
  for (int k = 0; k < per; ++k) {
    const int pi = t * per + k;
    unsigned len = 0;
    if (pi < nparams) {
      const size_t a = pi ? (size_t)ends[pi - 1] >> 2 : 0, b = (size_t)ends[pi] >> 2;
      const size_t lo = a > r_lo ? a : r_lo, hi = b < r_hi ? b : r_hi;
      const float act = hyp[pi].w;
      if (hi > lo && act != 0.f && act != 3.f) len = (unsigned)(hi - lo);
      pref[pi + 1] = len;                       // (own length for now)
    }
    loc += len;
  }
  tsum[t] = loc;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {           // inclusive scan of the 256 thread sums
    const unsigned v = t >= o ? tsum[t - o] : 0u;
    __syncthreads();
    tsum[t] += v;
    __syncthreads();
  }
  unsigned run = t ? tsum[t - 1] : 0u;
  if (t == 0) pref[0] = 0;
  for (int k = 0; k < per; ++k) {
    const int pi = t * per + k;
    if (pi < nparams) { run += pref[pi + 1]; pref[pi + 1] = run; }
  }
  __syncthreads();
  const unsigned total = pref[nparams];
  const unsigned share = (total + gridDim.x - 1) / gridDim.x;
  const unsigned lo = min(total, blockIdx.x * share), hi = min(total, lo + share);
  float s = 0.f;
  if (hi > lo) {
    int a = 0, b = nparams - 1;                 // first parameter whose prefix end lies behind lo
    while (a < b) { const int mid = (a + b) >> 1; if (pref[mid + 1] > lo) b = mid; else a = mid + 1; }
    for (int pi = a; pi < nparams && pref[pi] < hi; ++pi) {
      if (pref[pi + 1] == pref[pi]) continue;
      const size_t pa = pi ? (size_t)ends[pi - 1] >> 2 : 0;
      const size_t base = (pa > r_lo ? pa : r_lo) - r_lo;                 // float4 offset of the parameter's first in-range element in g
      const unsigned o0 = max(lo, pref[pi]) - pref[pi], o1 = min(hi, pref[pi + 1]) - pref[pi];
      for (size_t i = base + o0 + t; i < base + o1; i += 256 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + (size_t)u * 256 < base + o1 ? ((const float4*)g)[i + (size_t)u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
      }
    }
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// The same walk fused with the widening of a reduce-scattered bf16 gradient chunk: g[i] = float(y16[i]) for EVERY element of the
// chunk, sum of squares over the active parameters only -- one pass instead of hamt_wire_unpack_bf16 followed by hamt_sumsq_table
// (the exchange of a data-parallel step runs this once per arena range, behind the range's reduce-scatter).
__global__ __launch_bounds__(256) void unpack_sumsq_partial_kernel(size_t n, const bf16_t* __restrict__ y, float* __restrict__ g,
                                                                   const int* __restrict__ ends, const float4* __restrict__ hyp, int nparams,
                                                                   size_t first4, float* __restrict__ ws) {
  // (16-KiB chunks b, b + gridDim, ...: the blocks sweep the range together, see adamw_table_kernel)
  const size_t n4 = n >> 2;
  constexpr size_t CH = 256 * 4;
  float s = 0.f;
  int pi = -1;
  for (size_t lo = (size_t)blockIdx.x * CH; lo < n4; lo += (size_t)gridDim.x * CH) {
    const size_t hi = min(n4, lo + CH);
    const long e0 = (long)(lo + first4) * 4;
    if (pi < 0) {
      int a = 0, b = nparams - 1;
      while (a < b) { const int mid = (a + b) >> 1; if ((long)ends[mid] > e0) b = mid; else a = mid + 1; }
      pi = a;
    } else {
      while (pi < nparams - 1 && (long)ends[pi] <= e0) ++pi;
    }
    int k = pi;
    for (size_t seg = lo; seg < hi; ++k) {
      const size_t pend = k < nparams - 1 ? min(hi, ((size_t)ends[k] >> 2) - first4) : hi;
      const bool act = hyp[k].w != 0.f && hyp[k].w != 3.f;
      const size_t i = seg + threadIdx.x;
      uint2 u[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) u[q] = i + (size_t)q * 256 < pend ? ((const uint2*)y)[i + (size_t)q * 256] : make_uint2(0u, 0u);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (i + (size_t)q * 256 < pend) {
          const float4 v = make_float4(__uint_as_float(u[q].x << 16), __uint_as_float(u[q].x & 0xffff0000u), __uint_as_float(u[q].y << 16), __uint_as_float(u[q].y & 0xffff0000u));
          ((float4*)g)[i + (size_t)q * 256] = v;
          if (act) s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
      }
      seg = pend;
    }
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void clip_scale_kernel(size_t n, float* __restrict__ g, const float* __restrict__ gnorm_sq, float max_norm) {
  const float coef = clip_coef(gnorm_sq, max_norm);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] *= coef;
}

}  // namespace

extern "C" int hamt_sumsq(size_t n, const float* g, float* out, int accumulate, float* ws, void* stream) {
  HAMT_CHECK_ARG(g && out && ws && ((uintptr_t)g % 16) == 0, "hamt_sumsq: bad argument (ws needs 1024 floats, g 16-byte aligned)");
  size_t b = (n / 4 + 255) / 256;
  int nb = (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, n, g, ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, nb, ws, out, accumulate);
  HAMT_CHECK_LAUNCH("hamt_sumsq");
  return HAMT_OK;
}
extern "C" int hamt_sumsq_table(size_t first, size_t n, const float* g, const int* ends, const float* hyp, int nparams, float* out,
                               int accumulate, float* ws, void* stream) {
  HAMT_CHECK_ARG(g && out && ws && ends && hyp && nparams > 0 && n % 4 == 0 && first % 4 == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)hyp % 16) == 0,
                 "hamt_sumsq_table: bad argument (ws needs 1024 floats, g / hyp 16-byte aligned, first and n multiples of 4)");
  hipStream_t s = as_stream(stream);
  size_t b = (n / 4 + 1023) / 1024;
  int nb = (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
  static const bool balanced = getenv("HAMT_SUMSQ_UNBALANCED") == nullptr;
  const bool sparse = (accumulate & HAMT_SUMSQ_SPARSE) != 0;
  accumulate &= 1;
  if (sparse && balanced && nparams <= SS_NP_MAX && n / 4 < ((size_t)1 << 32)) {
    nb = nb < 512 ? nb : 512;
    hipLaunchKernelGGL(sumsq_table_balanced_kernel, dim3(nb), dim3(256), 0, s, n, g, ends, (const float4*)hyp, nparams, first / 4, ws);
  } else
    hipLaunchKernelGGL(sumsq_table_partial_kernel, dim3(nb), dim3(256), 0, s, n, g, ends, (const float4*)hyp, nparams, first / 4, ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, nb, ws, out, accumulate);
  HAMT_CHECK_LAUNCH("hamt_sumsq_table");
  return HAMT_OK;
}
extern "C" int hamt_wire_unpack_sumsq(size_t first, size_t n, const void* y16, float* g, const int* ends, const float* hyp, int nparams,
                                     float* out, int accumulate, float* ws, void* stream) {
  HAMT_CHECK_ARG(y16 && g && out && ws && ends && hyp && nparams > 0 && n % 4 == 0 && first % 4 == 0 && ((uintptr_t)g % 16) == 0 &&
                 ((uintptr_t)y16 % 8) == 0 && ((uintptr_t)hyp % 16) == 0,
                 "hamt_wire_unpack_sumsq: bad argument (ws needs 1024 floats, g / hyp 16-byte and y16 8-byte aligned, first and n multiples of 4)");
  hipStream_t s = as_stream(stream);
  size_t b = (n / 4 + 1023) / 1024;
  int nb = (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
  hipLaunchKernelGGL(unpack_sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, n, (const bf16_t*)y16, g, ends, (const float4*)hyp, nparams, first / 4, ws);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, nb, ws, out, accumulate);
  HAMT_CHECK_LAUNCH("hamt_wire_unpack_sumsq");
  return HAMT_OK;
}
// out (+)= sum of n floats, one block of 1024 threads, four independent 16-byte loads in flight per thread (the ~18 k tile
// slots of a step's weight gradients: 4 us instead of 15 with the 256-thread scalar loop); fixed order: deterministic
__global__ __launch_bounds__(1024) void sum_partials_kernel(size_t n, const float* __restrict__ x, float* __restrict__ out, int accumulate) {
  float s = 0.f;
  const size_t n4 = n >> 2;
  for (size_t i = threadIdx.x; i < n4; i += 4096) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i + (size_t)u * 1024 < n4 ? ((const float4*)x)[i + (size_t)u * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  if (threadIdx.x < (n & 3)) s += x[n4 * 4 + threadIdx.x];
  __shared__ float red[16];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    *out = accumulate ? *out + t : t;
  }
}
extern "C" int hamt_sumsq_partials(size_t n, const float* partials, float* out, int accumulate, void* stream) {
  HAMT_CHECK_ARG(partials && out && n < (size_t)1 << 30 && ((uintptr_t)partials % 16) == 0, "hamt_sumsq_partials: bad argument");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, as_stream(stream), n, partials, out, accumulate);
  HAMT_CHECK_LAUNCH("hamt_sumsq_partials");
  return HAMT_OK;
}
extern "C" int hamt_adamw_flat(size_t n, float* p, float* g, float* m, float* v, void* p16, const float* hyper,
                               const float* gnorm_sq, float beta1, float beta2, float eps, float weight_decay, int zero_grad,
                               void* stream) {
  HAMT_CHECK_ARG(p && g && m && v && hyper, "hamt_adamw_flat: null pointer");
  HAMT_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                 ((uintptr_t)p16 % 8) == 0, "hamt_adamw_flat: arenas must be 16-byte aligned");
  if (n == 0) return HAMT_OK;
  size_t b = (n / 4 + 255) / 256;
  int nb = (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
  hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, as_stream(stream), n, p, g, m, v, (bf16_t*)p16, hyper, gnorm_sq, beta1, beta2, eps, weight_decay, zero_grad);
  HAMT_CHECK_LAUNCH("hamt_adamw_flat");
  return HAMT_OK;
}
extern "C" int hamt_adamw_table_range(size_t first, size_t n, float* p, float* g, float* m, float* v, void* p16, const int* ends, const float* hyp,
                                      int nparams, const float* gnorm_sq, float max_norm, float beta1, float beta2, float eps,
                                      int zero_grad, void* stream);
extern "C" int hamt_adamw_table(size_t n, float* p, float* g, float* m, float* v, void* p16, const int* ends, const float* hyp,
                                int nparams, const float* gnorm_sq, float max_norm, float beta1, float beta2, float eps,
                                int zero_grad, void* stream) {
  return hamt_adamw_table_range(0, n, p, g, m, v, p16, ends, hyp, nparams, gnorm_sq, max_norm, beta1, beta2, eps, zero_grad, stream);
}
extern "C" int hamt_adamw_table_range(size_t first, size_t n, float* p, float* g, float* m, float* v, void* p16, const int* ends, const float* hyp,
                                      int nparams, const float* gnorm_sq, float max_norm, float beta1, float beta2, float eps,
                                      int zero_grad, void* stream) {
  HAMT_CHECK_ARG(p && g && m && v && ends && hyp && nparams > 0 && n % 4 == 0 && first % 4 == 0, "hamt_adamw_table: bad argument");
  HAMT_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                 ((uintptr_t)p16 % 8) == 0 && ((uintptr_t)hyp % 16) == 0, "hamt_adamw_table: arenas must be 16-byte aligned");
  if (n == 0) return HAMT_OK;
  size_t b = (n / 4 + 1023) / 1024;              // 16-KiB chunks of 256 x 4 float4; the grid sweeps them together (see the kernel)
  int nb = (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
  hipLaunchKernelGGL((adamw_table_kernel<4>), dim3(nb), dim3(256), 0, as_stream(stream), n, p, g, m, v, (bf16_t*)p16, ends,
                     (const float4*)hyp, nparams, gnorm_sq, max_norm, beta1, beta2, eps, zero_grad, first / 4);
  HAMT_CHECK_LAUNCH("hamt_adamw_table");
  return HAMT_OK;
}
extern "C" int hamt_clip_scale(size_t n, float* g, const float* gnorm_sq, float max_norm, void* stream) {
  HAMT_CHECK_ARG(g && gnorm_sq, "hamt_clip_scale: null pointer");
  if (n == 0) return HAMT_OK;
  size_t b = (n + 255) / 256;
  hipLaunchKernelGGL(clip_scale_kernel, dim3((int)(b > 4096 ? 4096 : b)), dim3(256), 0, as_stream(stream), n, g, gnorm_sq, max_norm);
  HAMT_CHECK_LAUNCH("hamt_clip_scale");
  return HAMT_OK;
}
