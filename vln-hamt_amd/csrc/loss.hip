// Proxy-task losses, reduction='none' exactly as the reference calls them:
//   F.cross_entropy  (pretrain_cmt.py:154-156, 180, 259)   -inf logits allowed (SAP masked_fill_)
//   F.mse_loss       (pretrain_cmt.py:197, 219)
//   F.kl_div(log_softmax(x), t).sum(1)  (pretrain_cmt.py:239-240), with 0*log 0 = 0
// One 256-thread workgroup per row; the 30 522-wide MLM rows are streamed twice (max, sum) from L2.
#include "common.h"

namespace {

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return v;
}
__device__ __forceinline__ float row_lse(const float* x, int C, float* red) {
  float m = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, x[c]);
  m = block_max(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += expf(x[c] - m);
  s = block_sum(s, red);
  return m + logf(s);
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(int C, const float* __restrict__ x, int ldx, const int64_t* __restrict__ label,
                                                     float* __restrict__ loss, float* __restrict__ lse) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* xr = x + (size_t)r * ldx;
  const float l = row_lse(xr, C, red);
  if (threadIdx.x == 0) { lse[r] = l; loss[r] = l - xr[label[r]]; }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(int C, const float* __restrict__ x, int ldx, const int64_t* __restrict__ label,
                                                     const float* __restrict__ lse, const float* __restrict__ g,
                                                     float* __restrict__ dx, int lddx) {
  const int r = blockIdx.x;
  const float* xr = x + (size_t)r * ldx;
  float* dr = dx + (size_t)r * lddx;
  const float l = lse[r], gr = g[r];
  const int lab = (int)label[r];
  for (int c = threadIdx.x; c < C; c += 256) dr[c] = gr * (expf(xr[c] - l) - (c == lab ? 1.0f : 0.0f));
}
__global__ void mse_fwd_kernel(size_t n, const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ loss) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = x[i] - t[i];
    loss[i] = d * d;
  }
}
__global__ void mse_bwd_kernel(size_t n, const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ g,
                               float* __restrict__ dx) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = 2.0f * g[i] * (x[i] - t[i]);
}
__global__ __launch_bounds__(256) void kl_fwd_kernel(int C, const float* __restrict__ x, int ldx, const float* __restrict__ t, int ldt,
                                                     float* __restrict__ loss, float* __restrict__ lse) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* xr = x + (size_t)r * ldx;
  const float* tr = t + (size_t)r * ldt;
  const float l = row_lse(xr, C, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float tv = tr[c];
    s += (tv > 0.f ? tv * logf(tv) : 0.f) - tv * (xr[c] - l);  // xlogy(t,t) - t*log_softmax(x)
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) { loss[r] = s; lse[r] = l; }
}
__global__ __launch_bounds__(256) void kl_bwd_kernel(int C, const float* __restrict__ x, int ldx, const float* __restrict__ t, int ldt,
                                                     const float* __restrict__ lse, const float* __restrict__ g,
                                                     float* __restrict__ dx, int lddx) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* xr = x + (size_t)r * ldx;
  const float* tr = t + (size_t)r * ldt;
  float ts = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) ts += tr[c];
  ts = block_sum(ts, red);
  const float l = lse[r], gr = g[r];
  for (int c = threadIdx.x; c < C; c += 256) dx[(size_t)r * lddx + c] = gr * (expf(xr[c] - l) * ts - tr[c]);
}
inline int nb(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

}  // namespace

extern "C" int hamt_ce_fwd(int R, int C, const float* x, int ldx, const int64_t* label, float* loss, float* lse, void* stream) {
  HAMT_CHECK_ARG(x && label && loss && lse && C > 0, "hamt_ce_fwd: bad argument");
  if (R == 0) return HAMT_OK;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(R), dim3(256), 0, as_stream(stream), C, x, ldx, label, loss, lse);
  HAMT_CHECK_LAUNCH("hamt_ce_fwd");
  return HAMT_OK;
}
extern "C" int hamt_ce_bwd(int R, int C, const float* x, int ldx, const int64_t* label, const float* lse, const float* g,
                           float* dx, int lddx, void* stream) {
  HAMT_CHECK_ARG(x && label && lse && g && dx, "hamt_ce_bwd: null pointer");
  if (R == 0) return HAMT_OK;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(R), dim3(256), 0, as_stream(stream), C, x, ldx, label, lse, g, dx, lddx);
  HAMT_CHECK_LAUNCH("hamt_ce_bwd");
  return HAMT_OK;
}
extern "C" int hamt_mse_fwd(size_t n, const float* x, const float* t, float* loss, void* stream) {
  HAMT_CHECK_ARG(x && t && loss, "hamt_mse_fwd: null pointer");
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(nb(n)), dim3(256), 0, as_stream(stream), n, x, t, loss);
  HAMT_CHECK_LAUNCH("hamt_mse_fwd");
  return HAMT_OK;
}
extern "C" int hamt_mse_bwd(size_t n, const float* x, const float* t, const float* g, float* dx, void* stream) {
  HAMT_CHECK_ARG(x && t && g && dx, "hamt_mse_bwd: null pointer");
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(nb(n)), dim3(256), 0, as_stream(stream), n, x, t, g, dx);
  HAMT_CHECK_LAUNCH("hamt_mse_bwd");
  return HAMT_OK;
}
extern "C" int hamt_kl_fwd(int R, int C, const float* x, int ldx, const float* t, int ldt, float* loss, float* lse, void* stream) {
  HAMT_CHECK_ARG(x && t && loss && lse && C > 0, "hamt_kl_fwd: bad argument");
  if (R == 0) return HAMT_OK;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(R), dim3(256), 0, as_stream(stream), C, x, ldx, t, ldt, loss, lse);
  HAMT_CHECK_LAUNCH("hamt_kl_fwd");
  return HAMT_OK;
}
extern "C" int hamt_kl_bwd(int R, int C, const float* x, int ldx, const float* t, int ldt, const float* lse, const float* g,
                           float* dx, int lddx, void* stream) {
  HAMT_CHECK_ARG(x && t && lse && g && dx, "hamt_kl_bwd: null pointer");
  if (R == 0) return HAMT_OK;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(R), dim3(256), 0, as_stream(stream), C, x, ldx, t, ldt, lse, g, dx, lddx);
  HAMT_CHECK_LAUNCH("hamt_kl_bwd");
  return HAMT_OK;
}
