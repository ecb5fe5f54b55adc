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
  if (threadIdx.x == 0) {
    // a NEGATIVE label is an ignored row (F.cross_entropy's ignore_index = -100; the reference's -1 "no label" never reaches a loss,
    // r2r_tasks.py:46 + pretrain_cmt.py:161-165): loss 0, zero gradient.  A label >= C is a corrupted input -- torch raises on it;
    // a kernel cannot, so the row's loss and gradient are NaN (never an out-of-bounds read, never a silent zero): the step's
    // loss goes NaN where a debugger can see it instead of training on.
    const int64_t lab = label[r];
    lse[r] = l;
    loss[r] = lab < 0 ? 0.f : (lab < C ? l - xr[lab] : __builtin_nanf(""));
  }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(int C, const float* __restrict__ x, int ldx, const int64_t* __restrict__ label,
                                                     const float* __restrict__ lse, const float* __restrict__ g,
                                                     float* __restrict__ dx, int lddx) {
  const int r = blockIdx.x;
  const float* xr = x + (size_t)r * ldx;
  float* dr = dx + (size_t)r * lddx;
  const float l = lse[r], gr = g[r];
  const int64_t lab64 = label[r];
  const bool ignored = lab64 < 0;                    // (an ignored row: zero gradient)
  const float bad = lab64 >= C ? __builtin_nanf("") : 0.f;      // (a label behind the last class: NaN, see ce_fwd_kernel)
  const int lab = (int)lab64;
  for (int c = threadIdx.x; c < C; c += 256) dr[c] = ignored ? 0.f : gr * (expf(xr[c] - l) - (c == lab ? 1.0f : 0.0f)) + bad;
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

// A2C rollout loss of the finetune agent (finetune_src/r2r/agent_cmt.py:476-518), all T steps and B episodes in one launch:
// one thread per episode runs the reverse discounted-return scan R_t = gamma R_{t+1} + r_t (seeded with the critic's value of
// the last state for episodes that have not ended, :480-484) and sums, per step, the policy term -log pi(a_t) (R_t - V_t),
// the critic term 1/2 (R_t - V_t)^2 and (feedback == 'sample') the entropy term -w H_t, each times mask_t (:489-500).  The
// advantage inside the policy term is a constant (`.detach()`, :493).  out[b] = {policy, critic, entropy} sums of episode b.
__global__ void a2c_fwd_kernel(int T, int B, const float* __restrict__ reward, const float* __restrict__ mask, const float* __restrict__ value,
                               const float* __restrict__ logp, const float* __restrict__ ent, const float* __restrict__ last_value,
                               float gamma, float ent_w, float* __restrict__ ret, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float R = last_value ? last_value[b] : 0.f, pol = 0.f, crit = 0.f, en = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    const size_t i = (size_t)t * B + b;
    R = R * gamma + reward[i];
    ret[i] = R;
    const float a = R - value[i], m = mask[i];
    pol += -logp[i] * a * m;
    crit += 0.5f * a * a * m;
    if (ent) en += -ent_w * ent[i] * m;
  }
  out[3 * b] = pol; out[3 * b + 1] = crit; out[3 * b + 2] = en;
}
// gradients for an upstream scale g (d loss / d out summed: the caller's normalisation): dlogp = -(R - V) m g,
// dV = -(R - V) m g (critic term only), dent = -w m g
__global__ void a2c_bwd_kernel(size_t n, const float* __restrict__ ret, const float* __restrict__ mask, const float* __restrict__ value,
                               float ent_w, const float* __restrict__ g, float* __restrict__ dlogp, float* __restrict__ dvalue,
                               float* __restrict__ dent) {
  const float gs = *g;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float a = ret[i] - value[i], m = mask[i];
    dlogp[i] = -a * m * gs;
    dvalue[i] = -a * m * gs;
    if (dent) dent[i] = -ent_w * m * gs;
  }
}

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

extern "C" int hamt_a2c_fwd(int T, int B, const float* reward, const float* mask, const float* value, const float* logp, const float* ent,
                            const float* last_value, float gamma, float ent_w, float* ret, float* out, void* stream) {
  HAMT_CHECK_ARG(T >= 0 && B >= 0 && reward && mask && value && logp && ret && out, "hamt_a2c_fwd: bad argument");
  if (B == 0) return HAMT_OK;
  hipLaunchKernelGGL(a2c_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), T, B, reward, mask, value, logp, ent, last_value, gamma, ent_w, ret, out);
  HAMT_CHECK_LAUNCH("hamt_a2c_fwd");
  return HAMT_OK;
}
extern "C" int hamt_a2c_bwd(int T, int B, const float* ret, const float* mask, const float* value, float ent_w, const float* g,
                            float* dlogp, float* dvalue, float* dent, void* stream) {
  HAMT_CHECK_ARG(T >= 0 && B >= 0 && ret && mask && value && g && dlogp && dvalue, "hamt_a2c_bwd: bad argument");
  const size_t n = (size_t)T * B;
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(a2c_bwd_kernel, dim3((int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, as_stream(stream), n, ret, mask, value, ent_w, g, dlogp, dvalue, dent);
  HAMT_CHECK_LAUNCH("hamt_a2c_bwd");
  return HAMT_OK;
}
