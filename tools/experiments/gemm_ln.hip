// Dense layer + bias + dropout + residual + LayerNorm in ONE kernel (forward):
//     o = A W^T + b ;  z = dropout(o) + res ;  y = LN(z) gamma + beta
// = BertSelfOutput / BertOutput / the output half of BertXAttention (vilmodel.py:139-143, 181-185, 351-360) without the
// round trip of `o` through HBM and without the second launch (the "proj_res_ln" row of SURVEY 2c; north_star's fused
// projection + LayerNorm).  The bf16 path otherwise runs hamt_gemm (bf16 o) followed by hamt_ln_fwd.
//
// Shape of the kernel.  LayerNorm needs whole rows, so a workgroup owns BM rows x ALL H = 768 columns: 8 waves, the 768 columns as
// three blocks of 256 whose accumulators all stay in registers (a wave: 32 rows x TN columns of each block; BM = 64: 2 x 4 waves,
// TN = 64, 96 accumulator registers; BM = 32: 1 x 8 waves, TN = 32, 48).  The k-loop runs once per column block -- A tile
// [BM x 64] and B tile [256 x 64] per step through a 3-deep LDS-DMA ring that is prefetched across the block boundaries -- so
// the weight matrix streams through every workgroup once (H x K x 2 bytes per workgroup: the kernel is bound by the CU's
// L2 -> LDS path, not by MFMA; that is why it wins where rows are many and K is short, see DESIGN).  Epilogue: bias, the dropout
// mask of hamt_ln_fwd's stream (call_id, row, column group), residual, two-pass row statistics across the four lane groups
// (shuffles) and the WN waves (LDS), then z (bf16, saved for backward), y (fp32 residual stream), y16 (the next GEMM's operand),
// mean and rstd -- the same outputs hamt_ln_fwd leaves, so hamt_ln_bwd serves both.
#include "common.h"

namespace {

#include "gemm_frag.h"

struct GemmLnArgs {
  int M, K, lda, Mpad16;
  float eps, p_pre;
  uint32_t call_id;
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;
  const float* res;
  const float* gamma;
  const float* beta;
  bf16_t* z16;
  float* y;
  bf16_t* y16;
  float* mean;
  float* rstd;
  const uint64_t* rng;
};

constexpr int LN_H = 768, LN_CB = 3, LN_BN = 256;

template <int BM>
__global__ __launch_bounds__(512) void gemm_ln_kernel(GemmLnArgs g) {
  constexpr int NW = 8, WM = BM / 32, WN = NW / WM, TN = LN_BN / WN, FM = 2, FN = TN / 16, NST = BM == 32 ? 4 : 3;
  constexpr int A_ELEMS = BM * BK, B_ELEMS = LN_BN * BK, STAGE = A_ELEMS + B_ELEMS;
  constexpr int NWA = BM / 8 >= NW ? NW : BM / 8;          // waves that carry A pieces (8 rows per 1 KiB piece)
  constexpr int NLD_B = LN_BN / (8 * NW), NLD_A = BM / (8 * NWA);
  __shared__ __attribute__((aligned(16))) bf16_t lds[NST * STAGE];
  __shared__ float red[2][BM][WN];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w / WN, wn = w % WN;
  const int m0 = blockIdx.x * BM;
  const int nk = g.K / BK, IT = LN_CB * nk;
  const bool has_a = w < NWA;                                // wave-uniform
  // Every workgroup reads the SAME weight tiles; marching through them in the same order, the workgroups of an XCD would all ask
  // its L2 for the same lines at the same moment (measured: 17 B/clk/CU).  So each workgroup starts somewhere else: its k-tiles
  // rotated by `krot`, its three column blocks by `crot` (accumulator slot c holds column block (c + crot) % 3) -- fp32 summation
  // order then depends on the row's workgroup, i.e. on the row index only: still deterministic.
  const int xi = blockIdx.x >> 3;                            // index within the XCD (blocks are dealt round-robin to the 8 XCDs)
  const int krot = xi % nk, crot = (xi / nk) % LN_CB;

  f32x4 acc[LN_CB][FM][FN];
#pragma unroll
  for (int c = 0; c < LN_CB; ++c)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[c][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = lds_base_of(lds);
  TileSrc<false, BM, NWA> src_a;
  TileSrc<false, LN_BN, NW> src_b[LN_CB];
  if (has_a) src_a.init(g.lda, m0, g.M - 1, w, lane);
#pragma unroll
  for (int c = 0; c < LN_CB; ++c) src_b[c].init(g.K, c * LN_BN, LN_H - 1, w, lane);
  auto stage = [&](int it) {       // it = slot * nk + step: accumulator slot `slot`, the step-th k-tile of its rotated order
    const unsigned dst = lds0 + (unsigned)((it % NST) * STAGE * 2);
    const int slot = it / nk;
    int kt = it - slot * nk + krot;
    kt = kt >= nk ? kt - nk : kt;
    int cb = slot + crot;
    cb = cb >= LN_CB ? cb - LN_CB : cb;
    if (has_a) src_a.issue(g.A, g.lda, kt * BK, 0, dst, w);
    if (cb == 0) src_b[0].issue(g.W, g.K, kt * BK, 0, dst + (unsigned)(A_ELEMS * 2), w);
    else if (cb == 1) src_b[1].issue(g.W, g.K, kt * BK, 0, dst + (unsigned)(A_ELEMS * 2), w);
    else src_b[2].issue(g.W, g.K, kt * BK, 0, dst + (unsigned)(A_ELEMS * 2), w);
  };
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < IT) stage(p);

  int it = 0;
#pragma unroll
  for (int cb = 0; cb < LN_CB; ++cb) {
    for (int kt = 0; kt < nk; ++kt, ++it) {
      // tile `it` has landed when at most the pieces of the NST - 2 younger tiles are outstanding
      const int younger = min(NST - 2, IT - 1 - it);
      if (younger >= 2) { if (has_a) wait_vmcnt_c<2 * (NLD_B + NLD_A)>(); else wait_vmcnt_c<2 * NLD_B>(); }
      else if (younger == 1) { if (has_a) wait_vmcnt_c<NLD_B + NLD_A>(); else wait_vmcnt_c<NLD_B>(); }
      else wait_vmcnt_c<0>();
      __builtin_amdgcn_s_barrier();             // everyone's share of tile `it` is in LDS; everyone is done with tile it - 1
      if (it + NST - 1 < IT) stage(it + NST - 1);
      const bf16_t* As = lds + (it % NST) * STAGE;
      const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = frag<false, BM>(As, wm * 32 + i * 16, s, lane);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = frag<false, LN_BN>(Bs, wn * TN + j * 16, s, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)   // swapped roles: a lane owns C[m = lane & 15][n = 4 (lane >> 4) .. + 3] of a fragment
            acc[cb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[cb][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }

  // ---------------------------------------------------------------- epilogue
  const int lr = lane & 15, lg = lane >> 4;
  const RngKey kpre = rng_key(g.rng, g.call_id);
  const float ik = g.p_pre > 0.f ? 1.0f / (1.0f - g.p_pre) : 1.0f;
  float s1[FM] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = m0 + wm * 32 + i * 16 + lr, rr = row < g.M ? row : g.M - 1;
    const uint32_t rowh = hamt_mix32((uint32_t)rr ^ kpre.k0);
#pragma unroll
    for (int c = 0; c < LN_CB; ++c)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = ((c + crot) % LN_CB) * LN_BN + wn * TN + j * 16 + lg * 4;
        const float4 b = *(const float4*)(g.bias + col);
        f32x4 v = acc[c][i][j];
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        if (g.p_pre > 0.f) {
          float f_[4];
          drop_scale4(kpre, rowh, (uint32_t)(col >> 2), g.p_pre, ik, f_);
          v[0] *= f_[0]; v[1] *= f_[1]; v[2] *= f_[2]; v[3] *= f_[3];
        }
        const float4 r = *(const float4*)(g.res + (size_t)rr * LN_H + col);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        acc[c][i][j] = v;
        s1[i] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    s1[i] += __shfl_xor(s1[i], 16, 64);
    s1[i] += __shfl_xor(s1[i], 32, 64);
    if (lg == 0) red[0][wm * 32 + i * 16 + lr][wn] = s1[i];
  }
  __syncthreads();
  float mean[FM], s2[FM] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < WN; ++q) tot += red[0][wm * 32 + i * 16 + lr][q];
    mean[i] = tot * (1.0f / LN_H);
#pragma unroll
    for (int c = 0; c < LN_CB; ++c)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f32x4 v = acc[c][i][j];
        const float a = v[0] - mean[i], b = v[1] - mean[i], cc = v[2] - mean[i], e = v[3] - mean[i];
        s2[i] += (a * a + b * b) + (cc * cc + e * e);
      }
    s2[i] += __shfl_xor(s2[i], 16, 64);
    s2[i] += __shfl_xor(s2[i], 32, 64);
    if (lg == 0) red[1][wm * 32 + i * 16 + lr][wn] = s2[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int row = m0 + wm * 32 + i * 16 + lr;
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < WN; ++q) tot += red[1][wm * 32 + i * 16 + lr][q];
    const float rstd = rsqrtf(tot * (1.0f / LN_H) + g.eps);
    if (row < g.M) {
      if (wn == 0 && lg == 0) { g.mean[row] = mean[i]; g.rstd[row] = rstd; }
#pragma unroll
      for (int c = 0; c < LN_CB; ++c)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int col = ((c + crot) % LN_CB) * LN_BN + wn * TN + j * 16 + lg * 4;
          const size_t o = (size_t)row * LN_H + col;
          const f32x4 v = acc[c][i][j];
          const float4 ga = *(const float4*)(g.gamma + col), be = *(const float4*)(g.beta + col);
          float4 r;
          r.x = (v[0] - mean[i]) * rstd * ga.x + be.x; r.y = (v[1] - mean[i]) * rstd * ga.y + be.y;
          r.z = (v[2] - mean[i]) * rstd * ga.z + be.z; r.w = (v[3] - mean[i]) * rstd * ga.w + be.w;
          *(uint2*)(g.z16 + o) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
          *(float4*)(g.y + o) = r;
          *(uint2*)(g.y16 + o) = make_uint2(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w));
        }
    }
  }
  // rows [M, Mpad16) of the bf16 image: zeros (reduction padding of the GEMMs that read it)
  if (blockIdx.x == gridDim.x - 1)
    for (int row = g.M + w; row < g.Mpad16; row += NW)
      for (int c = lane * 4; c < LN_H; c += 256) *(uint2*)(g.y16 + (size_t)row * LN_H + c) = make_uint2(0u, 0u);
}

}  // namespace

extern "C" int hamt_gemm_ln_fwd(const hamt_gemm_ln_desc* d, const void* a16, const void* w16, const float* bias,
                                const float* residual, const float* gamma, const float* beta, void* z16, float* y, void* y16,
                                float* mean, float* rstd, const uint64_t* rng, void* stream) {
  HAMT_CHECK_ARG(d && a16 && w16 && bias && residual && gamma && beta && z16 && y && y16 && mean && rstd, "hamt_gemm_ln_fwd: null pointer");
  HAMT_CHECK_ARG(d->H == LN_H, "hamt_gemm_ln_fwd: H = %d (this kernel is built for H = %d; use hamt_gemm + hamt_ln_fwd)", d->H, LN_H);
  HAMT_CHECK_ARG(d->M >= 1 && d->K >= 64 && d->K % 64 == 0 && d->lda >= d->K && d->lda % 8 == 0, "hamt_gemm_ln_fwd: bad M / K / lda (%d, %d, %d)", d->M, d->K, d->lda);
  HAMT_CHECK_ARG(d->Mpad16 >= d->M, "hamt_gemm_ln_fwd: Mpad16 < M");
  HAMT_CHECK_ARG(((uintptr_t)a16 % 16) == 0 && ((uintptr_t)w16 % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)residual % 16) == 0 &&
                 ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 && ((uintptr_t)z16 % 8) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)y16 % 8) == 0,
                 "hamt_gemm_ln_fwd: operands must be 16-byte aligned");
  HAMT_CHECK_ARG(2.0 * d->M * d->lda < 4294967296.0, "hamt_gemm_ln_fwd: A must be smaller than 4 GiB (32-bit DMA offsets)");
  HAMT_CHECK_ARG(d->p_pre >= 0.f && d->p_pre < 1.f && (d->p_pre == 0.f || rng), "hamt_gemm_ln_fwd: dropout needs the rng state");
  GemmLnArgs g{d->M, d->K, d->lda, d->Mpad16, d->eps, d->p_pre, d->call_id, (const bf16_t*)a16, (const bf16_t*)w16, bias, residual, gamma, beta,
               (bf16_t*)z16, y, (bf16_t*)y16, mean, rstd, rng};
  // rows per workgroup: every workgroup streams the whole weight through its CU whatever its height, so the shorter tile (twice
  // the workgroups, half the MFMA work each) is used until its grid no longer fits the chip in one round
  const int bm = d->tile_rows == 32 || d->tile_rows == 64 ? d->tile_rows : ((d->M + 31) / 32 <= 256 ? 32 : 64);
  hipStream_t s = as_stream(stream);
  if (bm == 32) hipLaunchKernelGGL((gemm_ln_kernel<32>), dim3((d->M + 31) / 32), dim3(512), 0, s, g);
  else hipLaunchKernelGGL((gemm_ln_kernel<64>), dim3((d->M + 63) / 64), dim3(512), 0, s, g);
  hamt_set_last_kernel("gemm_ln_kernel<%d>", bm);
  HAMT_CHECK_LAUNCH("hamt_gemm_ln_fwd");
  return HAMT_OK;
}
