// Fast path of hamt_gemm for the shapes that carry the FLOPs of the HAMT step:
//   C[M,N] = epi(A[M,K] * B[N,K]^T), A and B bf16, both K-contiguous ("NT"), K % 64 == 0.
// Every contraction of the step is brought to this form by the host side (forward: x16 * W16^T;
// dgrad: dY16 * (W^T)16^T; wgrad: (dY^T)16 * (X^T)16^T -- transposed bf16 copies come from
// hamt_cast_transpose), so ONE kernel needs tuning.
//
// Structure (cdna_hip_programming.md section 5, "step 3" + T2 swizzle):
//   * 128x128x64 tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 fragments of
//     v_mfma_f32_16x16x32_bf16, fp32 accumulators (64 VGPRs);
//   * operands go HBM/L2 -> LDS directly with global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPR
//     round trip), two LDS stages (2 x 32 KiB), next tile's DMA issued before the current tile's MFMAs;
//   * LDS image is lane-linear per DMA instruction (8 rows x 128 B); the 16-byte k-chunk index is XOR-swizzled
//     with (row & 7) on the SOURCE address and on the fragment read (conflict-free ds_read_b128);
//   * ragged M/N: row indices are clamped for the loads (no OOB), the epilogue masks the stores;
//   * blockIdx -> tile mapping keeps the tiles that share an A row-panel on one XCD (private L2).
#include "common.h"

struct GemmArgsF {
  int M, N, K, lda, ldb, ldc, ldaux;
  int dtype_c, dtype_aux, epi;
  float alpha;
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  void* aux;
};

namespace {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_ELEMS = BM * BK;  // per operand per stage (bf16 elements) = 16 KiB

__device__ __forceinline__ void epi_store_f(const GemmArgsF& g, int row, int col, float acc) {
  if (row >= g.M || col >= g.N) return;
  float v = acc * g.alpha;
  if (g.epi & HAMT_EPI_BIAS) v += g.bias[col];
  const size_t ia = (size_t)row * g.ldaux + col;
  if (g.epi & HAMT_EPI_SAVE_PRE) {
    if (g.dtype_aux == HAMT_BF16) ((bf16_t*)g.aux)[ia] = f2bf(v); else ((float*)g.aux)[ia] = v;
  }
  if (g.epi & HAMT_EPI_GELU) v = gelu_erf(v);
  if (g.epi & HAMT_EPI_RELU) v = fmaxf(v, 0.0f);
  if (g.epi & (HAMT_EPI_MUL_DGELU | HAMT_EPI_MUL_DRELU)) {
    const float h = (g.dtype_aux == HAMT_BF16) ? bf2f(((const bf16_t*)g.aux)[ia]) : ((const float*)g.aux)[ia];
    v *= (g.epi & HAMT_EPI_MUL_DGELU) ? dgelu_erf(h) : (h > 0.0f ? 1.0f : 0.0f);
  }
  const size_t ic = (size_t)row * g.ldc + col;
  if (g.dtype_c == HAMT_BF16) {
    bf16_t* c = (bf16_t*)g.C;
    if (g.epi & HAMT_EPI_ACCUM) v += bf2f(c[ic]);
    c[ic] = f2bf(v);
  } else {
    float* c = (float*)g.C;
    if (g.epi & HAMT_EPI_ACCUM) v += c[ic];
    c[ic] = v;
  }
}

// DMA one operand tile (128 rows x 64 k) into LDS: wave w moves rows [32w, 32w+32) with 4 instructions.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ P, int ld, int r0, int rmax, int k0, bf16_t* lds, int w,
                                           int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rbase = w * 32 + j * 8;              // wave-uniform
    const int r = rbase + (lane >> 3);             // tile row of this lane
    const int chunk = (lane & 7) ^ (r & 7);        // source k-chunk that lands in LDS slot (lane & 7)
    int gr = r0 + r;
    gr = gr < rmax ? gr : rmax;                    // clamp: rows past the edge re-read the last valid row
    const bf16_t* src = P + (size_t)gr * ld + k0 + chunk * 8;
    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(lds + rbase * BK), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 frag(const bf16_t* lds, int r, int chunk) {
  union { uint4 u; bf16x8 v; } f;
  f.u = *(const uint4*)(lds + r * BK + ((chunk ^ (r & 7)) << 3));
  return f.v;
}

__global__ __launch_bounds__(256) void gemm_nt_fast_kernel(GemmArgsF g) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 2 * TILE_ELEMS];  // [stage][A|B] = 64 KiB
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 1, wn = w & 1;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
  // XCD-aware remap (blocks are dealt round-robin to the 8 XCDs): give each XCD a contiguous run of tile ids
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  stage_tile(g.A, g.lda, m0, g.M - 1, 0, lds, w, lane);
  stage_tile(g.B, g.ldb, n0, g.N - 1, 0, lds + TILE_ELEMS, w, lane);
  __syncthreads();  // (compiler drains vmcnt before the barrier while LDS-DMA is in flight)
  for (int kt = 0; kt < nk; ++kt) {
    const bf16_t* As = lds + (kt & 1) * 2 * TILE_ELEMS;
    const bf16_t* Bs = As + TILE_ELEMS;
    if (kt + 1 < nk) {
      bf16_t* An = lds + ((kt + 1) & 1) * 2 * TILE_ELEMS;
      stage_tile(g.A, g.lda, m0, g.M - 1, (kt + 1) * BK, An, w, lane);
      stage_tile(g.B, g.ldb, n0, g.N - 1, (kt + 1) * BK, An + TILE_ELEMS, w, lane);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bfr[4];
      const int chunk = 4 * s + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag(As, wm * 64 + i * 16 + (lane & 15), chunk);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = frag(Bs, wn * 64 + j * 16 + (lane & 15), chunk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        epi_store_f(g, m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r, n0 + wn * 64 + j * 16 + (lane & 15), acc[i][j][r]);
}

// ---------------------------------------------------------------- fp32/bf16 [R][C] -> bf16 [C][Rpad] (zero padded)
// 64x64 tiles through LDS; reads coalesced along C, writes coalesced along R.
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(int R, int C, const T* __restrict__ x, int ldx, bf16_t* __restrict__ y,
                                                             int ldy, int Rpad) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      if constexpr (sizeof(T) == 2) v = bf2f(x[(size_t)r * ldx + c]); else v = x[(size_t)r * ldx + c];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) y[(size_t)c * ldy + r] = f2bf(tile[tx][i]);
  }
}

}  // namespace

bool hamt_gemm_fast_eligible(const hamt_gemm_desc* d, const void* A, const void* B) {
  return d->prec == HAMT_PREC_BF16 && d->dtype_a == HAMT_BF16 && d->dtype_b == HAMT_BF16 && !d->a_kmajor && !d->b_kmajor &&
         d->K >= 64 && d->K % 64 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 &&
         d->M >= 1 && d->N >= 1;
}

void hamt_gemm_fast_launch(const hamt_gemm_desc* d, const void* A, const void* B, void* C, const float* bias, void* aux,
                           hipStream_t s) {
  GemmArgsF g{d->M, d->N, d->K, d->lda, d->ldb, d->ldc, d->ldaux, d->dtype_c, d->dtype_aux, d->epilogue, d->alpha,
              (const bf16_t*)A, (const bf16_t*)B, C, bias, aux};
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  hipLaunchKernelGGL(gemm_nt_fast_kernel, dim3(tiles), dim3(256), 0, s, g);
}

extern "C" int hamt_cast_transpose(int R, int C, const void* x, int ldx, int dtype_x, void* y, int ldy, int Rpad, void* stream) {
  HAMT_CHECK_ARG(x && y && R >= 0 && C >= 0 && Rpad >= R && ldy >= Rpad, "hamt_cast_transpose: bad argument");
  if (C == 0 || Rpad == 0) return HAMT_OK;
  dim3 grid((C + 63) / 64, (Rpad + 63) / 64);
  if (dtype_x == HAMT_BF16) hipLaunchKernelGGL((cast_transpose_kernel<bf16_t>), grid, dim3(256), 0, as_stream(stream), R, C, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, Rpad);
  else hipLaunchKernelGGL((cast_transpose_kernel<float>), grid, dim3(256), 0, as_stream(stream), R, C, (const float*)x, ldx, (bf16_t*)y, ldy, Rpad);
  HAMT_CHECK_LAUNCH("hamt_cast_transpose");
  return HAMT_OK;
}
