// GEMM kernels for gfx950 (MI355X).
//
//   C[M,N] = epilogue(alpha * A[M,K] * B[K,N])
//
// Two arithmetic paths behind one entry point (hamt_gemm, include/hamt.h):
//   HAMT_PREC_BF16: v_mfma_f32_16x16x32_bf16, operands rounded to bf16 while they are staged
//                   global -> VGPR -> LDS (fp32 or bf16 sources), fp32 accumulate, fp32 epilogue.
//   HAMT_PREC_F32 : v_mfma_f32_16x16x4_f32 -- exact fp32 (an fmaf chain per output), used for the
//                   <=1e-3 parity mode and for gradient checks.
// Both handle the three operand layouts of forward / dgrad / wgrad (a_kmajor, b_kmajor) and ragged
// M/N/K by zero-filling the staging loads.  K-strided ("k-major") operands keep their global
// layout [k][r] in LDS and are turned into MFMA fragments with ds_read_b64_tr_b16.
//
// Replaces every nn.Linear forward/backward on the reference path (vilmodel.py:97-99, 140, 169,
// 182, 263, 284, 323-325, 497-498, 549-558; pretrain_cmt.py:16-68).
#include "common.h"

void hamt_reduce_partials(int R, int N, const float* ws, float* out, int accumulate, hipStream_t s);

namespace {

struct GemmArgs {
  int M, N, K, lda, ldb;
  int ka_lim, kb_lim;   // valid reduction extent of A / B (<= K); beyond it the staging loads return zeros
  int ldc, ldaux;
  int dtype_c, dtype_aux, epi;
  float alpha;
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  void* aux;
};

// ---------------------------------------------------------------- shared epilogue (one element)
__device__ __forceinline__ void epi_store(const GemmArgs& g, int row, int col, float acc) {
  if (row >= g.M || col >= g.N) return;
  float v = acc * g.alpha;
  if (g.epi & HAMT_EPI_BIAS) v += g.bias[col];
  size_t ia = (size_t)row * g.ldaux + col;
  if (g.epi & HAMT_EPI_SAVE_PRE) {
    if (g.dtype_aux == HAMT_BF16) ((bf16_t*)g.aux)[ia] = f2bf(v); else ((float*)g.aux)[ia] = v;
  }
  if (g.epi & HAMT_EPI_GELU) v = gelu_erf(v);
  if (g.epi & HAMT_EPI_GELU_GRAD) {
    float gg, dg;
    gelu_and_grad(v, gg, dg);
    v = gg;
    if (g.dtype_aux == HAMT_U8G) { const float d4[4] = {dg, 0.f, 0.f, 0.f}; ((uint8_t*)g.aux)[ia] = (uint8_t)(g8_pack4(d4) & 0xffu); }
    else if (g.dtype_aux == HAMT_BF16) ((bf16_t*)g.aux)[ia] = f2bf(dg); else ((float*)g.aux)[ia] = dg;
  }
  if (g.epi & HAMT_EPI_MUL_AUX)
    v *= g.dtype_aux == HAMT_U8G ? __builtin_fmaf((float)((const uint8_t*)g.aux)[ia], 0.005f, -0.13f)
                                 : ((g.dtype_aux == HAMT_BF16) ? bf2f(((const bf16_t*)g.aux)[ia]) : ((const float*)g.aux)[ia]);
  if (g.epi & HAMT_EPI_RELU) v = fmaxf(v, 0.0f);
  if (g.epi & (HAMT_EPI_MUL_DGELU | HAMT_EPI_MUL_DRELU)) {
    float h = (g.dtype_aux == HAMT_BF16) ? bf2f(((const bf16_t*)g.aux)[ia]) : ((const float*)g.aux)[ia];
    v *= (g.epi & HAMT_EPI_MUL_DGELU) ? dgelu_erf(h) : (h > 0.0f ? 1.0f : 0.0f);
  }
  if (g.epi & HAMT_EPI_ADD_AUX) v += (g.dtype_aux == HAMT_BF16) ? bf2f(((const bf16_t*)g.aux)[ia]) : ((const float*)g.aux)[ia];
  size_t ic = (size_t)row * g.ldc + col;
  if (g.dtype_c == HAMT_BF16) {
    bf16_t* c = (bf16_t*)g.C;
    if (g.epi & HAMT_EPI_ACCUM) v += bf2f(c[ic]);
    c[ic] = f2bf(v);
  } else if (g.dtype_c == HAMT_F16) {
    bf16_t* c = (bf16_t*)g.C;
    if (g.epi & HAMT_EPI_ACCUM) v += h2f(c[ic]);
    c[ic] = f2h(v);
  } else {
    float* c = (float*)g.C;
    if (g.epi & HAMT_EPI_ACCUM) v += c[ic];
    c[ic] = v;
  }
}

// =================================================================================================
// fp32 exact path: 64x64x16 tile, 4 waves (2x2), each wave 2x2 fragments of v_mfma_f32_16x16x4_f32.
// LDS tiles are kept k-major ([k][r], r contiguous, row stride R+16 floats => the two 16-lane halves
// of a 32-lane ds_read_b32 group hit disjoint banks).
// =================================================================================================
// F_BK = 32 since round 6 (two 16-row stage pieces per thread and operand): a tile's life is its chain of k-stages -- global load one
// stage ahead, LDS write, barrier, 0.8 us each -- and the callers of this path are few-tile problems (the fp32 prediction heads: 96
// tiles at the rollout's 456 x 768 x 768, 40 us for 48 stages of 16)
constexpr int F_BM = 64, F_BN = 64, F_BK = 32, F_LD = 64 + 16, F_NP = F_BK / 16;

template <bool KMAJOR>
__device__ __forceinline__ void f32_stage_load(const float* __restrict__ P, int ld, int r0, int rlim, int k0,
                                               int klim, int t, float (&v)[4]) {
  // KMAJOR == false: global [r][k] (k contiguous): thread -> r = t%64, k = (t/64)*4 .. +3
  // KMAJOR == true : global [k][r] (r contiguous): thread -> k = t/16, r = (t%16)*4 .. +3
  if (!KMAJOR) {
    int r = r0 + (t & 63), k = k0 + (t >> 6) * 4;
    if (r < rlim && k + 4 <= klim) {
      float4 x = *(const float4*)(P + (size_t)r * ld + k);
      v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (r < rlim && k + j < klim) ? P[(size_t)r * ld + k + j] : 0.0f;
    }
  } else {
    int k = k0 + (t >> 4), r = r0 + (t & 15) * 4;
    if (k < klim && r + 4 <= rlim) {
      float4 x = *(const float4*)(P + (size_t)k * ld + r);
      v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (k < klim && r + j < rlim) ? P[(size_t)k * ld + r + j] : 0.0f;
    }
  }
}
template <bool KMAJOR>
__device__ __forceinline__ void f32_stage_write(float* lds, int t, const float (&v)[4]) {
  if (!KMAJOR) {
    int r = t & 63, k = (t >> 6) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[(k + j) * F_LD + r] = v[j];
  } else {
    int k = t >> 4, r = (t & 15) * 4;
    *(float4*)(lds + k * F_LD + r) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][F_BK * F_LD];  // [buf][A|B]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int tiles_n = (g.N + F_BN - 1) / F_BN;
  const int m0 = (blockIdx.x / tiles_n) * F_BM, n0 = (blockIdx.x % tiles_n) * F_BN;
  const float* A = (const float*)g.A;
  const float* B = (const float*)g.B;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float ra[F_NP][4], rb[F_NP][4];        // a stage = F_NP pieces of 16 k-rows (the 256 threads cover 64 x 16 per piece)
  const int nk = (g.K + F_BK - 1) / F_BK;
#pragma unroll
  for (int h = 0; h < F_NP; ++h) {
    f32_stage_load<A_KM>(A, g.lda, m0, g.M, 16 * h, g.ka_lim, t, ra[h]);
    f32_stage_load<B_KM>(B, g.ldb, n0, g.N, 16 * h, g.kb_lim, t, rb[h]);
  }
#pragma unroll
  for (int h = 0; h < F_NP; ++h) {
    f32_stage_write<A_KM>(lds[0][0] + 16 * h * F_LD, t, ra[h]);
    f32_stage_write<B_KM>(lds[0][1] + 16 * h * F_LD, t, rb[h]);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int h = 0; h < F_NP; ++h) {
        f32_stage_load<A_KM>(A, g.lda, m0, g.M, (kt + 1) * F_BK + 16 * h, g.ka_lim, t, ra[h]);
        f32_stage_load<B_KM>(B, g.ldb, n0, g.N, (kt + 1) * F_BK + 16 * h, g.kb_lim, t, rb[h]);
      }
    }
    const float* As = lds[cur][0];
    const float* Bs = lds[cur][1];
#pragma unroll
    for (int s = 0; s < F_BK / 4; ++s) {
      const int kk = s * 4 + (lane >> 4);
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk * F_LD + wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk * F_LD + wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
#pragma unroll
      for (int h = 0; h < F_NP; ++h) {
        f32_stage_write<A_KM>(lds[cur ^ 1][0] + 16 * h * F_LD, t, ra[h]);
        f32_stage_write<B_KM>(lds[cur ^ 1][1] + 16 * h * F_LD, t, rb[h]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        epi_store(g, m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, n0 + wn * 32 + j * 16 + (lane & 15), acc[i][j][r]);
}

// =================================================================================================
// bf16 MFMA path: BMxBNx32 tile (128x128 or 64x64), 4 waves (2x2), v_mfma_f32_16x16x32_bf16.
//   k-contiguous operand  -> LDS [R][32+8] bf16, fragments by ds_read_b128 (8 consecutive k per lane)
//   k-strided operand     -> LDS [32][R+8] bf16 (global layout kept), fragments by 2x
//                            ds_read_b64_tr_b16 (hardware 4x16 transpose: lane (l&15) gets column
//                            (l&15), 4 consecutive k per read)
// Staging is global -> VGPR (issued before the MFMAs of the current tile) -> LDS (written after them),
// two LDS buffers, one barrier per k-step.
// =================================================================================================
constexpr int H_BK = 32, H_PAD = 8;

template <typename T> struct Ld8;  // load 8 consecutive elements as 8 bf16 packed in a uint4
template <> struct Ld8<float> {
  static __device__ __forceinline__ uint4 vec(const float* p) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    return make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
  }
  static __device__ __forceinline__ bf16_t one(const float* p) { return f2bf(*p); }
};
template <> struct Ld8<bf16_t> {
  static __device__ __forceinline__ uint4 vec(const bf16_t* p) { return *(const uint4*)p; }
  static __device__ __forceinline__ bf16_t one(const bf16_t* p) { return *p; }
};

// one 8-element chunk of a tile.  row-type: chunk c -> (r = c/4, k = (c%4)*8); col-type: (k = c/(R/8), r = (c%(R/8))*8)
template <typename T, bool KMAJOR, int R>
__device__ __forceinline__ uint4 h_stage_load(const T* __restrict__ P, int ld, int r0, int rlim, int k0, int klim, int c) {
  int r, k;
  const T* p;
  bool full;
  if (!KMAJOR) {
    r = r0 + (c >> 2); k = k0 + (c & 3) * 8;
    p = P + (size_t)r * ld + k;
    full = (r < rlim) && (k + 8 <= klim);
  } else {
    k = k0 + c / (R / 8); r = r0 + (c % (R / 8)) * 8;
    p = P + (size_t)k * ld + r;
    full = (k < klim) && (r + 8 <= rlim);
  }
  if (full) return Ld8<T>::vec(p);
  unsigned short e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bool ok = KMAJOR ? (k < klim && r + j < rlim) : (r < rlim && k + j < klim);
    e[j] = ok ? Ld8<T>::one(p + j) : (unsigned short)0;
  }
  return make_uint4(e[0] | (uint32_t)e[1] << 16, e[2] | (uint32_t)e[3] << 16, e[4] | (uint32_t)e[5] << 16,
                    e[6] | (uint32_t)e[7] << 16);
}
template <bool KMAJOR, int R>
__device__ __forceinline__ void h_stage_write(bf16_t* lds, int c, uint4 v) {
  if (!KMAJOR) *(uint4*)(lds + (c >> 2) * (H_BK + H_PAD) + (c & 3) * 8) = v;
  else *(uint4*)(lds + (c / (R / 8)) * (R + H_PAD) + (c % (R / 8)) * 8) = v;
}
// fragment for the 16 rows (or columns) starting at r16 of the tile
template <bool KMAJOR, int R, bool USE_TR>
__device__ __forceinline__ bf16x8 h_frag(const bf16_t* lds, int r16, int lane) {
  union { uint4 u; bf16x8 v; s16x4 h[2]; unsigned short e[8]; } f;
  if (!KMAJOR) {
    f.u = *(const uint4*)(lds + (r16 + (lane & 15)) * (H_BK + H_PAD) + (lane >> 4) * 8);
  } else if (USE_TR) {
    const bf16_t* p = lds + ((lane >> 4) * 8 + ((lane & 15) >> 2)) * (R + H_PAD) + r16 + (lane & 3) * 4;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * (R + H_PAD)));
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.e[j] = lds[((lane >> 4) * 8 + j) * (R + H_PAD) + r16 + (lane & 15)];
  }
  return f.v;
}

template <int BM, int BN, bool A_KM, bool B_KM, typename TA, typename TB, bool USE_TR>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  constexpr int A_SZ = A_KM ? H_BK * (BM + H_PAD) : BM * (H_BK + H_PAD);
  constexpr int B_SZ = B_KM ? H_BK * (BN + H_PAD) : BN * (H_BK + H_PAD);
  constexpr int CA = BM * H_BK / 8 / 256, CB = BN * H_BK / 8 / 256;  // chunks per thread
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (A_SZ + B_SZ)];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const TA* A = (const TA*)g.A;
  const TB* B = (const TB*)g.B;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[CA], rb[CB];
  const int nk = (g.K + H_BK - 1) / H_BK;
#pragma unroll
  for (int c = 0; c < CA; ++c) ra[c] = h_stage_load<TA, A_KM, BM>(A, g.lda, m0, g.M, 0, g.ka_lim, t + c * 256);
#pragma unroll
  for (int c = 0; c < CB; ++c) rb[c] = h_stage_load<TB, B_KM, BN>(B, g.ldb, n0, g.N, 0, g.kb_lim, t + c * 256);
#pragma unroll
  for (int c = 0; c < CA; ++c) h_stage_write<A_KM, BM>(lds, t + c * 256, ra[c]);
#pragma unroll
  for (int c = 0; c < CB; ++c) h_stage_write<B_KM, BN>(lds + A_SZ, t + c * 256, rb[c]);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bf16_t* As = lds + (kt & 1) * (A_SZ + B_SZ);
    const bf16_t* Bs = As + A_SZ;
    if (kt + 1 < nk) {
#pragma unroll
      for (int c = 0; c < CA; ++c) ra[c] = h_stage_load<TA, A_KM, BM>(A, g.lda, m0, g.M, (kt + 1) * H_BK, g.ka_lim, t + c * 256);
#pragma unroll
      for (int c = 0; c < CB; ++c) rb[c] = h_stage_load<TB, B_KM, BN>(B, g.ldb, n0, g.N, (kt + 1) * H_BK, g.kb_lim, t + c * 256);
    }
    bf16x8 af[FM], bf[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = h_frag<A_KM, BM, USE_TR>(As, wm * WM + i * 16, lane);
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[j] = h_frag<B_KM, BN, USE_TR>(Bs, wn * WN + j * 16, lane);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) {
      bf16_t* An = lds + ((kt + 1) & 1) * (A_SZ + B_SZ);
#pragma unroll
      for (int c = 0; c < CA; ++c) h_stage_write<A_KM, BM>(An, t + c * 256, ra[c]);
#pragma unroll
      for (int c = 0; c < CB; ++c) h_stage_write<B_KM, BN>(An + A_SZ, t + c * 256, rb[c]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        epi_store(g, m0 + wm * WM + i * 16 + (lane >> 4) * 4 + r, n0 + wn * WN + j * 16 + (lane & 15), acc[i][j][r]);
}

template <int BM, int BN, bool A_KM, bool B_KM, typename TA, typename TB>
void launch_bf16(const GemmArgs& g, bool use_tr, hipStream_t s) {
  int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  if (use_tr) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, A_KM, B_KM, TA, TB, true>), dim3(tiles), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, A_KM, B_KM, TA, TB, false>), dim3(tiles), dim3(256), 0, s, g);
}
template <bool A_KM, bool B_KM, typename TA, typename TB>
void pick_tile_bf16(const GemmArgs& g, bool use_tr, int force_tile, hipStream_t s) {
  long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
  bool big = force_tile ? (force_tile == 128) : (t128 >= 192);
  if (big) launch_bf16<128, 128, A_KM, B_KM, TA, TB>(g, use_tr, s);
  else launch_bf16<64, 64, A_KM, B_KM, TA, TB>(g, use_tr, s);
}
template <bool A_KM, bool B_KM>
void pick_types_bf16(const GemmArgs& g, int da, int db, bool use_tr, int force_tile, hipStream_t s) {
  if (da == HAMT_F32 && db == HAMT_F32) pick_tile_bf16<A_KM, B_KM, float, float>(g, use_tr, force_tile, s);
  else if (da == HAMT_F32) pick_tile_bf16<A_KM, B_KM, float, bf16_t>(g, use_tr, force_tile, s);
  else if (db == HAMT_F32) pick_tile_bf16<A_KM, B_KM, bf16_t, float>(g, use_tr, force_tile, s);
  else pick_tile_bf16<A_KM, B_KM, bf16_t, bf16_t>(g, use_tr, force_tile, s);
}

// ---------------------------------------------------------------- column sums (bias gradients)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(int M, int N, const T* __restrict__ x, int ldx,
                                                             float* __restrict__ ws, int rows_per_chunk) {
  // block = 64 columns x 4 row-phases; grid = (ceil(N/64), chunks)
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + ph; r < r1; r += 4) {
      if constexpr (sizeof(T) == 2) s += bf2f(x[(size_t)r * ldx + col]); else s += x[(size_t)r * ldx + col];
    }
  __shared__ float red[4][64];
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && col < N) ws[(size_t)blockIdx.y * N + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void reduce_partials_kernel(int R, int N, const float* __restrict__ ws, float* __restrict__ out,
                                                              int accumulate) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s = 0.f;
  if (n < N)
    for (int r = ph; r < R; r += 4) s += ws[(size_t)r * N + n];
  __shared__ float red[4][64];
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && n < N) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    out[n] = accumulate ? out[n] + t : t;
  }
}

// the same sum (slice order 0, 1, 2, ...) with 16-byte accesses and up to four slices in flight per thread
__global__ __launch_bounds__(256) void reduce_partials4_kernel(int R, size_t n4, size_t N, const float* __restrict__ ws, float* __restrict__ out,
                                                               int accumulate) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r0 = 0; r0 < R; r0 += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = r0 + u < R ? *(const float4*)(ws + (size_t)(r0 + u) * N + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    float4* o = (float4*)(out + i * 4);
    if (accumulate) { const float4 p = *o; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
    *o = s;
  }
}

// dW[n][k] = sum_m dy[m][n] * x[m][k] for tiny K (the 4-wide angle features): K weighted column sums of dy, exact fp32.
template <int KMAX>
__global__ __launch_bounds__(256) void smallk_wgrad_partial_kernel(int M, int N, int K, const float* __restrict__ dy, int lddy,
                                                                   const float* __restrict__ x, int ldx, float* __restrict__ ws,
                                                                   int rows_per_chunk) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
  if (col < N)
    for (int r = r0 + ph; r < r1; r += 4) {
      const float g = dy[(size_t)r * lddy + col];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) if (k < K) acc[k] += g * x[(size_t)r * ldx + k];
    }
  __shared__ float red[4][64][KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) red[ph][threadIdx.x & 63][k] = acc[k];
  __syncthreads();
  if (ph == 0 && col < N)
    for (int k = 0; k < K; ++k)
      ws[((size_t)blockIdx.y * N + col) * K + k] = (red[0][threadIdx.x][k] + red[1][threadIdx.x][k]) + (red[2][threadIdx.x][k] + red[3][threadIdx.x][k]);
}

}  // namespace

extern "C" int hamt_smallk_wgrad(int M, int N, int K, const float* dy, int lddy, const float* x, int ldx, float* dW, int accumulate,
                                 float* ws, void* stream) {
  HAMT_CHECK_ARG(dy && x && dW && ws && K >= 1 && K <= 8 && N >= 1 && M >= 0, "hamt_smallk_wgrad: bad argument (K <= 8)");
  hipStream_t s = as_stream(stream);
  int chunks = M >= 64 * 64 ? 64 : (M + 63) / 64;
  if (chunks < 1) chunks = 1;
  int rpc = (M + chunks - 1) / chunks;
  if (rpc < 1) rpc = 1;
  hipLaunchKernelGGL((smallk_wgrad_partial_kernel<8>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, M, N, K, dy, lddy, x, ldx, ws, rpc);
  hamt_reduce_partials(chunks, N * K, ws, dW, accumulate, s);
  HAMT_CHECK_LAUNCH("hamt_smallk_wgrad");
  return HAMT_OK;
}

bool hamt_gemm_fast_eligible(const hamt_gemm_desc* d, const void* A, const void* B);
int hamt_gemm_fast_ksplit(const hamt_gemm_desc* d, size_t ws_bytes);
void hamt_gemm_fast_launch(const hamt_gemm_desc* d, const void* A, const void* B, void* C, const float* bias, void* aux, float* ws,
                           size_t ws_bytes, hipStream_t s);

void hamt_reduce_partials(int R, int N, const float* ws, float* out, int accumulate, hipStream_t s) {
  if (N % 4 == 0 && (((uintptr_t)ws | (uintptr_t)out) % 16) == 0) {
    const size_t n4 = (size_t)N / 4;
    const size_t b = (n4 + 255) / 256;
    hipLaunchKernelGGL(reduce_partials4_kernel, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, s, R, n4, (size_t)N, ws, out, accumulate);
    return;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((N + 63) / 64), dim3(256), 0, s, R, N, ws, out, accumulate);
}

extern "C" int hamt_gemm_ksplit(const hamt_gemm_desc* d) {
  if (!d || d->prec != HAMT_PREC_BF16 || d->dtype_a != HAMT_BF16 || d->dtype_b != HAMT_BF16 || (d->a_kmajor && !d->b_kmajor) ||
      d->K < 64 || d->K % 64)
    return 1;
  return hamt_gemm_fast_ksplit(d, (size_t)-1);
}

extern "C" int hamt_gemm(const hamt_gemm_desc* d, const void* A, const void* B, void* C, const float* bias, void* aux,
                         void* stream) {
  return hamt_gemm_ws(d, A, B, C, bias, aux, nullptr, 0, stream);
}

extern "C" int hamt_gemm_ws(const hamt_gemm_desc* d, const void* A, const void* B, void* C, const float* bias, void* aux,
                            void* ws, size_t ws_bytes, void* stream) {
  HAMT_CHECK_ARG(d && A && B && C, "hamt_gemm: null pointer");
  HAMT_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0, "hamt_gemm: negative size");
  if (d->M == 0 || d->N == 0) return HAMT_OK;
  const int sa = d->dtype_a == HAMT_BF16 ? 2 : 4, sb = d->dtype_b == HAMT_BF16 ? 2 : 4;
  HAMT_CHECK_ARG((d->lda * sa) % 16 == 0 && (d->ldb * sb) % 16 == 0, "hamt_gemm: lda/ldb rows must be 16-byte aligned (lda=%d ldb=%d)", d->lda, d->ldb);
  HAMT_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "hamt_gemm: A/B must be 16-byte aligned");
  HAMT_CHECK_ARG(d->dtype_c == HAMT_F32 || d->dtype_c == HAMT_BF16 || d->dtype_c == HAMT_F16, "hamt_gemm: dtype_c = %d (C is fp32, bf16 or IEEE half)", d->dtype_c);
  HAMT_CHECK_ARG(d->dtype_c != HAMT_F16 || (d->epilogue & ~(HAMT_EPI_BIAS | HAMT_EPI_ACCUM)) == 0, "hamt_gemm: an IEEE-half C (HAMT_F16) takes the plain / bias / accumulate epilogues only");
  HAMT_CHECK_ARG((d->dtype_a == HAMT_F32 || d->dtype_a == HAMT_BF16) && (d->dtype_b == HAMT_F32 || d->dtype_b == HAMT_BF16), "hamt_gemm: operands are fp32 or bf16");
  HAMT_CHECK_ARG(!(d->epilogue & HAMT_EPI_BIAS) || bias, "hamt_gemm: EPI_BIAS without bias");
  HAMT_CHECK_ARG(!(d->epilogue & (HAMT_EPI_SAVE_PRE | HAMT_EPI_MUL_DGELU | HAMT_EPI_MUL_DRELU | HAMT_EPI_GELU_GRAD | HAMT_EPI_MUL_AUX | HAMT_EPI_ADD_AUX)) || aux, "hamt_gemm: epilogue needs aux");
  if (aux && d->dtype_aux == HAMT_U8G)
    HAMT_CHECK_ARG((d->epilogue & (HAMT_EPI_GELU_GRAD | HAMT_EPI_MUL_AUX)) && !(d->epilogue & (HAMT_EPI_SAVE_PRE | HAMT_EPI_MUL_DGELU | HAMT_EPI_MUL_DRELU | HAMT_EPI_ADD_AUX | HAMT_EPI_DROPOUT)),
                   "hamt_gemm: a HAMT_U8G aux is the gelu' image of HAMT_EPI_GELU_GRAD (without dropout) / HAMT_EPI_MUL_AUX only");
  if (d->epilogue & HAMT_EPI_DROPOUT) {
    HAMT_CHECK_ARG(d->rng && d->p_drop >= 0.0f && d->p_drop < 1.0f, "hamt_gemm: EPI_DROPOUT needs rng and 0 <= p_drop < 1");
    HAMT_CHECK_ARG(getenv("HAMT_NO_FAST") == nullptr && hamt_gemm_fast_eligible(d, A, B), "hamt_gemm: EPI_DROPOUT is implemented by the bf16 MFMA path only (bf16 operands, K %% 64 == 0)");
  }
  const int ka = (d->ka_rows > 0 && d->ka_rows < d->K) ? d->ka_rows : d->K, kb = (d->kb_rows > 0 && d->kb_rows < d->K) ? d->kb_rows : d->K;
  GemmArgs g{d->M, d->N, d->K, d->lda, d->ldb, ka, kb, d->ldc, d->ldaux, d->dtype_c, d->dtype_aux, d->epilogue, d->alpha, A, B, C, bias, aux};
  hipStream_t s = as_stream(stream);
  static const bool no_fast = getenv("HAMT_NO_FAST") != nullptr;
  if (!no_fast && hamt_gemm_fast_eligible(d, A, B)) {  // bf16 x bf16, K-contiguous operands, K % 64 == 0
    hamt_gemm_fast_launch(d, A, B, C, bias, aux, (float*)ws, ws_bytes, s);
    HAMT_CHECK_LAUNCH("hamt_gemm(fast)");
    return HAMT_OK;
  }
  hamt_set_last_kernel(d->prec == HAMT_PREC_F32 ? "gemm_f32_kernel<%s, %s>" : "gemm_bf16_kernel<%s, %s, ...>", d->a_kmajor ? "true" : "false", d->b_kmajor ? "true" : "false");
  if (d->prec == HAMT_PREC_F32) {
    HAMT_CHECK_ARG(d->dtype_a == HAMT_F32 && d->dtype_b == HAMT_F32, "hamt_gemm: PREC_F32 needs fp32 operands");
    int tiles = ((d->M + F_BM - 1) / F_BM) * ((d->N + F_BN - 1) / F_BN);
    if (!d->a_kmajor && !d->b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), dim3(tiles), dim3(256), 0, s, g);
    else if (!d->a_kmajor && d->b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), dim3(tiles), dim3(256), 0, s, g);
    else if (d->a_kmajor && !d->b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), dim3(tiles), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), dim3(tiles), dim3(256), 0, s, g);
  } else {
    static const bool use_tr = getenv("HAMT_NO_TR") == nullptr;
    static const int force_tile = getenv("HAMT_GEMM_TILE") ? atoi(getenv("HAMT_GEMM_TILE")) : 0;
    if (!d->a_kmajor && !d->b_kmajor) pick_types_bf16<false, false>(g, d->dtype_a, d->dtype_b, use_tr, force_tile, s);
    else if (!d->a_kmajor && d->b_kmajor) pick_types_bf16<false, true>(g, d->dtype_a, d->dtype_b, use_tr, force_tile, s);
    else if (d->a_kmajor && d->b_kmajor) pick_types_bf16<true, true>(g, d->dtype_a, d->dtype_b, use_tr, force_tile, s);
    else { hamt_set_error("hamt_gemm: a_kmajor=1,b_kmajor=0 is not used by the path"); return HAMT_ERR_UNSUPPORTED; }
  }
  HAMT_CHECK_LAUNCH("hamt_gemm");
  return HAMT_OK;
}

extern "C" int hamt_colsum(int M, int N, const void* x, int ldx, int dtype_x, float* out, int accumulate, float* ws,
                           void* stream) {
  HAMT_CHECK_ARG(x && out && ws && M >= 0 && N > 0, "hamt_colsum: bad argument");
  hipStream_t s = as_stream(stream);
  int chunks = M >= 64 * 64 ? 64 : (M + 63) / 64;
  if (chunks < 1) chunks = 1;
  int rpc = (M + chunks - 1) / chunks;
  if (rpc < 1) rpc = 1;
  dim3 grid((N + 63) / 64, chunks);
  if (dtype_x == HAMT_BF16) hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), grid, dim3(256), 0, s, M, N, (const bf16_t*)x, ldx, ws, rpc);
  else hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, dim3(256), 0, s, M, N, (const float*)x, ldx, ws, rpc);
  hamt_reduce_partials(chunks, N, ws, out, accumulate, s);
  HAMT_CHECK_LAUNCH("hamt_colsum");
  return HAMT_OK;
}
