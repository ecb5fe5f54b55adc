// gemm_q4_kernel: the 128 x 128 output tile for the NARROW outputs of the HAMT step -- N = 768 (attention output projection, FFN-2, the
// query projection of the cross attention, and every dgrad into the residual stream: K = 768 .. 3072) at 5120 / 2752 rows, where a
// 256-square tile leaves three quarters of the chip idle (5120 x 768 = 60 tiles) and the output has 240 tiles of 128 x 128: ONE tile per CU.
//
// What the tile is bound by is operand DELIVERY, not MFMA issue: 32 KiB of operands per 2.1 MFLOP k-tile (a 256-square tile: 64 KiB per 8.4),
// and every layer's weights arrive cold from HBM (profiles/r05_q4_probe.txt, tools/cold_probe.py: the two-deep rings of gemm_kg_kernel<128, 2, 2> /
// gemm_fast_kernel<64> lose 20 - 35 % when W, or W and A, were not read a moment ago -- which is how the step runs them; the 256-square
// two-phase tile with its deeper queue loses nothing).  So this kernel is the two-phase tile's schedule (gemm_fast.hip: p8_tile; guide
// section 5: counted vmcnt, raw s_barrier, load / multiply segments, the two wave rows one barrier apart) scaled to 128 x 128 with the
// LDS spent on DEPTH:
//   * 8 waves = 2 (M) x 4 (N), a wave's output 64 x 32 = 4 x 2 fragments: 16 MFMAs and 12 fragment reads per k-tile;
//   * a k-tile's operands are one 32 KiB buffer [A 128 x 64 | B 128 x 64]; the ring holds FOUR (128 KiB): the load segment of k-tile t
//     issues the DMA of k-tile t + 3 into the buffer k-tile t - 1 was read from (free since the barrier in front of this segment), and
//     waits only for k-tile t + 1 (vmcnt(8): two k-tiles = 8 pieces per wave stay in flight across the barriers) -- an operand piece has
//     two to three whole k-tile periods to arrive;
//   * ONE phase per k-tile: load segment (fragments of k-tile t -> registers, DMA issue, counted wait) | barrier | 16 MFMAs | barrier;
//     wave row 1 runs one barrier behind row 0, so on every SIMD one wave multiplies while the other reads LDS and issues DMA;
//   * epilogue through a workgroup-shared fp32 image of the tile (64 KiB of the ring): 16 lanes per output row, 8 columns per lane,
//     interior tiles on epi_fast8.
// K-contiguous A; B K-contiguous (forward, W[N][K]) or K-strided (dgrad, W[K][N]).
// Measured and dropped (profiles/r05_q4_probe.txt): the DMA pieces spread over the multiply segment (compiler-placed or pinned: +-1 %), and the
// same schedule on v_mfma_f32_32x32x16_bf16 (8 MFMAs per k-tile instead of 16: +-2 %) -- with warm operands every form of the 128-square tile,
// hipBLASLt's included, sits at ~0.55 us per k-tile (~900 cycles at the ~1.65 GHz the chip holds under this load; 16 bare MFMAs issue in ~350).
#include "common.h"
#include "gemm_args.h"
#include <type_traits>

namespace {

#include "gemm_frag.h"
#include "gemm_epi.h"

constexpr int Q4_STAGE = 2 * 128 * BK;         // elements per k-tile buffer: A unit + B unit (32 KiB)
constexpr int Q4_DEPTH = 4;

template <int N> __device__ __forceinline__ void q4_wait() {      // the load segment's end: this wave's older DMA pieces and its fragment reads
  static_assert(N == 0 || N == 4 || N == 8, "q4_wait");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// second half of the epilogue: the tile sits in `ct` as [128][128] fp32 (float4 slot ^= row & 7); 16 lanes per row, 8 columns per lane
template <int EPI>
__device__ __forceinline__ void q4_store_tile(const GemmArgsF& g, const float* ct, int m0, int n0, int t) {
  const int c8 = t & 15, col = n0 + c8 * 8;
  auto piece = [&](int rl, float* v8) {
    const f32x4 lo = *(const f32x4*)(ct + rl * 128 + (((2 * c8) ^ (rl & 7)) << 2));
    const f32x4 hi = *(const f32x4*)(ct + rl * 128 + (((2 * c8 + 1) ^ (rl & 7)) << 2));
    v8[0] = lo[0]; v8[1] = lo[1]; v8[2] = lo[2]; v8[3] = lo[3]; v8[4] = hi[0]; v8[5] = hi[1]; v8[6] = hi[2]; v8[7] = hi[3];
  };
  if (epi_fast_ok(g, EPI, m0, n0, 128, 128)) {      // (workgroup uniform) every column exists, 16-byte aligned rows: see gemm_epi.h
    float b8[8];
    epi_fast_bias<EPI>(g, col, b8);
    auto run = [&](auto c16) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int rl = 32 * p + (t >> 4);
        float v8[8];
        piece(rl, v8);
        epi_fast8<EPI, decltype(c16)::value>(g, m0 + rl, col, v8, b8, nullptr);
      }
    };
    if (g.dtype_c != HAMT_F32) run(std::true_type{}); else run(std::false_type{});
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rl = 32 * p + (t >> 4);
      float v8[8];
      piece(rl, v8);
      epi_store<EPI, 8>(g, m0 + rl, col, v8);
    }
  }
}

template <int EPI, bool B_KM>
__global__ __launch_bounds__(512) void gemm_q4_kernel(GemmArgsF g) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[Q4_DEPTH * Q4_STAGE];          // 128 KiB
  const int tiles_n = (g.N + 127) >> 7;
  const int bid = xcd_remap(blockIdx.x, ((g.M + 127) >> 7) * tiles_n);
  const int m0 = (bid / tiles_n) * 128, n0 = (bid % tiles_n) * 128;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;
  const int nk = g.K / BK;                                                           // >= 3 (launcher)
  const unsigned lds0 = lds_base_of(lds);

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bf_[2][2];

  TileSrc<false, 128, 8> sa;
  TileSrc<B_KM, 128, 8> sb;
  sa.init(g.lda, m0, g.M - 1, w, lane);
  sb.init(g.ldb, n0, g.N - 1, w, lane);
  auto issue = [&](int kt) {
    const unsigned dst = lds0 + (unsigned)((kt & (Q4_DEPTH - 1)) * Q4_STAGE * 2);
    sa.issue(g.A, g.lda, kt * BK, g.ka_max, dst, w);
    sb.issue(g.B, g.ldb, kt * BK, g.kb_max, dst + (unsigned)(128 * BK * 2), w);
  };
  issue(0); issue(1); issue(2);
  q4_wait<8>();                                  // k-tile 0 has landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();     // the second wave row runs one barrier behind the first

  // MODE 0: steady state (issues k-tile kt + 3); 1: kt == nk - 3; 2: kt == nk - 2; 3: kt == nk - 1
  auto phase = [&](int kt, auto modec) {
    constexpr int MODE = decltype(modec)::value;
    const bf16_t* As = lds + (kt & (Q4_DEPTH - 1)) * Q4_STAGE;
    const bf16_t* Bs = As + 128 * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) bf_[j][s] = frag<B_KM, 128>(Bs, 32 * wc + 16 * j, s, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i][s] = frag<false, 128>(As, 64 * wr + 16 * i, s, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 0) { issue(kt + 3); q4_wait<8>(); }      // k-tile kt + 1 landed; kt + 2, kt + 3 in flight
    else if constexpr (MODE == 1) q4_wait<4>();                    // kt + 1 landed; kt + 2 in flight
    else q4_wait<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)     // swapped roles: D[n][m] => a lane owns C[m = lane & 15][n = 4 (lane >> 4) .. + 3]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[j][s], af[i][s], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using std::integral_constant;
  int kt = 0;
  for (; kt < nk - 3; ++kt) phase(kt, integral_constant<int, 0>{});
  phase(kt, integral_constant<int, 1>{}); ++kt;
  phase(kt, integral_constant<int, 2>{}); ++kt;
  phase(kt, integral_constant<int, 3>{});
  if (wr == 0) __builtin_amdgcn_s_barrier();    // balances the second wave row's extra barrier: every wave is behind its last fragment read

  // epilogue: the tile as [128][128] fp32 in the first 64 KiB of the ring, float4 slot ^= row & 7 (conflict free for the 16-row b128 writes
  // of the MFMA layout and for the row-contiguous reads), then 16 lanes per row, 8 columns per lane, 32 rows per pass
  float* ct = (float*)lds;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rl = 64 * wr + 16 * i + (lane & 15), c4 = 8 * wc + 4 * j + (lane >> 4);
      *(f32x4*)(ct + rl * 128 + ((c4 ^ (rl & 7)) << 2)) = acc[i][j];
    }
  __syncthreads();
  q4_store_tile<EPI>(g, ct, m0, n0, t);
}


template <bool B_KM>
bool launch_q4(const GemmArgsF& g, hipStream_t s) {
  const dim3 grid(((g.M + 127) / 128) * ((g.N + 127) / 128));
  const int e = g.epi;
#define HAMT_L(E) do { hipLaunchKernelGGL((gemm_q4_kernel<E, B_KM>), grid, dim3(512), 0, s, g); \
                    hamt_set_last_kernel("gemm_q4_kernel<%d, %s>", (int)(E), B_KM ? "true" : "false"); } while (0)
  if (e == 0) HAMT_L(0);
  else if (e == HAMT_EPI_BIAS) HAMT_L(HAMT_EPI_BIAS);
  else if (e == HAMT_EPI_ACCUM) HAMT_L(HAMT_EPI_ACCUM);
  else return false;
#undef HAMT_L
  return true;
}

}  // namespace

// true: launched.  The caller (hamt_gemm_fast_launch) has checked the fast path's operand contract (bf16, K % 64 == 0, alignment, 32-bit
// offsets) and decides WHEN this tile is the right one; here only what the kernel itself needs.
bool hamt_gemm_q4_launch(const GemmArgsF& g, bool b_kmajor, hipStream_t s) {
  if (g.K < 3 * BK || g.ksplit > 1) return false;
  if (b_kmajor && g.ldb < 128) return false;
  return b_kmajor ? launch_q4<true>(g, s) : launch_q4<false>(g, s);
}
