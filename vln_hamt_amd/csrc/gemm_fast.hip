// Fast path of hamt_gemm and the grouped weight-gradient launch: the bf16 MFMA GEMMs that carry the FLOPs of the HAMT step.
//   C[M,N] = epi(op(A) * op(B)), A and B bf16; each operand either K-contiguous ([rows][K], forward activations and
//   weights) or K-strided ([K][cols]: the weight in the dgrad "NN" form, both operands in the wgrad "TN" form), so the
//   row-major bf16 images the producers wrote serve every contraction -- there are no transposed copies.
//
// Structure (cdna_hip_programming.md section 5: glds staging, swizzled LDS, explicit waits):
//   * gemm_tile<BM, BN, WM x WN waves>: 64/128 x 128 tiles with 4 waves (2-3 workgroups per CU) for forward / dgrad, and a
//     256 x 256 tile with 8 waves (one workgroup per CU, half the DMA pieces and 3/4 of the LDS reads per MFMA) for long
//     reductions over large grids (weight gradients; big forward GEMMs); v_mfma_f32_16x16x32_bf16 with the operand roles
//     SWAPPED (mfma(B,A)) so that a lane owns 4 consecutive columns of one C row;
//   * operands go L2 -> LDS directly with global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPR round trip) into a
//     2-deep ring: tile kt+1 is issued while tile kt is multiplied, waits are explicit s_waitcnt vmcnt in front of a raw
//     s_barrier (the tile body is generic in the ring depth; 2 measures best: independent workgroups sharing a CU overlap
//     better than a deeper ring in one);
//   * LDS images are lane-linear per DMA instruction; K-contiguous tiles XOR the 16-byte k-chunk index with (row & 7) on the
//     SOURCE address and on the ds_read_b128 fragment read (conflict free); K-strided tiles are read with
//     ds_read_b64_tr_b16 (hardware transpose) under a pair-preserving XOR;
//   * ragged M/N: row indices are clamped for the loads (no OOB access), the epilogue masks the stores;
//   * epilogue: the fp32 accumulators are re-tiled through an XOR-swizzled LDS tile so that a lane stores 8 consecutive
//     columns (one 16-byte store per bf16 output: stores are issue-bound on this chip), compiled per flag set (template --
//     a dynamic one unrolled 64x overflowed the I-cache); the 256-square tile stores fp32 straight from the MFMA layout;
//   * small outputs with a long reduction that are NOT weight gradients use deterministic split-K (grid.y slices writing
//     fp32 partial tiles, summed by a second kernel in a fixed order); weight gradients go through
//     hamt_wgrad_grouped instead: all problems of a backward pass in one launch per tile class, each problem pinned to
//     one XCD so that its operand panels cross the fabric once, bias sums fused in.
#include "common.h"
#include "gemm_args.h"
#include <algorithm>
#include <type_traits>
#include <vector>

void hamt_reduce_partials(int R, int N, const float* ws, float* out, int accumulate, hipStream_t s);
bool hamt_gemm_q4_launch(const GemmArgsF& g, bool b_kmajor, hipStream_t s);      // gemm_q4.hip: the 128-square tile with a four-deep operand ring


#ifdef HAMT_PROF   // cycle accounting of the main loop (tools/gemm_prof.py builds a private copy with -DHAMT_PROF)
__device__ unsigned long long hamt_prof_rec[65536 * 10];   // per-wave records (no atomics inside the timed regions)
__device__ unsigned int hamt_prof_n;
extern "C" int hamt_prof_fetch(unsigned long long* out8, int reset) {
  hipDeviceSynchronize();
  static unsigned long long rec[65536 * 10];     // out8: 16 sums over the recorded waves
  unsigned int n = 0;
  hipMemcpyFromSymbol(&n, HIP_SYMBOL(hamt_prof_n), 4);
  if (n > 65536) n = 65536;
  hipMemcpyFromSymbol(rec, HIP_SYMBOL(hamt_prof_rec), (size_t)n * 80);
  for (int k = 0; k < 16; ++k) out8[k] = 0;
  for (unsigned int i = 0; i < n; ++i) for (int k = 0; k < 9; ++k) out8[k] += rec[(size_t)i * 10 + k];
  if (reset) { n = 0; hipMemcpyToSymbol(HIP_SYMBOL(hamt_prof_n), &n, 4); }
  return 0;
}
#endif

#ifdef HAMT_PROF
__device__ unsigned long long hamt_p8_prof[40];    // [group 0/1][phase 0..3][load, barrier 1, multiply, barrier 2] cycles, then [32..33] waves, [34..35] phases
extern "C" int hamt_p8_prof_fetch(unsigned long long* out40, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out40, HIP_SYMBOL(hamt_p8_prof), 40 * 8);
  if (reset) { unsigned long long z[40] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(hamt_p8_prof), z, 40 * 8); }
  return 0;
}
#endif

namespace {

#include "gemm_frag.h"
#include "gemm_epi.h"

// One float per output tile: the sum of squares of what the tile stored (a weight-gradient tile's share of the global gradient
// norm: the clip needs it before the update, and reading 0.6 GB of gradients back for it costs ~0.13 ms per step).  Slot = the
// tile's origin in units of 64 rows x 128 columns, so the same array serves every tile size; summed in thread / wave order:
// deterministic.  `red`: >= (threads / 64) floats of LDS nobody else is using.
__device__ __forceinline__ void tile_sumsq_store(const GemmArgsF& g, int m0, int n0, float ssq, float* red) {
  const int t = threadIdx.x, nw = blockDim.x >> 6;
  ssq = wave_sum(ssq);
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = ssq;
  __syncthreads();
  if (t == 0) {
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += red[i];
    g.ss[(size_t)(m0 >> 6) * (g.ss_ld ? g.ss_ld : ((g.N + 127) >> 7)) + (n0 >> 7)] = tot;
  }
}

// One BMx128 output tile over k-tiles [kt0, kt0 + nk).  COLSUM (weight-gradient form, A = dY stored [K][M]): the wave
// column wn == 0 of the tiles with n0 == 0 also reduces A over k with one extra MFMA per fragment against a ones
// operand (D'[n][m] = sum_k 1 * A[m][k]) -- the bias gradient, for free of any extra pass over dY.
template <int BM, int EPI, bool A_KM, bool B_KM, int NSTAGE, bool COLSUM, int BN = 128, int WM = 2, int WN = 2>
__device__ __forceinline__ void gemm_tile(const GemmArgsF& g, int m0, int n0, int kt0, int nk, int slice, float* db, int db_accum) {
  // WM x WN waves, each a (BM/WM) x (BN/WN) output as FM x FN fragments of 16x16.  Instances: 64/128 x 128 with 2x2 waves
  // (two or three workgroups per CU), and 256 x 256 with 2x4 waves (one 512-thread workgroup per CU; half the DMA
  // pieces and 3/4 of the LDS fragment reads per MFMA of the 128-square tile) for grids that are large enough.
  constexpr int NW = WM * WN, TM = BM / WM, TN = BN / WN;
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
  constexpr int FM = TM / 16, FN = TN / 16;      // 16x16 fragments per wave along M and N
  constexpr int NLD = (BM + BN) / (8 * NW);      // glds instructions per wave per stage
  __shared__ __attribute__((aligned(16))) bf16_t lds[NSTAGE * STAGE];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w / WN, wn = w % WN;
#ifdef HAMT_PROF
  const unsigned long long pf_entry = __builtin_readcyclecounter();
#endif
  const bool do_cs = COLSUM && db != nullptr && n0 == 0 && wn == 0;   // wave-uniform
  f32x4 cs[FM];
  if constexpr (COLSUM) {
#pragma unroll
    for (int i = 0; i < FM; ++i) cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = lds_base_of(lds);
  TileSrc<A_KM, BM, NW> src_a;
  TileSrc<B_KM, BN, NW> src_b;
  src_a.init(g.lda, m0, g.M - 1, w, lane);
  src_b.init(g.ldb, n0, g.N - 1, w, lane);
  auto stage = [&](int kt, int slot) {
    const unsigned dst = lds0 + (unsigned)(slot * STAGE * 2);
    src_a.issue(g.A, g.lda, (kt0 + kt) * BK, g.ka_max, dst, w);
    src_b.issue(g.B, g.ldb, (kt0 + kt) * BK, g.kb_max, dst + (unsigned)(A_ELEMS * 2), w);
  };
  if constexpr (NSTAGE > 1) {
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p)
      if (p < nk) stage(p, p);
  }
#ifdef HAMT_PROF
  unsigned long long pf_wait = 0, pf_bar = 0, pf_dma = 0, pf_mma = 0, pf_t0 = __builtin_readcyclecounter();
  const unsigned long long pf_start = pf_t0;
#define HAMT_PF(acc) { const unsigned long long t_ = __builtin_readcyclecounter(); acc += t_ - pf_t0; pf_t0 = t_; }
#else
#define HAMT_PF(acc)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (NSTAGE == 1) {
      if (kt) __builtin_amdgcn_s_barrier();     // everyone is done reading tile kt-1
      stage(kt, 0);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    } else {
      // tile kt has landed when at most the loads of the (up to NSTAGE-2) younger tiles are still outstanding
      const int younger = min(NSTAGE - 2, nk - 1 - kt);
      if (younger >= 2) wait_vmcnt<2 * NLD>(); else if (younger == 1) wait_vmcnt<NLD>(); else wait_vmcnt<0>();
      HAMT_PF(pf_wait)
      __builtin_amdgcn_s_barrier();             // everyone's share of tile kt is in LDS; everyone is done with tile kt-1
      HAMT_PF(pf_bar)
      if (kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);   // overwrites the slot of tile kt-1
      HAMT_PF(pf_dma)
    }
    const bf16_t* As = lds + (kt % NSTAGE) * STAGE;
    const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = frag<A_KM, BM>(As, wm * TM + i * 16, s, lane);
      if constexpr (A_KM) {
        const int kbase = (kt0 + kt) * BK + 32 * s;
        if (kbase + 32 > g.ka_max + 1) {       // wave-uniform: this k-step crosses the operand's last valid reduction row
#pragma unroll
          for (int i = 0; i < FM; ++i) af[i] = mask_k_tail(af[i], kbase, lane, g.ka_max + 1);
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = frag<B_KM, BN>(Bs, wn * TN + j * 16, s, lane);
      __builtin_amdgcn_s_setprio(1);             // the co-resident wave (other workgroup / other half) is in its load phase
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)   // swapped roles: D[n][m] => lane owns C[m = lane&15][n = 4*(lane>>4) .. +3]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if constexpr (COLSUM) {
        if (do_cs) {
          const s16x8 one8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
          union { s16x8 s; bf16x8 v; } ones; ones.s = one8;
#pragma unroll
          for (int i = 0; i < FM; ++i) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, af[i], cs[i], 0, 0, 0);
        }
      }
    }
    HAMT_PF(pf_mma)
  }
#ifdef HAMT_PROF
  const unsigned long long pf_loop_end = __builtin_readcyclecounter();
#endif
  if constexpr (COLSUM) {
    if (do_cs && lane < 16) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = m0 + wm * TM + i * 16 + lane;
        if (row < g.M) db[row] = db_accum ? db[row] + cs[i][0] : cs[i][0];
      }
    }
  }
  float tile_ssq = 0.f;
  // Epilogue through LDS.  A wave's store instructions are issue-bound (~70 cycles each whatever their width, see
  // MI355X_MICROARCH.md "store tail"), and the MFMA layout gives a lane only 4 consecutive columns (8 bytes of a bf16
  // row).  Re-tiling the fp32 accumulators through LDS gives every lane 8 consecutive columns -- one 16-byte store per
  // bf16 output, two per fp32 output, 4 rows x 256-512 contiguous bytes per wave instruction -- and makes the aux /
  // accumulate reads coalesced as well.  The C tile is [BM][128] fp32 with the float4 slot index XOR-ed with (row & 7):
  // conflict-free for the 8-row groups of the b128 writes and for the row-contiguous reads, and exactly the size of
  // the 2-deep operand ring for BM = 128 (64 KiB).
  if constexpr (BM * BN * 4 > NSTAGE * STAGE * 2) {   // C tile larger than the ring (256-square tile): store from the MFMA
#pragma unroll                                        // layout -- 16 bytes per lane for fp32 outputs, which is all it is used for
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = m0 + wm * TM + i * 16 + (lane & 15), col = n0 + wn * TN + j * 16 + (lane >> 4) * 4;
        const float a4[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (g.ksplit > 1) {
          float* P = g.part + (size_t)slice * g.M * g.N;
          if (row < g.M) for (int e = 0; e < 4; ++e) if (col + e < g.N) P[(size_t)row * g.N + col + e] = a4[e];
        } else epi_store<EPI, 4>(g, row, col, a4, (COLSUM && g.ss) ? &tile_ssq : nullptr);
      }
    if constexpr (COLSUM) { if (g.ss) tile_sumsq_store(g, m0, n0, tile_ssq, (float*)lds); }
  } else {
    static_assert(BN == 128, "the staged epilogue is written for 128-column tiles");
    constexpr int RPP = NW * 4;                    // tile rows per pass: 16 threads per row
    float* ct = (float*)lds;
    const int c8 = t & 15, col = n0 + c8 * 8;      // this thread's 8 columns; rows (t >> 4) + RPP p
    // mulaux reads the saved gelu': all of the thread's row pieces are requested here, ahead of the two barriers and the
    // transpose (inside epi_store each load waits behind the previous piece's store): 7-8 % on the 64-row FFN-2 dgrads.
    // (The same for the fp32 C of `acc` and for the bias, here and in the K-group kernel: no measurable difference, not kept.)
    constexpr bool PRE_AUX = EPI >= 0 && (EPI & HAMT_EPI_MUL_AUX) != 0 && BM / RPP <= 8;
    uint4 pa[PRE_AUX ? BM / RPP : 1];
    bool pre_ok = false;
    if constexpr (PRE_AUX) {
      pre_ok = g.ksplit <= 1 && col + 8 <= g.N && (g.dtype_aux == HAMT_BF16 || g.dtype_aux == HAMT_U8G) && (g.ldaux % 8) == 0 && ((uintptr_t)g.aux % 16) == 0;
#pragma unroll
      for (int p = 0; p < BM / RPP; ++p) {
        const int row = m0 + p * RPP + (t >> 4), rr = row < g.M ? row : g.M - 1;
        if (pre_ok) pa[p] = ld_aux8(g, rr, col);
      }
    }
    __syncthreads();                               // every wave is done reading the last operand tile
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int rl = wm * TM + i * 16 + (lane & 15), c4 = wn * (TN / 4) + j * 4 + (lane >> 4);
        *(f32x4*)(ct + rl * BN + ((c4 ^ (rl & 7)) << 2)) = acc[i][j];
      }
    __syncthreads();
    constexpr bool FASTK = EPI >= 0 && (EPI & ~EPI_FAST_MASK) == 0 && !COLSUM;
    bool fast = false;
    if constexpr (FASTK) fast = epi_fast_ok(g, EPI, m0, n0, BM, BN);      // interior tile (workgroup uniform): see epi_fast8
    if (fast) {
      if constexpr (FASTK) {
        float b8[8];
        epi_fast_bias<EPI>(g, col, b8);
        // (no lambda here: one that captures `pa` by reference puts the array into scratch memory -- tests/test_kernel_resources.py)
#define HAMT_FAST_PIECES(C16_)                                                                                   \
        _Pragma("unroll") for (int p = 0; p < BM / RPP; ++p) {                                                   \
          const int rl = p * RPP + (t >> 4);                                                                      \
          const f32x4 lo = *(const f32x4*)(ct + rl * BN + (((2 * c8) ^ (rl & 7)) << 2));                          \
          const f32x4 hi = *(const f32x4*)(ct + rl * BN + (((2 * c8 + 1) ^ (rl & 7)) << 2));                      \
          const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                           \
          if constexpr (PRE_AUX) epi_fast8<EPI, C16_>(g, m0 + rl, col, v8, b8, &pa[p]);                           \
          else epi_fast8<EPI, C16_>(g, m0 + rl, col, v8, b8, nullptr);                                            \
        }
        if (g.dtype_c != HAMT_F32) { HAMT_FAST_PIECES(true) } else { HAMT_FAST_PIECES(false) }
#undef HAMT_FAST_PIECES
      }
    } else {
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
      const int rl = p * RPP + (t >> 4), row = m0 + rl;
      const f32x4 lo = *(const f32x4*)(ct + rl * BN + (((2 * c8) ^ (rl & 7)) << 2));
      const f32x4 hi = *(const f32x4*)(ct + rl * BN + (((2 * c8 + 1) ^ (rl & 7)) << 2));
      const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (g.ksplit > 1) {   // raw partial tile, combined (and epilogued) by the reduce pass
        float* P = g.part + (size_t)slice * g.M * g.N;
        if (row < g.M) {
          if (col + 8 <= g.N && (g.N & 3) == 0) st_f<8>(P + (size_t)row * g.N + col, v8);
          else for (int e = 0; e < 8; ++e) if (col + e < g.N) P[(size_t)row * g.N + col + e] = v8[e];
        }
      } else if constexpr (PRE_AUX) epi_store<EPI, 8>(g, row, col, v8, (COLSUM && g.ss) ? &tile_ssq : nullptr, &pa[p], nullptr, pre_ok);
      else epi_store<EPI, 8>(g, row, col, v8, (COLSUM && g.ss) ? &tile_ssq : nullptr);
    }
    }
    if constexpr (COLSUM) { if (g.ss) tile_sumsq_store(g, m0, n0, tile_ssq, (float*)lds); }
  }
#ifdef HAMT_PROF
  {   // one record per wave: [dma wait, barrier, dma issue, fragments + MFMA, loop, 1, k-tiles, prologue, epilogue] in shader cycles
    const unsigned long long pf_end = __builtin_readcyclecounter();
    if (lane == 0) {
      const unsigned int slot = atomicAdd(&hamt_prof_n, 1u);
      if (slot < 65536u) {
        unsigned long long* r = hamt_prof_rec + (size_t)slot * 10;
        r[0] = pf_wait; r[1] = pf_bar; r[2] = pf_dma; r[3] = pf_mma; r[4] = pf_loop_end - pf_start; r[5] = 1; r[6] = (unsigned long long)nk;
        r[7] = pf_start - pf_entry; r[8] = pf_end - pf_loop_end;
      }
    }
  }
#endif
}


// NST = depth of the operand ring: 2 when two or three workgroups share a CU (they hide each other's DMA latency), 3 - 4 for
// grids of about one workgroup per CU (the narrow N = 768 outputs), where nothing else covers the ~1000-cycle L2 round trip.
template <int BM, int EPI, bool A_KM, bool B_KM, int NST = 2>
__global__ __launch_bounds__(256) void gemm_fast_kernel(GemmArgsF g) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int nk_all = g.K / BK;   // K range of this slice
  const int kt0 = (int)((long)nk_all * blockIdx.y / g.ksplit), kt1 = (int)((long)nk_all * (blockIdx.y + 1) / g.ksplit);
  gemm_tile<BM, EPI, A_KM, B_KM, NST, false>(g, (bid / tiles_n) * BM, (bid % tiles_n) * BN, kt0, kt1 - kt0, blockIdx.y, nullptr, 0);
}

// ---------------------------------------------------------------- K groups inside a workgroup (small grids, long reductions)
// A BM x 128 output tile computed by G groups of four waves: group q multiplies the k-tiles q, q + G, q + 2G, ... from its own
// two-deep operand ring, and the G accumulator tiles are summed through LDS (group order 0, 1, 2: deterministic) in front of
// the usual epilogue.  For outputs with fewer tiles than CUs (the B = 16 text stream: 1280 x 768 = 120 tiles of 64 rows) a
// plain tile leaves each SIMD one wave that issues its DMA pieces, reads its fragments and multiplies strictly in turn
// (~1000 cycles per k-tile for 256 cycles of MFMA); split-K over the grid pays for its partial tiles in HBM traffic
// (measured: 6 slices of 1280 x 768 fp32 = 29 us, as slow as no split at all).  Here the G waves of a SIMD overlap each
// other's phases and the partial sums never leave the CU.  The epilogue flags are read at run time (one or two row pieces
// per thread: nothing to unroll).
template <int N> __device__ __forceinline__ void wait_vmcnt_n() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
template <int BM, int G, int D, bool B_KM>
__global__ __launch_bounds__(256 * G) void gemm_kg_kernel(GemmArgsF g) {
  constexpr int NW = 4, TM = BM / 2, TN = 64, FM = TM / 16, FN = 4;
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS, RING = D * STAGE;
  constexpr int NLD = (BM + BN) / (8 * NW);      // DMA pieces per wave per stage
  static_assert(BM * BN * 4 <= RING * 2, "a group's accumulator tile must fit its operand ring");
  __shared__ __attribute__((aligned(16))) bf16_t lds[G * RING];
  const int tiles_n = (g.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, ((g.M + BM - 1) / BM) * tiles_n);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, w = wave & 3, wm = w >> 1, wn = w & 1;
  const int nk = g.K / BK, iters = (nk + G - 1) / G, mine = (nk - grp + G - 1) / G;   // k-tiles grp + i G, i < mine

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16_t* ring = lds + grp * RING;
  const unsigned ring0 = lds_base_of(lds) + (unsigned)(grp * RING * 2);
  TileSrc<false, BM, NW> src_a;
  TileSrc<B_KM, BN, NW> src_b;
  src_a.init(g.lda, m0, g.M - 1, w, lane);
  src_b.init(g.ldb, n0, g.N - 1, w, lane);
  auto stage = [&](int i, int slot) {
    const unsigned dst = ring0 + (unsigned)(slot * STAGE * 2);
    const int k0 = (grp + i * G) * BK;
    src_a.issue(g.A, g.lda, k0, g.ka_max, dst, w);
    src_b.issue(g.B, g.ldb, k0, g.kb_max, dst + (unsigned)(A_ELEMS * 2), w);
  };
#pragma unroll
  for (int p = 0; p < D - 1; ++p)
    if (p < mine) stage(p, p);
  for (int i = 0; i < iters; ++i) {
    {   // tile i has landed when at most the pieces of the (up to D - 2) younger tiles are outstanding
      const int younger = min(D - 2, mine - 1 - i);
      if (D >= 6 && younger >= 4) wait_vmcnt_n<4 * NLD>();
      else if (D >= 5 && younger == 3) wait_vmcnt_n<3 * NLD>();
      else if (D >= 4 && younger == 2) wait_vmcnt_n<2 * NLD>();
      else if (D >= 3 && younger == 1) wait_vmcnt_n<NLD>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();               // tile i of every group is in LDS; everyone is done with tile i - 1
    if (i + D - 1 < mine) stage(i + D - 1, (i + D - 1) % D);
    if (i < mine) {
      const bf16_t* As = ring + (i % D) * STAGE;
      const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[FM], bfr[FN];
#pragma unroll
        for (int a = 0; a < FM; ++a) af[a] = frag<false, BM>(As, wm * TM + a * 16, s, lane);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = frag<B_KM, BN>(Bs, wn * TN + j * 16, s, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[a], acc[a][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  // accumulator tiles -> LDS (each group into its own ring), summed in group order, epilogue: 16 threads per row, 8 columns each
  float* ct = (float*)ring;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int rl = wm * TM + a * 16 + (lane & 15), c4 = wn * (TN / 4) + j * 4 + (lane >> 4);
      *(f32x4*)(ct + rl * BN + ((c4 ^ (rl & 7)) << 2)) = acc[a][j];
    }
  __syncthreads();
  auto piece_sum = [&](int rl, int c8, float* v8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = 0.f;
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const float* cq = (const float*)(lds + q * RING);
      const f32x4 lo = *(const f32x4*)(cq + rl * BN + (((2 * c8) ^ (rl & 7)) << 2));
      const f32x4 hi = *(const f32x4*)(cq + rl * BN + (((2 * c8 + 1) ^ (rl & 7)) << 2));
      v8[0] += lo[0]; v8[1] += lo[1]; v8[2] += lo[2]; v8[3] += lo[3]; v8[4] += hi[0]; v8[5] += hi[1]; v8[6] += hi[2]; v8[7] += hi[3];
    }
  };
  // interior tiles with the epilogues this kernel is launched with in the step (bias / accumulate / none): see epi_fast8.  A thread's
  // pieces all sit in the same 8 columns (256 G threads = a multiple of 16 pieces per row), so its bias is loaded once.
  const int fe = g.epi;
  if ((fe == HAMT_EPI_BIAS || fe == HAMT_EPI_ACCUM || fe == 0) && epi_fast_ok(g, fe, m0, n0, BM, BN)) {
    const int c8 = t & 15, col = n0 + c8 * 8;
    auto run = [&](auto ec, auto c16) {
      constexpr int E = decltype(ec)::value;
      float b8[8];
      epi_fast_bias<E>(g, col, b8);
#pragma unroll
      for (int rl = t >> 4; rl < BM; rl += 16 * G) {
        float v8[8];
        piece_sum(rl, c8, v8);
        epi_fast8<E, decltype(c16)::value>(g, m0 + rl, col, v8, b8, nullptr);
      }
    };
    using std::integral_constant;
    const bool c16 = g.dtype_c != HAMT_F32;
    if (fe == HAMT_EPI_BIAS) { if (c16) run(integral_constant<int, HAMT_EPI_BIAS>{}, std::true_type{}); else run(integral_constant<int, HAMT_EPI_BIAS>{}, std::false_type{}); }
    else if (fe == HAMT_EPI_ACCUM) { if (c16) run(integral_constant<int, HAMT_EPI_ACCUM>{}, std::true_type{}); else run(integral_constant<int, HAMT_EPI_ACCUM>{}, std::false_type{}); }
    else { if (c16) run(integral_constant<int, 0>{}, std::true_type{}); else run(integral_constant<int, 0>{}, std::false_type{}); }
    return;
  }
  for (int piece = t; piece < BM * 16; piece += 256 * G) {
    const int rl = piece >> 4, c8 = piece & 15;
    float v8[8];
    piece_sum(rl, c8, v8);
    epi_store<-1, 8>(g, m0 + rl, n0 + c8 * 8, v8);
  }
}

// ---------------------------------------------------------------- grouped weight gradients
// Up to WG_MAX independent problems dW_p[M_p,N_p] (+)= dY_p^T X_p (and db_p (+)= colsum dY_p) in ONE launch: the tile ids
// of all problems are concatenated (longest reductions first), so 768x768 outputs that alone would fill 36 CUs (or need
// split-K + a reduce pass) run as one chip-filling grid with full-length K loops.  Problem table by value in the kernarg.
constexpr int WG_MAX = 192;                      // table entries carried by one kernarg block (112 bytes each: 21 KiB of kernarg, one launch for ~190 units)
struct WgradProb {
  const bf16_t* dy; const bf16_t* x; float* dw; float* db; float* ss;
  int M, N, K, ldy, ldx, ldw, flags, tile_end;   // flags: 1 = dW +=, 2 = db +=, >> 8 = ss slots per 64-row band; tile_end = exclusive prefix end
  int kv;                                        // valid reduction rows (<= K): rows behind are padding of any content
  float wscale;                                  // != 0: dw is a bf16 array, the tile stores bf16(wscale * dW) (hamt_wgrad_desc.wire_scale)
  const bf16_t* dy2; const bf16_t* x2;           // second reduction into the same tile (hamt_wgrad_desc.dy2), or null
  int K2, ldy2, ldx2, kv2;
};
static_assert(sizeof(WgradProb) == HAMT_WGRAD_TABLE_ENTRY, "HAMT_WGRAD_TABLE_ENTRY");
struct WgradChunk { WgradProb p[WG_MAX]; };
// The problem table lives in caller-provided device memory and is WRITTEN BY KERNELS whose kernargs carry it WG_MAX entries
// at a time: no host buffer has to outlive the call, so the whole sequence is hipGraph-capturable as is.
__global__ void wgrad_table_write_kernel(WgradChunk c, WgradProb* tab, int off, int cnt) {
  if ((int)threadIdx.x < cnt) tab[off + threadIdx.x] = c.p[threadIdx.x];
}

struct WgradXcd { int start[9]; };               // table entries [start[x], start[x+1]) belong to XCD x

// Placement: blocks are dealt round-robin to the 8 XCDs (XCD = blockIdx & 7), each with its own 4 MiB L2.  Every table
// entry (a problem, or a band of tile rows of a large one) is processed ENTIRELY by one XCD, so the operand panels its
// tiles share (3 + 3 panels of 2.6 MB for a 768x768x5120 problem with 256-square tiles) cross the fabric once instead of
// once per tile: measured 8.4 GB -> see profiles/ of HBM/MALL fetch per launch, the kernel was fabric-bound.  The host
// balances the XCDs (longest-processing-time first) and orders each XCD's entries by K, longest first.
template <int BM, int BNT, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_grouped_kernel(const WgradProb* __restrict__ tab, WgradXcd xs) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  int lo = xs.start[xcd], hi = xs.start[xcd + 1] - 1;
  if (hi < lo || idx >= tab[hi].tile_end) return;   // this XCD has fewer tiles than the longest queue
  const int first = lo;
  while (lo < hi) {                              // first entry whose tile_end > idx (wave-uniform binary search)
    const int mid = (lo + hi) >> 1;
    if (idx >= tab[mid].tile_end) lo = mid + 1; else hi = mid;
  }
  const WgradProb q = tab[lo];
  const int local = idx - (lo > first ? tab[lo - 1].tile_end : 0);
  const int tiles_n = (q.N + BNT - 1) / BNT;
  const int nseg = q.dy2 ? 2 : 1;
  for (int seg = 0; seg < nseg; ++seg) {          // (a second pair of operands: the same tile again, accumulating; the tile sums of squares by the last)
    GemmArgsF g{q.M, q.N, seg ? q.K2 : q.K, seg ? q.ldy2 : q.ldy, seg ? q.ldx2 : q.ldx, q.ldw, 0, q.wscale != 0.f ? HAMT_BF16 : HAMT_F32, 0,
                ((q.flags & 1) || seg) ? HAMT_EPI_ACCUM : 0, q.wscale != 0.f ? q.wscale : 1.0f, seg ? q.dy2 : q.dy, seg ? q.x2 : q.x, q.dw, nullptr, nullptr, 1,
                nullptr, (seg ? q.kv2 : q.kv) - 1, (seg ? q.kv2 : q.kv) - 1, 0.f, 0u, nullptr, seg == nseg - 1 ? q.ss : nullptr, q.flags >> 8};
    gemm_tile<BM, -2, true, true, 2, true, BNT, WM, WN>(g, (local / tiles_n) * BM, (local % tiles_n) * BNT, 0, g.K / BK, 0, q.db, (q.flags & 2) || seg);
    if (nseg > 1) __syncthreads();
  }
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

// 256x256 tile, 8 waves, one workgroup per CU: for grids of about one (or several) 256-square tiles per CU
template <int EPI, bool A_KM, bool B_KM>
__global__ __launch_bounds__(512) void gemm_fast256_kernel(GemmArgsF g) {
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 255) / 256;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  gemm_tile<256, EPI, A_KM, B_KM, 2, false, 256, 2, 4>(g, (bid / tiles_n) * 256, (bid % tiles_n) * 256, 0, g.K / BK, 0, nullptr, 0);
}

// ---------------------------------------------------------------- 256 x 256 tile, 8 waves, two phases per k-tile ("p8")
// The schedule family of cdna_hip_programming.md section 5 (T3+T4: counted vmcnt, barrier-separated load / multiply
// segments, the two wave rows running one barrier apart so that on every SIMD one wave multiplies while the other reads
// LDS and issues DMA), with the phase length chosen from a cycle count of this kernel (tools/p8_prof.py): a multiply
// segment of 16 MFMAs took ~330 cycles and every barrier ~58, 8 of each per k-tile = 3100 cycles for 2048 of MFMA issue;
// 32 MFMAs per segment halves the barriers.
//   * waves 2 (M) x 4 (N), each a 128 x 64 output = a0/a1 (64 rows each) x b0/b1 (32 columns each); phase A multiplies
//     a0 x (b0, b1), phase B a1 x (b0, b1): 24 fragment reads per 64 MFMAs (a0 + b0 + b1, then a1);
//   * a k-tile's operands are staged as four 16 KiB units of 128 rows x 64 k: X0 = the a0 rows of both wave rows,
//     Y0 / Y1 = the b0 / b1 columns of the four wave columns, X1.  X0, Y0, Y1 die in phase A, X1 in phase B; every load
//     segment ends with lgkmcnt(0) before its barrier, so a slot may be re-staged from the phase after its last read:
//     phase B(t) issues X0, Y0 of k-tile t+2, phase A(t+1) issues Y1, X1 of k-tile t+2 -- every unit is in flight for at
//     least two whole phases, 3-4 units stay in flight across the barriers (vmcnt(8) / vmcnt(6));
//   * DMA addresses are 32-bit element offsets from a scalar base (one v_add / v_mad per piece);
//   * the epilogue re-tiles the accumulators through wave-private LDS (no workgroup barrier) so that a lane stores 8
//     consecutive columns.
constexpr int P8_UNIT = 128 * BK;                // elements per unit (16 KiB)
constexpr int P8_BUF = 4 * P8_UNIT;              // elements per k-tile buffer (64 KiB): [X0, Y0, Y1, X1]

// unit-local row / column r (0..127) of the `half`-th unit of an operand -> tile-local index.  GS = log2 of the group
// size: A units are two groups of 64 rows (one per wave row), B units four groups of 32 columns (one per wave column).
// GS = 7: the unit is one contiguous half of the tile (used for K-strided B operands, whose DMA pieces then read whole
// 128-byte lines of the [K][N] rows; wave column wc then owns columns [32 wc, +32) and 128 + [32 wc, +32)).
template <int GS> __device__ __forceinline__ int p8_index(int r, int half) {
  if constexpr (GS == 7) return half * 128 + r;
  else return ((r >> GS) << (GS + 1)) + (half << GS) + (r & ((1 << GS) - 1));
}

// Per-lane source description of one operand's two units (2 DMA pieces each), computed once per tile.
//   K-contiguous: off[h][j] = byte offset of (row, chunk) at k = 0; a k-tile adds 128 bytes.
//   K-strided:    col[h][j] = element column, kk[j] = k row within the tile; offset = min(k0 + kk, kmax) * ld + col.
template <bool KM> struct P8Src { unsigned off[2][2]; int kk[2]; int kk1; };

// B-operand column of unit-local index r of unit `half` when a wave column owns NB = 3 fragments (a 256 x 192 tile: unit 0 =
// the b0 columns, 2 fragments = 32 per wave column; unit 1 = the b1 columns, ONE fragment = 16 per wave column, a 64-wide unit):
// K-contiguous B: wave column g owns tile columns [48 g, 48 g + 48); K-strided B: unit 0 = columns [0, 128), unit 1 = [128, 192).
template <bool KM> __device__ __forceinline__ int p8_bcol3(int r, int half) {
  if constexpr (KM) return half * 128 + r;
  else return half == 0 ? 48 * (r >> 5) + (r & 31) : 48 * (r >> 4) + 32 + (r & 15);
}

// H1 = rows / columns of unit 1 (128, or 64 for the B operand of the 192-column tile: one DMA piece per wave instead of two)
template <bool KM, int GS, int H1 = 128>
__device__ __forceinline__ void p8_src_init(P8Src<KM>& sd, int ld, int o0, int omax, int w, int lane) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (H1 == 64 && h == 1) {
        if (j == 1) continue;
        if constexpr (!KM) {     // image [64][64 k], chunk ^= row & 7
          const int r = w * 8 + (lane >> 3);
          const int chunk = (lane & 7) ^ (r & 7);
          int gr = o0 + p8_bcol3<KM>(r, 1);
          gr = gr < omax ? gr : omax;
          sd.off[1][0] = ((unsigned)gr * (unsigned)ld + (unsigned)chunk * 8u) * 2u;
        } else {                 // image [64 k][64], 16-byte chunk ^= col_swz<64>(k row)
          const int kk = w * 8 + (lane >> 3);
          const int chunk = (lane & 7) ^ col_swz<64>(kk);
          int c = o0 + p8_bcol3<KM>(chunk * 8, 1);
          const int cmax = min(ld - 8, (omax >> 3) << 3);
          c = c < cmax ? c : cmax;
          sd.off[1][0] = (unsigned)c;
          sd.kk1 = kk;
        }
        continue;
      }
      if constexpr (!KM) {     // image [128][64], chunk ^= row & 7
        const int r = w * 16 + j * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (r & 7);
        int gr = o0 + (H1 == 64 ? p8_bcol3<KM>(r, h) : p8_index<GS>(r, h));
        gr = gr < omax ? gr : omax;
        sd.off[h][j] = ((unsigned)gr * (unsigned)ld + (unsigned)chunk * 8u) * 2u;
      } else {                 // image [64][128], 16-byte chunk ^= col_swz<128>(k row)
        const int kk = w * 8 + j * 4 + (lane >> 4);
        const int chunk = (lane & 15) ^ col_swz<128>(kk);
        int c = o0 + (H1 == 64 ? p8_bcol3<KM>(chunk * 8, h) : p8_index<GS>(chunk * 8, h));
        const int cmax = min(ld - 8, (omax >> 3) << 3);     // never behind the last 8-column chunk that holds a needed column:
        c = c < cmax ? c : cmax;                            // stays inside the operand (a slice of a wider buffer, the buffer's last row)
        sd.off[h][j] = (unsigned)c;
        sd.kk[j] = kk;
      }
    }
}

template <bool KM, int H1 = 128>
__device__ __forceinline__ void p8_issue(const P8Src<KM>& sd, const bf16_t* __restrict__ P, int ld, int kt, int kmax, unsigned dst, int h, int w) {
  if (H1 == 64 && h == 1) {      // the 64-wide unit: one piece per wave
    if constexpr (!KM) glds16_off(P, sd.off[1][0] + (unsigned)kt * (BK * 2), dst + (unsigned)((w * 8) * BK * 2));
    else {
      int gk = kt * BK + sd.kk1;
      gk = gk < kmax ? gk : kmax;
      glds16_off(P, ((unsigned)gk * (unsigned)ld + sd.off[1][0]) * 2u, dst + (unsigned)((w * 8) * 64 * 2));
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if constexpr (!KM) {
      glds16_off(P, sd.off[h][j] + (unsigned)kt * (BK * 2), dst + (unsigned)((w * 16 + j * 8) * BK * 2));
    } else {
      int gk = kt * BK + sd.kk[j];
      gk = gk < kmax ? gk : kmax;
      glds16_off(P, ((unsigned)gk * (unsigned)ld + sd.off[h][j]) * 2u, dst + (unsigned)((w * 8 + j * 4) * 128 * 2));
    }
  }
}

template <int N> __device__ __forceinline__ void p8_wait() {
  static_assert(N == 0 || N == 2 || N == 6 || N == 7 || N == 8, "p8_wait");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// COLSUM (weight-gradient form, A = dY stored [K][M]): the tiles with n0 == 0 also reduce A over k -- one extra MFMA per k-half
// and phase in EVERY wave: the four wave columns of a wave row hold the same eight A fragments, wave column c sums fragment c of
// a0 (phase A) and of a1 (phase B) against an operand that holds ones in row P only, so both land in rows 0 / 1 of ONE accumulator
// (4 registers): the bias gradient, for free of any extra pass over dY.  (Round 2 had wave column 0 sum all eight: +25 % MFMA
// issue on SIMD 0 alone, which every barrier of the tile then waited for -- 11 % of the grouped launch.)
// NB = B fragments per wave: 4 = the 256-column tile above; 3 = a 256 x 192 tile (wave output 128 x 48: b0 two fragments, b1 one;
// unit Y1 is 64 wide, one DMA piece per wave) for grids whose 256-square tiling leaves the last round mostly empty -- 5120 x 2304 is
// 180 tiles of 256 x 256 (0.70 of a round of 256 CUs) and 240 of 256 x 192 (0.94): the round is 0.8 as long and all of it is used.
template <int EPI, bool A_KM, bool B_KM, bool COLSUM, int NB = 4>
__device__ __forceinline__ void p8_tile(const GemmArgsF& g, int m0, int n0, float* db, int db_accum) {
  constexpr int H1B = NB == 3 ? 64 : 128;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * P8_BUF];          // 128 KiB
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;
  const bool do_cs = COLSUM && db != nullptr && n0 == 0;                   // workgroup-uniform
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  // COLSUM (= the weight-gradient instantiation): the reduction may run over TWO operand pairs back to back (GemmArgsF::A2): k-tiles
  // [0, nk1) come from (A, B), [nk1, nk) from (A2, B2) -- base pointer, row stride and row clamp are scalar selects at issue time
  const int nk1 = g.K / BK;
  const int nk = COLSUM ? nk1 + g.K2 / BK : nk1;
  const unsigned lds0 = lds_base_of(lds);

  f32x4 acc[8][NB];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bf_[NB][2];

  P8Src<A_KM> sa;
  P8Src<B_KM> sb;
  p8_src_init<A_KM, 6>(sa, g.lda, m0, g.M - 1, w, lane);
  constexpr int BGS = B_KM ? 7 : 5;
  p8_src_init<B_KM, BGS, H1B>(sb, g.ldb, n0, g.N - 1, w, lane);
  auto slot = [&](int kt, int ty) { return lds0 + (unsigned)(((kt & 1) * P8_BUF + ty * P8_UNIT) * 2); };
  auto issue_x = [&](int kt, int h) {
    if constexpr (COLSUM) {
      const bool s2 = kt >= nk1;
      p8_issue<A_KM>(sa, s2 ? g.A2 : g.A, s2 ? g.lda2 : g.lda, s2 ? kt - nk1 : kt, s2 ? g.k2_max : g.ka_max, slot(kt, h ? 3 : 0), h, w);
    } else p8_issue<A_KM>(sa, g.A, g.lda, kt, g.ka_max, slot(kt, h ? 3 : 0), h, w);
  };
  auto issue_y = [&](int kt, int h) {
    if constexpr (COLSUM) {
      const bool s2 = kt >= nk1;
      p8_issue<B_KM, H1B>(sb, s2 ? g.B2 : g.B, s2 ? g.ldb2 : g.ldb, s2 ? kt - nk1 : kt, s2 ? g.k2_max : g.kb_max, slot(kt, 1 + h), h, w);
    } else p8_issue<B_KM, H1B>(sb, g.B, g.ldb, kt, g.kb_max, slot(kt, 1 + h), h, w);
  };
  // prologue: k-tile 0 and X0, Y0 of k-tile 1; phase A(0) reads X0, Y0, Y1 of k-tile 0 (three younger units may be in flight)
  issue_x(0, 0); issue_y(0, 0); issue_y(0, 1); issue_x(0, 1); issue_x(1, 0); issue_y(1, 0);
  p8_wait<6>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first

#ifdef HAMT_PROF
  unsigned long long pf[4][4] = {}, pf_t = 0;
#define P8_STAMP(P, i) { const unsigned long long t_ = __builtin_readcyclecounter(); if (i) pf[P][i - 1] += t_ - pf_t; pf_t = t_; }
#else
#define P8_STAMP(P, i)
#endif
  // one phase: load segment | barrier | multiply segment | barrier.  MODE 0: steady state; 1: k-tile nk-2; 2: k-tile nk-1
  auto phase = [&](int kt, auto pc, auto modec) {
    constexpr int P = decltype(pc)::value, MODE = decltype(modec)::value;
    const bf16_t* buf = lds + (kt & 1) * P8_BUF;
    P8_STAMP(P, 0)
    if constexpr (P == 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bf_[j][s] = frag<B_KM, 128>(buf + 1 * P8_UNIT, 32 * wc + j * 16, s, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][s] = frag<A_KM, 128>(buf + 0 * P8_UNIT, 64 * wr + i * 16, s, lane);
        if constexpr (NB == 4) {
#pragma unroll
          for (int j = 0; j < 2; ++j) bf_[2 + j][s] = frag<B_KM, 128>(buf + 2 * P8_UNIT, 32 * wc + j * 16, s, lane);
        } else bf_[2][s] = frag<B_KM, 64>(buf + 2 * P8_UNIT, 16 * wc, s, lane);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i][s] = frag<A_KM, 128>(buf + 3 * P8_UNIT, 64 * wr + i * 16, s, lane);
    }
    if constexpr (A_KM && MODE == 2) {        // (the launcher shrinks K to the 64-row tile that holds the last valid row: only the
      // LAST k-tile can cross it -- masking code in every phase instance spilled registers; with a second operand pair the last k-tile
      // is ITS last one, and the first pair has no ragged tail: hamt_wgrad_desc)
      const int ktl = (COLSUM && g.K2 > 0) ? kt - nk1 : kt, kend = ((COLSUM && g.K2 > 0) ? g.k2_max : g.ka_max) + 1;
      if (ktl * BK + BK > kend) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i][s] = mask_k_tail(af[i][s], ktl * BK + 32 * s, lane, kend);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P == 0) {       // A(t): Y1, X1 of k-tile t+1; the next phase reads X1(t)
      if constexpr (MODE < 2) { issue_y(kt + 1, 1); issue_x(kt + 1, 1); p8_wait<NB == 4 ? 8 : 7>(); }
      else p8_wait<0>();
    } else {                      // B(t): X0, Y0 of k-tile t+2; the next phase reads X0, Y0, Y1 of k-tile t+1
      if constexpr (MODE == 0) { issue_x(kt + 2, 0); issue_y(kt + 2, 0); p8_wait<6>(); }
      else if constexpr (MODE == 1) p8_wait<2>();
      else p8_wait<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    P8_STAMP(P, 1)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    P8_STAMP(P, 2)
    __builtin_amdgcn_s_setprio(1);
    constexpr int I0 = P * 4;
    // COLSUM: this wave's fragment (af[wc], picked by selects that run under the MFMAs -- no branch inside the multiply run) against
    // ones in row P, one MFMA behind each k-half of the product.
    // Rows of A beyond M (a ragged last tile) hold whatever lies behind the operand's last column -- padding, the next
    // row -- possibly NaN bit patterns.  In the main product such a row only feeds its own (unstored) output row; HERE
    // both fragments accumulate into the same 16 x 16 block, and 0 x NaN from one fragment's row would poison the column
    // sums of the other's (first seen as a NaN gradient of the 30 522-row MLM decoder bias).
    union { uint4 u; bf16x8 v; } sel, csa[2];
    if constexpr (COLSUM) {
      const uint32_t o2 = (lane & 15) == P ? 0x3F803F80u : 0u;               // bf16 ones in row P of the operand
      sel.u = make_uint4(o2, o2, o2, o2);
      const bool live = m0 + 128 * wr + 64 * P + 16 * wc + (lane & 15) < g.M;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        union { uint4 u; bf16x8 v; } f0, f1, f2, f3;
        f0.v = af[0][s]; f1.v = af[1][s]; f2.v = af[2][s]; f3.v = af[3][s];
        const uint4 lo = wc & 1 ? f1.u : f0.u, hi = wc & 1 ? f3.u : f2.u;
        csa[s].u = wc & 2 ? hi : lo;
        if (!live) csa[s].u = make_uint4(0u, 0u, 0u, 0u);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[I0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[j][s], af[i][s], acc[I0 + i][j], 0, 0, 0);
      if constexpr (COLSUM) {
        if (do_cs) cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel.v, csa[s].v, cs, 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    P8_STAMP(P, 3)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    P8_STAMP(P, 4)
  };
  using std::integral_constant;
  int kt = 0;
  for (; kt < nk - 2; ++kt) {
    phase(kt, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    phase(kt, integral_constant<int, 1>{}, integral_constant<int, 0>{});
  }
  phase(kt, integral_constant<int, 0>{}, integral_constant<int, 1>{});
  phase(kt, integral_constant<int, 1>{}, integral_constant<int, 1>{});
  ++kt;
  phase(kt, integral_constant<int, 0>{}, integral_constant<int, 2>{});
  phase(kt, integral_constant<int, 1>{}, integral_constant<int, 2>{});
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balances the second wave row's extra barrier
#ifdef HAMT_PROF
  if (lane == 0) {
    for (int p_ = 0; p_ < 4; ++p_) for (int q_ = 0; q_ < 4; ++q_) atomicAdd(&hamt_p8_prof[wr * 16 + p_ * 4 + q_], pf[p_][q_]);
    atomicAdd(&hamt_p8_prof[32 + wr], 1ull);
    atomicAdd(&hamt_p8_prof[34 + wr], (unsigned long long)nk);
  }
#endif

  if constexpr (COLSUM) {
    if (do_cs) {   // cs[e] of lanes 0..15 = column sums of this wave's fragment of a0 (e = 0) and of a1 (e = 1) at row `lane`
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int row = m0 + 128 * wr + 64 * e + 16 * wc + lane;
        if (lane < 16 && row < g.M) db[row] = db_accum ? db[row] + cs[e] : cs[e];
      }
    }
  }
  // epilogue: two passes (a0 rows, a1 rows) through this wave's own 16 KiB of LDS: [64][64] fp32, float4 slot ^= row & 7
  float tile_ssq = 0.f;
  float* ct = (float*)lds + w * 4096;
  // An epilogue that READS (the saved gelu' of `mulaux`) issues the half's eight row-piece loads before
  // the transpose: inside epi_store each load sits behind the previous piece's store and its latency is paid 16 times per thread.
  constexpr bool PRE_AUX = EPI >= 0 && (EPI & HAMT_EPI_MUL_AUX) != 0;
  constexpr bool PRE_C = false;   // the same for the fp32 C of `acc` (16 registers per piece) measured 1 us SLOWER per launch: off
  const int c8 = lane & 7;
  int col;
  if constexpr (NB == 4) col = n0 + p8_index<BGS>(32 * wc + (c8 & 3) * 8, c8 >> 2);
  else col = c8 < 6 ? n0 + p8_bcol3<B_KM>(c8 < 4 ? 32 * wc + c8 * 8 : 16 * wc + (c8 - 4) * 8, c8 >> 2) : g.N;   // (chunks 6, 7: no columns)
  bool pre_ok = false;
  if constexpr (PRE_AUX) pre_ok = col + 8 <= g.N && (g.dtype_aux == HAMT_BF16 || g.dtype_aux == HAMT_U8G) && (g.ldaux % 8) == 0 && ((uintptr_t)g.aux % 16) == 0;
  if constexpr (PRE_C) pre_ok = col + 8 <= g.N && g.dtype_c == HAMT_F32 && (g.ldc % 8) == 0 && ((uintptr_t)g.C % 16) == 0;
  constexpr bool FASTK = EPI >= 0 && (EPI & ~EPI_FAST_MASK) == 0 && !COLSUM;
  constexpr bool FASTW = COLSUM && EPI == -2 && NB == 4;      // the weight-gradient tile: fp32 store or C +=, tile sum of squares
  bool fast = false;
  if constexpr (FASTK) fast = epi_fast_ok(g, EPI, m0, n0, 256, 64 * NB);      // interior tile (workgroup uniform): see epi_fast8
  if constexpr (FASTW) fast = g.dtype_c == HAMT_F32 && epi_fast_ok(g, g.epi & HAMT_EPI_ACCUM, m0, n0, 256, 256);
  if (fast) {
    if constexpr (FASTW) {
      const float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float* sq = g.ss ? &tile_ssq : nullptr;
      auto run = [&](auto accc) {
        constexpr int E = decltype(accc)::value ? HAMT_EPI_ACCUM : 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              const int rl = i * 16 + (lane & 15), c4 = j * 4 + (lane >> 4);
              *(f32x4*)(ct + rl * 64 + ((c4 ^ (rl & 7)) << 2)) = acc[4 * h + i][j];
            }
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int rl = it * 8 + (lane >> 3), row = m0 + 128 * wr + 64 * h + rl;
            const f32x4 lo = *(const f32x4*)(ct + rl * 64 + (((2 * c8) ^ (rl & 7)) << 2));
            const f32x4 hi = *(const f32x4*)(ct + rl * 64 + (((2 * c8 + 1) ^ (rl & 7)) << 2));
            const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            epi_fast8<E, false>(g, row, col, v8, b8, nullptr, sq);
          }
        }
      };
      if (g.epi & HAMT_EPI_ACCUM) run(std::true_type{}); else run(std::false_type{});
    }
    if constexpr (FASTK) {
      const bool live = NB == 4 || c8 < 6;         // (256 x 192 tile: row-piece chunks 6 and 7 of a wave's 64-column slab hold no columns)
      float b8[8];
      if (live) epi_fast_bias<EPI>(g, col, b8);
      auto run = [&](auto c16) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint4 pa[PRE_AUX ? 8 : 1];
          if constexpr (PRE_AUX) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
              if (live) pa[it] = *(const uint4*)((const bf16_t*)g.aux + (size_t)min(m0 + 128 * wr + 64 * h + it * 8 + (lane >> 3), g.M - 1) * g.ldaux + col);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              const int rl = i * 16 + (lane & 15), c4 = j * 4 + (lane >> 4);
              *(f32x4*)(ct + rl * 64 + ((c4 ^ (rl & 7)) << 2)) = acc[4 * h + i][j];
            }
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int rl = it * 8 + (lane >> 3), row = m0 + 128 * wr + 64 * h + rl;
            const f32x4 lo = *(const f32x4*)(ct + rl * 64 + (((2 * c8) ^ (rl & 7)) << 2));
            const f32x4 hi = *(const f32x4*)(ct + rl * 64 + (((2 * c8 + 1) ^ (rl & 7)) << 2));
            const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const uint4* pp = nullptr;
            if constexpr (PRE_AUX) pp = &pa[it];
            if (live) epi_fast8<EPI, decltype(c16)::value>(g, row, col, v8, b8, pp);
          }
        }
      };
      if (g.dtype_c != HAMT_F32) run(std::true_type{}); else run(std::false_type{});
    }
  } else {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint4 pa[PRE_AUX ? 8 : 1];
    f32x4 pc[PRE_C ? 16 : 2];
    if constexpr (PRE_AUX || PRE_C) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = m0 + 128 * wr + 64 * h + it * 8 + (lane >> 3);
        const int rr = row < g.M ? row : g.M - 1;            // (an out-of-range row's piece is loaded from the last row and never used)
        if constexpr (PRE_AUX) { if (pre_ok) pa[it] = ld_aux8(g, rr, col); }
        if constexpr (PRE_C) {
          if (pre_ok) { const f32x4* cp = (const f32x4*)((const float*)g.C + (size_t)rr * g.ldc + col); pc[2 * it] = cp[0]; pc[2 * it + 1] = cp[1]; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int rl = i * 16 + (lane & 15), c4 = j * 4 + (lane >> 4);
        *(f32x4*)(ct + rl * 64 + ((c4 ^ (rl & 7)) << 2)) = acc[4 * h + i][j];
      }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rl = it * 8 + (lane >> 3), row = m0 + 128 * wr + 64 * h + rl;
      const f32x4 lo = *(const f32x4*)(ct + rl * 64 + (((2 * c8) ^ (rl & 7)) << 2));
      const f32x4 hi = *(const f32x4*)(ct + rl * 64 + (((2 * c8 + 1) ^ (rl & 7)) << 2));
      const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if constexpr (PRE_AUX) epi_store<EPI, 8>(g, row, col, v8, (COLSUM && g.ss) ? &tile_ssq : nullptr, &pa[it], nullptr, pre_ok);
      else if constexpr (PRE_C) epi_store<EPI, 8>(g, row, col, v8, (COLSUM && g.ss) ? &tile_ssq : nullptr, nullptr, &pc[2 * it], pre_ok);
      else epi_store<EPI, 8>(g, row, col, v8, (COLSUM && g.ss) ? &tile_ssq : nullptr);
    }
  }
  }
  if constexpr (COLSUM) {
    if (g.ss) {
      __shared__ float p8_red[8];
      tile_sumsq_store(g, m0, n0, tile_ssq, p8_red);
    }
  }
}

template <int EPI, bool A_KM, bool B_KM, int NB = 4>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmArgsF g) {
  constexpr int TN = 64 * NB;
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + TN - 1) / TN;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  p8_tile<EPI, A_KM, B_KM, false, NB>(g, (bid / tiles_n) * 256, (bid % tiles_n) * TN, nullptr, 0);
}

// the grouped weight-gradient launch (see wgrad_grouped_kernel) on the two-phase tile
__global__ __launch_bounds__(512) void wgrad_grouped_p8_kernel(const WgradProb* __restrict__ tab, WgradXcd xs) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  int lo = xs.start[xcd], hi = xs.start[xcd + 1] - 1;
  if (hi < lo || idx >= tab[hi].tile_end) return;
  const int first = lo;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (idx >= tab[mid].tile_end) lo = mid + 1; else hi = mid;
  }
  const WgradProb q = tab[lo];
  const int local = idx - (lo > first ? tab[lo - 1].tile_end : 0);
  const int tiles_n = (q.N + 255) / 256;
  GemmArgsF g{q.M, q.N, q.K, q.ldy, q.ldx, q.ldw, 0, q.wscale != 0.f ? HAMT_BF16 : HAMT_F32, 0, (q.flags & 1) ? HAMT_EPI_ACCUM : 0,
              q.wscale != 0.f ? q.wscale : 1.0f, q.dy, q.x, q.dw, nullptr, nullptr, 1, nullptr, q.kv - 1, q.kv - 1, 0.f, 0u, nullptr, q.ss, q.flags >> 8,
              q.dy2, q.x2, q.dy2 ? q.K2 : 0, q.ldy2, q.ldx2, q.kv2 - 1};       // (a second operand pair: the tile's reduction simply goes on over it)
  p8_tile<-2, true, true, true>(g, (local / tiles_n) * 256, (local % tiles_n) * 256, q.db, q.flags & 2);
}

template <bool A_KM, bool B_KM, int NB = 4>
bool launch_p8(const GemmArgsF& g, hipStream_t s) {
  const dim3 grid(((g.M + 255) / 256) * ((g.N + 64 * NB - 1) / (64 * NB)));
  const int e = g.epi;
#define HAMT_L(E) do { hipLaunchKernelGGL((gemm_p8_kernel<E, A_KM, B_KM, NB>), grid, dim3(512), 0, s, g); \
                    hamt_set_last_kernel("gemm_p8_kernel<%d, %s, %s, %d>", (int)(E), A_KM ? "true" : "false", B_KM ? "true" : "false", NB); } while (0)
  if (e == 0) HAMT_L(0);
  else if (e == HAMT_EPI_BIAS) HAMT_L(HAMT_EPI_BIAS);
  else if (e == HAMT_EPI_ACCUM) HAMT_L(HAMT_EPI_ACCUM);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD);
  else if (e == HAMT_EPI_MUL_AUX) HAMT_L(HAMT_EPI_MUL_AUX);
  else if constexpr (!B_KM && NB == 4) {      // forward-only epilogues (the pre-LN ViT blocks: residual add / dropout in the epilogue)
    if (e == (HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX);
    else if (e == (HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX | HAMT_EPI_DROPOUT)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX | HAMT_EPI_DROPOUT);
    else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD | HAMT_EPI_DROPOUT)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD | HAMT_EPI_DROPOUT);
    else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU);
    else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU | HAMT_EPI_DROPOUT)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU | HAMT_EPI_DROPOUT);
    else return false;
  } else return false;
#undef HAMT_L
  return true;
}

template <bool A_KM, bool B_KM>
bool launch_256(const GemmArgsF& g, hipStream_t s) {
  const dim3 grid(((g.M + 255) / 256) * ((g.N + 255) / 256));
  const int e = g.epi;
#define HAMT_L(E) do { hipLaunchKernelGGL((gemm_fast256_kernel<E, A_KM, B_KM>), grid, dim3(512), 0, s, g); \
                    hamt_set_last_kernel("gemm_fast256_kernel<%d, %s, %s>", (int)(E), A_KM ? "true" : "false", B_KM ? "true" : "false"); } while (0)
  if (e == 0) HAMT_L(0);
  else if (e == HAMT_EPI_BIAS) HAMT_L(HAMT_EPI_BIAS);
  else if (e == HAMT_EPI_ACCUM) HAMT_L(HAMT_EPI_ACCUM);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD);
  else if (e == HAMT_EPI_MUL_AUX) HAMT_L(HAMT_EPI_MUL_AUX);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX | HAMT_EPI_DROPOUT)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX | HAMT_EPI_DROPOUT);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD | HAMT_EPI_DROPOUT)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD | HAMT_EPI_DROPOUT);
  else return false;
#undef HAMT_L
  return true;
}

template <int BM, bool A_KM, bool B_KM, int NST = 2>
void launch_bm(const GemmArgsF& g, dim3 grid, hipStream_t s) {
  const int e = g.epi;
#define HAMT_L(E) do { hipLaunchKernelGGL((gemm_fast_kernel<BM, E, A_KM, B_KM, NST>), grid, dim3(256), 0, s, g); \
                    hamt_set_last_kernel(NST == 2 ? "gemm_fast_kernel<%d, %d, %s, %s>" : "gemm_fast_kernel<%d, %d, %s, %s, %d>", BM, (int)(E), A_KM ? "true" : "false", B_KM ? "true" : "false", NST); } while (0)
  if (e == 0) HAMT_L(0);
  else if (e == HAMT_EPI_BIAS) HAMT_L(HAMT_EPI_BIAS);
  else if (e == HAMT_EPI_ACCUM) HAMT_L(HAMT_EPI_ACCUM);
  else if constexpr (BM == 32) launch_bm<64, A_KM, B_KM, NST>(g, dim3(((g.M + 63) / 64) * ((g.N + BN - 1) / BN), grid.y), s);   // 32-row tiles: narrow-output epilogues only
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU | HAMT_EPI_SAVE_PRE)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU | HAMT_EPI_SAVE_PRE);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_RELU)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_RELU);
  else if (e == HAMT_EPI_MUL_DGELU) HAMT_L(HAMT_EPI_MUL_DGELU);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD);
  else if (e == HAMT_EPI_MUL_AUX) HAMT_L(HAMT_EPI_MUL_AUX);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX | HAMT_EPI_DROPOUT)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_ADD_AUX | HAMT_EPI_DROPOUT);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD | HAMT_EPI_DROPOUT)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU_GRAD | HAMT_EPI_DROPOUT);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU);
  else if (e == (HAMT_EPI_BIAS | HAMT_EPI_GELU | HAMT_EPI_DROPOUT)) HAMT_L(HAMT_EPI_BIAS | HAMT_EPI_GELU | HAMT_EPI_DROPOUT);
  else HAMT_L(-1);
#undef HAMT_L
}

}  // namespace

// Layouts: NT (forward), NN (dgrad: B = W stored [K][N]), TN (wgrad: A = dY stored [K][M], B = X stored [K][N]).
// K-contiguous operands need K % 64 == 0 (the caller pads with zeros); K-strided operands are clamped to row K-1,
// so for a ragged K the OTHER operand must hold zeros there (NN: zero-padded dY columns; TN: zero-padded dY rows).
bool hamt_gemm_fast_eligible(const hamt_gemm_desc* d, const void* A, const void* B) {
  if (d->prec != HAMT_PREC_BF16 || d->dtype_a != HAMT_BF16 || d->dtype_b != HAMT_BF16) return false;
  if (d->a_kmajor && !d->b_kmajor) return false;
  if (d->K < 64 || d->M < 1 || d->N < 1) return false;
  if ((!d->a_kmajor || !d->b_kmajor) && d->K % 64 != 0) return false;
  if (d->a_kmajor && d->K % 64 != 0) return false;   // TN: the caller pads the reduction rows of A with zeros
  if (d->lda % 8 || d->ldb % 8 || ((uintptr_t)A % 16) || ((uintptr_t)B % 16)) return false;
  {   // the DMA pieces address their operand with 32-bit byte offsets from its base
    const double a_bytes = 2.0 * (d->a_kmajor ? (double)d->K : (double)d->M) * d->lda, b_bytes = 2.0 * (d->b_kmajor ? (double)d->K : (double)d->N) * d->ldb;
    if (a_bytes >= 4294967296.0 || b_bytes >= 4294967296.0) return false;
  }
  if (d->a_kmajor && d->lda < 64) return false;
  if (d->b_kmajor && d->ldb < 128) return false;
  return true;
}

// K groups inside the workgroup (gemm_kg_kernel) instead of plain tiles / split-K: when the whole output is at most one round
// of workgroups of that kernel and the reduction is long enough to amortise the extra LDS pass.  Returns 100 * tile rows +
// 10 * groups + ring depth, or 0.  Measured (tools/ksplit_sweep.py, B = 16 shapes, us): 1280x768x3072 29.4 -> 16.7,
// x2304 17.6 -> 14.5, 1968x768x3072 29.9 -> 22.2, x2304 22.9 -> 17.8; no gain once the plain tiles fill the chip (B = 64).
static int kg_variant(const hamt_gemm_desc* d) {
  static const int force = getenv("HAMT_KG") ? atoi(getenv("HAMT_KG")) : 0;   // 1 = never, > 1 = always this variant
  if (d->a_kmajor || force == 1) return 0;
  if (force) return force;
  const int nk = d->K / BK;
  if (nk < 12 || d->N > 1024) return 0;
  const long tn = (d->N + 127) / 128, t32 = (long)((d->M + 31) / 32) * tn, t64 = (long)((d->M + 63) / 64) * tn;
  // a VERY long reduction over a small output (the MLM decoder's dgrad: 768 x 768 x 30 528 at B = 64): 144 workgroups walking 477
  // k-tiles each took 157 us in the step (profiles/r04_*); split-K over the grid (10 slices of 128-row tiles, ~360 workgroups, the
  // partial tiles are 24 MB) is the better form there -- when the caller can take it (plain / accumulate epilogue, fp32 C)
  if (nk >= 192 && t64 <= 128 && !(d->epilogue & ~HAMT_EPI_ACCUM) && d->dtype_c == HAMT_F32 && d->ldc == d->N) return 0;
  if (t32 <= 256) return 3224;
  if (t64 <= 256 && nk >= 24) return 6432;
  // one 128-row tile per CU, two groups (B = 64 text / vision streams: 5120x768x3072 41.2 -> 35.3 us, 2752x768x2304 27.7 -> 24.2)
  const long t128 = (long)((d->M + 127) / 128) * tn;
  if (t128 <= 256 && nk >= 36) return 12822;
  return 0;
}

// K slices to use for this problem given `ws_bytes` of workspace (1 = no split).
int hamt_gemm_fast_ksplit(const hamt_gemm_desc* d, size_t ws_bytes) {
  if (d->epilogue & ~HAMT_EPI_ACCUM) return 1;
  if (d->dtype_c != HAMT_F32 || d->ldc != d->N) return 1;
  static const int force_bm = getenv("HAMT_FAST_BM") ? atoi(getenv("HAMT_FAST_BM")) : 0;
  static const int force = getenv("HAMT_KS") ? atoi(getenv("HAMT_KS")) : 0;
  if (!force_bm && !force && kg_variant(d)) return 1;
  const long tiles = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
  const int nk = d->K / BK;
  int s;
  if (force) s = force > nk ? nk : force;
  else {
    if (tiles >= 192 || nk < 16) return 1;
    s = (int)(384 / tiles);
    if (s > nk / 8) s = nk / 8;
    if (s > 16) s = 16;
  }
  while (s > 1 && (size_t)s * d->M * d->N * 4 > ws_bytes) --s;
  return s < 2 ? 1 : s;
}

// 256-square tiles (8 waves, one workgroup per CU, operands re-used twice as often per DMA piece and LDS read): when the
// reduction is long, the grid is 0.8 .. 1 tile per CU or at least 3 per CU, and the K-strided operand rows are wide enough.
static bool use256(const hamt_gemm_desc* d, int force_bm) {
  if (d->b_kmajor && d->ldb < 256) return false;
  if (force_bm) return force_bm == 256;
  if (d->K < 2048) return false;   // one workgroup per CU exposes the tile's prologue and store tail: long reductions only
  const long t = (long)((d->M + 255) / 256) * ((d->N + 255) / 256);   // (measured: 5120x3072x768 35.7 us with 128-row tiles, 40 with 256)
  return (t >= 208 && t <= 256) || t >= 768;
}

void hamt_gemm_fast_launch(const hamt_gemm_desc* d, const void* A, const void* B, void* C, const float* bias, void* aux,
                           float* ws, size_t ws_bytes, hipStream_t s) {
  GemmArgsF g{d->M, d->N, d->K, d->lda, d->ldb, d->ldc, d->ldaux, d->dtype_c, d->dtype_aux, d->epilogue, d->alpha,
              (const bf16_t*)A, (const bf16_t*)B, C, bias, aux, 1, nullptr,
              ((d->ka_rows > 0 && d->ka_rows < d->K) ? d->ka_rows : d->K) - 1, ((d->kb_rows > 0 && d->kb_rows < d->K) ? d->kb_rows : d->K) - 1,
              d->p_drop, d->call_id, d->rng};
  const int ks = ws ? hamt_gemm_fast_ksplit(d, ws_bytes) : 1;
  const long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
  static const int force_bm = getenv("HAMT_FAST_BM") ? atoi(getenv("HAMT_FAST_BM")) : 0;
  // Tile height.  The ring is 2 deep (64 / 48 KiB), so 2 workgroups of 128 rows or 3 of 64 rows share a CU and one's
  // prologue / epilogue / DMA issue overlaps another's MFMA loop (measured: 4096^3 894 TFLOP/s vs 692 with a 3-deep
  // ring at one workgroup per CU).  128-row tiles do ~1.4x the flops per LDS byte, 64-row tiles fill the chip when the
  // grid is small: pick the one with the lower (rounds of resident workgroups) x (time per tile) estimate.
  const long t64 = (long)((d->M + 63) / 64) * ((d->N + 127) / 128);
  const long r128 = (t128 * ks + 511) / 512, r64 = (t64 * ks + 767) / 768;
  const bool bm64 = force_bm ? force_bm == 64 : (t128 * ks < 1024 && r64 * 25 < r128 * 36);   // tile-time ratio 1 : 1.44
  g.ksplit = ks;
  g.part = ks > 1 ? ws : nullptr;
  const dim3 g64(((d->M + 63) / 64) * ((d->N + BN - 1) / BN), ks), g128(((d->M + 127) / 128) * ((d->N + BN - 1) / BN), ks);
  // The 128-square tile with the four-deep ring (gemm_q4.hip): narrow outputs whose 128-square tiling is about one round of the chip.
  // HAMT_Q4 = 0 / 1: never / whenever the kernel can run the problem (tuning and tests).
  static const int q4 = getenv("HAMT_Q4") ? atoi(getenv("HAMT_Q4")) : -1;
  if (ks == 1 && q4 != 0 && !force_bm && !d->a_kmajor) {
    // (measured, profiles/r05_q4_probe.txt: equal to the K-group / 64-row tiles on warm operands, 13 - 20 % ahead on cold ones -- the step's
    // case -- at 240 and at 132 tiles; behind them below ~100 tiles, behind the 256-row tiles beyond one round and on N = 1536)
    const bool pick = q4 == 1 || (d->N <= 1024 && d->K >= 192 && t128 >= 120 && t128 <= 256);
    if (pick && hamt_gemm_q4_launch(g, d->b_kmajor != 0, s)) return;
  }
  // 256-square two-phase kernel (one 8-wave workgroup per CU): by estimated time.  Measured on MI355X (tools/gemm_sweep.py,
  // tools/p8_probe.py): a p8 tile costs ~6 us + 1.55 us per k-tile whatever the grid, the 64/128-row tiles run the
  // step's shapes at ~620 TFLOP/s.  HAMT_P8=0 / 1 = never / whenever eligible.
  static const int p8 = getenv("HAMT_P8") ? atoi(getenv("HAMT_P8")) : -1;
  if (ks == 1 && p8 != 0 && !force_bm && d->K >= 128 && !d->a_kmajor && (!d->b_kmajor || d->ldb >= 256)) {
    const long t256 = (long)((d->M + 255) / 256) * ((d->N + 255) / 256), t192 = (long)((d->M + 255) / 256) * ((d->N + 191) / 192);
    const double t_p8 = (double)((t256 + 255) / 256) * (6.0 + 1.55 * (d->K / BK));
    // 256 x 192 tiles: 3/4 of the MFMA work and 7/8 of the operand bytes of a 256-square tile per k-tile -- but measured 0.92-0.94 of
    // its TIME at K = 768 .. 3072 (tools/gemm_bench.py with HAMT_P8_BN = 192 / 256: a tile's life is prologue, barriers and the
    // store tail as much as MFMA issue): worth it only where it does not add a round -- 5120 x 2304 x 768: 180 tiles -> 240, one
    // round either way, 27.6 -> 26.0 us; 11520 x 768 x 3072: 65.2 -> 59.8; 11520 x 768 x 2304 (NN, acc): 55.3 -> 50.9.  The FFN-1
    // epilogue (gelu + gelu', two stores per tile) measured 4 % SLOWER with the narrow tile at 720 tiles: kept on 256 columns there.
    static const int bn_force = getenv("HAMT_P8_BN") ? atoi(getenv("HAMT_P8_BN")) : 0;      // 192 / 256: tuning and tests
    const double t_p8n = (double)((t192 + 255) / 256) * (6.0 + 1.44 * (d->K / BK));
    const bool n192 = bn_force ? bn_force == 192 : (t_p8n < 0.97 * t_p8 && !((d->epilogue & HAMT_EPI_GELU_GRAD) && t192 > 512));
    const double t_small = 2.0 * d->M * d->N * d->K / 620e6;
    if (p8 == 1 || (n192 ? t_p8n : t_p8) < 0.95 * t_small) {
      bool ok = false;
      if (n192) ok = !d->b_kmajor ? launch_p8<false, false, 3>(g, s) : launch_p8<false, true, 3>(g, s);
      if (!ok) ok = !d->b_kmajor ? launch_p8<false, false>(g, s) : launch_p8<false, true>(g, s);
      if (ok) return;
    }
  }
  if (ks == 1 && !force_bm) {
    const int kg = kg_variant(d);
    if (kg) {
      const int bm = kg / 100, G = (kg / 10) % 10, D = kg % 10;
      const dim3 grid(((d->M + bm - 1) / bm) * ((d->N + BN - 1) / BN));
#define HAMT_KGL(BM_, G_, D_) do { if (d->b_kmajor) hipLaunchKernelGGL((gemm_kg_kernel<BM_, G_, D_, true>), grid, dim3(256 * G_), 0, s, g); \
                               else hipLaunchKernelGGL((gemm_kg_kernel<BM_, G_, D_, false>), grid, dim3(256 * G_), 0, s, g); \
                               hamt_set_last_kernel("gemm_kg_kernel<%d, %d, %d, %s>", BM_, G_, D_, d->b_kmajor ? "true" : "false"); } while (0)
      if (bm == 32 && G == 2 && D == 3) HAMT_KGL(32, 2, 3); else if (bm == 32 && G == 3) HAMT_KGL(32, 3, 2);
      else if (bm == 32) HAMT_KGL(32, 2, 4); else if (bm == 128) HAMT_KGL(128, 2, 2); else if (G == 2) HAMT_KGL(64, 2, 3); else HAMT_KGL(64, 3, 2);
#undef HAMT_KGL
      return;
    }
  }
  if (ks == 1 && use256(d, force_bm)) {
    const bool ok = !d->a_kmajor ? (!d->b_kmajor ? launch_256<false, false>(g, s) : launch_256<false, true>(g, s)) : false;
    if (ok) return;
  }
  // 32-row tiles: for grids so small (the B = 16 shapes: 1280 x 768 outputs are 120 tiles of 64 rows) that 64-row tiles leave every
  // SIMD with at most one wave, which then issues its DMA pieces, reads its fragments and multiplies strictly in turn
  const long t32 = (long)((d->M + 31) / 32) * ((d->N + 127) / 128);
  const bool bm32 = force_bm ? force_bm == 32 : (bm64 && ks == 1 && t64 <= 160 && d->K <= 1024);
  const dim3 g32(((d->M + 31) / 32) * ((d->N + BN - 1) / BN), ks);
  (void)t32;
#define HAMT_BMST(AK, BK_) do { \
    if (bm32) launch_bm<32, AK, BK_>(g, g32, s); else if (bm64) launch_bm<64, AK, BK_>(g, g64, s); else launch_bm<128, AK, BK_>(g, g128, s); } while (0)
  if (!d->a_kmajor && !d->b_kmajor) HAMT_BMST(false, false);
  else if (!d->a_kmajor) HAMT_BMST(false, true);
#undef HAMT_BMST
  else { if (bm64 && d->lda >= 64) launch_bm<64, true, true>(g, g64, s); else launch_bm<128, true, true>(g, g128, s); }
  if (ks > 1) hamt_reduce_partials(ks, d->M * d->N, ws, (float*)C, (d->epilogue & HAMT_EPI_ACCUM) ? 1 : 0, s);
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

static inline int kv_of(const hamt_wgrad_desc& d) { return (d.K_valid > 0 && d.K_valid < d.K) ? d.K_valid : d.K; }
static inline int keff(const hamt_wgrad_desc& d) { return (kv_of(d) + 63) / 64 * 64; }   // reduction rows actually multiplied
static inline int kv2_of(const hamt_wgrad_desc& d) { return !d.dy2 ? 0 : ((d.K2_valid > 0 && d.K2_valid < d.K2) ? d.K2_valid : d.K2); }
static inline int keff2(const hamt_wgrad_desc& d) { return (kv2_of(d) + 63) / 64 * 64; }  // ... of the second pair of operands (0: none)
static inline int kall(const hamt_wgrad_desc& d) { return keff(d) + keff2(d); }

// Measurement aid for bench.py's `roofline` object: per-launch durations of the grouped weight-gradient kernels, taken with HIP
// events recorded on the launch stream right around each kernel (not around the whole call: the table writes and the other tile
// class are separate kernels in rocprofv3's summary too).  Eager launches only -- events cannot be read back from a capture.
struct WgradTimed { hipEvent_t a, b; int bm; double flops; };
static bool wgrad_timing_on = false;
static std::vector<WgradTimed> wgrad_timing_ev;
extern "C" int hamt_debug_wgrad_timing(int on) {
  for (auto& t : wgrad_timing_ev) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
  wgrad_timing_ev.clear();
  wgrad_timing_on = on != 0;
  return HAMT_OK;
}
// us[i], tile_rows[i], flops[i] (2 M N K over the launch's table, K = the k-tiles actually multiplied) of the launches since hamt_debug_wgrad_timing(1) (waits for them); returns their number (<= cap are written)
extern "C" int hamt_debug_wgrad_times(float* us, int* tile_rows, double* flops, int cap) {
  int n = 0;
  for (auto& t : wgrad_timing_ev) {
    float ms = 0.f;
    if (hipEventSynchronize(t.b) != hipSuccess || hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) return -1;
    if (n < cap) { if (us) us[n] = ms * 1e3f; if (tile_rows) tile_rows[n] = t.bm; if (flops) flops[n] = t.flops; }
    ++n;
  }
  return n;
}

// phase: 3 = write the launch table and launch (hamt_wgrad_grouped); 1 = write the table only; 2 = launch only (the table was written
// by an earlier phase-1 call with the SAME problems: the host part -- classes, units, XCD queues, grid -- is recomputed, identical)
static int wgrad_grouped_impl(int n, const hamt_wgrad_desc* probs, void* table, size_t table_bytes, int phase, void* stream) {
  HAMT_CHECK_ARG(n >= 0 && (n == 0 || probs) && phase >= 1 && phase <= 3, "hamt_wgrad_grouped: bad argument");
  std::vector<int> order;
  for (int i = 0; i < n; ++i) {
    const hamt_wgrad_desc& d = probs[i];
    HAMT_CHECK_ARG(d.dy && d.x && d.dw && d.M >= 1 && d.N >= 1, "hamt_wgrad_grouped: problem %d: null pointer or empty output", i);
    HAMT_CHECK_ARG(d.K >= 0 && d.K % 64 == 0, "hamt_wgrad_grouped: problem %d: K = %d is not a multiple of 64", i, d.K);
    HAMT_CHECK_ARG(d.ldy % 8 == 0 && d.ldy >= 64 && d.ldy >= d.M && d.ldx % 8 == 0 && d.ldx >= 128 && d.ldx >= d.N && d.ldw >= d.N,
                   "hamt_wgrad_grouped: problem %d: bad leading dimension (ldy %d, ldx %d, ldw %d)", i, d.ldy, d.ldx, d.ldw);
    HAMT_CHECK_ARG((uintptr_t)d.dy % 16 == 0 && (uintptr_t)d.x % 16 == 0, "hamt_wgrad_grouped: problem %d: operands must be 16-byte aligned", i);
    HAMT_CHECK_ARG(2.0 * d.K * d.ldy < 4294967296.0 && 2.0 * d.K * d.ldx < 4294967296.0, "hamt_wgrad_grouped: problem %d: operands must be smaller than 4 GiB (32-bit DMA offsets)", i);
    HAMT_CHECK_ARG(d.K_valid >= 0 && d.K_valid <= d.K, "hamt_wgrad_grouped: problem %d: K_valid = %d outside [0, K = %d]", i, d.K_valid, d.K);
    HAMT_CHECK_ARG(d.wire_scale == 0.f || (!d.accum_dw && d.K > 0 && d.ldw % 8 == 0 && (uintptr_t)d.dw % 16 == 0),
                   "hamt_wgrad_grouped: problem %d: a bf16 wire output (wire_scale != 0) is store-only and needs ldw %% 8 == 0 and a 16-byte aligned dw", i);
    if (d.dy2) {
      HAMT_CHECK_ARG(d.x2 && d.K > 0 && d.K2 > 0 && d.K2 % 64 == 0 && d.wire_scale == 0.f && (d.K_valid == 0 || d.K_valid == d.K),
                     "hamt_wgrad_grouped: problem %d: a second operand pair needs x2, K and K2 > 0, K2 %% 64 == 0, K_valid 0 or K, wire_scale 0", i);
      HAMT_CHECK_ARG(d.ldy2 % 8 == 0 && d.ldy2 >= 64 && d.ldy2 >= d.M && d.ldx2 % 8 == 0 && d.ldx2 >= 128 && d.ldx2 >= d.N && (uintptr_t)d.dy2 % 16 == 0 &&
                     (uintptr_t)d.x2 % 16 == 0 && d.K2_valid >= 0 && d.K2_valid <= d.K2 && 2.0 * d.K2 * d.ldy2 < 4294967296.0 && 2.0 * d.K2 * d.ldx2 < 4294967296.0,
                     "hamt_wgrad_grouped: problem %d: bad second operand pair (ldy2 %d, ldx2 %d, K2 %d, K2_valid %d)", i, d.ldy2, d.ldx2, d.K2, d.K2_valid);
    }
    if (d.K > 0) order.push_back(i);
    else HAMT_CHECK_ARG(d.accum_dw && (!d.db || d.accum_db), "hamt_wgrad_grouped: problem %d: K = 0 with store semantics", i);
  }
  if (order.empty()) return HAMT_OK;
  HAMT_CHECK_ARG(table && (uintptr_t)table % 16 == 0, "hamt_wgrad_grouped: table must be 16-byte aligned device memory");
  hipStream_t s = as_stream(stream);
  // Two launch classes: problems whose operand rows are >= 256 elements wide can use 256-square tiles; the rest 128 / 64
  // rows.  Within a class: longest reductions first, the short tail tiles fill in behind them.
  std::vector<int> cls[2];
  for (int i : order) {
    const hamt_wgrad_desc& d = probs[i];
    const bool wide2 = !d.dy2 || (d.ldy2 >= 256 && d.ldx2 >= 256 && keff2(d) >= 128);
    cls[(d.ldy >= 256 && d.ldx >= 256 && keff(d) >= 128 && wide2) ? 0 : 1].push_back(i);
  }
  static const int use_p8 = getenv("HAMT_WGRAD_P8") ? atoi(getenv("HAMT_WGRAD_P8")) : 1;   // 0: the one-phase 256-square tile
  const char* uenv = getenv("HAMT_WGRAD_UNIT_TILES");   // tuning: tiles per XCD-pinned unit (read per call)
  const int unit_tiles = uenv && atoi(uenv) > 0 ? atoi(uenv) : 12;
  const char* u2env = getenv("HAMT_WGRAD_UNIT_2D");
  const int unit_2d = u2env ? atoi(u2env) : 1;
  const char* fenv = getenv("HAMT_WGRAD_TILE");   // test / tuning override: 256, 128 or 64 (read per call)
  const int force = fenv ? atoi(fenv) : 0;
  {
    long t256 = 0;
    for (int i : cls[0]) t256 += (long)((probs[i].M + 255) / 256) * ((probs[i].N + 255) / 256);
    // too few 256-square tiles to occupy half the chip: use the smaller tiles for everything
    const char* menv = getenv("HAMT_WGRAD_MIN256");     // tuning: fewest 256-square tiles for which the 256-square launch is used (read per call)
    // (fewer than one tile per CU: the 64-row tiles fill the chip better -- the second use of the shared cross-attention weights in an ITM
    // step, 144 tiles of 256 x 256 with reductions of up to 25 600 rows: 785 us against 496 with 64 x 128 tiles; 128 was the threshold before)
    const long min256 = menv && atoi(menv) > 0 ? atoi(menv) : 256;
    if (force == 128 || force == 64 || (t256 < min256 && force != 256)) { cls[1].insert(cls[1].end(), cls[0].begin(), cls[0].end()); cls[0].clear(); }
  }
  WgradProb* tab = (WgradProb*)table;
  int off = 0;
  for (int c = 0; c < 2; ++c) {
    std::vector<int>& v = cls[c];
    if (v.empty()) continue;
    std::stable_sort(v.begin(), v.end(), [&](int a, int b) { return kall(probs[a]) > kall(probs[b]); });
    long t128 = 0, t256 = 0;
    for (int i : v) {
      t128 += (long)((probs[i].M + 127) / 128) * ((probs[i].N + 127) / 128);
      t256 += (long)((probs[i].M + 255) / 256) * ((probs[i].N + 255) / 256);
    }
    // 256-square tiles (one 8-wave workgroup per CU) when they still give every CU >= 3 tiles; else 128 / 64 rows
    const char* m128 = getenv("HAMT_WGRAD_MIN128");     // tuning: fewest 128-square tiles for which 128-row tiles are used (read per call)
    int bm = c == 0 ? 256 : (t128 >= (m128 && atoi(m128) > 0 ? atoi(m128) : 1024) ? 128 : 64);
    if (force == 128 || force == 64) bm = force;
    const int bn = bm == 256 ? 256 : 128;
    // units: a problem, or rectangles of r x c tiles of a problem with many tiles (so that one XCD's share stays ~<= 12 tiles).  A unit's
    // tiles read r + c operand panels between them: the rectangle is the one that covers the problem with the fewest panel reads
    // (3 x 4 for the 3 x 12 tiles of an FFN-2 weight, where a row band reads 13 panels for its 12 tiles).
    struct Unit { int prob, m_lo, m_rows, n_lo, n_cols, tiles; double cost; };
    std::vector<Unit> units;
    for (int i : v) {
      const hamt_wgrad_desc& d = probs[i];
      const int tm = (d.M + bm - 1) / bm, tn = (d.N + bn - 1) / bn;
      int ur = 1, uc = 1;
      long best = -1;
      for (int r = 1; r <= tm; ++r)
        for (int c = 1; c <= tn; ++c) {
          if (r * c > unit_tiles && !(r == 1 && c == tn)) continue;                                  // (a single tile row is always allowed)
          if ((long)((tm + r - 1) / r) * ((tn + c - 1) / c) > (d.M + 63) / 64) continue;             // the caller's table: one entry per 64 output rows
          long panels = 0;                         // sum over the covering rectangles of (rows + columns)
          for (int r0 = 0; r0 < tm; r0 += r)
            for (int c0 = 0; c0 < tn; c0 += c) panels += std::min(r, tm - r0) + std::min(c, tn - c0);
          if (best < 0 || panels < best || (panels == best && r * c > ur * uc)) { best = panels; ur = r; uc = c; }
        }
      if (unit_2d == 0) { uc = tn; ur = tn >= unit_tiles ? 1 : unit_tiles / tn; }   // (row bands only: the round-2 placement, for measurements)
      for (int r0 = 0; r0 < tm; r0 += ur)
        for (int c0 = 0; c0 < tn; c0 += uc) {
          const int rows = std::min(ur * bm, d.M - r0 * bm), cols = std::min(uc * bn, d.N - c0 * bn);
          const int t = ((rows + bm - 1) / bm) * ((cols + bn - 1) / bn);
          units.push_back(Unit{i, r0 * bm, rows, c0 * bn, cols, t, (double)t * kall(d)});
        }
    }
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.cost > b.cost; });
    std::vector<int> xq[8];
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int u = 0; u < (int)units.size(); ++u) {           // longest processing time first onto the least loaded XCD
      int best = 0;
      for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
      xq[best].push_back(u);
      load[best] += units[u].cost;
    }
    const int cn = (int)units.size();
    HAMT_CHECK_ARG((size_t)(off + cn) * sizeof(WgradProb) <= table_bytes, "hamt_wgrad_grouped: table too small (%zu bytes needed)",
                   (size_t)(off + cn) * sizeof(WgradProb));
    WgradXcd xs;
    std::vector<WgradProb> flat;
    flat.reserve(cn);
    int max_tiles = 0;
    double launch_flops = 0.0;
    for (int x = 0; x < 8; ++x) {
      xs.start[x] = (int)flat.size();
      std::stable_sort(xq[x].begin(), xq[x].end(), [&](int a, int b) { return kall(probs[units[a].prob]) > kall(probs[units[b].prob]); });
      int tiles = 0;
      for (int u : xq[x]) {
        const Unit& un = units[u];
        const hamt_wgrad_desc& d = probs[un.prob];
        tiles += un.tiles;
        const bool wire = d.wire_scale != 0.f;     // bf16 output: the band's first row is m_lo * ldw ELEMENTS of 2 bytes further
        const size_t w_off = (size_t)un.m_lo * d.ldw + un.n_lo;
        float* dw_band = wire ? (float*)((bf16_t*)d.dw + w_off) : d.dw + w_off;
        const int ss_ld = (d.N + 127) >> 7;        // (units start on multiples of 64 rows and 128 columns)
        flat.push_back(WgradProb{(const bf16_t*)d.dy + un.m_lo, (const bf16_t*)d.x + un.n_lo, dw_band,
                                 (d.db && un.n_lo == 0) ? d.db + un.m_lo : nullptr,       // the column sums of dY: by the unit that holds column 0
                                 (d.ss && !wire) ? d.ss + (size_t)(un.m_lo >> 6) * ss_ld + (un.n_lo >> 7) : nullptr,
                                 un.m_rows, un.n_cols, d.K, d.ldy, d.ldx, d.ldw,
                                 (d.accum_dw ? 1 : 0) | (d.accum_db ? 2 : 0) | (ss_ld << 8), tiles, kv_of(d), d.wire_scale,
                                 d.dy2 ? (const bf16_t*)d.dy2 + un.m_lo : nullptr, d.dy2 ? (const bf16_t*)d.x2 + un.n_lo : nullptr,
                                 keff2(d), d.ldy2, d.ldx2, kv2_of(d)});
        flat.back().K = keff(d);                        // whole k-tiles behind the last valid row are not multiplied at all
        launch_flops += 2.0 * un.m_rows * un.n_cols * kall(d);
      }
      max_tiles = std::max(max_tiles, tiles);
    }
    xs.start[8] = (int)flat.size();
    for (int b0 = 0; b0 < cn; b0 += WG_MAX) {
      WgradChunk ch;
      const int cnt = std::min(WG_MAX, cn - b0);
      if (!(phase & 1)) break;
      for (int i = 0; i < cnt; ++i) ch.p[i] = flat[b0 + i];
      hipLaunchKernelGGL(wgrad_table_write_kernel, dim3(1), dim3(WG_MAX), 0, s, ch, tab, off + b0, cnt);
    }
    if (!(phase & 2)) { HAMT_CHECK_LAUNCH("hamt_wgrad_grouped (table)"); off += cn; continue; }
    const dim3 grid(8 * max_tiles);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (wgrad_timing_on) {     // measurement aid (hamt_debug_wgrad_timing): HIP events around the kernel, on its own stream
      hipEventCreate(&ev0); hipEventCreate(&ev1);
      hipEventRecord(ev0, s);
    }
    if (bm == 256 && use_p8) hipLaunchKernelGGL(wgrad_grouped_p8_kernel, grid, dim3(512), 0, s, tab + off, xs);
    else if (bm == 256) hipLaunchKernelGGL((wgrad_grouped_kernel<256, 256, 2, 4>), grid, dim3(512), 0, s, tab + off, xs);
    else if (bm == 128) hipLaunchKernelGGL((wgrad_grouped_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, tab + off, xs);
    else hipLaunchKernelGGL((wgrad_grouped_kernel<64, 128, 2, 2>), grid, dim3(256), 0, s, tab + off, xs);
    if (ev0) { hipEventRecord(ev1, s); wgrad_timing_ev.push_back(WgradTimed{ev0, ev1, bm, launch_flops}); }
    HAMT_CHECK_LAUNCH("hamt_wgrad_grouped");
    off += cn;
  }
  return HAMT_OK;
}
extern "C" int hamt_wgrad_grouped(int n, const hamt_wgrad_desc* probs, void* table, size_t table_bytes, void* stream) {
  return wgrad_grouped_impl(n, probs, table, table_bytes, 3, stream);
}
extern "C" int hamt_wgrad_grouped_ex(int n, const hamt_wgrad_desc* probs, void* table, size_t table_bytes, int phase, void* stream) {
  return wgrad_grouped_impl(n, probs, table, table_bytes, phase, stream);
}
