// Kernel argument block of the bf16 fast-path GEMM kernels (gemm_fast.hip, gemm_q4.hip).
#pragma once
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
  int ksplit;      // number of K slices (grid.y); > 1 => raw fp32 partials to `part`
  float* part;
  int ka_max, kb_max;   // last valid reduction row of a K-strided A / B (rows beyond are clamped to it)
  float p_drop;         // HAMT_EPI_DROPOUT
  uint32_t call_id;
  const uint64_t* rng;
  float* ss;            // weight-gradient tiles: slot array for the sum of squares of each tile's FINAL values (or nullptr)
  int ss_ld;            // slots per 64-row band of `ss` (0: those of this N; a column band of a wider output passes the full width's)
  // weight-gradient tiles only: a second pair of K-strided operands reduced into the same tile behind the first (K2 = 0: none)
  const bf16_t* A2;
  const bf16_t* B2;
  int K2, lda2, ldb2, k2_max;
};
