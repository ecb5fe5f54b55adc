// Epilogue of the bf16 GEMM kernels (gemm_fast.hip, gemm_q4.hip): the kernel argument block, the general per-piece epilogue (epi_store),
// the interior-tile epilogue (epi_fast8) and the XCD-aware tile order.  Include inside an anonymous namespace after common.h / gemm_frag.h;
// `struct GemmArgsF` must be visible (gemm_args.h).
#pragma once

// One row segment of W (4 or 8) consecutive columns: v = alpha * acc, then the epilogue flags, then the store.
// Vector paths need the segment inside the row (full) and W-element alignment of the row stride and base.
template <int W> __device__ __forceinline__ void ld_bf(const bf16_t* p, float* o) {   // W bf16 -> fp32
  if constexpr (W == 8) {
    const uint4 u = *(const uint4*)p;
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { o[2 * q] = __uint_as_float(w[q] << 16); o[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
  } else {
    const uint2 u = *(const uint2*)p;
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
  }
}
template <int W> __device__ __forceinline__ void st_bf(bf16_t* p, const float* v) {   // W fp32 -> bf16, one store
  if constexpr (W == 8) *(uint4*)p = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
  else *(uint2*)p = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}
// the same for a 2-byte C that is either bf16 or IEEE half (`f16`: wave uniform).  Both conversions are computed and one is selected
// per pair (a v_cndmask on a scalar condition) -- a branch around each format kept both sets of temporaries alive across the general
// epilogue and put 528 bytes of it into scratch (tests/test_kernel_resources.py)
__device__ __forceinline__ uint32_t pack_16(float lo, float hi, bool f16) { return f16 ? pack_h2(lo, hi) : pack_bf2(lo, hi); }
__device__ __forceinline__ void unpack_16(uint32_t w, float& lo, float& hi, bool f16) {
  float a, b;
  unpack_h2(w, a, b);
  lo = f16 ? a : __uint_as_float(w << 16);
  hi = f16 ? b : __uint_as_float(w & 0xffff0000u);
}
template <int W> __device__ __forceinline__ void ld_16(const bf16_t* p, float* o, bool f16) {
  if constexpr (W == 8) {
    const uint4 u = *(const uint4*)p;
    unpack_16(u.x, o[0], o[1], f16); unpack_16(u.y, o[2], o[3], f16); unpack_16(u.z, o[4], o[5], f16); unpack_16(u.w, o[6], o[7], f16);
  } else {
    const uint2 u = *(const uint2*)p;
    unpack_16(u.x, o[0], o[1], f16); unpack_16(u.y, o[2], o[3], f16);
  }
}
template <int W> __device__ __forceinline__ void st_16(bf16_t* p, const float* v, bool f16) {
  if constexpr (W == 8) *(uint4*)p = make_uint4(pack_16(v[0], v[1], f16), pack_16(v[2], v[3], f16), pack_16(v[4], v[5], f16), pack_16(v[6], v[7], f16));
  else *(uint2*)p = make_uint2(pack_16(v[0], v[1], f16), pack_16(v[2], v[3], f16));
}
template <int W> __device__ __forceinline__ void ld_f(const float* p, float* o) {
#pragma unroll
  for (int q = 0; q < W / 4; ++q) { const float4 f = *(const float4*)(p + 4 * q); o[4 * q] = f.x; o[4 * q + 1] = f.y; o[4 * q + 2] = f.z; o[4 * q + 3] = f.w; }
}
template <int W> __device__ __forceinline__ void st_f(float* p, const float* v) {
#pragma unroll
  for (int q = 0; q < W / 4; ++q) *(float4*)(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// the 8 aux values (bf16: 16 bytes; HAMT_U8G: 8 bytes in .x / .y) of row piece (row, col .. col + 7), for the epilogues that request
// their aux reads ahead of the accumulator transpose (wave-uniform dtype branch)
__device__ __forceinline__ uint4 ld_aux8(const GemmArgsF& g, int row, int col) {
  if (g.dtype_aux == HAMT_U8G) { const uint2 u = *(const uint2*)((const uint8_t*)g.aux + (size_t)row * g.ldaux + col); return make_uint4(u.x, u.y, 0u, 0u); }
  return *(const uint4*)((const bf16_t*)g.aux + (size_t)row * g.ldaux + col);
}

template <int EPI, int W>
__device__ __forceinline__ void epi_store(const GemmArgsF& g, int row, int col, const float* acc, float* ssq = nullptr,
                                          const uint4* pre_aux = nullptr, const f32x4* pre_c = nullptr, bool pre_ok = false) {
  // pre_aux / pre_c (W = 8, pre_ok): this row piece's 8 bf16 of aux / 8 fp32 of C, loaded by the caller ahead of its LDS transpose
  if (row >= g.M || col >= g.N) return;
  const int epi = EPI >= 0 ? EPI : (EPI == -2 ? (g.epi & HAMT_EPI_ACCUM) : g.epi);   // -2: plain store or C += only
  float v[W];
#pragma unroll
  for (int j = 0; j < W; ++j) v[j] = acc[j] * g.alpha;
  const bool full = col + W <= g.N;
  const size_t ia = (size_t)row * g.ldaux + col, ic = (size_t)row * g.ldc + col;
  const bool aux16 = g.dtype_aux == HAMT_BF16, aux8 = g.dtype_aux == HAMT_U8G;   // (aux8: GELU_GRAD / MUL_AUX only, checked by hamt_gemm)
  // vector access to aux / C: whole segment in range, row stride and base aligned to the vector
  const bool vaux = full && g.aux && (g.ldaux % W) == 0 && ((uintptr_t)g.aux % 16) == 0;
  const bool vc = full && (g.ldc % W) == 0 && ((uintptr_t)g.C % 16) == 0;
  if (epi & HAMT_EPI_BIAS) {
    if (full && ((uintptr_t)g.bias % 16) == 0) { float b[W]; ld_f<W>(g.bias + col, b); _Pragma("unroll") for (int j = 0; j < W; ++j) v[j] += b[j]; }
    else _Pragma("unroll") for (int j = 0; j < W; ++j) if (col + j < g.N) v[j] += g.bias[col + j];
  }
  if (epi & HAMT_EPI_SAVE_PRE) {
    if (vaux) { if (aux16) st_bf<W>((bf16_t*)g.aux + ia, v); else st_f<W>((float*)g.aux + ia, v); }
    else _Pragma("unroll") for (int j = 0; j < W; ++j) if (col + j < g.N) { if (aux16) ((bf16_t*)g.aux)[ia + j] = f2bf(v[j]); else ((float*)g.aux)[ia + j] = v[j]; }
  }
  if (epi & HAMT_EPI_GELU) { _Pragma("unroll") for (int j = 0; j < W; ++j) v[j] = gelu_erf(v[j]); }
  float keep[W];
  if (epi & HAMT_EPI_DROPOUT) {   // col % 4 == 0: W / 4 groups of the row's mask stream
    const RngKey key = rng_key(g.rng, g.call_id);
    const uint32_t rowh = hamt_mix32((uint32_t)row ^ key.k0);
    const float inv_keep = 1.0f / (1.0f - g.p_drop);
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      float f[4];
      drop_scale4(key, rowh, (uint32_t)(col >> 2) + q, g.p_drop, inv_keep, f);
      _Pragma("unroll") for (int j = 0; j < 4; ++j) keep[q * 4 + j] = f[j];
    }
  }
  if (epi & HAMT_EPI_GELU_GRAD) {
    float dg[W];
    _Pragma("unroll") for (int j = 0; j < W; ++j) gelu_and_grad(v[j], v[j], dg[j]);
    if (epi & HAMT_EPI_DROPOUT) _Pragma("unroll") for (int j = 0; j < W; ++j) dg[j] *= keep[j];
    if (aux8) {
      uint8_t* a8 = (uint8_t*)g.aux + ia;
      if (vaux) { if constexpr (W == 8) *(uint2*)a8 = make_uint2(g8_pack4(dg), g8_pack4(dg + 4)); else *(uint32_t*)a8 = g8_pack4(dg); }
      else {
        const uint32_t w0 = g8_pack4(dg), w1 = W == 8 ? g8_pack4(dg + (W == 8 ? 4 : 0)) : 0u;
        _Pragma("unroll") for (int j = 0; j < W; ++j) if (col + j < g.N) a8[j] = (uint8_t)(((j < 4 ? w0 : w1) >> (8 * (j & 3))) & 0xffu);
      }
    } else if (vaux) { if (aux16) st_bf<W>((bf16_t*)g.aux + ia, dg); else st_f<W>((float*)g.aux + ia, dg); }
    else _Pragma("unroll") for (int j = 0; j < W; ++j) if (col + j < g.N) { if (aux16) ((bf16_t*)g.aux)[ia + j] = f2bf(dg[j]); else ((float*)g.aux)[ia + j] = dg[j]; }
  }
  if (epi & (HAMT_EPI_MUL_AUX | HAMT_EPI_MUL_DGELU | HAMT_EPI_MUL_DRELU)) {
    float h[W];
    if (W == 8 && pre_aux && pre_ok) {
      if (aux8) { g8_unpack4(pre_aux->x, h); g8_unpack4(pre_aux->y, h + (W == 8 ? 4 : 0)); }
      else {
        const uint32_t w4[4] = {pre_aux->x, pre_aux->y, pre_aux->z, pre_aux->w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[2 * q] = __uint_as_float(w4[q] << 16); h[2 * q + 1] = __uint_as_float(w4[q] & 0xffff0000u); }
      }
    } else if (aux8) {
      const uint8_t* a8 = (const uint8_t*)g.aux + ia;
      if (vaux) {
        if constexpr (W == 8) { const uint2 u = *(const uint2*)a8; g8_unpack4(u.x, h); g8_unpack4(u.y, h + (W == 8 ? 4 : 0)); }
        else g8_unpack4(*(const uint32_t*)a8, h);
      } else _Pragma("unroll") for (int j = 0; j < W; ++j) h[j] = (col + j < g.N) ? __builtin_fmaf((float)a8[j], 0.005f, -0.13f) : 0.f;
    } else if (vaux) { if (aux16) ld_bf<W>((const bf16_t*)g.aux + ia, h); else ld_f<W>((const float*)g.aux + ia, h); }
    else _Pragma("unroll") for (int j = 0; j < W; ++j) h[j] = (col + j < g.N) ? (aux16 ? bf2f(((const bf16_t*)g.aux)[ia + j]) : ((const float*)g.aux)[ia + j]) : 0.f;
    _Pragma("unroll") for (int j = 0; j < W; ++j)
      v[j] *= (epi & HAMT_EPI_MUL_AUX) ? h[j] : ((epi & HAMT_EPI_MUL_DGELU) ? dgelu_erf(h[j]) : (h[j] > 0.0f ? 1.0f : 0.0f));
  }
  if (epi & HAMT_EPI_RELU) { _Pragma("unroll") for (int j = 0; j < W; ++j) v[j] = fmaxf(v[j], 0.0f); }
  if (epi & HAMT_EPI_DROPOUT) { _Pragma("unroll") for (int j = 0; j < W; ++j) v[j] *= keep[j]; }
  if (epi & HAMT_EPI_ADD_AUX) {   // residual add
    float h[W];
    if (vaux) { if (aux16) ld_bf<W>((const bf16_t*)g.aux + ia, h); else ld_f<W>((const float*)g.aux + ia, h); }
    else _Pragma("unroll") for (int j = 0; j < W; ++j) h[j] = (col + j < g.N) ? (aux16 ? bf2f(((const bf16_t*)g.aux)[ia + j]) : ((const float*)g.aux)[ia + j]) : 0.f;
    _Pragma("unroll") for (int j = 0; j < W; ++j) v[j] += h[j];
  }
  if (g.dtype_c != HAMT_F32) {      // two bytes per element: bf16, or IEEE half (HAMT_F16)
    // (half only behind a plain / bias / accumulate epilogue -- what hamt_gemm admits: the dense layer in front of a LayerNorm.  In the
    // instantiations with the long epilogues the extra selects pushed a 32-fold unrolled loop over clang's pragma-unroll budget: the
    // accumulators then went to scratch, tests/test_kernel_resources.py)
    constexpr bool F16_OK = EPI < 0 || (EPI & ~(HAMT_EPI_BIAS | HAMT_EPI_ACCUM)) == 0;
    const bool ch = F16_OK && g.dtype_c == HAMT_F16;
    bf16_t* c = (bf16_t*)g.C + ic;
    if (vc) {
      if (epi & HAMT_EPI_ACCUM) { float p[W]; ld_16<W>(c, p, ch); _Pragma("unroll") for (int j = 0; j < W; ++j) v[j] += p[j]; }
      st_16<W>(c, v, ch);
    } else {
      _Pragma("unroll") for (int j = 0; j < W; ++j) if (col + j < g.N) {       // (unrolled: a rolled loop indexes v[] dynamically, i.e. through scratch)
        const float f = (epi & HAMT_EPI_ACCUM) ? v[j] + (ch ? h2f(c[j]) : bf2f(c[j])) : v[j];
        c[j] = ch ? f2h(f) : f2bf(f);
      }
    }
  } else {
    float* c = (float*)g.C + ic;
    if (vc) {
      if (epi & HAMT_EPI_ACCUM) {
        float p[W];
        if (W == 8 && pre_c && pre_ok) { _Pragma("unroll") for (int j = 0; j < 4; ++j) { p[j] = pre_c[0][j]; p[4 + j] = pre_c[1][j]; } }
        else ld_f<W>(c, p);
        _Pragma("unroll") for (int j = 0; j < W; ++j) v[j] += p[j];
      }
      st_f<W>(c, v);
      if (ssq) _Pragma("unroll") for (int j = 0; j < W; ++j) *ssq += v[j] * v[j];
    } else _Pragma("unroll") for (int j = 0; j < W; ++j) if (col + j < g.N) {
      const float f = (epi & HAMT_EPI_ACCUM) ? c[j] + v[j] : v[j];
      c[j] = f;
      if (ssq) *ssq += f * f;
    }
  }
}

// ---- the epilogue of an INTERIOR tile.  epi_store above is general (ragged rows / columns, any alignment, every flag, two or three
// dtypes per operand) and hipcc compiles it to ~50 scalar branches per 8-column row piece -- 800-1500 branches and 2-5 k scalar
// instructions per thread for the 16-32 pieces of a tile, as much issue time as the arithmetic (profiles/r05_epilogue_isa.txt).  A tile
// whose columns all exist (rows may be ragged), with 16-byte aligned rows, takes this path instead: the tile-level test is made once (workgroup
// uniform), the dtype of C is a template argument, the bias of a lane's 8 columns is loaded once per tile, and a piece is two LDS reads,
// the arithmetic and one or two 16-byte stores.
// (round 6: + dropout and the residual add -- the epilogues of the ViT blocks' proj / fc1 / fc2, vision_transformer.py:148-150, 176-177,
// 196-197, which ran the general path: 35 % of the image-input step's kernel time in three GEMM instantiations)
constexpr int EPI_FAST_MASK = HAMT_EPI_BIAS | HAMT_EPI_GELU | HAMT_EPI_GELU_GRAD | HAMT_EPI_ACCUM | HAMT_EPI_MUL_AUX | HAMT_EPI_DROPOUT | HAMT_EPI_ADD_AUX;
__device__ __forceinline__ bool epi_fast_ok(const GemmArgsF& g, int epi, int m0, int n0, int bm, int bn) {
  (void)m0; (void)bm;     // (rows may be ragged: epi_fast8 skips rows >= M, one compare per piece)
  bool ok = (epi & ~EPI_FAST_MASK) == 0 && g.ksplit <= 1 && n0 + bn <= g.N && (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0;
  if (epi & HAMT_EPI_BIAS) ok = ok && ((uintptr_t)g.bias & 15) == 0;
  if (epi & (HAMT_EPI_GELU_GRAD | HAMT_EPI_MUL_AUX)) ok = ok && g.dtype_aux == HAMT_BF16 && (g.ldaux & 7) == 0 && ((uintptr_t)g.aux & 15) == 0;
  if (epi & HAMT_EPI_ADD_AUX)      // the residual: fp32 or bf16 rows, 16-byte aligned; `aux` cannot also be the gelu' image
    ok = ok && !(epi & (HAMT_EPI_GELU_GRAD | HAMT_EPI_MUL_AUX)) && (g.dtype_aux == HAMT_BF16 || g.dtype_aux == HAMT_F32) && (g.ldaux & 7) == 0 && ((uintptr_t)g.aux & 15) == 0;
  return ok;
}
// EPI: compile-time flag set (subset of EPI_FAST_MASK); b8: the bias of columns col .. col + 7 (BIAS); pre_aux: the piece's 8 bf16 of aux
// when the caller requested them ahead of its transpose (MUL_AUX), else nullptr
template <int EPI, bool C16>
__device__ __forceinline__ void epi_fast8(const GemmArgsF& g, int row, int col, const float* acc, const float* b8, const uint4* pre_aux, float* ssq = nullptr) {
  if (row >= g.M) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (EPI & HAMT_EPI_BIAS) ? __builtin_fmaf(acc[j], g.alpha, b8[j]) : acc[j] * g.alpha;
  if constexpr ((EPI & HAMT_EPI_GELU) != 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
  }
  float keep[8];
  if constexpr ((EPI & HAMT_EPI_DROPOUT) != 0) {      // the row's mask stream, two groups of four columns (as epi_store draws them)
    const RngKey key = rng_key(g.rng, g.call_id);     // (kernel-argument address: scalar loads)
    const uint32_t rowh = hamt_mix32((uint32_t)row ^ key.k0);
    const float inv_keep = 1.0f / (1.0f - g.p_drop);
    float f0[4], f1[4];
    drop_scale4(key, rowh, (uint32_t)(col >> 2), g.p_drop, inv_keep, f0);
    drop_scale4(key, rowh, (uint32_t)(col >> 2) + 1u, g.p_drop, inv_keep, f1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { keep[j] = f0[j]; keep[4 + j] = f1[j]; }
  }
  if constexpr ((EPI & HAMT_EPI_GELU_GRAD) != 0) {
    float dg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gelu_and_grad(v[j], v[j], dg[j]);
    if constexpr ((EPI & HAMT_EPI_DROPOUT) != 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dg[j] *= keep[j];
    }
    st_bf<8>((bf16_t*)g.aux + (size_t)row * g.ldaux + col, dg);
  }
  if constexpr ((EPI & HAMT_EPI_MUL_AUX) != 0) {
    const uint4 u = pre_aux ? *pre_aux : *(const uint4*)((const bf16_t*)g.aux + (size_t)row * g.ldaux + col);
    const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] *= __uint_as_float(w4[q] << 16); v[2 * q + 1] *= __uint_as_float(w4[q] & 0xffff0000u); }
  }
  if constexpr ((EPI & HAMT_EPI_DROPOUT) != 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= keep[j];
  }
  if constexpr ((EPI & HAMT_EPI_ADD_AUX) != 0) {      // the residual add behind the dropout
    float h[8];
    const size_t ia = (size_t)row * g.ldaux + col;
    if (g.dtype_aux == HAMT_BF16) ld_bf<8>((const bf16_t*)g.aux + ia, h); else ld_f<8>((const float*)g.aux + ia, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += h[j];
  }
  const size_t ic = (size_t)row * g.ldc + col;
  if constexpr (C16) {
    const bool ch = g.dtype_c == HAMT_F16;      // (kernel-argument compare: scalar, one branch per piece)
    bf16_t* c = (bf16_t*)g.C + ic;
    if constexpr ((EPI & HAMT_EPI_ACCUM) != 0) { float p[8]; ld_16<8>(c, p, ch); for (int j = 0; j < 8; ++j) v[j] += p[j]; }
    st_16<8>(c, v, ch);
  } else {
    float* c = (float*)g.C + ic;
    if constexpr ((EPI & HAMT_EPI_ACCUM) != 0) { float p[8]; ld_f<8>(c, p); for (int j = 0; j < 8; ++j) v[j] += p[j]; }
    st_f<8>(c, v);
    if (ssq) { for (int j = 0; j < 8; ++j) *ssq += v[j] * v[j]; }       // (weight-gradient tiles: the sum of squares of what was stored)
  }
}
template <int EPI> __device__ __forceinline__ void epi_fast_bias(const GemmArgsF& g, int col, float* b8) {
  if constexpr (EPI >= 0 && (EPI & HAMT_EPI_BIAS) != 0) ld_f<8>(g.bias + col, b8);
  else { for (int j = 0; j < 8; ++j) b8[j] = 0.f; }
}

// tile id -> (m0, n0) with the XCD-aware remap (blocks are dealt round-robin to the 8 XCDs: give each XCD a contiguous
// run of tile ids so that neighbouring tiles share operand rows in one L2)
__device__ __forceinline__ int xcd_remap(int bid, int ntiles) {
  const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
