// Gather / scatter / broadcast / elementwise kernels (all HBM- or L2-bound; 16-byte accesses, one row
// per 64..256 threads).  They carry the integer/index work of the path, which must be bit-exact:
// embedding lookups (vilmodel.py:62-66, 501, 543, 569), boolean-mask compaction
// (pretrain_cmt.py:161-165), SPREL anchor gather + concat (pretrain_cmt.py:211-214), hist/obs
// concat + slice (vilmodel.py:467-468, 475-477, 618), panorama mean (vilmodel.py:563-564), the SAP/ITM
// fusion products (pretrain_cmt.py:176, vilmodel.py:722) and the -inf fill (pretrain_cmt.py:177).
#include "common.h"

void hamt_reduce_partials(int R, int N, const float* ws, float* out, int accumulate, hipStream_t s);

namespace {

__global__ void gather_rows_kernel(int R, int W, const float* __restrict__ src, int ld_src,
                                   const int64_t* __restrict__ idx, const float* base, int ld_base, float* out, int ld_out,
                                   int col0) {
  const int w4 = W >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)R * w4; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / w4), c = (int)(i % w4) * 4;
    const int64_t sr = idx ? idx[r] : r;
    float4 v = *(const float4*)(src + (size_t)sr * ld_src + c);
    float* o = out + (size_t)r * ld_out + col0 + c;
    if (base) { const float4 p = *(const float4*)(base + (size_t)r * ld_base + col0 + c); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
    *(float4*)o = v;
  }
}
__global__ void scatter_add_rows_kernel(int R, int W, const float* __restrict__ src, int ld_src, int col0,
                                        const int64_t* __restrict__ idx, float* __restrict__ dst, int ld_dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)R * W; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / W), c = (int)(i % W);
    const int64_t dr = idx ? idx[r] : r;
    atomicAdd(dst + (size_t)dr * ld_dst + c, src[(size_t)r * ld_src + col0 + c]);
  }
}
__global__ void embed_sum_fwd_kernel(int B, int L, int H, const int64_t* __restrict__ ids, const float* __restrict__ word,
                                     const float* __restrict__ pos, const float* __restrict__ type_row, float* __restrict__ z) {
  const int h4 = H >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)B * L * h4; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / h4), c = (int)(i % h4) * 4, l = row % L;
    const float4 a = *(const float4*)(word + (size_t)ids[row] * H + c), p = *(const float4*)(pos + (size_t)l * H + c),
                 ty = *(const float4*)(type_row + c);
    // same association as the reference: (word + position) + token_type  (vilmodel.py:66)
    *(float4*)(z + (size_t)row * H + c) = make_float4((a.x + p.x) + ty.x, (a.y + p.y) + ty.y, (a.z + p.z) + ty.z, (a.w + p.w) + ty.w);
  }
}
__global__ void embed_sum_bwd_kernel(int B, int L, int H, int V, const int64_t* __restrict__ ids, const float* __restrict__ dz,
                                     float* __restrict__ dword) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)B * L * H; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / H), c = (int)(i % H);
    const int64_t id = ids[row];
    if ((uint64_t)id < (uint64_t)V) atomicAdd(dword + (size_t)id * H + c, dz[i]);
  }
}
// x[B,S,H]: mode 0 -> partial sums over (b,s) per block-row-chunk; mode 1 -> sums over b per (s,h)
__global__ void sum_rows_partial_kernel(int R, int H, const float* __restrict__ x, float* __restrict__ ws, int rows_per_chunk) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += x[(size_t)r * H + c];
  ws[(size_t)blockIdx.y * H + c] = s;
}
__global__ void sum_over_b_kernel(int B, int SH, const float* __restrict__ x, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= SH) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += x[(size_t)b * SH + i];
  out[i] += s;
}
__global__ void mean_mid_fwd_kernel(int B, int S, int H, const float* __restrict__ x, float* __restrict__ y) {
  const int h4 = H >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * h4) return;
  const int b = (int)(i / h4), c = (int)(i % h4) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = 0; j < S; ++j) {
    const float4 v = *(const float4*)(x + ((size_t)b * S + j) * H + c);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float inv = 1.0f / (float)S;  // torch.mean: sum then divide
  *(float4*)(y + (size_t)b * H + c) = make_float4(s.x / (float)S, s.y / (float)S, s.z / (float)S, s.w / (float)S);
  (void)inv;
}
__global__ void mean_mid_bwd_kernel(int B, int S, int H, const float* __restrict__ dy, float* __restrict__ dx) {
  const int h4 = H >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)B * S * h4; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / h4), c = (int)(i % h4) * 4, b = row / S;
    const float4 v = *(const float4*)(dy + (size_t)b * H + c);
    const float inv = 1.0f / (float)S;
    *(float4*)(dx + (size_t)row * H + c) = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
  }
}
__global__ void mul_bcast_fwd_kernel(int B, int S, int H, const float* __restrict__ a, const float* __restrict__ c, int ldc_rows,
                                     float* __restrict__ y) {
  const int h4 = H >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)B * S * h4; i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / h4), col = (int)(i % h4) * 4, b = row / S;
    const float4 x = *(const float4*)(a + (size_t)row * H + col), m = *(const float4*)(c + (size_t)b * ldc_rows + col);
    *(float4*)(y + (size_t)row * H + col) = make_float4(x.x * m.x, x.y * m.y, x.z * m.z, x.w * m.w);
  }
}
// da = dy * c ; dc[b] = sum_s dy * a   (one block per (b, 256-column chunk))
__global__ void mul_bcast_bwd_kernel(int B, int S, int H, const float* __restrict__ a, const float* __restrict__ c, int ldc_rows,
                                     const float* __restrict__ dy, float* __restrict__ da, float* __restrict__ dc) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (col >= H) return;
  const float m = c[(size_t)b * ldc_rows + col];
  float acc = 0.f;
  for (int s = 0; s < S; ++s) {
    const size_t o = ((size_t)b * S + s) * H + col;
    const float g = dy[o];
    da[o] = g * m;
    acc += g * a[o];
  }
  dc[(size_t)b * H + col] = acc;
}
__global__ void add3_kernel(size_t n4, const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                            float4* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 x = a[i], y = b[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    if (c) { float4 z = c[i]; x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w; }
    out[i] = x;
  }
}
__global__ void dropout_kernel(size_t n, const float* __restrict__ x, float* __restrict__ y, float p, uint32_t call_id,
                               const uint64_t* __restrict__ rng) {
  const RngKey k = rng_key(rng, call_id);
  const float ik = 1.0f / (1.0f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = x[i] * drop_scale(k, i, p, ik);
}
__global__ void cast_bf16_kernel(size_t n8, size_t n, const float* __restrict__ x, bf16_t* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = *(const float4*)(x + i * 8), b = *(const float4*)(x + i * 8 + 4);
    *(uint4*)(y + i * 8) = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) y[n8 * 8 + threadIdx.x] = f2bf(x[n8 * 8 + threadIdx.x]);
}
// Device-side batch collation (data/collate.py; replaces the host loops of pretrain_src/data/common.py:5-29): ragged rows
// packed back to back -> zero/pattern padded [B][maxlen][row]; prefix[b] = first packed row of sample b.
template <typename V>
__global__ void unpack_padded_kernel(const V* __restrict__ src, const int32_t* __restrict__ prefix, int B, int maxlen, int row_v, V padv,
                                     V* __restrict__ dst) {
  const size_t per = (size_t)maxlen * row_v, total = (size_t)B * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const size_t r = i - (size_t)b * per;
    const int t = (int)(r / row_v), c = (int)(r - (size_t)t * row_v);
    const int p0 = prefix[b], len = prefix[b + 1] - p0;
    dst[i] = t < len ? src[(size_t)(p0 + t) * row_v + c] : padv;
  }
}
// mask[b][t] = t < len_b + add (common.py:22-29), lens_out[b] = len_b + add
__global__ void seq_masks_kernel(const int32_t* __restrict__ prefix, int add, int B, int maxlen, uint8_t* __restrict__ mask, int64_t* __restrict__ lens_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * maxlen) return;
  const int b = i / maxlen, t = i - b * maxlen, len = prefix[b + 1] - prefix[b] + add;
  mask[i] = t < len ? 1 : 0;
  if (t == 0 && lens_out) lens_out[b] = len;
}
// Gradient wire format (parallel.py): y = bf16(x * scale) and back.  x and y share their element offset inside mirrored,
// 256-byte aligned arenas, so `head` scalar elements bring both to vector alignment (float4 / 4 x bf16).
__global__ void wire_pack_kernel(size_t n, size_t head, const float* __restrict__ x, bf16_t* __restrict__ y, float scale) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = (n - head) / 4, tail0 = head + n4 * 4;
  for (size_t i = tid; i < n4; i += nth) {
    const float4 a = *(const float4*)(x + head + i * 4);
    *(uint2*)(y + head + i * 4) = make_uint2(pack_bf2(a.x * scale, a.y * scale), pack_bf2(a.z * scale, a.w * scale));
  }
  if (tid < head) y[tid] = f2bf(x[tid] * scale);
  if (tid < n - tail0) y[tail0 + tid] = f2bf(x[tail0 + tid] * scale);
}
__global__ void wire_unpack_kernel(size_t n, size_t head, const bf16_t* __restrict__ y, float* __restrict__ x) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = (n - head) / 4, tail0 = head + n4 * 4;
  for (size_t i = tid; i < n4; i += nth) {
    const uint2 u = *(const uint2*)(y + head + i * 4);
    *(float4*)(x + head + i * 4) = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  }
  if (tid < head) x[tid] = bf2f(y[tid]);
  if (tid < n - tail0) x[tail0 + tid] = bf2f(y[tail0 + tid]);
}
// fp32 [R][C] -> bf16 [R][Cpad], columns >= C zero filled (reduction-dimension padding for the fast GEMM)
__global__ void cast_pad_kernel(int R, int C, int Rpad, int Cpad, const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy) {
  const int c8 = Cpad >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)Rpad * c8; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c8), c = (int)(i % c8) * 8;
    const float* xr = x + (size_t)r * ldx + c;
    uint4 o;
    if (r >= R) o = make_uint4(0u, 0u, 0u, 0u);
    else if (c + 8 <= C && (((uintptr_t)xr) & 15) == 0) {
      const float4 a = *(const float4*)xr, b = *(const float4*)(xr + 4);
      o = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
    } else {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (c + j < C) ? xr[j] : 0.f;
      o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
    *(uint4*)(y + (size_t)r * ldy + c) = o;
  }
}
__global__ void cast_pad_dropout_kernel(int R, int C, int Rpad, const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, float p,
                                        uint32_t call_id, const uint64_t* __restrict__ rng) {
  const RngKey key = rng_key(rng, call_id);
  const float inv_keep = 1.0f / (1.0f - p);
  const int c8 = C >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)Rpad * c8; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c8), c = (int)(i % c8) * 8;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (r < R) {
      const float* xr = x + (size_t)r * ldx + c;
      float v[8], f0[4], f1[4];
      if ((((uintptr_t)xr) & 15) == 0) { const float4 a = *(const float4*)xr, b = *(const float4*)(xr + 4); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
      else for (int j = 0; j < 8; ++j) v[j] = xr[j];
      const uint32_t rowh = hamt_mix32((uint32_t)r ^ key.k0);
      drop_scale4(key, rowh, (uint32_t)(c >> 2), p, inv_keep, f0);
      drop_scale4(key, rowh, (uint32_t)(c >> 2) + 1u, p, inv_keep, f1);
      o = make_uint4(pack_bf2(v[0] * f0[0], v[1] * f0[1]), pack_bf2(v[2] * f0[2], v[3] * f0[3]), pack_bf2(v[4] * f1[0], v[5] * f1[1]), pack_bf2(v[6] * f1[2], v[7] * f1[3]));
    }
    *(uint4*)(y + (size_t)r * ldy + c) = o;
  }
}
__global__ void fill_where_zero_kernel(size_t n, const int64_t* __restrict__ flag, float* __restrict__ x, float value) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (flag[i] == 0) x[i] = value;
}
// dx = dy * act'(h): mode 1 = erf-GELU (h = pre-activation), mode 2 = ReLU (h = pre-activation or output)
__global__ void act_bwd_kernel(size_t n4, const float4* __restrict__ dy, const float4* __restrict__ h, int mode, float4* __restrict__ dx) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 g = dy[i], x = h[i];
    float4 r;
    if (mode == 1) { r.x = g.x * dgelu_erf(x.x); r.y = g.y * dgelu_erf(x.y); r.z = g.z * dgelu_erf(x.z); r.w = g.w * dgelu_erf(x.w); }
    else { r.x = x.x > 0.f ? g.x : 0.f; r.y = x.y > 0.f ? g.y : 0.f; r.z = x.z > 0.f ? g.z : 0.f; r.w = x.w > 0.f ? g.w : 0.f; }
    dx[i] = r;
  }
}
__global__ void rng_advance_kernel(uint64_t* rng) { rng[1] += 1; }

inline int nblocks(size_t work, int bs = 256, int cap = 2048) {
  size_t b = (work + bs - 1) / bs;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

extern "C" int hamt_gather_rows(int R, int W, const float* src, int ld_src, const int64_t* idx, const float* base, int ld_base,
                                float* out, int ld_out, int col0, void* stream) {
  HAMT_CHECK_ARG(src && out && W % 4 == 0 && ld_src % 4 == 0 && ld_out % 4 == 0 && col0 % 4 == 0 && (!base || ld_base % 4 == 0),
                 "hamt_gather_rows: bad argument");
  if (R == 0) return HAMT_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nblocks((size_t)R * W / 4)), dim3(256), 0, as_stream(stream), R, W, src, ld_src, idx, base, ld_base, out, ld_out, col0);
  HAMT_CHECK_LAUNCH("hamt_gather_rows");
  return HAMT_OK;
}
// A table of a few rows (token types, navigability types, a cls token: T <= 8) gathered by thousands of rows: the atomic scatter above
// adds them in whatever order the hardware serves the collisions, i.e. the table's gradient differs in the last bit from run to run.
// Here every block sums its chunk of source rows per table row in a fixed order (block = 64 columns x 4 row phases, T accumulators
// per thread), and the chunks are summed in order by hamt_reduce_partials: deterministic, and no serialised atomics.
constexpr int SCAT_T_MAX = 8;
__global__ __launch_bounds__(256) void scatter_small_partial_kernel(int R, int W, int T, const float* __restrict__ src, int ld_src, int col0,
                                                                    const int64_t* __restrict__ idx, float* __restrict__ ws, int rows_per_chunk) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float acc[SCAT_T_MAX];
#pragma unroll
  for (int t = 0; t < SCAT_T_MAX; ++t) acc[t] = 0.f;
  if (col < W)
    for (int r = r0 + ph; r < r1; r += 4) {
      const int t_of = (int)idx[r];
      const float v = src[(size_t)r * ld_src + col0 + col];
#pragma unroll
      for (int t = 0; t < SCAT_T_MAX; ++t) acc[t] += t == t_of ? v : 0.f;
    }
  __shared__ float red[4][64];
  for (int t = 0; t < T; ++t) {
    red[ph][threadIdx.x & 63] = acc[t];
    __syncthreads();
    if (ph == 0 && col < W)
      ws[((size_t)blockIdx.y * T + t) * W + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}
extern "C" int hamt_scatter_add_rows_small(int R, int W, const float* src, int ld_src, int col0, const int64_t* idx, int T, float* dst,
                                           float* ws, void* stream) {
  HAMT_CHECK_ARG(src && dst && idx && ws && T >= 1 && T <= SCAT_T_MAX, "hamt_scatter_add_rows_small: bad argument (1 <= T <= %d table rows)", SCAT_T_MAX);
  if (R == 0) return HAMT_OK;
  hipStream_t s = as_stream(stream);
  int chunks = R >= 64 * 64 ? 64 : (R + 63) / 64;
  const int rpc = (R + chunks - 1) / chunks;
  hipLaunchKernelGGL(scatter_small_partial_kernel, dim3((W + 63) / 64, chunks), dim3(256), 0, s, R, W, T, src, ld_src, col0, idx, ws, rpc);
  hamt_reduce_partials(chunks, T * W, ws, dst, 1, s);
  HAMT_CHECK_LAUNCH("hamt_scatter_add_rows_small");
  return HAMT_OK;
}
// ---- dst[idx[r]] += src[r] in a FIXED order for any table (the word embeddings: 30 522 rows, a few thousand source rows with repeated
// tokens -- [MASK] alone is 15 % of an MLM batch).  The atomic scatter adds colliding rows in whatever order the hardware serves them: the
// table's gradient then differs in the last bit from run to run (tools/grad_bitwise_repeat.py: up to 30 distinct values in 30 repetitions),
// which is enough to flip a rounded two-rank average.  Here the source rows are ranked by (index, row) -- a counting
// sort by comparison: R x R integer comparisons through LDS tiles, 26 M for the step's 5120 rows, spread over up to 32 slices of the row range --
// and every table row that is hit is summed over its (now contiguous) list of source rows by one wave (up to 8 rows, in row order) or
// by the eight waves of a block (wave k: the members j = k (mod 8) in order; the partial sums added in wave order).  One writer per
// table row, no atomics, the same association for the same indices: bit-reproducible.  (A first version walked a linked list of equal
// rows per table row: the 770-row chain of [MASK] cost 0.4 ms of dependent loads.)
constexpr int SCAT_SLICES = 32;
__global__ __launch_bounds__(256) void scatter_rank_kernel(int R, int per, const int64_t* __restrict__ idx, int* __restrict__ cnt) {
  __shared__ int tile[256];                          // (table rows < 2^31: 32-bit compares -- 64-bit integer compares run at a quarter of the rate)
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int mine = r < R ? (int)idx[r] : -1;
  const int q0 = blockIdx.y * per, q1 = min(R, q0 + per);
  int below = 0;                                     // rows q of this slice with (idx[q], q) < (mine, r)
  for (int t0 = q0; t0 < q1; t0 += 256) {
    const int nv = min(256, q1 - t0);
    if ((int)threadIdx.x < nv) tile[threadIdx.x] = (int)idx[t0 + threadIdx.x];
    __syncthreads();
    if (r < R) {
#pragma unroll 8
      for (int i = 0; i < nv; ++i) below += (tile[i] < mine) + ((tile[i] == mine) & (t0 + i < r));
    }
    __syncthreads();
  }
  if (r < R) cnt[blockIdx.y * R + r] = below;
}
__global__ __launch_bounds__(256) void scatter_perm_kernel(int R, int ns, const int64_t* __restrict__ idx, const int* __restrict__ cnt, int* __restrict__ perm,
                                                           int* __restrict__ sid) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  int pos = 0;
  for (int y = 0; y < ns; ++y) pos += cnt[y * R + r];
  perm[pos] = r;                                    // (a permutation: (idx, row) pairs are distinct)
  sid[pos] = (int)idx[r];
}
constexpr int SCAT_WAVES = 8, SCAT_UNROLL = 16;      // (512-thread blocks; 16 source rows x one float4 per lane in flight per wave)
// A block covers 256 COLUMNS (one float4 per lane; grid.y = column chunks of a wider row): against one block per 768 columns this is three
// times the blocks for the one long list of a batch -- [MASK], ~770 of 5120 source rows, summed by a single block -- and twice the rows
// in flight per wave: 42 -> ~20 us at the step's shape.  The order of the additions is unchanged (and independent of the chunking).
// one wave's sum of the members first, first + stride, ... < e of a table row's source rows, columns cb + 4 lane .. + 3:
// the loads of SCAT_UNROLL members are requested together, the adds stay in member order
__device__ __forceinline__ void scatter_members_sum(const float* __restrict__ src, int ld_src, int col0, int W, int cb, int ln, bool v4,
                                                    const int* __restrict__ perm, int first, int stride, int e, float* a) {
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = 0.f;
  const int c = cb + 4 * ln;
  if (c >= W) return;                               // (lanes beyond the row: nothing to sum; wave-divergent exit is fine, no shuffles below)
  int nxt[SCAT_UNROLL];                             // (the next round's row numbers are requested while this round's rows arrive)
#pragma unroll
  for (int u = 0; u < SCAT_UNROLL; ++u) { const int j = first + u * stride; nxt[u] = j < e ? perm[j] : -1; }
  for (int j0 = first; j0 < e; j0 += stride * SCAT_UNROLL) {
    int rows[SCAT_UNROLL];
#pragma unroll
    for (int u = 0; u < SCAT_UNROLL; ++u) rows[u] = nxt[u];
#pragma unroll
    for (int u = 0; u < SCAT_UNROLL; ++u) { const int j = j0 + (SCAT_UNROLL + u) * stride; nxt[u] = j < e ? perm[j] : -1; }
    float v[SCAT_UNROLL][4];
#pragma unroll
    for (int u = 0; u < SCAT_UNROLL; ++u) {
      if (rows[u] < 0) continue;                    // (wave uniform)
      const float* sp = src + (size_t)rows[u] * ld_src + col0 + c;
      if (v4 && c + 4 <= W) { const float4 f = *(const float4*)sp; v[u][0] = f.x; v[u][1] = f.y; v[u][2] = f.z; v[u][3] = f.w; }
      else for (int k = 0; k < 4; ++k) v[u][k] = c + k < W ? sp[k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SCAT_UNROLL; ++u)
      if (rows[u] >= 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += v[u][k];
      }
  }
}
__device__ __forceinline__ void scatter_row_add(float* __restrict__ drow, int W, int cb, int ln, bool v4, const float* a) {
  const int c = cb + 4 * ln;
  if (c >= W) return;
  if (v4 && c + 4 <= W) {
    float4 d = *(float4*)(drow + c);
    d.x += a[0]; d.y += a[1]; d.z += a[2]; d.w += a[3];
    *(float4*)(drow + c) = d;
  } else for (int k = 0; k < 4; ++k) { if (c + k < W) drow[c + k] += a[k]; }
}
__global__ __launch_bounds__(64 * SCAT_WAVES) void scatter_segment_add_kernel(int R, int W, const float* __restrict__ src, int ld_src, int col0,
                                                                             const int* __restrict__ perm, const int* __restrict__ sid,
                                                                             float* __restrict__ dst, int ld_dst, int T) {
  // A block looks at SCAT_WAVES consecutive sorted positions, one per wave.  A position that starts a table row's list [p, e) of source rows:
  //   n = e - p <= SCAT_WAVES members (nearly all: most tokens occur once or a few times): THAT wave sums them in row order and adds the
  //     sum to the table row -- no LDS, no barrier;
  //   longer lists ([MASK]: 15 % of an MLM batch, a position table row: every sample): afterwards, the whole block: wave k sums the
  //     members j = k (mod SCAT_WAVES) in order, the partial sums are added in wave order, then to the table row.
  // Either way ONE writer per table row and an order of additions that depends on the indices only.
  __shared__ float part[SCAT_WAVES - 1][256];
  __shared__ int longs[SCAT_WAVES];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int cb = blockIdx.y * 256;                  // this block's columns
  const bool v4 = ((ld_src | col0 | W | ld_dst) & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  const int p = blockIdx.x * SCAT_WAVES + wv;
  int e = p, id = -1;
  bool start = false;
  if (p < R) {
    id = sid[p];
    // (a table has T rows: an id outside [0, T) -- a padding index, a corrupted batch -- would make one wave add a whole row out of bounds;
    // its source rows are skipped, as the sorted list keeps them together at either end)
    start = (p == 0 || sid[p - 1] != id) && (unsigned)id < (unsigned)T;
    if (start) {      // end of the list: a few steps of a linear scan (most lists have one or two rows), then -- sid is sorted -- a binary search
      e = p + 1;      // (a linear scan of a 750-row list is 750 dependent L2 round trips; a binary search of every 1-row list is 13)
      int k = 0;
      while (e < R && k < 4 && sid[e] == id) { ++e; ++k; }
      if (k == 4 && e < R && sid[e] == id) {
        int lo = e + 1, hi = R;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sid[mid] == id) lo = mid + 1; else hi = mid; }
        e = lo;
      }
    }
  }
  const int n = e - p;
  if (ln == 0) longs[wv] = (start && n > SCAT_WAVES) ? e : 0;
  if (start && n <= SCAT_WAVES) {
    float a[4];
    scatter_members_sum(src, ld_src, col0, W, cb, ln, v4, perm, p, 1, e, a);
    scatter_row_add(dst + (size_t)id * ld_dst, W, cb, ln, v4, a);
  }
  __syncthreads();
  for (int k = 0; k < SCAT_WAVES; ++k) {
    const int e2 = longs[k];
    if (!e2) continue;                              // (block uniform)
    const int p2 = blockIdx.x * SCAT_WAVES + k;
    float* drow = dst + (size_t)sid[p2] * ld_dst;
    float a[4];
    scatter_members_sum(src, ld_src, col0, W, cb, ln, v4, perm, p2 + wv, SCAT_WAVES, e2, a);
    __syncthreads();
    if (wv) *(float4*)&part[wv - 1][4 * ln] = make_float4(a[0], a[1], a[2], a[3]);
    __syncthreads();
    if (wv == 0) {
      for (int k2 = 0; k2 < SCAT_WAVES - 1; ++k2) {
        const float4 f = *(const float4*)&part[k2][4 * ln];
        a[0] += f.x; a[1] += f.y; a[2] += f.z; a[3] += f.w;
      }
      scatter_row_add(drow, W, cb, ln, v4, a);
    }
  }
}
// ws: (SCAT_SLICES + 2) R ints.  false: no index / no scratch / R too large for the quadratic ranking (the caller falls back to atomics)
static bool scatter_add_ordered(int R, int W, const float* src, int ld_src, int col0, const int64_t* idx, float* dst, int ld_dst, int T, int* ws, hipStream_t s) {
  static const bool off = getenv("HAMT_ATOMIC_SCATTER") != nullptr;      // (measurement: the atomic kernels instead)
  // The ranking compares every pair of source rows: 26 M compares (11 us) at the step's 5 120 rows, 1 G (~0.4 ms) at 32 768, 69 G at
  // SCAT_MAX_ROWS -- a pass of tens of milliseconds that no batch of this path comes near (RxR: 32 x 250 = 8 000 rows); beyond that
  // the atomic kernels, LOUDLY (once): the sums are then correct but not bit-reproducible
  constexpr int SCAT_MAX_ROWS = 262144;
  if (R > SCAT_MAX_ROWS && idx && ws && !off) {
    static bool told = false;
    if (!told) { told = true; fprintf(stderr, "[hamt] scatter_add over %d rows (> %d): atomic adds instead of the ordered sums -- results are not bit-reproducible\n", R, SCAT_MAX_ROWS); }
  }
  if (R > SCAT_MAX_ROWS || !idx || !ws || off) return false;
  const int ns = (R + 255) / 256 < SCAT_SLICES ? (R + 255) / 256 : SCAT_SLICES;      // slices of the row range (grid.y): 256 rows each up to 8192 rows
  const int per = ((R + ns - 1) / ns + 255) / 256 * 256;
  int* cnt = ws;
  int* perm = ws + (size_t)SCAT_SLICES * R;
  int* sid = perm + R;
  hipLaunchKernelGGL(scatter_rank_kernel, dim3((R + 255) / 256, ns), dim3(256), 0, s, R, per, idx, cnt);
  hipLaunchKernelGGL(scatter_perm_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, ns, idx, cnt, perm, sid);
  hipLaunchKernelGGL(scatter_segment_add_kernel, dim3((R + SCAT_WAVES - 1) / SCAT_WAVES, (W + 255) / 256), dim3(64 * SCAT_WAVES), 0, s, R, W, src, ld_src, col0, perm, sid, dst, ld_dst, T);
  return true;
}
extern "C" int hamt_scatter_add_rows_ordered(int R, int W, const float* src, int ld_src, int col0, const int64_t* idx, float* dst,
                                             int ld_dst, int T, int* ws, void* stream) {
  HAMT_CHECK_ARG(src && dst && idx && ws && T >= 0, "hamt_scatter_add_rows_ordered: null pointer / negative T");
  if (R == 0) return HAMT_OK;
  if (!scatter_add_ordered(R, W, src, ld_src, col0, idx, dst, ld_dst, T, ws, as_stream(stream)))
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(nblocks((size_t)R * W)), dim3(256), 0, as_stream(stream), R, W, src, ld_src, col0, idx, dst, ld_dst);
  HAMT_CHECK_LAUNCH("hamt_scatter_add_rows_ordered");
  return HAMT_OK;
}
extern "C" int hamt_scatter_add_rows(int R, int W, const float* src, int ld_src, int col0, const int64_t* idx, float* dst,
                                     int ld_dst, void* stream) {
  HAMT_CHECK_ARG(src && dst, "hamt_scatter_add_rows: null pointer");
  if (R == 0) return HAMT_OK;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(nblocks((size_t)R * W)), dim3(256), 0, as_stream(stream), R, W, src, ld_src, col0, idx, dst, ld_dst);
  HAMT_CHECK_LAUNCH("hamt_scatter_add_rows");
  return HAMT_OK;
}
extern "C" int hamt_embed_sum_fwd(int B, int L, int H, const int64_t* ids, const float* word, const float* pos,
                                  const float* type_row, float* z, void* stream) {
  HAMT_CHECK_ARG(ids && word && pos && type_row && z && H % 4 == 0, "hamt_embed_sum_fwd: bad argument");
  if (B * L == 0) return HAMT_OK;
  hipLaunchKernelGGL(embed_sum_fwd_kernel, dim3(nblocks((size_t)B * L * H / 4)), dim3(256), 0, as_stream(stream), B, L, H, ids, word, pos, type_row, z);
  HAMT_CHECK_LAUNCH("hamt_embed_sum_fwd");
  return HAMT_OK;
}
extern "C" int hamt_sum_rows(int B, int S, int H, const float* x, int mode, float* out, float* ws, void* stream) {
  HAMT_CHECK_ARG(x && out, "hamt_sum_rows: null pointer");
  if (B * S == 0) return HAMT_OK;
  hipStream_t s = as_stream(stream);
  if (mode == 0) {
    HAMT_CHECK_ARG(ws, "hamt_sum_rows: mode 0 needs ws (64*H floats)");
    const int R = B * S;
    int chunks = R >= 64 * 16 ? 64 : (R + 15) / 16;
    if (chunks < 1) chunks = 1;
    const int rpc = (R + chunks - 1) / chunks;
    hipLaunchKernelGGL(sum_rows_partial_kernel, dim3((H + 255) / 256, chunks), dim3(256), 0, s, R, H, x, ws, rpc);
    hamt_reduce_partials(chunks, H, ws, out, 1, s);
  } else {
    hipLaunchKernelGGL(sum_over_b_kernel, dim3((S * H + 255) / 256), dim3(256), 0, s, B, S * H, x, out);
  }
  HAMT_CHECK_LAUNCH("hamt_sum_rows");
  return HAMT_OK;
}
extern "C" int hamt_colsum(int M, int N, const void* x, int ldx, int dtype_x, float* out, int accumulate, float* ws, void* stream);
extern "C" int hamt_embed_sum_bwd(int B, int L, int H, int V, const int64_t* ids, const float* dz, float* dword, float* dpos,
                                  float* dtype_row, float* ws, size_t ws_bytes, void* stream) {
  // Three passes over dz (16 MB at the step's shape, L2 / MALL resident), each on the kernel built for it: the word rows by atomic adds
  // over the whole grid (a one-pass kernel with one block per position was measured: 87 us -- 80 blocks cannot issue 3.9 M atomics as
  // fast as 15 000 can -- against 26 here), the positions by the slice-sum kernel (dz = B slices of L * H), the token-type row by the
  // column-sum kernels.  The last two replaced a serial loop per thread (26 + 35 us) at the tail of the backward chain.
  HAMT_CHECK_ARG(ids && dz, "hamt_embed_sum_bwd: null pointer");
  HAMT_CHECK_ARG(!dtype_row || ws, "hamt_embed_sum_bwd: dtype_row needs ws (HAMT_WS_EMBED_BWD {B * L, H} bytes)");
  if (B * L == 0) return HAMT_OK;
  hipStream_t s = as_stream(stream);
  // (the word rows: in a fixed order when ws is given AND large enough for the ranking's (SCAT_SLICES + 2) R ints -- scatter_add_ordered
  // above -- else, or beyond 262 144 rows, by atomic adds.  ws_bytes: ABI 1 took the size on trust; a caller that sized ws by the older
  // HAMT_WS_COLSUM rule would have been overrun by the ranking arrays)
  const bool ws_ok = ws && ws_bytes >= (size_t)(SCAT_SLICES + 2) * (size_t)(B * L) * sizeof(int);
  if (dtype_row) {
    const int shp[2] = {B * L, H};
    HAMT_CHECK_ARG(ws_bytes >= hamt_workspace_bytes(HAMT_WS_COLSUM, shp, 2), "hamt_embed_sum_bwd: ws_bytes = %zu is below HAMT_WS_COLSUM {B * L, H}", ws_bytes);
  }
  if (dword && !scatter_add_ordered(B * L, H, dz, H, 0, ids, dword, H, V, ws_ok ? (int*)ws : nullptr, s))
    hipLaunchKernelGGL(embed_sum_bwd_kernel, dim3(nblocks((size_t)B * L * H)), dim3(256), 0, s, B, L, H, V, ids, dz, dword);
  if (dpos) hamt_reduce_partials(B, L * H, dz, dpos, 1, s);
  HAMT_CHECK_LAUNCH("hamt_embed_sum_bwd");
  if (dtype_row) return hamt_colsum(B * L, H, dz, H, HAMT_F32, dtype_row, 1, ws, stream);
  return HAMT_OK;
}
extern "C" int hamt_mean_mid_fwd(int B, int S, int H, const float* x, float* y, void* stream) {
  HAMT_CHECK_ARG(x && y && H % 4 == 0 && S > 0, "hamt_mean_mid_fwd: bad argument");
  if (B == 0) return HAMT_OK;
  hipLaunchKernelGGL(mean_mid_fwd_kernel, dim3(((size_t)B * H / 4 + 255) / 256), dim3(256), 0, as_stream(stream), B, S, H, x, y);
  HAMT_CHECK_LAUNCH("hamt_mean_mid_fwd");
  return HAMT_OK;
}
extern "C" int hamt_mean_mid_bwd(int B, int S, int H, const float* dy, float* dx, void* stream) {
  HAMT_CHECK_ARG(dy && dx && H % 4 == 0 && S > 0, "hamt_mean_mid_bwd: bad argument");
  if (B == 0) return HAMT_OK;
  hipLaunchKernelGGL(mean_mid_bwd_kernel, dim3(nblocks((size_t)B * S * H / 4)), dim3(256), 0, as_stream(stream), B, S, H, dy, dx);
  HAMT_CHECK_LAUNCH("hamt_mean_mid_bwd");
  return HAMT_OK;
}
extern "C" int hamt_mul_bcast_fwd(int B, int S, int H, const float* a, const float* c, int ldc_rows, float* y, void* stream) {
  HAMT_CHECK_ARG(a && c && y && H % 4 == 0 && ldc_rows % 4 == 0, "hamt_mul_bcast_fwd: bad argument");
  if (B * S == 0) return HAMT_OK;
  hipLaunchKernelGGL(mul_bcast_fwd_kernel, dim3(nblocks((size_t)B * S * H / 4)), dim3(256), 0, as_stream(stream), B, S, H, a, c, ldc_rows, y);
  HAMT_CHECK_LAUNCH("hamt_mul_bcast_fwd");
  return HAMT_OK;
}
extern "C" int hamt_mul_bcast_bwd(int B, int S, int H, const float* a, const float* c, int ldc_rows, const float* dy,
                                  float* da, float* dc, void* stream) {
  HAMT_CHECK_ARG(a && c && dy && da && dc, "hamt_mul_bcast_bwd: null pointer");
  if (B * S == 0) return HAMT_OK;
  hipLaunchKernelGGL(mul_bcast_bwd_kernel, dim3((H + 255) / 256, B), dim3(256), 0, as_stream(stream), B, S, H, a, c, ldc_rows, dy, da, dc);
  HAMT_CHECK_LAUNCH("hamt_mul_bcast_bwd");
  return HAMT_OK;
}
// ---------------------------------------------------------------- image -> patch rows (ViT patch embedding as a GEMM)
// y[(n, py, px)][c*P*P + ky*P + kx] = x[n][c][py*P + ky][px*P + kx]: the column order of the flattened conv weight
// [D][C][P][P], so that conv2d(kernel = stride = P) (vision_transformer.py:216-221) is y @ W.view(D, C*P*P)^T + b.
// One thread per 4 consecutive kx (16-byte loads along the image row, 8-byte bf16 stores along the patch row).
template <typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(int N, int C, int H, int W, int P, const float* __restrict__ x, TO* __restrict__ y,
                                                       int ldy, int Rpad) {
  const int gw = W / P, gh = H / P, K = C * P * P, q = K / 4;
  const size_t total = (size_t)Rpad * q;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / q), col = (int)(i % q) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < N * gh * gw) {
      const int n = row / (gh * gw), py = (row / gw) % gh, px = row % gw;
      const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
      v = *(const float4*)(x + (((size_t)n * C + c) * H + py * P + ky) * W + px * P + kx);
    }
    if constexpr (sizeof(TO) == 2) *(uint2*)(y + (size_t)row * ldy + col) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    else *(float4*)(y + (size_t)row * ldy + col) = v;
  }
}
extern "C" int hamt_patchify(int N, int C, int H, int W, int P, const float* x, void* y, int ldy, int dtype_y, int Rpad, void* stream) {
  HAMT_CHECK_ARG(x && y && N >= 0 && C > 0 && P > 0 && P % 4 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0 && ldy >= C * P * P && ldy % 4 == 0 &&
                 Rpad >= N * (H / P) * (W / P) && ((uintptr_t)x % 16) == 0, "hamt_patchify: bad argument");
  if (Rpad == 0) return HAMT_OK;
  const size_t total = (size_t)Rpad * (C * P * P / 4);
  if (dtype_y == HAMT_BF16) hipLaunchKernelGGL((patchify_kernel<bf16_t>), dim3(nblocks(total)), dim3(256), 0, as_stream(stream), N, C, H, W, P, x, (bf16_t*)y, ldy, Rpad);
  else hipLaunchKernelGGL((patchify_kernel<float>), dim3(nblocks(total)), dim3(256), 0, as_stream(stream), N, C, H, W, P, x, (float*)y, ldy, Rpad);
  HAMT_CHECK_LAUNCH("hamt_patchify");
  return HAMT_OK;
}
extern "C" int hamt_add3(size_t n, const float* a, const float* b, const float* c, float* out, void* stream) {
  HAMT_CHECK_ARG(a && b && out && n % 4 == 0, "hamt_add3: bad argument");
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(add3_kernel, dim3(nblocks(n / 4)), dim3(256), 0, as_stream(stream), n / 4, (const float4*)a, (const float4*)b, (const float4*)c, (float4*)out);
  HAMT_CHECK_LAUNCH("hamt_add3");
  return HAMT_OK;
}
extern "C" int hamt_dropout(size_t n, const float* x, float* y, float p, uint32_t call_id, const uint64_t* rng, void* stream) {
  HAMT_CHECK_ARG(x && y && p >= 0.f && p < 1.f, "hamt_dropout: bad argument");
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(nblocks(n)), dim3(256), 0, as_stream(stream), n, x, y, p, call_id, rng);
  HAMT_CHECK_LAUNCH("hamt_dropout");
  return HAMT_OK;
}
extern "C" int hamt_cast_f32_bf16(size_t n, const float* x, void* y, void* stream) {
  HAMT_CHECK_ARG(x && y, "hamt_cast_f32_bf16: null pointer");
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(nblocks(n / 8 + 1)), dim3(256), 0, as_stream(stream), n / 8, n, x, (bf16_t*)y);
  HAMT_CHECK_LAUNCH("hamt_cast_f32_bf16");
  return HAMT_OK;
}
extern "C" int hamt_unpack_padded(const void* src, const int32_t* prefix, int B, int maxlen, int row_bytes, int pad_byte, void* dst, void* stream) {
  HAMT_CHECK_ARG(prefix && dst && B >= 0 && maxlen >= 0 && row_bytes > 0 && pad_byte >= 0 && pad_byte < 256, "hamt_unpack_padded: bad argument");
  if (B == 0 || maxlen == 0) return HAMT_OK;
  HAMT_CHECK_ARG(src, "hamt_unpack_padded: null source");
  const uintptr_t al = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)row_bytes;
  const uint32_t p4 = 0x01010101u * (uint32_t)pad_byte;
  hipStream_t s = as_stream(stream);
  const size_t total = (size_t)B * maxlen * row_bytes;
  if (al % 16 == 0) hipLaunchKernelGGL((unpack_padded_kernel<uint4>), dim3(nblocks(total / 16)), dim3(256), 0, s, (const uint4*)src, prefix, B, maxlen, row_bytes / 16, make_uint4(p4, p4, p4, p4), (uint4*)dst);
  else if (al % 8 == 0) hipLaunchKernelGGL((unpack_padded_kernel<uint2>), dim3(nblocks(total / 8)), dim3(256), 0, s, (const uint2*)src, prefix, B, maxlen, row_bytes / 8, make_uint2(p4, p4), (uint2*)dst);
  else if (al % 4 == 0) hipLaunchKernelGGL((unpack_padded_kernel<uint32_t>), dim3(nblocks(total / 4)), dim3(256), 0, s, (const uint32_t*)src, prefix, B, maxlen, row_bytes / 4, p4, (uint32_t*)dst);
  else hipLaunchKernelGGL((unpack_padded_kernel<uint8_t>), dim3(nblocks(total)), dim3(256), 0, s, (const uint8_t*)src, prefix, B, maxlen, row_bytes, (uint8_t)pad_byte, (uint8_t*)dst);
  HAMT_CHECK_LAUNCH("hamt_unpack_padded");
  return HAMT_OK;
}
extern "C" int hamt_seq_masks(const int32_t* prefix, int add, int B, int maxlen, uint8_t* mask, int64_t* lens_out, void* stream) {
  HAMT_CHECK_ARG(prefix && mask && B >= 0 && maxlen >= 0, "hamt_seq_masks: bad argument");
  if (B == 0 || maxlen == 0) return HAMT_OK;
  hipLaunchKernelGGL(seq_masks_kernel, dim3((B * maxlen + 255) / 256), dim3(256), 0, as_stream(stream), prefix, add, B, maxlen, mask, lens_out);
  HAMT_CHECK_LAUNCH("hamt_seq_masks");
  return HAMT_OK;
}
extern "C" int hamt_wire_pack_bf16(size_t n, const float* x, void* y, float scale, void* stream) {
  HAMT_CHECK_ARG(x && y, "hamt_wire_pack_bf16: null pointer");
  if (n == 0) return HAMT_OK;
  size_t head = (4 - (((uintptr_t)x / 4) & 3)) & 3;
  if (head > n) head = n;
  HAMT_CHECK_ARG(((uintptr_t)((const bf16_t*)y + head) % 8) == 0 || n - head < 4, "hamt_wire_pack_bf16: x and y must share their offset modulo 4 elements");
  hipLaunchKernelGGL(wire_pack_kernel, dim3(nblocks(n / 4 + 1)), dim3(256), 0, as_stream(stream), n, head, x, (bf16_t*)y, scale);
  HAMT_CHECK_LAUNCH("hamt_wire_pack_bf16");
  return HAMT_OK;
}
extern "C" int hamt_wire_unpack_bf16(size_t n, const void* y, float* x, void* stream) {
  HAMT_CHECK_ARG(x && y, "hamt_wire_unpack_bf16: null pointer");
  if (n == 0) return HAMT_OK;
  size_t head = (4 - (((uintptr_t)x / 4) & 3)) & 3;
  if (head > n) head = n;
  HAMT_CHECK_ARG(((uintptr_t)((const bf16_t*)y + head) % 8) == 0 || n - head < 4, "hamt_wire_unpack_bf16: x and y must share their offset modulo 4 elements");
  hipLaunchKernelGGL(wire_unpack_kernel, dim3(nblocks(n / 4 + 1)), dim3(256), 0, as_stream(stream), n, head, (const bf16_t*)y, x);
  HAMT_CHECK_LAUNCH("hamt_wire_unpack_bf16");
  return HAMT_OK;
}
extern "C" int hamt_cast_pad_bf16(int R, int C, int Rpad, int Cpad, const float* x, int ldx, void* y, int ldy, void* stream) {
  HAMT_CHECK_ARG(x && y && Cpad % 8 == 0 && Cpad >= C && Rpad >= R && ldy >= Cpad && ldy % 8 == 0, "hamt_cast_pad_bf16: bad argument");
  if (Rpad == 0 || Cpad == 0) return HAMT_OK;
  hipLaunchKernelGGL(cast_pad_kernel, dim3(nblocks((size_t)Rpad * Cpad / 8)), dim3(256), 0, as_stream(stream), R, C, Rpad, Cpad, x, ldx, (bf16_t*)y, ldy);
  HAMT_CHECK_LAUNCH("hamt_cast_pad_bf16");
  return HAMT_OK;
}
extern "C" int hamt_cast_pad_bf16_dropout(int R, int C, int Rpad, const float* x, int ldx, void* y, int ldy, float p, uint32_t call_id,
                                          const uint64_t* rng, void* stream) {
  HAMT_CHECK_ARG(x && y && rng && C % 8 == 0 && Rpad >= R && ldy >= C && ldy % 8 == 0 && p >= 0.0f && p < 1.0f, "hamt_cast_pad_bf16_dropout: bad argument");
  if (Rpad == 0 || C == 0) return HAMT_OK;
  hipLaunchKernelGGL(cast_pad_dropout_kernel, dim3(nblocks((size_t)Rpad * C / 8)), dim3(256), 0, as_stream(stream), R, C, Rpad, x, ldx, (bf16_t*)y, ldy, p, call_id, rng);
  HAMT_CHECK_LAUNCH("hamt_cast_pad_bf16_dropout");
  return HAMT_OK;
}
extern "C" int hamt_fill_where_zero(size_t n, const int64_t* flag, float* x, float value, void* stream) {
  HAMT_CHECK_ARG(flag && x, "hamt_fill_where_zero: null pointer");
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(fill_where_zero_kernel, dim3(nblocks(n)), dim3(256), 0, as_stream(stream), n, flag, x, value);
  HAMT_CHECK_LAUNCH("hamt_fill_where_zero");
  return HAMT_OK;
}
extern "C" int hamt_act_bwd(size_t n, const float* dy, const float* h, int mode, float* dx, void* stream) {
  HAMT_CHECK_ARG(dy && h && dx && n % 4 == 0 && (mode == 1 || mode == 2), "hamt_act_bwd: bad argument");
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(nblocks(n / 4)), dim3(256), 0, as_stream(stream), n / 4, (const float4*)dy, (const float4*)h, mode, (float4*)dx);
  HAMT_CHECK_LAUNCH("hamt_act_bwd");
  return HAMT_OK;
}
extern "C" int hamt_rng_advance(uint64_t* rng, void* stream) {
  HAMT_CHECK_ARG(rng, "hamt_rng_advance: null pointer");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), rng);
  HAMT_CHECK_LAUNCH("hamt_rng_advance");
  return HAMT_OK;
}

// additive attention mask of a boolean (1 byte per element) keep-mask: out = (1 - m) * -10000 (vilmodel.py:597-599, 604-606, 626-628)
namespace {
__global__ void extend_mask_kernel(size_t n, const unsigned char* __restrict__ m, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (1.0f - (m[i] ? 1.0f : 0.0f)) * -10000.0f;
}
}  // namespace
extern "C" int hamt_extend_mask(size_t n, const void* mask_u8, float* out, void* stream) {
  HAMT_CHECK_ARG(mask_u8 && out, "hamt_extend_mask: null pointer");
  if (n == 0) return HAMT_OK;
  hipLaunchKernelGGL(extend_mask_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, as_stream(stream), n, (const unsigned char*)mask_u8, out);
  HAMT_CHECK_LAUNCH("hamt_extend_mask");
  return HAMT_OK;
}

// Test aid: fill the LDS of every CU with `pattern` (e.g. 0x7FC07FC0 = two bf16 NaNs) -- a kernel that reads LDS it has not
// written (a missing wait in front of a DMA-filled tile) then produces NaNs instead of silently re-using whatever the
// previous launch left there (tests/test_gpu_ops.py: the tiled GEMM kernels after a poisoned LDS).
__global__ __launch_bounds__(256) void lds_fill_kernel(uint32_t pattern, int words) {
  extern __shared__ uint32_t lds_fill_sm[];
  for (int i = threadIdx.x; i < words; i += 256) lds_fill_sm[i] = pattern;
  __syncthreads();
  if (lds_fill_sm[(threadIdx.x * 7) % words] != pattern) __builtin_trap();   // keeps the stores
}
extern "C" int hamt_debug_fill_lds(uint32_t pattern, void* stream) {
  static bool raised = false;
  const int bytes = 150 * 1024;
  if (!raised) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    raised = true;
  }
  hipLaunchKernelGGL(lds_fill_kernel, dim3(1024), dim3(256), bytes, as_stream(stream), pattern, bytes / 4);
  HAMT_CHECK_LAUNCH("hamt_debug_fill_lds");
  return HAMT_OK;
}
