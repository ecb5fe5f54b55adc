// ABI bookkeeping: version + thread-local last-error text.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void hamt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int hamt_version(void) { return HAMT_ABI_VERSION; }

extern "C" int hamt_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return (int)strlen(g_err);
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = 0;
  return (int)strlen(buf);
}

static thread_local char g_kernel[160] = "";

void hamt_set_last_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

extern "C" int hamt_last_kernel(char* buf, size_t n) {
  if (!buf || n == 0) return (int)strlen(g_kernel);
  strncpy(buf, g_kernel, n - 1);
  buf[n - 1] = 0;
  return (int)strlen(buf);
}

extern "C" int hamt_gemm_ksplit(const hamt_gemm_desc* d);
size_t hamt_vis_embed_ws_bytes(int M, int H);   // vis_embed.hip

extern "C" size_t hamt_workspace_bytes(int op, const int* shape, int nshape) {
  switch (op) {
    case HAMT_WS_GEMM_SPLITK: {
      if (nshape < 3 || !shape) return 0;
      hamt_gemm_desc d{};
      d.M = shape[0]; d.N = shape[1]; d.K = shape[2];
      d.lda = d.K; d.ldb = d.K; d.ldc = d.N;
      d.dtype_a = d.dtype_b = HAMT_BF16; d.dtype_c = HAMT_F32; d.prec = HAMT_PREC_BF16;
      const int ks = hamt_gemm_ksplit(&d);
      return ks > 1 ? (size_t)ks * d.M * d.N * 4 : 0;
    }
    case HAMT_WS_COLSUM: return nshape >= 2 && shape ? (size_t)64 * shape[1] * 4 : 0;
    case HAMT_WS_SUMSQ: return 1024 * 4;
    case HAMT_WS_LN_BWD: return nshape >= 2 && shape ? (size_t)3 * 256 * shape[1] * 4 : 0;
    case HAMT_WS_WGRAD_TABLE: {
      size_t e = 0;
      for (int i = 0; i < nshape; ++i) e += (size_t)(shape[i] + 63) / 64;
      return e * HAMT_WGRAD_TABLE_ENTRY;
    }
    case HAMT_WS_LNRED_TABLE: return nshape >= 1 && shape ? (size_t)shape[0] * HAMT_LNRED_TABLE_ENTRY : 0;
    case HAMT_WS_VIS_EMBED_BWD: return nshape >= 2 && shape ? hamt_vis_embed_ws_bytes(shape[0], shape[1]) : 0;
    case HAMT_WS_EMBED_BWD: {
      if (nshape < 2 || !shape) return 0;
      const size_t a = (size_t)64 * shape[1] * 4, b = (size_t)136 * shape[0];
      return a > b ? a : b;
    }
    default: return 0;
  }
}
