// L2 -> LDS fill rate of global_load_lds_dwordx4 per CU and chip-wide: every workgroup (8 waves) streams its own L2-resident slab
// into a 64 KiB LDS ring, keeping PIECES 1-KiB pieces per wave in flight.  usage: ta_bench   (prints a table)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ void glds16_off(const void* base, unsigned off, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(dst) : "memory", "m0");
}
template <int INFLIGHT>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, size_t slab_bytes, int iters, unsigned long long* cycles) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (size_t)blockIdx.x * slab_bytes;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned off = (unsigned)(w * 8192 + lane * 16);           // each wave walks its own 8 KiB lane of the slab ring
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < INFLIGHT; ++p) {
      glds16_off(base, (off + p * 1024) % (unsigned)slab_bytes, lds0 + (unsigned)(w * 8192 + p * 1024));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    off += 65536;                                            // next 64 KiB window of the slab
    if (off >= slab_bytes) off -= (unsigned)slab_bytes;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (lds[threadIdx.x] == 77 && iters < 0) cycles[0] = 0;    // keep LDS alive
}
int main(int argc, char** argv) {
  const size_t slab = (argc > 1 ? atoi(argv[1]) : 256) * 1024;   // per-workgroup slab in KiB (256 WGs x 128 KiB = all 32 MiB of L2)
  char* src; unsigned long long* cyc;
  hipMalloc(&src, slab * 1024); hipMemset(src, 1, slab * 1024); hipMalloc(&cyc, 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%6s %9s %12s %14s %12s\n", "WGs", "inflight", "us", "B/clk/CU(evt)", "TB/s chip");
  for (int wgs : {32, 64, 128, 256}) {
    for (int inflight : {4, 8}) {
      const int iters = 2000;
      auto launch = [&]() {
        if (inflight == 4) hipLaunchKernelGGL(fill_kernel<4>, dim3(wgs), dim3(512), 0, 0, src, slab, iters, cyc);
        else hipLaunchKernelGGL(fill_kernel<8>, dim3(wgs), dim3(512), 0, 0, src, slab, iters, cyc);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[1024]; hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost);
      double mc = 0; for (int i = 0; i < wgs; ++i) mc += h[i]; mc /= wgs;
      const double bytes_per_wg = (double)iters * inflight * 8 * 1024;
      printf("%6d %9d %12.1f %14.1f %12.2f   (cycles/WG %.0f => %.1f B/clk by s_memtime)\n", wgs, inflight, ms * 1e3, bytes_per_wg / (ms * 1e-3 * 2.1e9),
             bytes_per_wg * wgs / (ms * 1e-3) / 1e12, mc, bytes_per_wg / mc);
    }
  }
  return 0;
}
