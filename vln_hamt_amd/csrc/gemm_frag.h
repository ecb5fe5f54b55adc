// Operand staging and MFMA fragment helpers shared by the bf16 GEMM kernels (gemm_fast.hip, gemm_q4.hip): LDS-DMA pieces
// (global_load_lds_dwordx4 through inline asm), per-tile source offsets, swizzled fragment reads, counted waits.
// Include inside an anonymous namespace after common.h.
#pragma once

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int BN = 128, BK = 64;

__device__ __forceinline__ void glds16(const bf16_t* src, unsigned dst_uniform) {
  // Issued through inline asm on purpose: hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of the first ds_read
  // that follows a __builtin_amdgcn_global_load_lds it can see, which drains the prefetch of tile kt+1 before tile kt
  // is multiplied.  The DMA is ordered by the explicit vmcnt + s_barrier in the main loop instead.  M0 (the LDS
  // destination) is declared clobbered rather than saved/restored: 3 instructions per piece instead of 5.
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst_uniform) : "memory", "m0");
}
// the same with the source as scalar base + 32-bit per-lane byte offset (one VGPR, no 64-bit address arithmetic per piece)
__device__ __forceinline__ void glds16_off(const bf16_t* base_uniform, unsigned byte_off, unsigned dst_uniform) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(byte_off), "s"(base_uniform), "s"(dst_uniform) : "memory", "m0");
}
// LDS byte address of a __shared__ object as a plain integer (taken ONCE: every use of the pointer cast costs a null
// check, s_cmp + s_cselect, per DMA piece otherwise); piece destinations are integer offsets from it.
__device__ __forceinline__ unsigned lds_base_of(const bf16_t* p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t*)p); }

// XOR applied to the 16-byte chunk index of a k-strided ("col") tile row kk: keeps 32-byte pairs together (a tr-read
// quad reads 32 contiguous bytes) and sends the 4 rows of a tr-read (and, for 256-byte rows, the sibling 16-lane
// group 8 rows further) to different bank groups.
template <int R> __device__ __forceinline__ int col_swz(int kk) {
  return R == 256 ? (((kk & 3) | (((kk >> 3) & 3) << 2)) << 1) : R == 128 ? (((kk & 3) | (((kk >> 3) & 1) << 2)) << 1) : ((kk & 3) << 1);
}

// DMA one operand tile into LDS; every wave-instruction moves 1 KiB (64 lanes x 16 B).
//   KM == false: operand stored [rows][K] (K contiguous): tile image [R][64], 8 rows per instruction, chunk ^= row & 7
//   KM == true : operand stored [K][cols] (K strided):    tile image [64][R], 1 KiB = 1024/(2R) k-rows per instruction
// Out-of-range rows are clamped (re-read a valid row); see the callers for why that is harmless.
// The per-lane source offsets are computed ONCE per output tile: the address arithmetic of a piece (row clamp,
// swizzle, 64-bit multiply-add: ~20 VALU instructions, measured ~100 cycles per piece in a load segment) shrinks to one
// v_add (K-contiguous) or add + min + mad (K-strided) per piece and k-tile.  Offsets are 32-bit: operands < 4 GiB.
template <bool KM, int R, int NW> struct TileSrc {
  static constexpr int CH = R / 8, RPI = KM ? 64 / CH : 8, PER_WAVE = KM ? BK / NW : R / NW, NP = PER_WAVE / RPI;
  unsigned off[NP];
  int kk0;
  __device__ __forceinline__ void init(int ld, int r0, int rmax, int w, int lane) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if constexpr (!KM) {
        const int r = w * PER_WAVE + j * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (r & 7);
        int gr = r0 + r;
        gr = gr < rmax ? gr : rmax;
        off[j] = ((unsigned)gr * (unsigned)ld + (unsigned)chunk * 8u) * 2u;
      } else {
        const int kk = w * PER_WAVE + j * RPI + lane / CH;
        const int chunk = (lane % CH) ^ col_swz<R>(kk);
        int c = r0 + chunk * 8;
        const int cmax = min(ld - 8, (rmax >> 3) << 3);     // (see p8_src_init: clamp inside the operand, not inside the row stride)
        c = c < cmax ? c : cmax;
        off[j] = (unsigned)c * 2u;
      }
    }
    kk0 = KM ? w * PER_WAVE + lane / CH : 0;
  }
  __device__ __forceinline__ void issue(const bf16_t* __restrict__ P, int ld, int k0, int kmax, unsigned lds_bytes, int w) const {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if constexpr (!KM) {
        glds16_off(P, off[j] + (unsigned)k0 * 2u, lds_bytes + (unsigned)((w * PER_WAVE + j * 8) * BK * 2));
      } else {
        int gk = k0 + kk0 + j * RPI;
        gk = gk < kmax ? gk : kmax;
        glds16_off(P, (unsigned)gk * ((unsigned)ld * 2u) + off[j], lds_bytes + (unsigned)((w * PER_WAVE + j * RPI) * R * 2));
      }
    }
  }
};

// MFMA fragment (8 bf16 along k for one row/column) of the 16 rows/columns starting at r16, k-step s (32 k)
template <bool KM, int R>
__device__ __forceinline__ bf16x8 frag(const bf16_t* lds, int r16, int s, int lane) {
  union { uint4 u; bf16x8 v; s16x4 h[2]; } f;
  if constexpr (!KM) {
    const int r = r16 + (lane & 15), chunk = 4 * s + (lane >> 4);
    f.u = *(const uint4*)(lds + r * BK + ((chunk ^ (r & 7)) << 3));
  } else {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int i = lane & 15, col = r16 + (i & 3) * 4;
    const int k0 = 32 * s + 8 * (lane >> 4) + (i >> 2), k1 = k0 + 4;
    const bf16_t* p0 = lds + k0 * R + ((((col >> 3) ^ col_swz<R>(k0)) << 3) | (col & 7));
    const bf16_t* p1 = lds + k1 * R + ((((col >> 3) ^ col_swz<R>(k1)) << 3) | (col & 7));
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
  }
  return f.v;
}

// A K-strided A operand (weight-gradient form: A = dY stored [K][M]) whose reduction rows [kvalid, K) are padding: the DMA
// re-reads row kvalid - 1 for them (finite values, never beyond the operand), and the fragments of the k-tile that holds the
// boundary are zeroed HERE for k >= kvalid -- so the padding rows of NEITHER operand need to hold anything in particular (they
// used to have to be zero in at least one and finite in the other; one uninitialised NaN there is a NaN gradient).  `kbase` =
// first reduction row of this 32-wide k-step; a lane of group g = lane >> 4 holds k = kbase + 8 g .. + 7 in element order.
__device__ __forceinline__ bf16x8 mask_k_tail(bf16x8 v, int kbase, int lane, int kvalid) {
  union { bf16x8 v; uint32_t w[4]; } f;
  f.v = v;
  const int k0 = kbase + 8 * (lane >> 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t lo = k0 + 2 * j < kvalid ? 0x0000ffffu : 0u, hi = k0 + 2 * j + 1 < kvalid ? 0xffff0000u : 0u;
    f.w[j] &= lo | hi;
  }
  return f.v;
}

template <int N> __device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vmcnt();
template <> __device__ __forceinline__ void wait_vmcnt<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<12>() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<16>() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }

