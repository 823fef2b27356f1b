// Device helpers shared by the W4A16 decode engines (gemm_w4_v3.hip: stream-K / split-K unit loop; gemm_w4_v4.hip: N-split
// row-group loop): the exact nibble unpack + fp16 affine map, the LDS-DMA statements and the counted vmcnt wait.
// One definition on purpose: both engines must dequantise bit-identically (tests compare them).
#pragma once
#include "common.h"

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ uint32_t v3_pk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t v3_pk_fma(uint32_t a, uint32_t b, uint32_t c) {
  f16x2 r = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b),
                                      __builtin_bit_cast(f16x2, c));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t v3_and_or(uint32_t w, uint32_t mask, uint32_t magic) {
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mask), "v"(magic));
  return r;
}
// One packed word -> eight fp16 weights in natural k order (the packer stores nibble 2i of a k-octet at
// position i and nibble 2i+1 at position 4+i).  Exact unpack: (w & 0x000F000F) | 0x6400 = (1024 + q_i,
// 1024 + q_{4+i}); the offset is removed exactly, then ONE fp16 fma with (s, -z*s): the same arithmetic as
// gemm_wq.hip.  Error against the reference's fp16(fp32((q - z) * s)): <= 3 fp16 ulps of the
// weight (roundings of s, of z*s and of the fma), stated in DESIGN_NOTEBOOK.md.
__device__ __forceinline__ f16x8 v3_dequant(uint32_t w, uint32_t s, uint32_t nzs, uint32_t magic) {
  const uint32_t w2 = w >> 8;
  uint32_t a = v3_and_or(w, 0x000F000Fu, magic);
  uint32_t b = v3_and_or(w, 0x00F000F0u, magic);
  uint32_t c = v3_and_or(w2, 0x000F000Fu, magic);
  uint32_t d = v3_and_or(w2, 0x00F000F0u, magic);
  a = v3_pk_add(a, 0xE400E400u);
  b = v3_pk_fma(b, 0x2C002C00u, 0xD400D400u);
  c = v3_pk_add(c, 0xE400E400u);
  d = v3_pk_fma(d, 0x2C002C00u, 0xD400D400u);
  u32x4 o;
  o.x = v3_pk_fma(a, s, nzs);
  o.y = v3_pk_fma(b, s, nzs);
  o.z = v3_pk_fma(c, s, nzs);
  o.w = v3_pk_fma(d, s, nzs);
  return __builtin_bit_cast(f16x8, o);
}



// LDS-DMA: 64 lanes x 16 B (or 4 B) from saddr + voff to the wave-uniform LDS byte address lds_dst + lane * size.
// M0 carries the LDS address and is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ const void* v3_uniform_ptr(const void* p) {  // provably wave-uniform for the "s" constraint
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (const void*)(((uint64_t)hi << 32) | lo);
}
// NT: the non-temporal cache policy for data this CU reads once (the weight stream) -- never for the activation tile,
// which every workgroup re-reads from L2
#ifndef V3_NT_W
#define V3_NT_W 1
#endif
template <bool NT = false>
__device__ __forceinline__ void v3_dma16(uint32_t lds_dst, const void* sbase, uint32_t voff) {
  uint32_t keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  sbase = v3_uniform_ptr(sbase);
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
template <bool NT = false>
__device__ __forceinline__ void v3_dma4(uint32_t lds_dst, const void* sbase, uint32_t voff) {
  uint32_t keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  sbase = v3_uniform_ptr(sbase);
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

template <int N>
__device__ __forceinline__ void v3_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// host-side dispatch between the two engines (gemm_w4_v4.hip)
extern "C" int ll_w4a16_prepacked_supported(int64_t m, int64_t n, int64_t k, int group_size);
int v4_wants(int64_t m, int64_t n, int64_t k, int group_size, int epilogue);
int v4_launch(void* out, const void* x, const void* wpacked, const void* spacked, const void* bias, int64_t m, int64_t n, int64_t k,
              int group_size, int64_t x_stride_m, int epilogue, int kslices, void* stream);
// ... and the short-stream engine for split-K partial launches of a few tens of KB per CU (gemm_short.hip)
int ss_partials_slices(int64_t m, int64_t n, int64_t k, int group_size);  // planes it leaves; 0: not its launch
// the short-stream engine with finished outputs (gemm_short_full.hip): narrow projections that must leave fp16 rows
int sf_wants(int64_t m, int64_t n, int64_t k, int group_size, int epilogue);
int sf_launch(void* out, const void* x, const void* wpacked, const void* spacked, const void* bias, int64_t m, int64_t n, int64_t k,
              int group_size, int64_t x_stride_m, int epilogue, void* stream);
int ss_launch(void* out, const void* x, const void* wpacked, const void* spacked, int64_t m, int64_t n, int64_t k, int group_size,
              int64_t x_stride_m, void* stream);
