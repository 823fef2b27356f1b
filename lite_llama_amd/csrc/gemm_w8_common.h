// Exact widening of 8-bit weights to fp16 pairs, shared by the decode-shaped engine (gemm_w8_skinny.hip) and the M-tiled prefill
// engine (gemm_w8_prefill.hip) -- ONE definition: the two must dequantise bit-identically (w8a16.py:48-74 semantics).
#pragma once
#include "common.h"

__device__ __forceinline__ uint32_t d8_mul(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) * __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t d8_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
__device__ __forceinline__ uint32_t d8_bcast(float v) {
  const uint32_t h = f32_to_f16_bits(v);
  return h | (h << 16);
}
// 4 int8 -> 2 x (2 fp16) * s: bias to unsigned, place each byte under the 1024 exponent, remove 1152 (w8a16.py:66-74 semantics)
__device__ __forceinline__ void d8_i8(uint32_t w, uint32_t s, uint32_t& o0, uint32_t& o1) {
  const uint32_t u = w ^ 0x80808080u;
  o0 = d8_mul(d8_add(__builtin_amdgcn_perm(0x64646464u, u, 0x04010400u), 0xE480E480u), s);
  o1 = d8_mul(d8_add(__builtin_amdgcn_perm(0x64646464u, u, 0x04030402u), 0xE480E480u), s);
}
// 4 fp8 e4m3 bytes -> 2 x (2 fp16) * (s * 256): the reference's bit surgery (w8a16.py:48-62)
__device__ __forceinline__ void d8_fp8(uint32_t w, uint32_t s256, uint32_t& o0, uint32_t& o1) {
  uint32_t p0 = __builtin_amdgcn_perm(0u, w, 0x010C000Cu);
  uint32_t p1 = __builtin_amdgcn_perm(0u, w, 0x030C020Cu);
  p0 = (p0 & 0x80008000u) | ((p0 >> 1) & 0x3F803F80u);
  p1 = (p1 & 0x80008000u) | ((p1 >> 1) & 0x3F803F80u);
  o0 = d8_mul(p0, s256);
  o1 = d8_mul(p1, s256);
}

