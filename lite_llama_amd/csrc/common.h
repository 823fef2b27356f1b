// Shared device helpers for the gfx950 kernels (wave64, 16-byte vector I/O).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lite_llama_amd.h"

#define LL_WAVE 64

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
// round-to-nearest-even, NaN preserved (quiet)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
  return (float)__builtin_bit_cast(f16, h);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (f16)f);
}

// 16-bit storage type <-> fp32, selected by the LL_F16 / LL_BF16 code.
template <int DT>
__device__ __forceinline__ float to_f32(uint16_t b) {
  if constexpr (DT == LL_F16) return f16_bits_to_f32(b);
  else return bf16_bits_to_f32(b);
}
template <int DT>
__device__ __forceinline__ uint16_t from_f32(float f) {
  if constexpr (DT == LL_F16) return f32_to_f16_bits(f);
  else return f32_to_bf16_bits(f);
}

// Product of two storage-dtype values rounded once to the storage dtype (what
// a fp16*fp16 / bf16*bf16 multiply yields: the fp32 product is exact).
template <int DT>
__device__ __forceinline__ uint16_t mul_storage(uint16_t a, uint16_t b) {
  return from_f32<DT>(to_f32<DT>(a) * to_f32<DT>(b));
}

struct alignas(16) U16x8 {
  uint16_t v[8];
};

template <int VEC>
struct VecIO;
template <>
struct VecIO<8> {
  __device__ static __forceinline__ void load(const uint16_t* p, uint16_t (&o)[8]) {
    U16x8 t = *reinterpret_cast<const U16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = t.v[i];
  }
  __device__ static __forceinline__ void store(uint16_t* p, const uint16_t (&o)[8]) {
    U16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = o[i];
    *reinterpret_cast<U16x8*>(p) = t;
  }
};
template <>
struct VecIO<1> {
  __device__ static __forceinline__ void load(const uint16_t* p, uint16_t (&o)[1]) { o[0] = p[0]; }
  __device__ static __forceinline__ void store(uint16_t* p, const uint16_t (&o)[1]) { p[0] = o[0]; }
};

// Wave reductions on DPP row moves + permlane swaps (a few cycles per step) -- NOT __shfl_xor, which hipcc lowers to
// ds_bpermute_b32 (~100 cycles each; six dependent ones per reduction: ~0.3 us of every row-per-workgroup norm launch, ~5 us
// of a router tail's eight selection rounds, round 4).  Every lane of the wave must be active (DPP reads disabled lanes as
// garbage): all callers reduce right before a workgroup barrier.  The result is lane 0's value in every lane.
template <int CTRL>
__device__ __forceinline__ float ll_dpp_f32(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  float t;
  t = ll_dpp_f32<0xB1>(v); v += t;   // quad_perm [1,0,3,2]
  t = ll_dpp_f32<0x4E>(v); v += t;   // quad_perm [2,3,0,1]
  t = ll_dpp_f32<0x124>(v); v += t;  // row_ror:4
  t = ll_dpp_f32<0x128>(v); v += t;  // row_ror:8
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}
__device__ __forceinline__ float wave_max(float v) {
  float t;
  t = ll_dpp_f32<0xB1>(v); v = fmaxf(v, t);
  t = ll_dpp_f32<0x4E>(v); v = fmaxf(v, t);
  t = ll_dpp_f32<0x124>(v); v = fmaxf(v, t);
  t = ll_dpp_f32<0x128>(v); v = fmaxf(v, t);
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
  return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}

// shared by swiglu / silu_and_mul and the GEMM's fused gate-up epilogue (must stay ONE definition:
// the fused path is tested bit-identical against the separate kernels)
// Round 4: the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each -- an order below the fp16 rounding of the
// product that follows; swiglu.py:45-65 computes tl.sigmoid in fp32 and Triton lowers it to the same two approximations)
// instead of libm expf + an IEEE division: ~35 dependent VALU operations per value, 2.6 us of the fused gate|up launch's
// tail (in-kernel stamps, DESIGN_NOTEBOOK.md 4.3).  Saturates correctly: exp2(+big) = inf -> 0, exp2(-big) = 0 -> 1; NaN stays NaN.
__device__ __forceinline__ float ll_sigmoidf(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

// silu(g) * u rounded to fp16 the way the reference rounds it: the fp32 product first (swiglu.py:45-65 computes in fp32 and
// stores with .to(dtype)).  The product is pinned in a register before the narrowing conversion: hipcc may otherwise select
// v_fma_mixlo_f16 for fptrunc(fmul) in ONE of several kernels that must agree bit for bit (single instead of double rounding,
// different once in ~2^13 values -- found in round 4 on the smoothquant norms, again in round 5 on the M-tiled GEMM's epilogue).
__device__ __forceinline__ float ll_silu_mul_f32(float g, float u) {
  float p = g * ll_sigmoidf(g) * u;
  asm volatile("" : "+v"(p));
  return p;
}

static inline bool ll_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define LL_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? LL_OK : LL_ERR_LAUNCH)

// Reductions over the four 16-lane rows of a wave (lanes t, t + 16, t + 32, t + 48) with the gfx950 row-swap VALU
// instructions instead of two ds_bpermute round trips through the LDS pipe: v_permlane16_swap exchanges the odd rows of
// its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half
// of the second -- fed the same value twice, the two results hold a lane's own value and its partner's.
__device__ __forceinline__ float rows_max(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

