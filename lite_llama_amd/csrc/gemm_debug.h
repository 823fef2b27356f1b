// Measurement instruments of the W4A16 decode engines (gemm_w4_v3.hip, gemm_w4_v4.hip, gemm_short.hip) -- NOT product code.
// The regular build (lite_llama_amd/build.py) defines none of the switches below: every macro expands to nothing and the
// kernels carry no debug field, branch or argument.  tools/build_variant.py builds an A/B library with one of
//   -DV3_TIMELINE   benchmarks/gemm3_timeline.py   stamps of wave LL_GEMM3_TL_WAVE per workgroup ([workgroup][64])
//   -DV4_TIMELINE   benchmarks/gemm4_timeline.py   stamps per (workgroup, wave) ([workgroup][12][64], shader clock)
//   -DSS_TIMELINE   benchmarks/gemm_short_timeline.py  ([workgroup][12][16])
// and the host side reads the buffer's address from LL_GEMM{3,4,_SS}_TIMELINE.  The ABLATION builds of rounds 3 - 6 (parts of a
// kernel's work removed to time the rest: no DMA, no dequantisation, no MFMA, no barriers ...) are no longer in the sources: the
// hooks are kept as a patch (benchmarks/probes/gemm_ablate_hooks.patch, applies to this tree), their results in DESIGN_NOTEBOOK.md.
#pragma once
#include <stdlib.h>

// ---- gemm_w4_v3.hip (unit loop): 0 entry, 1 ranges decoded, 2 prologue loads issued, 3 first x tile staged + barrier, 4 + u:
//      barrier that ends unit u (u < 40), 50..55 last segment end: begin / exchanged / counter seen / slabs added / stores issued,
//      60 wave done; TLC = shader-clock stamps (s_memtime), TL = s_memrealtime ----
#ifdef V3_TIMELINE
#define V3_DEBUG_FIELDS unsigned long long* tl; int tlwave;
#define V3_DEBUG_SET(P)                                                                                                    \
  (P).tlwave = getenv("LL_GEMM3_TL_WAVE") ? atoi(getenv("LL_GEMM3_TL_WAVE")) : 0;                                          \
  (P).tl = getenv("LL_GEMM3_TIMELINE") ? (unsigned long long*)strtoull(getenv("LL_GEMM3_TIMELINE"), nullptr, 16) : nullptr;
#define V3_TL(IDX) if (p.tl && lane == 0 && wv == p.tlwave) p.tl[(size_t)blockIdx.x * 64 + (IDX)] = __builtin_amdgcn_s_memrealtime();
#define V3_TLC(IDX) if (p.tl && lane == 0 && wv == p.tlwave) p.tl[(size_t)blockIdx.x * 64 + (IDX)] = __builtin_amdgcn_s_memtime();
// pin the step's MFMAs BEFORE the stamp that follows: the accumulators are made opaque to the scheduler
#define V3_TL_FENCE(CUR) asm volatile("" : "+v"(acc0), "+v"(acc1));
#else
#define V3_DEBUG_FIELDS
#define V3_DEBUG_SET(P)
#define V3_TL(IDX)
#define V3_TLC(IDX)
#define V3_TL_FENCE(CUR)
#endif

// ---- gemm_w4_v4.hip (row-group loop): p.tl[(workgroup * 12 + wave) * 64 + idx] = s_memtime; 62 / 63 = s_memrealtime at entry / exit ----
#ifdef V4_TIMELINE
#define V4_DEBUG_FIELDS unsigned long long* tl;
#define V4_DEBUG_SET(P) (P).tl = getenv("LL_GEMM4_TIMELINE") ? (unsigned long long*)strtoull(getenv("LL_GEMM4_TIMELINE"), nullptr, 16) : nullptr;
#define V4_TL(IDX) if (p.tl && lane == 0) p.tl[((size_t)blockIdx.x * 12 + wv) * 64 + (IDX)] = __builtin_amdgcn_s_memtime();
#define V4_TL_REAL(IDX) if (p.tl && lane == 0) p.tl[((size_t)blockIdx.x * 12 + wv) * 64 + (IDX)] = __builtin_amdgcn_s_memrealtime();
#else
#define V4_DEBUG_FIELDS
#define V4_DEBUG_SET(P)
#define V4_TL(IDX)
#define V4_TL_REAL(IDX)
#endif

// ---- gemm_short.hip (short-stream engine): the kernel's last argument p_tl is always there (nullptr in the product build) ----
#ifdef SS_TIMELINE
#define SS_DEBUG_SET(P) (P).tl = getenv("LL_GEMM_SS_TIMELINE") ? (unsigned long long*)strtoull(getenv("LL_GEMM_SS_TIMELINE"), nullptr, 16) : nullptr;
#define SS_TL(IDX) if (p_tl && lane == 0) p_tl[((size_t)blockIdx.x * 12 + wv) * 16 + (IDX)] = __builtin_amdgcn_s_memrealtime();
#define SS_TLC(IDX) if (p_tl && lane == 0) p_tl[((size_t)blockIdx.x * 12 + wv) * 16 + (IDX)] = __builtin_amdgcn_s_memtime();
#define SS_TL_PIECE_LANDED(I, WORD, S0) if ((I) < 4) { asm volatile("" : "+v"(WORD), "+v"(S0)); SS_TL(3 + 2 * (I)) }
#define SS_TL_PIECE_DONE(I, ACC) if ((I) < 4) { asm volatile("v_mov_b32 %0, %0" : "+v"(ACC)); SS_TL(4 + 2 * (I)) }
#else
#define SS_DEBUG_SET(P)
#define SS_TL(IDX)
#define SS_TLC(IDX)
#define SS_TL_PIECE_LANDED(I, WORD, S0)
#define SS_TL_PIECE_DONE(I, ACC)
#endif
