// Probe: gfx950 v_cvt_scalef32_pk_f16_fp8 against the reference's bit surgery (w8a16.py:48-62) for all 256 e4m3 codes,
// scale 2^-8 (what the surgery implies), 1.0 and 0.37 (is a non-power-of-two scale applied in full?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t* out, float s) {
  const uint32_t c = threadIdx.x;           // code in byte 0, ~code in byte 1, code in byte 2, 0 in byte 3
  const uint32_t w = c | ((c ^ 0x55u) << 8) | (c << 16);
  const h2 a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, s, false);
  const h2 b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, s, true);
  uint32_t p0 = __builtin_amdgcn_perm(0u, w, 0x010C000Cu);
  uint32_t p1 = __builtin_amdgcn_perm(0u, w, 0x030C020Cu);
  p0 = (p0 & 0x80008000u) | ((p0 >> 1) & 0x3F803F80u);
  p1 = (p1 & 0x80008000u) | ((p1 >> 1) & 0x3F803F80u);
  out[4 * c] = __builtin_bit_cast(uint32_t, a);
  out[4 * c + 1] = __builtin_bit_cast(uint32_t, b);
  out[4 * c + 2] = p0;
  out[4 * c + 3] = p1;
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 256 * 16);
  uint32_t h[1024];
  const float scales[3] = {0x1p-8f, 1.0f, 0.37f};
  for (int si = 0; si < 3; ++si) {
    k<<<1, 256>>>(d, scales[si]);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int diff = 0;
    for (int c = 0; c < 256; ++c) {
      if (si == 0 && (h[4 * c] != h[4 * c + 2] || h[4 * c + 1] != h[4 * c + 3])) {
        if (diff < 12) printf("  code 0x%02x: hw %08x %08x  surgery %08x %08x\n", c, h[4 * c], h[4 * c + 1], h[4 * c + 2], h[4 * c + 3]);
        ++diff;
      }
    }
    if (si == 0) printf("scale 2^-8: %d of 256 codes differ from the bit surgery\n", diff);
    else {
      _Float16 v; uint16_t bits = (uint16_t)(h[4 * 0x38] & 0xffff);  // code 0x38 = 1.0
      __builtin_memcpy(&v, &bits, 2);
      bits = (uint16_t)(h[4 * 0x3C] & 0xffff);  // 0x3C = 1.5
      _Float16 v2; __builtin_memcpy(&v2, &bits, 2);
      printf("scale %g: code 0x38 (1.0) -> %g, code 0x3C (1.5) -> %g\n", scales[si], (float)v, (float)v2);
    }
  }
  return 0;
}
