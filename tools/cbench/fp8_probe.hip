// Standalone probe (no torch): conventions of gfx950's scaled FP8 conversions, before the pair BPTT's fp8 lo plane relies on them.
//   v_cvt_scalef32_pk_fp8_f32 (f32 x2 -> fp8 x2 into one 16-bit half)   and   v_cvt_scalef32_pk_f16_fp8 (fp8 x2 -> f16 x2)
// Prints, per (value pair, scale): the 16-bit code and the two f16 values the inverse returns -- round trip = identity (up to
// e4m3's 4 significant bits) iff "pack divides by the scale, unpack multiplies" (or the other way round: visible here).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, const float* scales, unsigned* codes, float* out, int n, int ns) {
  const int i = threadIdx.x;
  if (i >= n * ns) return;
  const int v = i % n, s = i / n;
  const float a = in[2 * v], b = in[2 * v + 1], sc = scales[s];
  s2 old = {0, 0};
  const s2 p = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, sc, false);
  const unsigned c = __builtin_bit_cast(unsigned, p);
  codes[i] = c;
  const h2 r = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c, sc, false);
  const h2 r1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(c, 1.0f, false);
  out[4 * i] = (float)r[0], out[4 * i + 1] = (float)r[1], out[4 * i + 2] = (float)r1[0], out[4 * i + 3] = (float)r1[1];
}
int main() {
  const int n = 8, ns = 4;
  float hin[2 * n] = {1.0f, 2.0f, 0.3f, -0.7f, 3.0e-3f, 1.1e-4f, 100.f, 500.f, 0.0625f, 0.0156f, 17.f, -25.f, 1e-6f, 448.f, 0.4f, 0.45f};
  float hsc[ns] = {1.0f, 0.25f, 4.0f, 1.0f / 4096.f};
  float *din, *dsc, *dout;
  unsigned* dc;
  hipMalloc(&din, sizeof(hin)), hipMalloc(&dsc, sizeof(hsc)), hipMalloc(&dc, n * ns * 4), hipMalloc(&dout, n * ns * 16);
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice), hipMemcpy(dsc, hsc, sizeof(hsc), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dsc, dc, dout, n, ns);
  unsigned hc[n * ns];
  float ho[4 * n * ns];
  hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost), hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  for (int s = 0; s < ns; ++s)
    for (int v = 0; v < n; ++v) {
      const int i = s * n + v;
      printf("scale %-10g in (%-9g, %-9g) code 0x%04x -> unpack(same scale) (%-9g, %-9g)  unpack(scale 1) (%-9g, %-9g)\n", hsc[s], hin[2 * v],
             hin[2 * v + 1], hc[i] & 0xffffu, ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3]);
    }
  return 0;
}
