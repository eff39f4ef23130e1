// Host-side operand splitting shared by the weight packers (dyn_nets.hip, dyn_encoder.hip): fp32 -> 16-bit parts of the split-product
// MFMA engine (dyn_mlp.h).
#pragma once
#include <math.h>
#include <string.h>

namespace {

// round-to-nearest-even bf16 bits of an fp32 value (no NaN inputs here)
unsigned short bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
float bf16_to_f32(unsigned short b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// round-to-nearest-even IEEE half bits of an fp32 value (subnormal halves included; |x| >= 65520 -> inf, refused by the packer)
unsigned short f16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  const unsigned sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);  // >= 65520: rounds to inf
  if (u < 0x38800000u) {                                          // < 2^-14: subnormal half = round(|x| * 2^24)
    float a;
    memcpy(&a, &u, 4);
    const float scaled = a * 16777216.0f;                          // exact
    const float r = nearbyintf(scaled);                            // default rounding mode: to nearest even
    return (unsigned short)(sign | (unsigned)r);
  }
  u += 0xc8000000u;                                               // rebias exponent 127 -> 15
  u += 0x0fffu + ((u >> 13) & 1u);
  return (unsigned short)(sign | (u >> 13));
}
float f16_to_f32(unsigned short h) {
  const unsigned sign = (unsigned)(h & 0x8000u) << 16;
  const unsigned e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float f;
  if (e == 0) {
    f = (float)m * (1.0f / 16777216.0f);
  } else {
    const unsigned u = ((e + 112u) << 23) | (m << 13);
    memcpy(&f, &u, 4);
  }
  unsigned u2;
  memcpy(&u2, &f, 4);
  u2 |= sign;
  memcpy(&f, &u2, 4);
  return f;
}
static thread_local bool g_pack_range_error = false;  // set when a weight does not fit the half-float range (per host thread: pack calls may run concurrently)


// one weight -> the engine's parts (hi | mid | lo as 16-bit patterns)
inline void split_weight(float w, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
#if DYN_SPLIT_F16
  if (!(fabsf(w) < 65504.0f)) g_pack_range_error = true;
  hi = f16_rne(w);
  const float r1 = w - f16_to_f32(hi);
  mid = f16_rne(r1);
  lo = 0;
#else
  hi = bf16_rne(w);
  const float r1 = w - bf16_to_f32(hi);
  mid = bf16_rne(r1);
  lo = bf16_rne(r1 - bf16_to_f32(mid));
#endif
}

}  // namespace
