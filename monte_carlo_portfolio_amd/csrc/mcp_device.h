// mcp_device.h -- device-side building blocks of the Monte Carlo path kernel (gfx950 only).
//
// SPEC.md sections 2-4: Philox4x32-10 counter layout, the exact-arithmetic Box-Muller pair and the
// key transform used by the radix select.  Every floating-point operation below is an explicit IEEE
// binary32 op (the translation unit is compiled with -ffp-contract=off), so the CPU oracle
// (oracle/mc_oracle.c) reproduces terminal values bit for bit.
//
// The reference has no counterpart for this file (app.py contains no normal draws, SURVEY.md
// section 0.2); the conventions it inherits from the reference are cited in mcp_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcp {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u;
constexpr uint32_t PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u;
constexpr uint32_t PHILOX_W1 = 0xBB67AE85u;

#ifndef MCP_EXP_BITOP3
#define MCP_EXP_BITOP3 1
#endif

// a ^ b ^ c in one VALU instruction (v_bitop3_b32, truth table 0x96).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if MCP_EXP_BITOP3
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

// Round keys of Philox4x32-10: key_r = key_0 + r*(W0, W1).  They depend on the seed only.
struct PhiloxKeys {
  uint32_t k0[10], k1[10];
};
__device__ __forceinline__ PhiloxKeys philox_keys(uint32_t k0, uint32_t k1) {
  PhiloxKeys ks;
#pragma unroll
  for (int r = 0; r < 10; r++) { ks.k0[r] = k0 + (uint32_t)r * PHILOX_W0; ks.k1[r] = k1 + (uint32_t)r * PHILOX_W1; }
  return ks;
}

// One Philox4x32-10 block: per round two v_mad_u64_u32 and two three-input xors per lane.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              const PhiloxKeys& ks, uint32_t (&x)[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, ks.k0[r]);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, ks.k1[r]);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
  }
  x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
}

// -2*log1p(f) = -2 f + f^2 Q(f) on [sqrt(.5)-1, sqrt(2)-1]; coefficients from tools/fit_coeffs.py.
constexpr float LQ0 = 0x1.fffff4p-1f, LQ1 = -0x1.5557acp-1f, LQ2 = 0x1.000688p-1f, LQ3 = -0x1.98a664p-2f;
constexpr float LQ4 = 0x1.52fdf6p-2f, LQ5 = -0x1.32c6c8p-2f, LQ6 = 0x1.27c4a8p-2f, LQ7 = -0x1.65b8e2p-3f;
constexpr float NEG_2LN2 = -0x1.62e43p+0f;
// sin(a) = a + a^3 S(a^2), cos(a) = 1 - a^2/2 + a^4 C(a^2) on |a| <= pi/4.
constexpr float SS0 = -0x1.55554p-3f, SS1 = 0x1.1105b4p-7f, SS2 = -0x1.98da62p-13f;
constexpr float CC0 = 0x1.55554ap-5f, CC1 = -0x1.6c0c8cp-10f, CC2 = 0x1.9a0256p-16f;
constexpr float TWO_PI_2M32 = 0x1.921fb6p-30f;

__device__ __forceinline__ float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Correctly rounded sqrt for t in [0, 64): v_sqrt_f32 (<= 1 ulp) followed by the neighbour test LLVM
// uses for IEEE sqrt, without the denormal pre-scaling (t is either 0 or >= 2^-24, never subnormal).
__device__ __forceinline__ float sqrt_rn(float t) {
  float s = __builtin_amdgcn_sqrtf(t);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
  const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float e_dn = fma32(-s_dn, s, t);
  const float e_up = fma32(-s_up, s, t);
  s = (e_dn <= 0.0f) ? s_dn : s;
  s = (e_up > 0.0f) ? s_up : s;
  return s;
}

// Exact-arithmetic Box-Muller pair (SPEC.md section 3).  (xa, xb) -> (s sin(theta), s cos(theta)),
// u = fl(xa) 2^-32 + 2^-32, s = sqrt(-2 ln u), theta = 2 pi xb 2^-32.
template <bool NATIVE>
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float& z_sin, float& z_cos) {
  const float u = fma32((float)xa, 0x1p-32f, 0x1p-32f);
  if constexpr (NATIVE) {
    // hardware approximations: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (input in turns)
    const float t = __builtin_amdgcn_logf(u) * NEG_2LN2;
    const float s = __builtin_amdgcn_sqrtf(t);
    const float turns = (float)xb * 0x1p-32f;
    z_sin = s * __builtin_amdgcn_sinf(turns);
    z_cos = s * __builtin_amdgcn_cosf(turns);
  } else {
    const uint32_t ib = __float_as_uint(u) - 0x3f3504f3u;
    const int32_t k = (int32_t)ib >> 23;
    const float m = __uint_as_float((ib & 0x007fffffu) + 0x3f3504f3u);
    const float f = m - 1.0f;
    float q = LQ7;
    q = fma32(q, f, LQ6); q = fma32(q, f, LQ5); q = fma32(q, f, LQ4); q = fma32(q, f, LQ3);
    q = fma32(q, f, LQ2); q = fma32(q, f, LQ1); q = fma32(q, f, LQ0);
    const float ff = f * f;
    const float tm = fma32(f, -2.0f, ff * q);
    const float t = fma32((float)k, NEG_2LN2, tm);
    const float s = sqrt_rn(t);
    const uint32_t y = xb + 0x20000000u;
    const int32_t r = (int32_t)(xb << 2) >> 2;
    const float a = (float)r * TWO_PI_2M32;
    const float a2 = a * a;
    float ps = fma32(a2, SS2, SS1); ps = fma32(a2, ps, SS0);
    const float sn = fma32(a * a2, ps, a);
    float pc = fma32(a2, CC2, CC1); pc = fma32(a2, pc, CC0);
    const float cs = fma32(a2 * a2, pc, fma32(a2, -0.5f, 1.0f));
    const bool swap = (y & 0x40000000u) != 0u;
    const float vs = swap ? cs : sn;
    const float vc = swap ? sn : cs;
    const uint32_t sign_s = y & 0x80000000u;
    const uint32_t sign_c = (y ^ (y << 1)) & 0x80000000u;
    z_sin = __uint_as_float(__float_as_uint(s) ^ sign_s) * vs;
    z_cos = __uint_as_float(__float_as_uint(s) ^ sign_c) * vc;
  }
}

// Order-preserving map float -> uint32 (ascending floats <-> ascending keys), used by the select.
__host__ __device__ __forceinline__ uint32_t float_to_key(float v) {
  uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = __float_as_uint(v);
#else
  __builtin_memcpy(&b, &v, 4);
#endif
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ __forceinline__ float key_to_float(uint32_t k) {
  const uint32_t b = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(b);
#else
  float v;
  __builtin_memcpy(&v, &b, 4);
  return v;
#endif
}

}  // namespace mcp
