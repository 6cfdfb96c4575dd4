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

#ifndef MCP_EXP_SQRT_NEIGHBOUR
#define MCP_EXP_SQRT_NEIGHBOUR 1
#endif
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

constexpr float NEG_2LN2 = -0x1.62e43p+0f;       // -2 ln 2
constexpr float TWO_PI_2M32 = 0x1.921fb6p-30f;   // 2 pi / 2^32
constexpr float PI_1024 = 0x1.921fb6p-9f;        // 2^21 * TWO_PI_2M32 = pi/1024
// sin(a) = a + a^3 S(a^2), cos(a) = 1 - a^2/2 + a^4 C(a^2) on |a| <= pi/4 (tools/fit_coeffs.py); used
// only to BUILD the sin/cos table.
constexpr float SS0 = -0x1.55554p-3f, SS1 = 0x1.1105b4p-7f, SS2 = -0x1.98da62p-13f;
constexpr float CC0 = 0x1.55554ap-5f, CC1 = -0x1.6c0c8cp-10f, CC2 = 0x1.9a0256p-16f;

constexpr int BM_TAB = 1024;   // entries of each Box-Muller table (float2): 8 KiB + 8 KiB of LDS per block

__device__ __forceinline__ float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Correctly rounded sqrt for t in {0} U [2^-24, 64).  Default: v_sqrt_f32 (<= 1 ulp) and the two-neighbour
// residual test LLVM uses for IEEE sqrt (no denormal pre-scaling needed in this range).  Alternative
// (MCP_EXP_SQRT_NEIGHBOUR=0): v_rsq_f32 seed + fma refinement.  Both are checked exhaustively against sqrtf
// over the whole range on the device (tests/test_gpu_parity.py::test_device_sqrt_is_correctly_rounded);
// they time within 1 % of each other.
__device__ __forceinline__ float sqrt_rn(float t) {
#if MCP_EXP_SQRT_NEIGHBOUR
  float s = __builtin_amdgcn_sqrtf(t);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
  const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float e_dn = fma32(-s_dn, s, t);
  const float e_up = fma32(-s_up, s, t);
  s = (e_dn <= 0.0f) ? s_dn : s;
  s = (e_up > 0.0f) ? s_up : s;
  return s;
#else
  const float y = __builtin_amdgcn_rsqf(t);          // +inf at t == 0
  float g = t * y;                                    // ~ sqrt(t)
  float h = y * 0.5f;
  const float e = fma32(-h, g, 0.5f);
  h = fma32(h, e, h);
  g = fma32(g, e, g);
  const float d = fma32(-g, g, t);
  g = fma32(d, h, g);
  return t == 0.0f ? 0.0f : g;                        // 0 * inf = NaN at t == 0
#endif
}

// ---- table construction (SPEC.md section 3.1; run once per device by tables_init_kernel) -----------
// (sin, cos)(2 pi xb / 2^32): exact integer quadrant reduction + fixed fp32 polynomials.
__device__ __forceinline__ void sincos_poly(uint32_t xb, float& sn_out, float& cs_out) {
  const uint32_t y = xb + 0x20000000u;
  const int32_t r = (int32_t)(xb << 2) >> 2;
  const float a = (float)r * TWO_PI_2M32;
  const float a2 = a * a;
  float ps = fma32(a2, SS2, SS1); ps = fma32(a2, ps, SS0);
  const float sn = fma32(a * a2, ps, a);
  float pc = fma32(a2, CC2, CC1); pc = fma32(a2, pc, CC0);
  const float cs = fma32(a2 * a2, pc, fma32(a2, -0.5f, 1.0f));
  const uint32_t kq = y >> 30;
  float vs = (kq & 1u) ? cs : sn;
  float vc = (kq & 1u) ? sn : cs;
  if (kq & 2u) vs = -vs;
  if (kq == 1u || kq == 2u) vc = -vc;
  sn_out = vs + 0.0f;
  cs_out = vc + 0.0f;
}

// ln(x), x in [0.7, 1.42], binary64 atanh series: IEEE +,*,/ only (identical on host and device).
__device__ __forceinline__ double ln_series(double x) {
  const double y = (x - 1.0) / (x + 1.0), y2 = y * y;
  double s = 0.0;
  for (int n = 17; n >= 0; n--) s = s * y2 + 1.0 / (double)(2 * n + 1);
  return 2.0 * y * s;
}

__device__ __forceinline__ float2 log_table_entry(uint32_t j) {
  const uint32_t lo = 0x3f3504f3u + (j << 13);
  float c = __uint_as_float(lo + 0x1000u);
  if (lo <= 0x3f800000u && 0x3f800000u < lo + 0x2000u) c = 1.0f;
  const float inv_c = 1.0f / c;
  const float l2 = (c == 1.0f) ? 0.0f : (float)(-2.0 * ln_series(1.0 / (double)inv_c));
  return make_float2(inv_c, l2);
}

// Exact-arithmetic, table-driven Box-Muller pair (SPEC.md section 3).  (xa, xb) -> (s sin th, s cos th),
// u = fl(xa) 2^-32 + 2^-32, s = sqrt(-2 ln u), th = 2 pi xb 2^-32.  sc / lg point at the LDS copies of the
// tables.  NATIVE: hardware v_log/v_sqrt/v_sin/v_cos approximations, no tables (not bit-reproducible).
template <bool NATIVE>
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, const float2* sc, const float2* lg,
                                           float& z_sin, float& z_cos) {
  const float u = fma32((float)xa, 0x1p-32f, 0x1p-32f);
  if constexpr (NATIVE) {
    const float t = __builtin_amdgcn_logf(u) * NEG_2LN2;
    const float s = __builtin_amdgcn_sqrtf(t);
    const float turns = (float)xb * 0x1p-32f;
    z_sin = s * __builtin_amdgcn_sinf(turns);
    z_cos = s * __builtin_amdgcn_cosf(turns);
  } else {
    // radius: u = 2^k m, m in [sqrt(.5), sqrt(2)); -2 ln u = k(-2 ln 2) + LG[j].y - 2 log1p(r), r = m LG[j].x - 1
    const uint32_t ib = __float_as_uint(u) - 0x3f3504f3u;
    const int32_t k = (int32_t)ib >> 23;
    const uint32_t mant = ib & 0x007fffffu;
    const float m = __uint_as_float(mant + 0x3f3504f3u);
    const float2 e = lg[mant >> 13];
    const float r = fma32(m, e.x, -1.0f);
    const float w = r * (r - 2.0f);
    float t = fma32((float)k, NEG_2LN2, e.y);
    t = t + w;
    const float s = sqrt_rn(t);
    // angle: table bin i = xb >> 22 (midpoint theta_i), residual d in [-pi/1024, pi/1024): sin d ~ d, cos d ~ 1 - d^2/2
    const float2 p = sc[xb >> 22];
    const float d = fma32((float)(xb & 0x003fffffu), TWO_PI_2M32, -PI_1024);
    const float cd = fma32(d * -0.5f, d, 1.0f);
    const float sn = fma32(p.y, d, p.x * cd);
    const float cs = fma32(-p.x, d, p.y * cd);
    z_sin = s * sn;
    z_cos = s * cs;
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
