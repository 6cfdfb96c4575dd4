// mcp_device.h -- device-side building blocks of the Monte Carlo path kernel (gfx950 only).
//
// SPEC.md sections 2-4: Philox4x32-10 counter layout, the exact-arithmetic inverse-CDF normal transform and
// the key transform used by the radix select.  Every floating-point operation below is an explicit IEEE
// binary32 op (the translation unit is compiled with -ffp-contract=off), so the CPU oracle
// (oracle/mc_oracle.c) reproduces terminal values bit for bit.
//
// The reference has no counterpart for this file (app.py contains no normal draws, SURVEY.md
// section 0.2); the conventions it inherits from the reference are cited in mcp_paths.h.
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
#ifndef MCP_EXP_VMUL
#define MCP_EXP_VMUL 0
#endif
#ifndef MCP_EXP_NORMALS4
#define MCP_EXP_NORMALS4 1
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
  uint32_t m0, m1;      // the two multipliers (MCP_EXP_VMUL pins them in VGPRs)
};
__device__ __forceinline__ PhiloxKeys philox_keys(uint32_t k0, uint32_t k1) {
  PhiloxKeys ks;
#pragma unroll
  for (int r = 0; r < 10; r++) { ks.k0[r] = k0 + (uint32_t)r * PHILOX_W0; ks.k1[r] = k1 + (uint32_t)r * PHILOX_W1; }
  ks.m0 = PHILOX_M0; ks.m1 = PHILOX_M1;
#if MCP_EXP_VMUL
  asm volatile("" : "+v"(ks.m0), "+v"(ks.m1));
#endif
  return ks;
}

// One Philox4x32-10 block: per round two v_mad_u64_u32 and two three-input xors per lane.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              const PhiloxKeys& ks, uint32_t (&x)[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
#if MCP_EXP_VMUL
    const uint64_t p0 = (uint64_t)ks.m0 * c0;
    const uint64_t p1 = (uint64_t)ks.m1 * c2;
#else
    const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
#endif
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, ks.k0[r]);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, ks.k1[r]);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
  }
  x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
}

constexpr float NEG_2LN2 = -0x1.62e43p+0f;       // -2 ln 2 (native Box-Muller only)

constexpr int ICDF_ENTRIES = 1056;   // 33 octaves x 32 mantissa bins of float4 {c0,c1,c2,c3}: 16.5 KiB of LDS per workgroup
constexpr uint32_t ICDF_E_LO = 94;   // u in [2^-33, 1/2]: binary32 exponents 94..126

__device__ __forceinline__ float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// The LDS copy of the table carries ICDF_PAD unused entries in front: u is formed with its exponent pre-scaled by 2^-93
// (binary32 exponent field 1..33 instead of 94..126; same significand, so the same table entry and the same delta), and
// then bits(u) >> 18 indexes the padded table directly -- the spec's subtraction of (94 << 23) costs no instruction.
constexpr int ICDF_PAD = 32;
constexpr int ICDF_LDS_ENTRIES = ICDF_ENTRIES + ICDF_PAD;

// Constants of the transform that must sit in VGPRs: an SGPR (or, in VOP3, any non-inline) operand halves the issue
// rate of a VALU instruction on gfx950 (profiles/r01_valu_rates.txt).
struct IcdfConsts {
  uint32_t m18, m31;
};
__device__ __forceinline__ IcdfConsts icdf_consts() {
  IcdfConsts c = {0x0003ffffu, 0x7fffffffu};
  asm volatile("" : "+v"(c.m18), "+v"(c.m31));
  return c;
}

// (a & m) | (b & ~m) in one v_bfi_b32
__device__ __forceinline__ uint32_t bitselect(uint32_t m, uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}

// Exact-arithmetic normal transform (SPEC.md section 3): one 32-bit word -> one N(0,1) draw by table-driven inverse
// CDF.  bit 31 = sign; v = low 31 bits, u = (v + 1/2) 2^-32; u's exponent and top 5 mantissa bits pick a cubic in the
// remaining 18 mantissa bits: one ds_read_b128 from the LDS copy of the table, three fma, no transcendental.
// `tab` is the PADDED LDS table (entry ICDF_PAD holds T[0]).  10.5 VALU instructions per normal.
__device__ __forceinline__ float normal_icdf(uint32_t x, const float4* tab, const IcdfConsts& k) {
  const float us = fma32((float)(x & 0x7fffffffu), 0x1p-125f, 0x1p-126f);          // u * 2^-93: exponent field 1..33
  const uint32_t b = __float_as_uint(us);
  const float4 c = *(const float4*)((const char*)tab + ((b >> 14) & 0x0003fff0u));   // tab[b >> 18]
  const float dc = __uint_as_float((b & k.m18) | 0x3f800000u) - 0x1.04p+0f;
  float a = fma32(c.w, dc, c.z);
  a = fma32(a, dc, c.y);
  a = fma32(a, dc, c.x);
  return __uint_as_float(bitselect(k.m31, __float_as_uint(a), x));                  // |a| with the sign of bit 31
}

// MCP_FLAG_NATIVE_MATH: Box-Muller on the hardware approximations (v_log_f32, v_sqrt_f32, v_sin_f32, v_cos_f32) of
// word pairs.  Statistically equivalent N(0,1) draws from the same Philox stream, but NOT the spec's normals: results
// are comparable to the oracle only in distribution (tests check moments and Monte-Carlo-level agreement).
__device__ __forceinline__ void box_muller_native(uint32_t xa, uint32_t xb, float& z_sin, float& z_cos) {
  const float u = fma32((float)xa, 0x1p-32f, 0x1p-32f);
  const float t = __builtin_amdgcn_logf(u) * NEG_2LN2;
  const float s = __builtin_amdgcn_sqrtf(t);
  const float turns = (float)xb * 0x1p-32f;
  z_sin = s * __builtin_amdgcn_sinf(turns);
  z_cos = s * __builtin_amdgcn_cosf(turns);
}

// The four normals of one Philox block.
template <bool NATIVE>
__device__ __forceinline__ void block_normals(const uint32_t (&x)[4], const float4* tab, const IcdfConsts& k, float& z0, float& z1,
                                              float& z2, float& z3) {
  if constexpr (NATIVE) {
    box_muller_native(x[0], x[1], z0, z1);
    box_muller_native(x[2], x[3], z2, z3);
  } else {
#if MCP_EXP_NORMALS4
    // the same four transforms with the four table reads issued together, ahead of everything that depends on them
    // (one s_waitcnt per block instead of one per normal), and the four Horner chains interleaved
    uint32_t b[4];
    float4 c[4];
    float dc[4], a[4];
#pragma unroll
    for (int i = 0; i < 4; i++) b[i] = __float_as_uint(fma32((float)(x[i] & 0x7fffffffu), 0x1p-125f, 0x1p-126f));
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = *(const float4*)((const char*)tab + ((b[i] >> 14) & 0x0003fff0u));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; i++) dc[i] = __uint_as_float((b[i] & k.m18) | 0x3f800000u) - 0x1.04p+0f;   // (as v_pk_add_f32 pairs: +4 %)
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = fma32(c[i].w, dc[i], c[i].z);
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = fma32(a[i], dc[i], c[i].y);
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = fma32(a[i], dc[i], c[i].x);
    z0 = __uint_as_float(bitselect(k.m31, __float_as_uint(a[0]), x[0]));
    z1 = __uint_as_float(bitselect(k.m31, __float_as_uint(a[1]), x[1]));
    z2 = __uint_as_float(bitselect(k.m31, __float_as_uint(a[2]), x[2]));
    z3 = __uint_as_float(bitselect(k.m31, __float_as_uint(a[3]), x[3]));
#else
    z0 = normal_icdf(x[0], tab, k);
    z1 = normal_icdf(x[1], tab, k);
    z2 = normal_icdf(x[2], tab, k);
    z3 = normal_icdf(x[3], tab, k);
#endif
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
