// mcp_sweep_paths.hip -- K-portfolio path kernel on the matrix cores (BASELINE configs[4], SURVEY.md section 7
// step 7b): all portfolios see the same normals (common random numbers), so per step the correlated
// returns r[16 x 64 paths] of a wave are drawn ONCE and the K portfolio returns are the dense fp32 product
//     rho[K x 64] = W[K x 16] . r[16 x 64]
// issued as v_mfma_f32_32x32x2_f32: exact fp32, and bit for bit the k-ordered fma chain of SPEC.md section 4
// (rho = fma(w_15, r_15, ... fma(w_0, r_0, 0))), so this kernel and mc_paths_kernel agree bitwise.
//
// Layout per wave: 64 paths (one per lane for the draw) x 32*MT portfolios.
//   B operand of k-step kk for the path tile nt: lanes 0-31 carry r[2kk], lanes 32-63 carry r[2kk+1] of
//   paths 32nt..32nt+31 -- obtained from the lane-per-path registers with ONE v_permlane32_swap per k-step
//   (X = r[2kk], Y = r[2kk+1]: the swap exchanges X[32:63] with Y[0:31]; X becomes tile 0, Y tile 1).
//   A operand: lane l holds W[portfolio 32mt + (l&31)][asset 2kk + (l>>5)], resident in VGPRs for all T steps.
//   C/D: lane l, register g = portfolio 32mt + (g&3) + 8(g>>2) + 4(l>>5), path 32nt + (l&31); V is kept in
//   the same layout for the whole walk (MT*2*16 accumulators) and compounded elementwise, V = fma(V, rho, V).
// The W.r product is matrix-bound (2*K*16 flops per path-step); RNG and GEMV are amortised over 32*MT
// portfolios.  N <= 16 (NB = 1..4).
#include "mcp_paths.h"
#include "mcp_stats_kernels.h"

#ifndef MCP_SWEEP_PART
#error "compile with -DMCP_SWEEP_PART=<0..3> (part p instantiates the shared-draw kernel for NB = 4p+1 .. 4p+4)"
#endif
#define MCP_CAT_(a, b) a##b
#define MCP_CAT(a, b) MCP_CAT_(a, b)

namespace mcp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- lane exchanges of the fused moment reduction (no LDS round trip: DPP inside a 16-lane row, v_permlane16_swap across) ----
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
constexpr int DPP_XOR1 = 0xB1;           // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;           // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;   // lane i <-> 7 - i inside each group of 8
constexpr int DPP_MIRROR = 0x140;        // lane i <-> 15 - i inside the row
// sum / min / max over the 16 lanes of each row, the same value in every lane; fixed order (deterministic)
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_f64<DPP_XOR1>(v); v += dpp_f64<DPP_XOR2>(v); v += dpp_f64<DPP_HALF_MIRROR>(v); v += dpp_f64<DPP_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float row_min(float v) {
  v = fminf(v, dpp_f32<DPP_XOR1>(v)); v = fminf(v, dpp_f32<DPP_XOR2>(v)); v = fminf(v, dpp_f32<DPP_HALF_MIRROR>(v)); v = fminf(v, dpp_f32<DPP_MIRROR>(v));
  return v;
}
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, dpp_f32<DPP_XOR1>(v)); v = fmaxf(v, dpp_f32<DPP_XOR2>(v)); v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f32<DPP_MIRROR>(v));
  return v;
}
// a, b: two values per lane.  v_permlane16_swap exchanges the odd rows of a with the even rows of b; afterwards a + b (min, max)
// is, in the lanes of an EVEN row, the combination of the two a of lane l and lane l + 16, and in the lanes of an ODD row that of
// the two b: the first step of two butterflies at once, without a select.
__device__ __forceinline__ void swap16_f32(float& a, float& b) {
  const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(sw[0]); b = __uint_as_float(sw[1]);
}
__device__ __forceinline__ void swap16_f64(double& a, double& b) {
  const uint64_t ba = (uint64_t)__double_as_longlong(a), bb = (uint64_t)__double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)ba, (uint32_t)bb, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(ba >> 32), (uint32_t)(bb >> 32), false, false);
  a = __longlong_as_double((long long)(((uint64_t)hi[0] << 32) | lo[0]));
  b = __longlong_as_double((long long)(((uint64_t)hi[1] << 32) | lo[1]));
}

// Fused moments of the sweep kernels (N3): the wave holds V for 32 MT portfolios x 64 paths in the MFMA C/D layout (lane l,
// register g of tile (mt, nt): portfolio k_base + 32 mt + (g&3) + 8 (g>>2) + 4 (l>>5), path path0 + 32 nt + (l&31)).  Per
// portfolio: x = V/v0 - 1 (or expm1 S), d = x - c_k, {sum d, sum d^2, min V, max V} over the wave's 64 paths -- the two path
// tiles in the lane, then over the 32 lanes of the half-wave: registers g and g + 8 are reduced together (swap16: lanes of
// even rows end up with g, lanes of odd rows with g + 8), then four DPP steps inside the 16-lane row -- and ONE MomentPartial
// per portfolio and wave tile, slot `tile` (= global 64-path tile index).  Dead paths (>= n_paths) contribute nothing.
template <int MT, bool LOGC>
__device__ __forceinline__ void sweep_moments(const PathArgs& a, const f32x16 (&V)[MT][2], int k_base, uint64_t path0) {
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, odd_row = (lane >> 4) & 1;
  if (path0 >= a.n_paths) return;                          // wave-uniform: a wave tile beyond the range has no slot
  const bool live0 = path0 + col < a.n_paths, live1 = path0 + 32 + col < a.n_paths;
  const uint64_t left = a.n_paths - path0;
  const unsigned long long n_tile = left < 64 ? left : 64;
  const uint64_t tile = path0 / SWEEP_TILE_PATHS;
  const float inf = __builtin_inff();
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
#pragma unroll
    for (int gp = 0; gp < 8; gp++) {
      double s1[2], s2[2];
      float mn[2], mx[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int g = gp + 8 * h;
        const int k = k_base + 32 * mt + (g & 3) + 8 * (g >> 2) + 4 * half;
        const double c = (k < a.n_portfolios && a.pivot) ? a.pivot[k] : 0.0;      // rows beyond K are zero-weight padding
        const float v0 = V[mt][0][g], v1 = V[mt][1][g];
        double x0, x1;
        if constexpr (LOGC) { x0 = expm1((double)v0); x1 = expm1((double)v1); }
        else if (a.v0_pow2) { x0 = __builtin_fma((double)v0, a.inv_v0d, -1.0); x1 = __builtin_fma((double)v1, a.inv_v0d, -1.0); }
        else { x0 = (double)v0 / a.v0d - 1.0; x1 = (double)v1 / a.v0d - 1.0; }
        const double d0 = live0 ? x0 - c : 0.0, d1 = live1 ? x1 - c : 0.0;
        s1[h] = d0 + d1;
        s2[h] = __builtin_fma(d0, d0, d1 * d1);
        mn[h] = fminf(live0 ? v0 : inf, live1 ? v1 : inf);
        mx[h] = fmaxf(live0 ? v0 : -inf, live1 ? v1 : -inf);
      }
      swap16_f64(s1[0], s1[1]); swap16_f64(s2[0], s2[1]); swap16_f32(mn[0], mn[1]); swap16_f32(mx[0], mx[1]);
      const double r1 = row_sum(s1[0] + s1[1]), r2 = row_sum(s2[0] + s2[1]);
      const float rmn = row_min(fminf(mn[0], mn[1])), rmx = row_max(fmaxf(mx[0], mx[1]));
      const int g = gp + 8 * odd_row;                      // what this row reduced
      const int k = k_base + 32 * mt + (g & 3) + 8 * (g >> 2) + 4 * half;
      if ((lane & 15) == 0 && k < a.n_portfolios) {
        MomentPartial m;
        m.s1 = r1; m.s2 = r2; m.vmin = rmn; m.vmax = rmx; m.n = n_tile;
        a.partials[(size_t)k * a.slots + tile] = m;
      }
    }
  }
}

#if MCP_SWEEP_PART == 0
template <int NB, int MT, bool NATIVE, bool LOGC>
__global__ void __launch_bounds__(PATH_BLOCK, 2) mc_sweep_kernel(const PathArgs a) {
  constexpr int N4 = 4 * NB, KS = N4 / 2;   // KS k-steps of 2 assets
  typedef const __attribute__((address_space(4))) float* cfloat_p;
  cfloat_p mu = (cfloat_p)a.packed;
  cfloat_p Lp = mu + N4;
  const float* __restrict__ Wg = a.packed + N4 + N4 * (N4 / 2 + 1);   // [Kpad][N4], rows >= K are zero

  __shared__ float4 s_tab[ICDF_LDS_ENTRIES];
  if constexpr (!NATIVE) {
    for (int i = threadIdx.x; i < ICDF_ENTRIES; i += PATH_BLOCK) s_tab[ICDF_PAD + i] = a.tables[i];
    __syncthreads();
  }
  const IcdfConsts kc = icdf_consts();
  const PhiloxKeys ks = philox_keys((uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t p = ((uint64_t)blockIdx.x * (PATH_BLOCK / 64) + wave) * 64 + lane;     // local path of this lane (draw)
  const uint64_t g = a.path_begin + p;
  const uint32_t plo = (uint32_t)g, phi = (uint32_t)(g >> 32);
  const int k_base = a.k_begin + blockIdx.y * 32 * MT;
  constexpr bool logc = LOGC;      // compile-time: a run-time flag turns the compounding into fma + add + select

  float areg[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int kk = 0; kk < KS; kk++)
      areg[mt][kk] = Wg[(size_t)(k_base + 32 * mt + (lane & 31)) * N4 + 2 * kk + (lane >> 5)];

  f32x16 V[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) V[mt][nt][r] = logc ? 0.0f : a.v0;

  for (int t = 0; t < a.n_steps; t++) {
    asm volatile("" : "+s"(mu), "+s"(Lp));
    float z[N4];
#pragma unroll
    for (int q = 0; q < NB; q++) {
      uint32_t x[4];
      philox4x32_10((uint32_t)t * NB + q, 0u, plo, phi, ks, x);
      block_normals<NATIVE>(x, s_tab, kc, z[0 * NB + q], z[1 * NB + q], z[2 * NB + q], z[3 * NB + q]);
    }
    float r[N4];
#pragma unroll
    for (int m = 0; m < N4 / 2; m++) {
      f32x2 acc = {mu[2 * m], mu[2 * m + 1]};
#pragma unroll
      for (int j = 0; j <= 2 * m + 1; j++) {
        const f32x2 l2 = {Lp[2 * m * (m + 1) + 2 * j], Lp[2 * m * (m + 1) + 2 * j + 1]};
        acc = __builtin_elementwise_fma(l2, (f32x2){z[j], z[j]}, acc);
      }
      r[2 * m] = acc.x;
      r[2 * m + 1] = acc.y;
    }
    // lane-per-path registers -> MFMA B operands of both 32-path tiles
    float b0[KS], b1[KS];
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[2 * kk]), __float_as_uint(r[2 * kk + 1]), false, false);
      b0[kk] = __uint_as_float(sw[0]);
      b1[kk] = __uint_as_float(sw[1]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        f32x16 rho = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
          rho = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[mt][kk], nt ? b1[kk] : b0[kk], rho, 0, 0, 0);
        if constexpr (logc) V[mt][nt] = V[mt][nt] + rho;
        else V[mt][nt] = __builtin_elementwise_fma(V[mt][nt], rho, V[mt][nt]);      // 8 v_pk_fma_f32
      }
    }
  }

  const uint64_t wave_path0 = ((uint64_t)blockIdx.x * (PATH_BLOCK / 64) + wave) * 64;
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const uint64_t path = wave_path0 + 32 * nt + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int k = k_base + 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (path < a.n_paths && k < a.n_portfolios) a.terminal[(size_t)k * a.stride + path] = V[mt][nt][q];
      }
    }
  if (a.partials) sweep_moments<MT, LOGC>(a, V, k_base, wave_path0);
}

#endif  // MCP_SWEEP_PART == 0

// ---- shared-draw variant: the four waves of a workgroup own the SAME 64 paths and 4 x 32 MT portfolios.
// The per-step draw is split four ways and exchanged through LDS: wave w runs Philox blocks q = w, w+4, ...
// and their normals (z rows to LDS), then row pairs m = w, w+4, ... of the GEMV (r rows to LDS); after
// the second barrier every wave reads its MFMA B operands straight from the r image (lane l: r[2kk + (l>>5)]
// [32nt + (l&31)], conflict-free), so no permlane is needed.  Two barriers per step suffice without double buffering:
// z(t+1) is written after barrier 2 of step t (all reads of z(t) precede it), r(t+1) after barrier 1 of step t+1
// (every wave loads its B operands of step t before reaching it).  Same arithmetic, same order: bit-identical to
// mc_sweep_kernel and to the oracle.  MT = 4 or 2 (512 / 256 portfolios per workgroup) for N <= 16, MT = 2 or 1 (256 / 128) up to
// N = 64; the smaller workgroup takes the remainder behind whole big ones (mcp_api.cpp: sweep_plan).
#ifndef MCP_SHARED_WAVES_SMALL
#define MCP_SHARED_WAVES_SMALL 4     // __launch_bounds__ minimum waves per SIMD of the shared kernel with MT <= 2, N <= 16: 128 VGPRs (56 B of
                                     // spills outside the step loop) at 4 waves measured 112.0 against 108.2 TFLOP/s at 3 waves / 164 VGPRs
                                     // (K = 256 x 262,144 paths, profiles/r03_lab_sweep256.txt)
#endif
template <int NB, int MT, bool NATIVE, bool LOGC>
__global__ void __launch_bounds__(PATH_BLOCK, (MT <= 2 && NB <= 4) ? MCP_SHARED_WAVES_SMALL : 2) mc_sweep_shared_kernel(const PathArgs a) {
  constexpr int N4 = 4 * NB, KS = N4 / 2;
  typedef const __attribute__((address_space(4))) float* cfloat_p;
  cfloat_p mu = (cfloat_p)a.packed;
  cfloat_p Lp = mu + N4;
  const float* __restrict__ Wg = a.packed + N4 + N4 * (N4 / 2 + 1);

  // MCP_SWEEP_PIPE=1 (experiment, N <= 16; measured: bit-identical, 0.4 % SLOWER, profiles/r03_lab_pipe.txt -- the barriers are
  // not where the time goes): the step software-pipelined over three steps -- at iteration t a wave draws its share of step t+2,
  // does its row pairs of step t+1 and the matrix work of step t -- so that every hand-off between waves crosses exactly ONE
  // barrier (double-buffered z and r images) instead of two barriers per step.
#ifndef MCP_SWEEP_PIPE
#define MCP_SWEEP_PIPE 0
#endif
  constexpr bool PIPE = MCP_SWEEP_PIPE && NB <= 4;
  __shared__ float4 s_tab[ICDF_LDS_ENTRIES];
  __shared__ float s_z[PIPE ? 2 : 1][N4][64], s_r[PIPE ? 2 : 1][N4][64];
  if constexpr (!NATIVE) {
    for (int i = threadIdx.x; i < ICDF_ENTRIES; i += PATH_BLOCK) s_tab[ICDF_PAD + i] = a.tables[i];
  }
  __syncthreads();
  const IcdfConsts kc = icdf_consts();
  const PhiloxKeys ks = philox_keys((uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t p = (uint64_t)blockIdx.x * 64 + lane;             // all four waves: the same 64 paths
  const uint64_t g = a.path_begin + p;
  const uint32_t plo = (uint32_t)g, phi = (uint32_t)(g >> 32);
  const int k_base = a.k_begin + (blockIdx.y * 4 + wave) * 32 * MT;
  constexpr bool logc = LOGC;      // compile-time: a run-time flag turns the compounding into fma + add + select

  float areg[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int kk = 0; kk < KS; kk++)
      areg[mt][kk] = Wg[(size_t)(k_base + 32 * mt + (lane & 31)) * N4 + 2 * kk + (lane >> 5)];

  f32x16 V[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) V[mt][nt][r] = logc ? 0.0f : a.v0;

  // phase A: this wave's share of the normals of step t -> z image
  auto phase_a = [&](int t, float (*zb)[64]) {
#pragma unroll
    for (int q0 = 0; q0 < NB; q0 += 4) {
      const int q = q0 + wave;                                      // wave-uniform
      if (q < NB) {
        uint32_t x[4];
        philox4x32_10((uint32_t)t * NB + q, 0u, plo, phi, ks, x);
        float z0, z1, z2, z3;
        block_normals<NATIVE>(x, s_tab, kc, z0, z1, z2, z3);
        zb[0 * NB + q][lane] = z0;
        zb[1 * NB + q][lane] = z1;
        zb[2 * NB + q][lane] = z2;
        zb[3 * NB + q][lane] = z3;
      }
    }
  };
  // phase B: this wave's row pairs of r = mu + L z -> r image
  auto phase_b = [&](const float (*zb)[64], float (*rb)[64]) {
#pragma unroll
    for (int m0 = 0; m0 < N4 / 2; m0 += 4) {
#pragma unroll
      for (int wv = 0; wv < 4; wv++) {                              // unrolled so that m is a compile-time constant
        const int m = m0 + wv;
        // pair m costs 2m + 2 fmas: within a group of 8 pairs wave w takes pairs w and 7 - w (18 fmas for every wave) instead of
        // w and w + 4 (12 ... 24): +0.25 % at N = 16, +1.0 % at N = 64 (profiles/r03_lab_wide.txt, "unbal" = the old assignment)
        const int owner = (m0 & 4) ? 3 - wv : wv;
        if (m < N4 / 2 && owner == wave) {
          f32x2 acc = {mu[2 * m], mu[2 * m + 1]};
#pragma unroll
          for (int j = 0; j <= 2 * m + 1; j++) {
            const f32x2 l2 = {Lp[2 * m * (m + 1) + 2 * j], Lp[2 * m * (m + 1) + 2 * j + 1]};
            const float zj = zb[j][lane];
            acc = __builtin_elementwise_fma(l2, (f32x2){zj, zj}, acc);
          }
          rb[2 * m][lane] = acc.x;
          rb[2 * m + 1][lane] = acc.y;
        }
      }
    }
  };
  // phase C: rho = W . r on the matrix cores, then compounding
  auto phase_c = [&](const float (*rb)[64]) {
    float b[2][KS];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int kk = 0; kk < KS; kk++) b[nt][kk] = rb[2 * kk + (lane >> 5)][32 * nt + (lane & 31)];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        f32x16 rho = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kk = 0; kk < KS; kk++) rho = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[mt][kk], b[nt][kk], rho, 0, 0, 0);
        if constexpr (logc) V[mt][nt] = V[mt][nt] + rho;
        else V[mt][nt] = __builtin_elementwise_fma(V[mt][nt], rho, V[mt][nt]);      // 8 v_pk_fma_f32
      }
    }
  };

  const int T = a.n_steps;
  if constexpr (PIPE) {
    if (T > 0) phase_a(0, s_z[0]);
    if (T > 1) phase_a(1, s_z[1]);
    __syncthreads();
    if (T > 0) phase_b(s_z[0], s_r[0]);
    __syncthreads();
    for (int t = 0; t < T; t++) {
      asm volatile("" : "+s"(mu), "+s"(Lp));
      if (t + 2 < T) phase_a(t + 2, s_z[t & 1]);                 // z(t+2) replaces z(t), whose last readers passed the barrier below
      if (t + 1 < T) phase_b(s_z[(t + 1) & 1], s_r[(t + 1) & 1]);
      phase_c(s_r[t & 1]);
      __syncthreads();
    }
  } else {
    for (int t = 0; t < T; t++) {
      asm volatile("" : "+s"(mu), "+s"(Lp));
      phase_a(t, s_z[0]);
#ifndef MCP_EXP_NOBARRIER     // experiment only (wrong results): what the two barriers per step cost
      __syncthreads();
#endif
      phase_b(s_z[0], s_r[0]);
#ifndef MCP_EXP_NOBARRIER
      __syncthreads();
#endif
      phase_c(s_r[0]);
    }
  }

  const uint64_t path0 = (uint64_t)blockIdx.x * 64;
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const uint64_t path = path0 + 32 * nt + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int k = k_base + 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (path < a.n_paths && k < a.n_portfolios) a.terminal[(size_t)k * a.stride + path] = V[mt][nt][q];
      }
    }
  if (a.partials) sweep_moments<MT, LOGC>(a, V, k_base, path0);
}

// Workgroups of 128 MT portfolios: MT = 4 or 2 for N <= 16 (512 / 256), MT = 2 or 1 for 16 < N <= 64 (256 / 128).
template <int NB, int MT>
static hipError_t go_shared_mt(bool native, const PathArgs& args, hipStream_t stream) {
  const dim3 grid((unsigned)((args.n_paths + 63) / 64), (unsigned)((args.k_count + 128 * MT - 1) / (128 * MT)));
  const bool lg = args.compounding == MCP_COMPOUND_LOG;
  if (native) { if (lg) mc_sweep_shared_kernel<NB, MT, true, true><<<grid, PATH_BLOCK, 0, stream>>>(args); else mc_sweep_shared_kernel<NB, MT, true, false><<<grid, PATH_BLOCK, 0, stream>>>(args); }
  else { if (lg) mc_sweep_shared_kernel<NB, MT, false, true><<<grid, PATH_BLOCK, 0, stream>>>(args); else mc_sweep_shared_kernel<NB, MT, false, false><<<grid, PATH_BLOCK, 0, stream>>>(args); }
  return hipGetLastError();
}
template <int NB>
static hipError_t go_shared(int mt, bool native, const PathArgs& args, hipStream_t stream) {
  constexpr int BIG = NB <= 4 ? 4 : 2;
  if (mt == BIG) return go_shared_mt<NB, BIG>(native, args, stream);
  if (mt == BIG / 2) return go_shared_mt<NB, BIG / 2>(native, args, stream);
  return hipErrorInvalidValue;
}

// W rows must be zero-padded to a multiple of 512 (mcp_pack_params pads to K_PAD).
hipError_t MCP_CAT(launch_sweep_shared_p, MCP_SWEEP_PART)(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream) {
  switch (nb) {
#define MCP_CASE(n) case n: return go_shared<n>(mt, native, args, stream);
#if MCP_SWEEP_PART == 0
    MCP_CASE(1) MCP_CASE(2) MCP_CASE(3) MCP_CASE(4)
#elif MCP_SWEEP_PART == 1
    MCP_CASE(5) MCP_CASE(6) MCP_CASE(7) MCP_CASE(8)
#elif MCP_SWEEP_PART == 2
    MCP_CASE(9) MCP_CASE(10) MCP_CASE(11) MCP_CASE(12)
#else
    MCP_CASE(13) MCP_CASE(14) MCP_CASE(15) MCP_CASE(16)
#endif
#undef MCP_CASE
    default: return hipErrorInvalidValue;
  }
}

#if MCP_SWEEP_PART == 0
hipError_t launch_sweep_shared(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream) {
  if (nb < 1 || nb > 16) return hipErrorInvalidValue;
  switch ((nb - 1) / 4) {
    case 0: return launch_sweep_shared_p0(nb, mt, native, args, stream);
    case 1: return launch_sweep_shared_p1(nb, mt, native, args, stream);
    case 2: return launch_sweep_shared_p2(nb, mt, native, args, stream);
    default: return launch_sweep_shared_p3(nb, mt, native, args, stream);
  }
}

// grid.x = ceil(n_paths / 256), grid.y = ceil(K / (32 MT))
template <int NB>
static hipError_t go_nb(int mt, bool native, const PathArgs& args, const dim3 grid, hipStream_t stream) {
#define MCP_GO(M, NAT)                                                                               \
  do {                                                                                               \
    if (args.compounding == MCP_COMPOUND_LOG) mc_sweep_kernel<NB, M, NAT, true><<<grid, PATH_BLOCK, 0, stream>>>(args); \
    else mc_sweep_kernel<NB, M, NAT, false><<<grid, PATH_BLOCK, 0, stream>>>(args);                  \
  } while (0)
  if (mt == 1) { if (native) MCP_GO(1, true); else MCP_GO(1, false); }
  else if (mt == 2) { if (native) MCP_GO(2, true); else MCP_GO(2, false); }
  else if (mt == 4) { if (native) MCP_GO(4, true); else MCP_GO(4, false); }
  else return hipErrorInvalidValue;
#undef MCP_GO
  return hipGetLastError();
}

hipError_t launch_sweep_paths(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream) {
  const unsigned gx = (unsigned)((args.n_paths + PATH_BLOCK - 1) / PATH_BLOCK);
  const unsigned gy = (unsigned)((args.k_count + 32 * mt - 1) / (32 * mt));
  const dim3 grid(gx, gy);
  switch (nb) {
    case 1: return go_nb<1>(mt, native, args, grid, stream);
    case 2: return go_nb<2>(mt, native, args, grid, stream);
    case 3: return go_nb<3>(mt, native, args, grid, stream);
    case 4: return go_nb<4>(mt, native, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

#endif  // MCP_SWEEP_PART == 0

}  // namespace mcp
