// mcp_paths.h -- the fused Monte Carlo path kernel (template; instantiated per NB in mcp_paths_inst.hip).
//
//   mc_paths_kernel   N1+N2 of SURVEY.md section 8(a): Philox4x32-10 -> Box-Muller -> r = mu + L z ->
//                     rho = w.r -> V <- V(1+rho) over T steps, entirely in registers; writes V_T
//                     (4 B/path, coalesced) and per-block fp64 moment partials.
//                     Conventions inherited from the reference: fixed-weight portfolio return
//                     `returns_df @ ws` (app.py:710), compounding prod(1+r) (app.py:249, app.py:253).
//
// Layout: one lane = one path (PPT independent paths per lane for ILP); the Cholesky factor, drift
// and weights are wave-uniform and are read with scalar loads (s_load_dword*) straight into SGPR
// operands of the FMAs -- they never occupy VGPRs or LDS bandwidth.  Roofline: VALU issue
// (DESIGN.md section 4); HBM traffic is 4 B per path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcport.h"
#include "mcp_device.h"

namespace mcp {

struct PathArgs {
  const float* __restrict__ packed;   // [mu N4][L packed lower N4(N4+1)/2][W K*N4]
  float* __restrict__ terminal;       // [K][stride]
  mcp_moments* __restrict__ partials; // [K][gridDim.x]
  uint64_t seed, path_begin, n_paths, stride;
  int32_t n_steps, n_portfolios, k_begin, compounding;
  float v0;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// x = V_T/V0 - 1 (simple) or expm1(S_T) (log); double, as the host computes it (mcp_terminal_to_x).
__device__ __forceinline__ double terminal_to_x(float term, double v0, int compounding) {
  return compounding == MCP_COMPOUND_LOG ? expm1((double)term) : (double)term / v0 - 1.0;
}

constexpr int PATH_BLOCK = 256;

// NB = N4/4 Philox blocks per path-step; KT portfolios per pass; PPT paths per lane.
template <int NB, int KT, int PPT, bool NATIVE>
__global__ void __launch_bounds__(PATH_BLOCK) mc_paths_kernel(const PathArgs a) {
  constexpr int N4 = 4 * NB;
  // wave-uniform parameters through the constant address space -> s_load_dword* into SGPRs
  typedef const __attribute__((address_space(4))) float* cfloat_p;
  cfloat_p mu = (cfloat_p)a.packed;
  cfloat_p Lp = mu + N4;
  cfloat_p Wk = mu + N4 + N4 * (N4 + 1) / 2 + (size_t)a.k_begin * N4;
  const int kt = min(KT, a.n_portfolios - a.k_begin);   // live portfolios in this pass (uniform)
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const int T = a.n_steps;
  const bool logc = a.compounding == MCP_COMPOUND_LOG;
  const double v0d = (double)a.v0;

  double s1[KT], s2[KT], mn[KT], mx[KT];
  double cnt = 0.0;
#pragma unroll
  for (int k = 0; k < KT; k++) { s1[k] = 0.0; s2[k] = 0.0; mn[k] = __builtin_inf(); mx[k] = -__builtin_inf(); }

  const uint64_t tile = (uint64_t)PATH_BLOCK * PPT;
  const uint64_t n_tiles = (a.n_paths + tile - 1) / tile;
  for (uint64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    uint64_t p[PPT];
    bool live[PPT];
    uint32_t plo[PPT], phi[PPT];
    float V[PPT][KT];
#pragma unroll
    for (int e = 0; e < PPT; e++) {
      p[e] = tl * tile + (uint64_t)e * PATH_BLOCK + threadIdx.x;
      live[e] = p[e] < a.n_paths;
      const uint64_t g = a.path_begin + p[e];
      plo[e] = (uint32_t)g; phi[e] = (uint32_t)(g >> 32);
#pragma unroll
      for (int k = 0; k < KT; k++) V[e][k] = logc ? 0.0f : a.v0;
    }

    for (int t = 0; t < T; t++) {
      // keep the (loop-invariant) parameter loads inside the step: hoisted, they would pin ~170 registers
      asm volatile("" : "+s"(mu), "+s"(Lp), "+s"(Wk));
      float z[PPT][N4];
#pragma unroll
      for (int q = 0; q < NB; q++) {
        const uint32_t blk = (uint32_t)t * NB + q;     // counter.x; counter.y = 0 (T*NB < 2^32)
#pragma unroll
        for (int e = 0; e < PPT; e++) {
          uint32_t x[4];
          philox4x32_10(blk, 0u, plo[e], phi[e], k0, k1, x);
          box_muller<NATIVE>(x[0], x[1], z[e][0 * NB + q], z[e][1 * NB + q]);
          box_muller<NATIVE>(x[2], x[3], z[e][2 * NB + q], z[e][3 * NB + q]);
        }
      }
      // r = mu + L z (row i: acc = mu_i, then j ascending), rho_k = sum_i w_ki r_i (i ascending)
      float rho[PPT][KT];
#pragma unroll
      for (int e = 0; e < PPT; e++)
#pragma unroll
        for (int k = 0; k < KT; k++) rho[e][k] = 0.0f;
#pragma unroll
      for (int i = 0; i < N4; i++) {
        float acc[PPT];
        const float mui = mu[i];
#pragma unroll
        for (int e = 0; e < PPT; e++) acc[e] = mui;
#pragma unroll
        for (int j = 0; j <= i; j++) {
          const float lij = Lp[i * (i + 1) / 2 + j];
#pragma unroll
          for (int e = 0; e < PPT; e++) acc[e] = fma32(lij, z[e][j], acc[e]);
        }
#pragma unroll
        for (int k = 0; k < KT; k++) {
          const float wki = Wk[k * N4 + i];              // rows >= kt are zero-padded by pack_params
#pragma unroll
          for (int e = 0; e < PPT; e++) rho[e][k] = fma32(wki, acc[e], rho[e][k]);
        }
      }
#pragma unroll
      for (int e = 0; e < PPT; e++)
#pragma unroll
        for (int k = 0; k < KT; k++)
          V[e][k] = logc ? (V[e][k] + rho[e][k]) : fma32(V[e][k], rho[e][k], V[e][k]);
    }

#pragma unroll
    for (int e = 0; e < PPT; e++) {
      if (live[e]) {
        cnt += 1.0;
#pragma unroll
        for (int k = 0; k < KT; k++) {
          if (k < kt) {
            a.terminal[(size_t)(a.k_begin + k) * a.stride + p[e]] = V[e][k];
            const double x = terminal_to_x(V[e][k], v0d, a.compounding);
            s1[k] += x; s2[k] += x * x; mn[k] = fmin(mn[k], x); mx[k] = fmax(mx[k], x);
          }
        }
      }
    }
  }

  // block reduction of the moment partials: wave shuffles, then 4 waves through LDS
  __shared__ double red[PATH_BLOCK / 64][5];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double c_w = wave_sum(cnt);
  for (int k = 0; k < kt; k++) {
    double v1 = s1[0], v2 = s2[0], vmn = mn[0], vmx = mx[0];
#pragma unroll
    for (int kk = 1; kk < KT; kk++)
      if (kk == k) { v1 = s1[kk]; v2 = s2[kk]; vmn = mn[kk]; vmx = mx[kk]; }
    v1 = wave_sum(v1); v2 = wave_sum(v2); vmn = wave_min(vmn); vmx = wave_max(vmx);
    __syncthreads();
    if (lane == 0) { red[wv][0] = c_w; red[wv][1] = v1; red[wv][2] = v2; red[wv][3] = vmn; red[wv][4] = vmx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      mcp_moments m = {red[0][0], red[0][1], red[0][2], red[0][3], red[0][4]};
      for (int w = 1; w < PATH_BLOCK / 64; w++) {
        m.n += red[w][0]; m.sum += red[w][1]; m.sumsq += red[w][2];
        m.min = fmin(m.min, red[w][3]); m.max = fmax(m.max, red[w][4]);
      }
      a.partials[(size_t)(a.k_begin + k) * gridDim.x + blockIdx.x] = m;
    }
  }
}

}  // namespace mcp
