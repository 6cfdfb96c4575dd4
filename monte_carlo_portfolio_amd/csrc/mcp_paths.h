// mcp_paths.h -- the fused Monte Carlo path kernel (template; instantiated per NB in mcp_paths_inst.hip).
//
//   mc_paths_kernel   N1+N2+N3 of SURVEY.md section 8(a): Philox4x32-10 -> normals (inverse CDF) -> r = mu + L z ->
//                     rho = w.r -> V <- V(1+rho) over T steps, entirely in registers; writes V_T
//                     (4 B/path, coalesced).  Epilogue (when the launch carries a statistics workspace): with V still in
//                     registers, x = V/v0 - 1, the shifted moments {n, sum (x-c), sum (x-c)^2, min, max} in fp64 by
//                     wavefront shuffle reduction -> ONE partial per workgroup and portfolio, and the digit-0 histogram
//                     of the radix select in LDS -> global.  The select's two remaining digits are streaming passes.
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
#include "mcp_stats_kernels.h"

#ifndef MCP_MIN_WAVES
#define MCP_MIN_WAVES 6     // __launch_bounds__ 2nd argument for N <= 16, one portfolio: at least 6 waves/SIMD (the kernel needs 74
                            // VGPRs; forced into 72 for 7 waves it spills 24 B per lane outside the loop and is 0.5 % slower)
#endif
#ifndef MCP_MIN_WAVES_BIG
#define MCP_MIN_WAVES_BIG 1 // the same for 16 < N <= 64, one portfolio.  4 (<= 128 VGPRs) was measured: the 64 live normals plus the
                            // Philox / transform state do not fit, 108-148 B per lane spill into the step loop, -10 % (profiles/r03_lab_n64.txt)
#endif
#ifndef MCP_EXP_VKEYS
#define MCP_EXP_VKEYS 1
#endif
#ifndef MCP_EXP_LDSPAR      // 1: drift from LDS (default); 2: drift and (one portfolio) weights from LDS; 0: both from SGPRs
#define MCP_EXP_LDSPAR 1
#endif

namespace mcp {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef MCP_DIAG_CLOCK
// Diagnostic build only (tools/clock_probe.py; MI355X_MICROARCH.md, DVFS give-back item 6): every workgroup of mc_paths_kernel
// stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its step loops into a buffer of its own that nothing
// else reads; in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz.  No stamp executes in the product build.
extern __device__ unsigned long long mcp_diag_stamps[2 * 8192];
#endif

struct PathArgs {
  const float* __restrict__ packed;   // [mu N4][L row pairs N4(N4/2+1)][W Kpad*N4]  (mcp_pack_params)
  float* __restrict__ terminal;       // [K][stride]
  const float4* __restrict__ tables;  // [ICDF_ENTRIES] inverse-CDF coefficient table (device copy of mcp_icdf_table.inc)
  // fused statistics epilogue (all three NULL: terminal values only)
  const double* __restrict__ pivot;   // [K] shift c of the moments (mcp_pivots), NULL = 0
  MomentPartial* __restrict__ partials;   // [K][slots]
  unsigned long long* __restrict__ hist;  // [K][2][MCP_SELECT_BINS]: digit-0 histogram into [k][0] (K <= 16 kernels only)
  uint64_t slots;                     // MomentPartial slots per portfolio (= gridDim.x of mc_paths_kernel, n/64 tiles of the sweeps)
  double v0d;                         // (double)(float)v0
  double inv_v0d;                     // 1 / v0d; exact when v0 is a power of two (v0_pow2), then x = fma(V, inv_v0d, -1) IS V/v0 - 1
  int32_t v0_pow2;
  uint64_t seed, path_begin, n_paths, stride;
  int32_t n_steps, n_portfolios, k_begin, compounding;
  int32_t k_count;                    // sweep kernels: portfolios [k_begin, k_begin + k_count) belong to this launch
  float v0;
  uint32_t fold_offset;               // float index of [c, v_0 .. v_{N4-1}] (portfolio 0 folded through L, SPEC.md 4.1)
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

__device__ __forceinline__ float wave_minf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// h[digit] += 1 for the active lanes.  Terminal values cluster (V_T ~ 1 +- 0.2 hits a handful of digit-0 bins), and 64
// lanes on one LDS address serialise; so up to four distinct digits per wave are counted by ballot and added once.
// Every lane of the wave must call it (ballots inside).
__device__ __forceinline__ void lds_hist_add(uint32_t* h, uint32_t digit, bool active) {
  unsigned long long todo = __ballot(active);
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int it = 0; it < 4 && todo; it++) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)digit, leader);
    const bool same = active && digit == d0;
    const unsigned long long m = __ballot(same);
    if (lane == leader) atomicAdd(&h[d0], (uint32_t)__popcll(m));
    todo &= ~m;
    active = active && !same;
  }
  if (active) atomicAdd(&h[digit], 1u);
}

// x = V_T/V0 - 1 (simple) or expm1(S_T) (log); double, as the host computes it (mcp_terminal_to_x).
__device__ __forceinline__ double terminal_to_x(float term, double v0, int compounding) {
  return compounding == MCP_COMPOUND_LOG ? expm1((double)term) : (double)term / v0 - 1.0;
}

constexpr int PATH_BLOCK = 256;

// NB = N4/4 Philox blocks per path-step; KT portfolios per pass; PPT paths per lane; FOLD: rho = c + v.z with
// v = L^T w precomputed on the host (SPEC.md 4.1, one portfolio) instead of the triangular GEMV.
// LOGC: compounding mode at compile time (as a run-time flag the compiler if-converts the step into fma + add + select).
template <int NB, int KT, int PPT, bool NATIVE, bool FOLD = false, bool LOGC = false>
__global__ void __launch_bounds__(PATH_BLOCK, (NB <= 4 && KT == 1 && PPT == 1) ? MCP_MIN_WAVES : ((KT == 1 && PPT == 1) ? MCP_MIN_WAVES_BIG : 1))
mc_paths_kernel(const PathArgs a) {
  constexpr int N4 = 4 * NB;
  // wave-uniform parameters through the constant address space -> s_load_dword* into SGPRs
  typedef const __attribute__((address_space(4))) float* cfloat_p;
  cfloat_p mu = (cfloat_p)a.packed;
  cfloat_p Lp = mu + N4;
  cfloat_p Wk = mu + N4 + N4 * (N4 / 2 + 1) + (size_t)a.k_begin * N4;
  const int kt = min(KT, a.n_portfolios - a.k_begin);   // live portfolios in this pass (uniform)
  // inverse-CDF table: 16.5 KiB of LDS per block, filled once from the device-resident copy
  __shared__ float4 s_tab[ICDF_LDS_ENTRIES];
  if constexpr (!NATIVE) {
    for (int i = threadIdx.x; i < ICDF_ENTRIES; i += PATH_BLOCK) s_tab[ICDF_PAD + i] = a.tables[i];
  }
  // The 512 B of padding in front of the table hold the drift (and, for one portfolio, the weights): read from LDS they
  // land in VGPRs without a VALU instruction (a v_mov from an SGPR costs an issue slot, an SGPR operand halves the
  // issue rate of the weight-dot FMAs).
  constexpr bool LDS_MU = MCP_EXP_LDSPAR >= 1 && !NATIVE && !FOLD;
  constexpr bool LDS_W = MCP_EXP_LDSPAR >= 2 && !NATIVE && !FOLD && KT == 1;
  float* const s_par0 = (float*)&s_tab[0];
  if constexpr (LDS_MU) {
    if (threadIdx.x < N4) { s_par0[threadIdx.x] = mu[threadIdx.x]; if (LDS_W) s_par0[N4 + threadIdx.x] = Wk[threadIdx.x]; }
  }
  // statistics epilogue (N3): per-wave moment accumulators and the digit-0 histogram of one portfolio at a time
  __shared__ uint32_t s_hist[MCP_SELECT_BINS];
  __shared__ double s_mom[PATH_BLOCK / 64][KT][2];
  __shared__ float s_ext[PATH_BLOCK / 64][KT][2];
  __shared__ unsigned long long s_cnt[PATH_BLOCK / 64];
  // The epilogue's own arguments (pivot, partials, hist, slots, v0d: 13 dwords) are read from the kernel-argument segment
  // AFTER the step loop, through a pointer the compiler cannot see through: loaded up front they would sit in SGPRs for the
  // whole walk, and the kernel has none to spare (the Cholesky factor is fed from SGPRs): they spilled into VGPR lanes.
  typedef const __attribute__((address_space(4))) PathArgs* cargs_p;
  cargs_p kargs = (cargs_p)__builtin_amdgcn_kernarg_segment_ptr();
  for (int i = threadIdx.x; i < MCP_SELECT_BINS; i += PATH_BLOCK) s_hist[i] = 0u;
  if (threadIdx.x < (PATH_BLOCK / 64) * KT) {
    (&s_mom[0][0][0])[2 * threadIdx.x] = 0.0; (&s_mom[0][0][0])[2 * threadIdx.x + 1] = 0.0;
    (&s_ext[0][0][0])[2 * threadIdx.x] = __builtin_inff(); (&s_ext[0][0][0])[2 * threadIdx.x + 1] = -__builtin_inff();
  }
  if (threadIdx.x < PATH_BLOCK / 64) s_cnt[threadIdx.x] = 0ull;
  __syncthreads();
  const IcdfConsts kc = icdf_consts();
  PhiloxKeys ks = philox_keys((uint32_t)a.seed, (uint32_t)(a.seed >> 32));
#if MCP_EXP_VKEYS
  // pin the 20 round keys in VGPRs: an SGPR operand halves the issue rate of the xor (profiles/r01_valu_rates.txt)
#pragma unroll
  for (int r = 0; r < 10; r++) { asm volatile("" : "+v"(ks.k0[r])); asm volatile("" : "+v"(ks.k1[r])); }
#endif
  const int T = a.n_steps;
  constexpr bool logc = LOGC;

  const uint64_t tile = (uint64_t)PATH_BLOCK * PPT;
  const uint64_t n_tiles = (a.n_paths + tile - 1) / tile;
#ifdef MCP_DIAG_CLOCK
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
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
      uint32_t par_off = 0;                                        // opaque zero: keeps the LDS reads inside the step too
      if constexpr (LDS_MU) asm volatile("" : "+v"(par_off));
      const float* s_par = s_par0 + par_off;
      float z[PPT][N4];
#pragma unroll
      for (int q = 0; q < NB; q++) {
        const uint32_t blk = (uint32_t)t * NB + q;     // counter.x; counter.y = 0 (T*NB < 2^32)
#pragma unroll
        for (int e = 0; e < PPT; e++) {
          uint32_t x[4];
          philox4x32_10(blk, 0u, plo[e], phi[e], ks, x);
          block_normals<NATIVE>(x, s_tab, kc, z[e][0 * NB + q], z[e][1 * NB + q], z[e][2 * NB + q], z[e][3 * NB + q]);
        }
      }
      float rho[PPT][KT];
      if constexpr (FOLD) {
        cfloat_p fv = mu + a.fold_offset;
        asm volatile("" : "+s"(fv));
#pragma unroll
        for (int e = 0; e < PPT; e++) {
          float acc = fv[0];
#pragma unroll
          for (int j = 0; j < N4; j++) acc = fma32(fv[1 + j], z[e][j], acc);
          rho[e][0] = acc;
        }
      } else {
      // r = mu + L z (row i: acc = mu_i, then j ascending), rho_k = sum_i w_ki r_i (i ascending)
#pragma unroll
      for (int e = 0; e < PPT; e++)
#pragma unroll
        for (int k = 0; k < KT; k++) rho[e][k] = 0.0f;
      // Rows are processed in pairs (2m, 2m+1): one v_pk_fma_f32 per column does both rows, its L operand
      // an SGPR pair straight from the row-pair-interleaved parameter block, z_j broadcast by op_sel.
#pragma unroll
      for (int m = 0; m < N4 / 2; m++) {
        f32x2 acc[PPT];
        f32x2 mu2;
        if constexpr (LDS_MU) mu2 = *(const f32x2*)&s_par[2 * m];
        else mu2 = f32x2{mu[2 * m], mu[2 * m + 1]};
#pragma unroll
        for (int e = 0; e < PPT; e++) acc[e] = mu2;
#pragma unroll
        for (int j = 0; j <= 2 * m + 1; j++) {
          const f32x2 l2 = {Lp[2 * m * (m + 1) + 2 * j], Lp[2 * m * (m + 1) + 2 * j + 1]};   // (L[2m][j], L[2m+1][j])
#pragma unroll
          for (int e = 0; e < PPT; e++) acc[e] = __builtin_elementwise_fma(l2, (f32x2){z[e][j], z[e][j]}, acc[e]);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int i = 2 * m + h;
#pragma unroll
          for (int k = 0; k < KT; k++) {
            const float wki = LDS_W ? s_par[N4 + i] : Wk[k * N4 + i];   // rows >= kt are zero-padded by pack_params
#pragma unroll
            for (int e = 0; e < PPT; e++) rho[e][k] = fma32(wki, h ? acc[e].y : acc[e].x, rho[e][k]);
          }
        }
      }
      }  // !FOLD
#pragma unroll
      for (int e = 0; e < PPT; e++)
#pragma unroll
        for (int k = 0; k < KT; k++)
          V[e][k] = logc ? (V[e][k] + rho[e][k]) : fma32(V[e][k], rho[e][k], V[e][k]);
    }

#pragma unroll
    for (int e = 0; e < PPT; e++) {
      if (live[e]) {
#pragma unroll
        for (int k = 0; k < KT; k++)
          if (k < kt) a.terminal[(size_t)(a.k_begin + k) * a.stride + p[e]] = V[e][k];
      }
    }

    // ---- fused statistics epilogue: V is still in registers ----
    asm volatile("" : "+s"(kargs));
    if (kargs->partials != nullptr) {                      // wave-uniform (kernel argument)
      const double* __restrict__ e_pivot = kargs->pivot;
      unsigned long long* __restrict__ e_hist = kargs->hist;
      const double e_v0d = kargs->v0d;
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));                        // nothing derived from it (LDS addresses) is hoisted above the step loop
      const int lane = tid & 63, wv = tid >> 6;
      unsigned long long cnt = 0;
#pragma unroll
      for (int e = 0; e < PPT; e++) cnt += (unsigned long long)__popcll(__ballot(live[e]));
      if (lane == 0) s_cnt[wv] += cnt;
#pragma unroll 1
      for (int k = 0; k < kt; k++) {
        const double c = e_pivot ? e_pivot[a.k_begin + k] : 0.0;
        double d1 = 0.0, d2 = 0.0;
        float mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
        for (int e = 0; e < PPT; e++) {
          float v = V[e][0];
#pragma unroll
          for (int kk = 1; kk < KT; kk++) v = (kk == k) ? V[e][kk] : v;      // register select (k is a run-time index)
          if (live[e]) {
            const double d = terminal_to_x(v, e_v0d, logc ? MCP_COMPOUND_LOG : MCP_COMPOUND_SIMPLE) - c;
            d1 += d;
            d2 = __builtin_fma(d, d, d2);
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
          }
          if (e_hist) lds_hist_add(s_hist, float_to_key(v) >> 21, live[e]);
        }
        d1 = wave_sum(d1); d2 = wave_sum(d2); mn = wave_minf(mn); mx = wave_maxf(mx);
        if (lane == 0) {                                   // this wave's own slot: no race, fixed order over the tiles
          s_mom[wv][k][0] += d1; s_mom[wv][k][1] += d2;
          s_ext[wv][k][0] = fminf(s_ext[wv][k][0], mn); s_ext[wv][k][1] = fmaxf(s_ext[wv][k][1], mx);
        }
        if (e_hist) {                                      // flush portfolio k's digit-0 counts (read-and-clear)
          __syncthreads();
          unsigned long long* out = e_hist + (size_t)(a.k_begin + k) * 2 * MCP_SELECT_BINS;
          for (int i = tid; i < MCP_SELECT_BINS; i += PATH_BLOCK) {
            const uint32_t h = s_hist[i];
            if (h) { atomicAdd(&out[i], (unsigned long long)h); s_hist[i] = 0u; }
          }
          __syncthreads();
        }
      }
    }
  }

#ifdef MCP_DIAG_CLOCK
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    mcp_diag_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - diag_t0;
    mcp_diag_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - diag_r0;
  }
#endif
  asm volatile("" : "+s"(kargs));
  if (kargs->partials != nullptr) {
    __syncthreads();
    if ((int)threadIdx.x < kt) {                           // one partial per workgroup and portfolio, waves in order
      const int k = threadIdx.x;
      MomentPartial o;
      o.s1 = (s_mom[0][k][0] + s_mom[1][k][0]) + (s_mom[2][k][0] + s_mom[3][k][0]);
      o.s2 = (s_mom[0][k][1] + s_mom[1][k][1]) + (s_mom[2][k][1] + s_mom[3][k][1]);
      o.vmin = fminf(fminf(s_ext[0][k][0], s_ext[1][k][0]), fminf(s_ext[2][k][0], s_ext[3][k][0]));
      o.vmax = fmaxf(fmaxf(s_ext[0][k][1], s_ext[1][k][1]), fmaxf(s_ext[2][k][1], s_ext[3][k][1]));
      o.n = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
      kargs->partials[(size_t)(a.k_begin + k) * kargs->slots + blockIdx.x] = o;
    }
  }
}

}  // namespace mcp
