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

// Fused moments of the sweep kernels (N3): the wave holds V for 32 MT portfolios x 64 paths in the MFMA C/D layout (lane l,
// register g of tile (mt, nt): portfolio k_base + 32 mt + (g&3) + 8 (g>>2) + 4 (l>>5), path path0 + 32 nt + (l&31)).  Per
// portfolio: x = V/v0 - 1 (or expm1 S), d = x - c_k, {sum d, sum d^2, min V, max V} over the wave's 64 paths -- the two
// path tiles in the lane, then a shuffle reduction over the 32 lanes of the half-wave -- and ONE MomentPartial per portfolio
// and wave tile, slot `tile` (= global 64-path tile index).  Dead paths (>= n_paths) contribute nothing.
template <int MT, bool LOGC>
__device__ __forceinline__ void sweep_moments(const PathArgs& a, const f32x16 (&V)[MT][2], int k_base, uint64_t path0) {
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  if (path0 >= a.n_paths) return;                          // wave-uniform: a wave tile beyond the range has no slot
  const bool live0 = path0 + col < a.n_paths, live1 = path0 + 32 + col < a.n_paths;
  const uint64_t left = a.n_paths - path0;
  const unsigned long long n_tile = left < 64 ? left : 64;
  const uint64_t tile = path0 / SWEEP_TILE_PATHS;
  const float inf = __builtin_inff();
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const int k = k_base + 32 * mt + (g & 3) + 8 * (g >> 2) + 4 * half;
      const bool mine = k < a.n_portfolios;                // rows beyond K are zero-weight padding
      const double c = (mine && a.pivot) ? a.pivot[k] : 0.0;
      const float v0 = V[mt][0][g], v1 = V[mt][1][g];
      double x0, x1;
      if constexpr (LOGC) { x0 = expm1((double)v0); x1 = expm1((double)v1); }
      else if (a.v0_pow2) { x0 = __builtin_fma((double)v0, a.inv_v0d, -1.0); x1 = __builtin_fma((double)v1, a.inv_v0d, -1.0); }
      else { x0 = (double)v0 / a.v0d - 1.0; x1 = (double)v1 / a.v0d - 1.0; }
      const double d0 = live0 ? x0 - c : 0.0, d1 = live1 ? x1 - c : 0.0;
      double s1 = d0 + d1;
      double s2 = __builtin_fma(d0, d0, d1 * d1);
      float mn = fminf(live0 ? v0 : inf, live1 ? v1 : inf);
      float mx = fmaxf(live0 ? v0 : -inf, live1 ? v1 : -inf);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {                   // xor offsets < 32 stay inside the half-wave
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      }
      if (col == 0 && mine) {
        MomentPartial m;
        m.s1 = s1; m.s2 = s2; m.vmin = mn; m.vmax = mx; m.n = n_tile;
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
// mc_sweep_kernel and to the oracle.  MT = 4 (512 portfolios per workgroup) for N <= 16, MT = 2 (256) up to N = 64.
template <int NB, int MT, bool NATIVE, bool LOGC>
__global__ void __launch_bounds__(PATH_BLOCK, 2) mc_sweep_shared_kernel(const PathArgs a) {
  constexpr int N4 = 4 * NB, KS = N4 / 2;
  typedef const __attribute__((address_space(4))) float* cfloat_p;
  cfloat_p mu = (cfloat_p)a.packed;
  cfloat_p Lp = mu + N4;
  const float* __restrict__ Wg = a.packed + N4 + N4 * (N4 / 2 + 1);

  __shared__ float4 s_tab[ICDF_LDS_ENTRIES];
  __shared__ float s_z[N4][64], s_r[N4][64];
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

  for (int t = 0; t < a.n_steps; t++) {
    asm volatile("" : "+s"(mu), "+s"(Lp));
    // phase A: this wave's share of the normals
#pragma unroll
    for (int q0 = 0; q0 < NB; q0 += 4) {
      const int q = q0 + wave;                                      // wave-uniform
      if (q < NB) {
        uint32_t x[4];
        philox4x32_10((uint32_t)t * NB + q, 0u, plo, phi, ks, x);
        float z0, z1, z2, z3;
        block_normals<NATIVE>(x, s_tab, kc, z0, z1, z2, z3);
        s_z[0 * NB + q][lane] = z0;
        s_z[1 * NB + q][lane] = z1;
        s_z[2 * NB + q][lane] = z2;
        s_z[3 * NB + q][lane] = z3;
      }
    }
#ifndef MCP_EXP_NOBARRIER     // experiment only (wrong results): what the two barriers per step cost
    __syncthreads();
#endif
    // phase B: this wave's row pairs of r = mu + L z
#pragma unroll
    for (int m0 = 0; m0 < N4 / 2; m0 += 4) {
#pragma unroll
      for (int wv = 0; wv < 4; wv++) {                              // unrolled so that m is a compile-time constant
        const int m = m0 + wv;
        if (m < N4 / 2 && wv == wave) {
          f32x2 acc = {mu[2 * m], mu[2 * m + 1]};
#pragma unroll
          for (int j = 0; j <= 2 * m + 1; j++) {
            const f32x2 l2 = {Lp[2 * m * (m + 1) + 2 * j], Lp[2 * m * (m + 1) + 2 * j + 1]};
            const float zj = s_z[j][lane];
            acc = __builtin_elementwise_fma(l2, (f32x2){zj, zj}, acc);
          }
          s_r[2 * m][lane] = acc.x;
          s_r[2 * m + 1][lane] = acc.y;
        }
      }
    }
#ifndef MCP_EXP_NOBARRIER
    __syncthreads();
#endif
    // phase C: rho = W . r on the matrix cores, then compounding
    float b[2][KS];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int kk = 0; kk < KS; kk++) b[nt][kk] = s_r[2 * kk + (lane >> 5)][32 * nt + (lane & 31)];
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

template <int NB>
static hipError_t go_shared(bool native, const PathArgs& args, hipStream_t stream) {
  constexpr int MT = NB <= 4 ? 4 : 2;
  const dim3 grid((unsigned)((args.n_paths + 63) / 64), (unsigned)((args.k_count + 128 * MT - 1) / (128 * MT)));
  const bool lg = args.compounding == MCP_COMPOUND_LOG;
  if (native) { if (lg) mc_sweep_shared_kernel<NB, MT, true, true><<<grid, PATH_BLOCK, 0, stream>>>(args); else mc_sweep_shared_kernel<NB, MT, true, false><<<grid, PATH_BLOCK, 0, stream>>>(args); }
  else { if (lg) mc_sweep_shared_kernel<NB, MT, false, true><<<grid, PATH_BLOCK, 0, stream>>>(args); else mc_sweep_shared_kernel<NB, MT, false, false><<<grid, PATH_BLOCK, 0, stream>>>(args); }
  return hipGetLastError();
}

// K rows of W must be zero-padded to a multiple of 512 (mcp_pack_params pads to K_PAD).
hipError_t MCP_CAT(launch_sweep_shared_p, MCP_SWEEP_PART)(int nb, bool native, const PathArgs& args, hipStream_t stream) {
  switch (nb) {
#define MCP_CASE(n) case n: return go_shared<n>(native, args, stream);
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
hipError_t launch_sweep_shared(int nb, bool native, const PathArgs& args, hipStream_t stream) {
  if (nb < 1 || nb > 16) return hipErrorInvalidValue;
  switch ((nb - 1) / 4) {
    case 0: return launch_sweep_shared_p0(nb, native, args, stream);
    case 1: return launch_sweep_shared_p1(nb, native, args, stream);
    case 2: return launch_sweep_shared_p2(nb, native, args, stream);
    default: return launch_sweep_shared_p3(nb, native, args, stream);
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
