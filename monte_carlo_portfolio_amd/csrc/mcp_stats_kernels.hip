// mcp_stats_kernels.hip -- reductions behind the path kernel (gfx950).
//
//   moments_*         n, sum x, sum x^2, min, max of x over V_T (fp64, deterministic two-stage reduction).
//   select_*          exact order statistics by 3-pass radix select on the float bit pattern; they
//                     feed np.percentile's linear interpolation (app.py:258-259, numpy 2.2 _lerp).
//   tail_*            count / sum of x <= VaR (app.py:261-263).
//   stats_kernel      mean, std(ddof=1), Sharpe (app.py:711), VaR, CVaR per portfolio.
// All of them are HBM-bound streaming passes over V_T (4 B/path) or trivially small.
#include "mcp_paths.h"
#include "mcp_stats_kernels.h"

namespace mcp {

// Moments of x over V_T: partial[k][b] = {n, sum x, sum x^2, min, max} of a grid-stride slice (fp64), then
// a fixed-order sum of the MOMENTS_GRID partials -> run-to-run deterministic.  grid = (MOMENTS_GRID, K).
__global__ void __launch_bounds__(256) moments_partial_kernel(const mcp_params prm, const float* __restrict__ terminal,
                                                              uint64_t stride, uint64_t n, mcp_moments* __restrict__ partial) {
  const int k = blockIdx.y;
  const double v0d = (double)(float)prm.v0;
  const float* __restrict__ src = terminal + (size_t)k * stride;
  double c = 0.0, s1 = 0.0, s2 = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const double x = terminal_to_x(src[i], v0d, prm.compounding);
    c += 1.0; s1 += x; s2 += x * x; mn = fmin(mn, x); mx = fmax(mx, x);
  }
  __shared__ double red[4][5];
  c = wave_sum(c); s1 = wave_sum(s1); s2 = wave_sum(s2); mn = wave_min(mn); mx = wave_max(mx);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv][0] = c; red[wv][1] = s1; red[wv][2] = s2; red[wv][3] = mn; red[wv][4] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mcp_moments m = {red[0][0], red[0][1], red[0][2], red[0][3], red[0][4]};
    for (int w = 1; w < 4; w++) {
      m.n += red[w][0]; m.sum += red[w][1]; m.sumsq += red[w][2];
      m.min = fmin(m.min, red[w][3]); m.max = fmax(m.max, red[w][4]);
    }
    partial[(size_t)k * gridDim.x + blockIdx.x] = m;
  }
}

// partials [K][grid] -> moments [K]; one block per portfolio, fixed summation order.
__global__ void __launch_bounds__(256) moments_kernel(const mcp_moments* __restrict__ partials, int grid,
                                                      mcp_moments* __restrict__ out) {
  const int k = blockIdx.x;
  double n = 0, s1 = 0, s2 = 0, mn = __builtin_inf(), mx = -__builtin_inf();
  for (int b = threadIdx.x; b < grid; b += blockDim.x) {
    const mcp_moments m = partials[(size_t)k * grid + b];
    n += m.n; s1 += m.sum; s2 += m.sumsq; mn = fmin(mn, m.min); mx = fmax(mx, m.max);
  }
  __shared__ double red[4][5];
  n = wave_sum(n); s1 = wave_sum(s1); s2 = wave_sum(s2); mn = wave_min(mn); mx = wave_max(mx);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv][0] = n; red[wv][1] = s1; red[wv][2] = s2; red[wv][3] = mn; red[wv][4] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mcp_moments m = {red[0][0], red[0][1], red[0][2], red[0][3], red[0][4]};
    for (int w = 1; w < 4; w++) {
      m.n += red[w][0]; m.sum += red[w][1]; m.sumsq += red[w][2];
      m.min = fmin(m.min, red[w][3]); m.max = fmax(m.max, red[w][4]);
    }
    out[k] = m;
  }
}

// gathered [world][K] moment records of all ranks -> merged [K] (SUM on n, sum, sumsq; MIN; MAX), rank order fixed
__global__ void moments_merge_kernel(int K, int world, const mcp_moments* __restrict__ gathered, mcp_moments* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  mcp_moments m = gathered[k];
  for (int r = 1; r < world; r++) {
    const mcp_moments g = gathered[(size_t)r * K + k];
    m.n += g.n; m.sum += g.sum; m.sumsq += g.sumsq; m.min = fmin(m.min, g.min); m.max = fmax(m.max, g.max);
  }
  out[k] = m;
}

hipError_t launch_moments_merge(int K, int world, const mcp_moments* gathered, mcp_moments* out, hipStream_t s) {
  moments_merge_kernel<<<(K + 63) / 64, 64, 0, s>>>(K, world, gathered, out);
  return hipGetLastError();
}

// ---- radix select -------------------------------------------------------------------------------
// state[k][w] (w = 0: rank lo, w = 1: rank hi): {prefix, rank within the prefix}.

__device__ __forceinline__ void pass_shape(int pass, int& shift, int& bits, int& pshift) {
  // pass 0: key[31:21]; pass 1: key[20:10]; pass 2: key[9:0]
  shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
  bits = pass == 2 ? 10 : 11;
  pshift = pass == 0 ? 32 : (pass == 1 ? 21 : 10);   // prefix = key >> pshift (pass 0: no prefix)
}

constexpr int SELECT_BLOCK = 256;

// hist[k][w][bin] += #{keys of portfolio k matching state[k][w].prefix with digit == bin}.
// Pass 0 fills w = 0 only (no prefix yet).  grid = (blocks, K).
__global__ void __launch_bounds__(SELECT_BLOCK) select_hist_kernel(
    const float* __restrict__ terminal, uint64_t stride, uint64_t n, int pass,
    const SelectState* __restrict__ state, unsigned long long* __restrict__ hist) {
  __shared__ uint32_t h[2][MCP_SELECT_BINS];
  const int k = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * MCP_SELECT_BINS; i += SELECT_BLOCK) (&h[0][0])[i] = 0u;
  __syncthreads();
  int shift, bits, pshift;
  pass_shape(pass, shift, bits, pshift);
  const uint32_t mask = (1u << bits) - 1u;
  const uint32_t pa = state[2 * k + 0].prefix, pb = state[2 * k + 1].prefix;
  const float* __restrict__ src = terminal + (size_t)k * stride;
  for (uint64_t i = (uint64_t)blockIdx.x * SELECT_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * SELECT_BLOCK) {
    const uint32_t key = float_to_key(src[i]);
    const uint32_t d = (key >> shift) & mask;
    if (pass == 0) {
      atomicAdd(&h[0][d], 1u);
    } else {
      const uint32_t pre = key >> pshift;
      if (pre == pa) atomicAdd(&h[0][d], 1u);
      if (pre == pb) atomicAdd(&h[1][d], 1u);
    }
  }
  __syncthreads();
  unsigned long long* out = hist + (size_t)k * 2 * MCP_SELECT_BINS;
  for (int i = threadIdx.x; i < 2 * MCP_SELECT_BINS; i += SELECT_BLOCK) {
    const uint32_t c = (&h[0][0])[i];
    if (c) atomicAdd(&out[i], (unsigned long long)c);
  }
}

// Find, per (k, w), the digit whose bin contains state.rank; descend into it.  grid = (2, K).
__global__ void __launch_bounds__(SELECT_BLOCK) select_scan_kernel(int pass, const unsigned long long* __restrict__ hist,
                                                                   SelectState* __restrict__ state) {
  const int w = blockIdx.x, k = blockIdx.y;
  const unsigned long long* hh = hist + ((size_t)k * 2 + (pass == 0 ? 0 : w)) * MCP_SELECT_BINS;
  constexpr int PER = MCP_SELECT_BINS / SELECT_BLOCK;   // 8 consecutive bins per thread
  unsigned long long c[PER], tot = 0;
#pragma unroll
  for (int i = 0; i < PER; i++) { c[i] = hh[threadIdx.x * PER + i]; tot += c[i]; }
  __shared__ unsigned long long part[SELECT_BLOCK];
  part[threadIdx.x] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {   // 256-entry serial exclusive scan: negligible, and order-exact
    unsigned long long run = 0;
    for (int i = 0; i < SELECT_BLOCK; i++) { const unsigned long long v = part[i]; part[i] = run; run += v; }
  }
  __syncthreads();
  SelectState st = state[2 * k + w];
  unsigned long long before = part[threadIdx.x];
  int shift, bits, pshift;
  pass_shape(pass, shift, bits, pshift);
  if (st.rank >= before && st.rank < before + tot) {   // exactly one thread owns the rank
#pragma unroll
    for (int i = 0; i < PER; i++) {
      if (st.rank < before + c[i]) {
        st.prefix = (pass == 0 ? 0u : (st.prefix << bits)) | (uint32_t)(threadIdx.x * PER + i);
        st.rank -= before;
        state[2 * k + w] = st;
        break;
      }
      before += c[i];
    }
  }
}

// ---- tail (CVaR) --------------------------------------------------------------------------------

// np.percentile(method='linear') on the two order statistics (numpy 2.2 _lerp).
__global__ void quantile_kernel(const mcp_params prm, int K, double gamma, const SelectState* __restrict__ state,
                                Quantile* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const double v0d = (double)(float)prm.v0;
  const double a = terminal_to_x(key_to_float(state[2 * k + 0].prefix), v0d, prm.compounding);
  const double b = terminal_to_x(key_to_float(state[2 * k + 1].prefix), v0d, prm.compounding);
  const double diff = b - a;
  double r = a + diff * gamma;
  if (gamma >= 0.5) r = b - diff * (1.0 - gamma);
  out[k] = Quantile{a, b, r};
}


// partial[k][b] = {count, sum} of x <= var_k over a grid-stride slice.  grid = (TAIL_GRID, K).
__global__ void __launch_bounds__(256) tail_kernel(const mcp_params prm, const float* __restrict__ terminal,
                                                   uint64_t stride, uint64_t n, const Quantile* __restrict__ quant,
                                                   double* __restrict__ partial) {
  const int k = blockIdx.y;
  const double v0d = (double)(float)prm.v0;
  const double thr = quant[k].var;
  const float* __restrict__ src = terminal + (size_t)k * stride;
  double c = 0.0, s = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const double x = terminal_to_x(src[i], v0d, prm.compounding);
    if (x <= thr) { c += 1.0; s += x; }
  }
  c = wave_sum(c); s = wave_sum(s);
  __shared__ double red[4][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv][0] = c; red[wv][1] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + ((size_t)k * gridDim.x + blockIdx.x) * 2;
    o[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    o[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  }
}

// tail[k] = {count, sum}: fixed-order sum of the TAIL_GRID partials.  grid = K, block = 64.
__global__ void __launch_bounds__(64) tail_sum_kernel(const double* __restrict__ partial, int grid, double* __restrict__ tail) {
  const int k = blockIdx.x;
  double c = 0.0, s = 0.0;
  for (int b = threadIdx.x; b < grid; b += 64) {
    c += partial[((size_t)k * grid + b) * 2 + 0];
    s += partial[((size_t)k * grid + b) * 2 + 1];
  }
  c = wave_sum(c); s = wave_sum(s);
  if (threadIdx.x == 0) { tail[2 * k] = c; tail[2 * k + 1] = s; }
}

// moments + quantile + tail -> mcp_stats (app.py:711 Sharpe, app.py:263 CVaR fallback).
__global__ void stats_kernel(const mcp_params prm, int K, const mcp_moments* __restrict__ mom,
                             const Quantile* __restrict__ quant, const double* __restrict__ tail,
                             mcp_stats* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const mcp_moments m = mom[k];
  mcp_stats s;
  s.n = (uint64_t)m.n;
  s.mean = m.n > 0 ? m.sum / m.n : 0.0;
  double m2 = m.sumsq - m.sum * s.mean;
  if (m2 < 0.0) m2 = 0.0;
  s.m2 = m2;
  s.std = m.n > 1 ? sqrt(m2 / (m.n - 1.0)) : 0.0;
  s.sharpe = s.std > 0.0 ? (s.mean - prm.rf) / s.std : 0.0;
  s.var = quant[k].var; s.x_lo = quant[k].x_lo; s.x_hi = quant[k].x_hi;
  s.n_tail = (uint64_t)tail[2 * k];
  s.sum_tail = tail[2 * k + 1];
  s.cvar = s.n_tail > 0 ? s.sum_tail / (double)s.n_tail : s.var;
  s.min = m.min; s.max = m.max;
  out[k] = s;
}

// The normal generator on its own: z[i] = inverse-CDF normal of word x[i] (SPEC.md section 3); also what tests use
// to hit edge inputs (both ends of every octave, u == 1/2, the deepest tail).
__global__ void __launch_bounds__(256) normals_kernel(const uint32_t* __restrict__ x, uint64_t n,
                                                      const float4* __restrict__ table, float* __restrict__ z) {
  __shared__ float4 s_tab[ICDF_ENTRIES];
  for (int i = threadIdx.x; i < ICDF_ENTRIES; i += 256) s_tab[i] = table[i];
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) z[i] = normal_icdf(x[i], s_tab);
}

hipError_t launch_normals(const uint32_t* x, uint64_t n, const float4* table, float* z, hipStream_t s) {
  uint64_t g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  normals_kernel<<<(unsigned)g, 256, 0, s>>>(x, n, table, z);
  return hipGetLastError();
}

__global__ void select_init_kernel(int K, uint64_t rank_lo, uint64_t rank_hi, SelectState* __restrict__ state) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * K) return;
  state[i] = SelectState{0u, 0u, (i & 1) ? rank_hi : rank_lo};
}

// blocks per portfolio of a streaming pass over n values: >= 8 Ki values per 256-thread block, at most `cap`
static int stream_grid(uint64_t n, int cap) {
  uint64_t g = (n + 8191) / 8192;
  if (g < 1) g = 1;
  return (int)(g > (uint64_t)cap ? (uint64_t)cap : g);
}

__global__ void __launch_bounds__(256) zero_u64_kernel(unsigned long long* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0ull;
}

// ---- launch wrappers (enqueue only) -------------------------------------------------------------
hipError_t launch_moments(const mcp_params& prm, int K, const float* terminal, uint64_t stride, uint64_t n,
                          mcp_moments* partials, mcp_moments* out, hipStream_t s) {
  const int gx = stream_grid(n, MOMENTS_GRID);
  moments_partial_kernel<<<dim3((unsigned)gx, (unsigned)K), 256, 0, s>>>(prm, terminal, stride, n, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  moments_kernel<<<K, 256, 0, s>>>(partials, gx, out);
  return hipGetLastError();
}

hipError_t launch_select_init(int K, uint64_t rank_lo, uint64_t rank_hi, SelectState* state, hipStream_t s) {
  select_init_kernel<<<(2 * K + 255) / 256, 256, 0, s>>>(K, rank_lo, rank_hi, state);
  return hipGetLastError();
}

hipError_t launch_select_hist(int K, const float* terminal, uint64_t stride, uint64_t n, int pass,
                              const SelectState* state, unsigned long long* hist, hipStream_t s) {
  // own zero-fill kernel rather than hipMemsetAsync: a captured memset node replayed wrongly from the second
  // hipGraphLaunch on (ROCm 7.0 runtime bundled with torch); a plain kernel node replays exactly
  const size_t words = (size_t)K * 2 * MCP_SELECT_BINS;
  zero_u64_kernel<<<(unsigned)((words + 1023) / 1024 > 4096 ? 4096 : (words + 1023) / 1024), 256, 0, s>>>(hist, words);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // >= 16 Ki elements per block: zeroing and flushing the 2 x 2048-bin LDS histograms costs as much as ~4 Ki elements
  uint64_t bx = (n + 16383) / 16384;
  if (bx < 1) bx = 1;
  if (bx > 1024) bx = 1024;
  select_hist_kernel<<<dim3((unsigned)bx, (unsigned)K), SELECT_BLOCK, 0, s>>>(terminal, stride, n, pass, state, hist);
  return hipGetLastError();
}

hipError_t launch_select_scan(int K, int pass, const unsigned long long* hist, SelectState* state, hipStream_t s) {
  select_scan_kernel<<<dim3(2, (unsigned)K), SELECT_BLOCK, 0, s>>>(pass, hist, state);
  return hipGetLastError();
}

hipError_t launch_quantile(const mcp_params& prm, int K, double gamma, const SelectState* state, Quantile* out, hipStream_t s) {
  quantile_kernel<<<(K + 63) / 64, 64, 0, s>>>(prm, K, gamma, state, out);
  return hipGetLastError();
}

hipError_t launch_tail(const mcp_params& prm, int K, const float* terminal, uint64_t stride, uint64_t n,
                       const Quantile* quant, double* partial, double* tail, hipStream_t s) {
  const int gx = stream_grid(n, TAIL_GRID);
  tail_kernel<<<dim3((unsigned)gx, (unsigned)K), 256, 0, s>>>(prm, terminal, stride, n, quant, partial);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  tail_sum_kernel<<<K, 64, 0, s>>>(partial, gx, tail);
  return hipGetLastError();
}

hipError_t launch_stats(const mcp_params& prm, int K, const mcp_moments* mom, const Quantile* quant,
                        const double* tail, mcp_stats* out, hipStream_t s) {
  stats_kernel<<<(K + 63) / 64, 64, 0, s>>>(prm, K, mom, quant, tail, out);
  return hipGetLastError();
}

}  // namespace mcp
