// mcp_sweep_kernels.hip -- the reference's own "Monte Carlo": the random-weight sweep over HISTORICAL
// returns, loop body of app.py:708-713 evaluated for P weight vectors at once, in binary64 like the
// reference (NumPy/pandas defaults).
//
//   port_return = w . mean_returns                     app.py:708
//   port_std    = sqrt(w^T (cov w))                    app.py:709
//   port_series = returns_df @ w                       app.py:710   [R]
//   sharpe      = (port_return - rf)/port_std or 0     app.py:711   (rf in the reference's units, Q2)
//   var_95      = np.percentile(port_series, (1-a)*100) app.py:712 -> 258-259
//   cvar_95     = port_series[port_series <= var].mean() app.py:713 -> 261-263
//
// Up to 256 rows (the reference's own sizes: 13 monthly ... 252 daily rows): one 64-lane wave per portfolio, the series in
// LDS, the two order statistics by rank counting (O(R^2) compares, a few microseconds for 2,500 portfolios x 13 rows).
// Beyond: one 256-thread workgroup per portfolio and an in-LDS bitonic sort of the series (O(R log^2 R)): 2,500 portfolios x
// 4,096 rows in milliseconds instead of 16 M compares per portfolio.  Same values either way (order statistics are exact;
// the tail sum differs in association only).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcport.h"
#include "mcp_stats_kernels.h"

namespace mcp {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// grid = P, block = 64, dynamic LDS = (R + N) doubles
__global__ void __launch_bounds__(64) sweep_hist_kernel(int N, int R, const double* __restrict__ returns,
                                                        const double* __restrict__ mean, const double* __restrict__ cov,
                                                        const double* __restrict__ W, double rf, uint64_t rank_lo,
                                                        uint64_t rank_hi, double gamma, double* __restrict__ out_ret,
                                                        double* __restrict__ out_std, double* __restrict__ out_sharpe,
                                                        double* __restrict__ out_var, double* __restrict__ out_cvar) {
  extern __shared__ double lds[];
  double* series = lds;        // [R]
  double* w = lds + R;         // [N]
  const int p = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < N; i += 64) w[i] = W[(size_t)p * N + i];
  __syncthreads();

  // analytic moments
  double pr = 0.0, pv = 0.0;
  for (int i = lane; i < N; i += 64) {
    pr += w[i] * mean[i];
    double cw = 0.0;
    for (int j = 0; j < N; j++) cw += cov[(size_t)i * N + j] * w[j];
    pv += w[i] * cw;
  }
  pr = wsum(pr);
  pv = wsum(pv);
  const double sd = sqrt(pv);

  // historical portfolio return series
  for (int r = lane; r < R; r += 64) {
    double s = 0.0;
    for (int i = 0; i < N; i++) s += returns[(size_t)r * N + i] * w[i];
    series[r] = s;
  }
  __syncthreads();

  // order statistics rank_lo / rank_hi by (stable) rank counting
  double a_part = 0.0, b_part = 0.0;
  for (int r = lane; r < R; r += 64) {
    const double x = series[r];
    uint64_t rk = 0;
    for (int q = 0; q < R; q++) {
      const double y = series[q];
      rk += (y < x) || (y == x && q < r);
    }
    if (rk == rank_lo) a_part = x;
    if (rk == rank_hi) b_part = x;
  }
  // exactly one lane holds each; all others contribute +0.0 (x + 0.0 == x, also for -0.0 + 0.0 -> +0.0: harmless)
  const double a = wsum(a_part), b = wsum(b_part);
  const double diff = b - a;
  double v = a + diff * gamma;                    // numpy _lerp
  if (gamma >= 0.5) v = b - diff * (1.0 - gamma);

  double cnt = 0.0, sum = 0.0;
  for (int r = lane; r < R; r += 64) {
    const double x = series[r];
    if (x <= v) { cnt += 1.0; sum += x; }
  }
  cnt = wsum(cnt);
  sum = wsum(sum);
  if (lane == 0) {
    out_ret[p] = pr;
    out_std[p] = sd;
    out_sharpe[p] = sd > 0.0 ? (pr - rf) / sd : 0.0;
    out_var[p] = v;
    out_cvar[p] = cnt > 0.0 ? sum / cnt : v;
  }
}

// ---- R > 256: sort instead of counting ranks ---------------------------------------------------------------------------
constexpr int SORT_BLOCK = 256;
__device__ __forceinline__ double bsum(double v, double* red /* [4] LDS */) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// grid = P, block = 256, dynamic LDS = (R2 + N) doubles, R2 = R rounded up to a power of two
__global__ void __launch_bounds__(SORT_BLOCK) sweep_hist_sorted_kernel(int N, int R, int R2, const double* __restrict__ returns,
                                                                       const double* __restrict__ mean, const double* __restrict__ cov,
                                                                       const double* __restrict__ W, double rf, uint64_t rank_lo,
                                                                       uint64_t rank_hi, double gamma, double* __restrict__ out_ret,
                                                                       double* __restrict__ out_std, double* __restrict__ out_sharpe,
                                                                       double* __restrict__ out_var, double* __restrict__ out_cvar) {
  extern __shared__ double lds[];
  __shared__ double red[4];
  double* series = lds;        // [R2], sorted in place
  double* w = lds + R2;        // [N]
  const int p = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < N; i += SORT_BLOCK) w[i] = W[(size_t)p * N + i];
  __syncthreads();

  // analytic moments (the same per-lane expressions as the one-wave kernel)
  double pr = 0.0, pv = 0.0;
  for (int i = tid; i < N; i += SORT_BLOCK) {
    pr += w[i] * mean[i];
    double cw = 0.0;
    for (int j = 0; j < N; j++) cw += cov[(size_t)i * N + j] * w[j];
    pv += w[i] * cw;
  }
  pr = bsum(pr, red);
  pv = bsum(pv, red);
  const double sd = sqrt(pv);

  // historical portfolio return series (app.py:710), padded with +inf up to the power of two
  for (int r = tid; r < R2; r += SORT_BLOCK) {
    double x = __builtin_inf();
    if (r < R) {
      x = 0.0;
      for (int i = 0; i < N; i++) x += returns[(size_t)r * N + i] * w[i];
    }
    series[r] = x;
  }
  __syncthreads();

  // bitonic sort, ascending
  for (int k = 2; k <= R2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < R2 / 2; t += SORT_BLOCK) {
        const int i = 2 * t - (t & (j - 1));           // lower index of the t-th pair at distance j
        const int l = i + j;
        const bool up = (i & k) == 0;
        const double a = series[i], b = series[l];
        if ((a > b) == up) { series[i] = b; series[l] = a; }
      }
      __syncthreads();
    }
  }
  const double a = series[rank_lo], b = series[rank_hi];
  const double diff = b - a;
  double v = a + diff * gamma;                    // numpy _lerp
  if (gamma >= 0.5) v = b - diff * (1.0 - gamma);

  double cnt = 0.0, sum = 0.0;
  for (int r = tid; r < R; r += SORT_BLOCK) {
    const double x = series[r];
    if (x <= v) { cnt += 1.0; sum += x; }
  }
  cnt = bsum(cnt, red);
  sum = bsum(sum, red);
  if (tid == 0) {
    out_ret[p] = pr;
    out_std[p] = sd;
    out_sharpe[p] = sd > 0.0 ? (pr - rf) / sd : 0.0;
    out_var[p] = v;
    out_cvar[p] = cnt > 0.0 ? sum / cnt : v;
  }
}

hipError_t launch_sweep_hist(int N, int R, int P, const double* returns, const double* mean, const double* cov,
                             const double* W, double rf, uint64_t rank_lo, uint64_t rank_hi, double gamma,
                             double* out5 /* [5][P] */, hipStream_t s) {
  if (R > 256) {
    int R2 = 512;
    while (R2 < R) R2 <<= 1;
    const size_t lds = (size_t)(R2 + N) * sizeof(double);          // <= (4096 + 64) * 8 = 33,280 B
    sweep_hist_sorted_kernel<<<P, SORT_BLOCK, lds, s>>>(N, R, R2, returns, mean, cov, W, rf, rank_lo, rank_hi, gamma, out5, out5 + P,
                                                        out5 + 2 * (size_t)P, out5 + 3 * (size_t)P, out5 + 4 * (size_t)P);
    return hipGetLastError();
  }
  const size_t lds = (size_t)(R + N) * sizeof(double);
  sweep_hist_kernel<<<P, 64, lds, s>>>(N, R, returns, mean, cov, W, rf, rank_lo, rank_hi, gamma, out5, out5 + P,
                                       out5 + 2 * (size_t)P, out5 + 3 * (size_t)P, out5 + 4 * (size_t)P);
  return hipGetLastError();
}

}  // namespace mcp
