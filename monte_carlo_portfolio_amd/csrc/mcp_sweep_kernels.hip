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
// One 64-lane wave per portfolio; the R-row series lives in LDS; the two order statistics are found by
// rank counting (R is the number of historical rows: 13 ... a few thousand).  Tiny next to the path
// kernel: 2,500 portfolios x 13 rows is a few microseconds; it exists for drop-in completeness.
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

hipError_t launch_sweep_hist(int N, int R, int P, const double* returns, const double* mean, const double* cov,
                             const double* W, double rf, uint64_t rank_lo, uint64_t rank_hi, double gamma,
                             double* out5 /* [5][P] */, hipStream_t s) {
  const size_t lds = (size_t)(R + N) * sizeof(double);
  sweep_hist_kernel<<<P, 64, lds, s>>>(N, R, returns, mean, cov, W, rf, rank_lo, rank_hi, gamma, out5, out5 + P,
                                       out5 + 2 * (size_t)P, out5 + 3 * (size_t)P, out5 + 4 * (size_t)P);
  return hipGetLastError();
}

}  // namespace mcp
