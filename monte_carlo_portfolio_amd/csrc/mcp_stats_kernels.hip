// mcp_stats_kernels.hip -- reductions behind the path kernel (gfx950): six launches, three streaming reads of V_T.
//
//   pass0_kernel   one read of V_T: {n, sum x, sum x^2, min, max} partials (fp64) AND the digit-0 histogram of the
//                  radix select (key bits 31..21).
//   scan_kernel    pass 0: partials -> moment record, ranks -> select state, descend into the digit holding each rank;
//                  pass 1: the same descent plus the "below" partials of hist pass 1.  Clears the histogram it consumed.
//   hist_kernel    pass 1 / 2: one read of V_T: digit histograms of the keys that match the prefixes found so far
//                  (bits 20..10, then 9..0) AND the fp64 sum of x over the elements that sort strictly below the bucket
//                  of the low order statistic (what the CVaR tail mean needs; no separate tail pass).
//   final_kernel   last descent -> the two order statistics -> np.percentile's linear interpolation (app.py:258-259,
//                  numpy 2.2 _lerp); tail {x <= var} = keys below the low bucket (count from the ranks, sum from the hist
//                  passes) + the digits of the last bucket(s) whose x <= var (counts x values) (app.py:261-263); one rank:
//                  mean, std(ddof=1), Sharpe (app.py:711), CVaR.
//   stats_kernel   several ranks: merges the all-gathered records in rank order and finishes the same way.
//
// Exchanges of a multi-GPU host (include/mcport.h): all-reduce(SUM, u64) of the histogram after pass0 / hist(1) /
// hist(2); all-gather of the [K] records after final.  Everything is fixed-order fp64 or integer: run-to-run
// deterministic.  HBM-bound streaming passes over V_T (4 B/path each) or trivially small.
#include <cstdlib>

#include "mcp_paths.h"
#include "mcp_stats_kernels.h"

namespace mcp {

constexpr int SB = 256;                               // threads per block of every kernel here
constexpr int PER = MCP_SELECT_BINS / SB;             // 8 consecutive bins per thread in the scans

__device__ __forceinline__ void pass_shape(int pass, int& shift, int& bits, int& pshift) {
  // pass 0: key[31:21]; pass 1: key[20:10]; pass 2: key[9:0]
  shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
  bits = pass == 2 ? 10 : 11;
  pshift = pass == 0 ? 32 : (pass == 1 ? 21 : 10);   // prefix = key >> pshift (pass 0: no prefix)
}

// h[digit] += 1 for the active lanes.  Terminal values cluster (V_T ~ 1 +- 0.2 hits a handful of digit-0 bins), and 64
// lanes on one LDS address serialise; so up to four distinct digits per wave are counted by ballot and added once.
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

// blocks per portfolio of a streaming pass over n values: >= 4 Ki values per 256-thread block, at most stream_slots(K).
// The passes are latency-bound (a load, an fp64 divide, an LDS atomic per element and lane), so they want every wave
// slot of the chip before they want long per-thread loops (measured: 57 -> 9 us for pass 0 at 10^6 values).
static int stream_grid(uint64_t n, int K) {
  static const uint64_t per = [] { const char* e = getenv("MCP_STREAM_ELEMS"); const long v = e ? atol(e) : 0; return (uint64_t)(v > 0 ? v : 4096); }();
  uint64_t g = (n + per - 1) / per;
  if (g < 1) g = 1;
  const uint64_t cap = (uint64_t)stream_slots(K);
  return (int)(g > cap ? cap : g);
}

// ---- pass 0: moments + digit-0 histogram ----------------------------------------------------------------------------
// grid = K * G (block b of portfolio k at blockIdx.x = k*G + b: no 65,535 limit on K).  hist[k][0][*] must be zero on
// entry (the scans clear what they consume; buffers start zeroed).
__global__ void __launch_bounds__(SB) pass0_kernel(const mcp_params prm, const float* __restrict__ terminal, uint64_t stride,
                                                   uint64_t n, int G, int slots, double* __restrict__ partials,
                                                   unsigned long long* __restrict__ hist) {
  __shared__ uint32_t h[MCP_SELECT_BINS];
  __shared__ double red[4][5];
  const int k = blockIdx.x / G, b = blockIdx.x % G;
  for (int i = threadIdx.x; i < MCP_SELECT_BINS; i += SB) h[i] = 0u;
  __syncthreads();
  const double v0d = (double)(float)prm.v0;
  const float* __restrict__ src = terminal + (size_t)k * stride;
  double c = 0.0, s1 = 0.0, s2 = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
  const uint64_t step = (uint64_t)G * SB;
  for (uint64_t i0 = (uint64_t)b * SB; i0 < n; i0 += step) {     // uniform trip count: ballots inside see whole waves
    const uint64_t i = i0 + threadIdx.x;
    const bool live = i < n;
    const float v = live ? src[i] : 0.0f;
    if (live) {
      const double x = terminal_to_x(v, v0d, prm.compounding);
      c += 1.0; s1 += x; s2 += x * x; mn = fmin(mn, x); mx = fmax(mx, x);
    }
    lds_hist_add(h, float_to_key(v) >> 21, live);
  }
  c = wave_sum(c); s1 = wave_sum(s1); s2 = wave_sum(s2); mn = wave_min(mn); mx = wave_max(mx);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv][0] = c; red[wv][1] = s1; red[wv][2] = s2; red[wv][3] = mn; red[wv][4] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partials + ((size_t)k * slots + b) * PARTIAL_DOUBLES;
    o[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    o[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    o[2] = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    o[3] = fmin(fmin(red[0][3], red[1][3]), fmin(red[2][3], red[3][3]));
    o[4] = fmax(fmax(red[0][4], red[1][4]), fmax(red[2][4], red[3][4]));
  }
  unsigned long long* out = hist + (size_t)k * 2 * MCP_SELECT_BINS;
  for (int i = threadIdx.x; i < MCP_SELECT_BINS; i += SB) {
    const uint32_t cnt = h[i];
    if (cnt) atomicAdd(&out[i], (unsigned long long)cnt);
  }
}

// ---- hist pass 1 / 2 ------------------------------------------------------------------------------------------------
// hist[k][w][digit] += #{keys of portfolio k matching state[k][w].prefix}; partial "below" = sum of x over the elements
// that sort below state[k][0]'s bucket and were not already summed by the previous pass.  grid = K * G.
__global__ void __launch_bounds__(SB) hist_kernel(const mcp_params prm, int pass, const float* __restrict__ terminal,
                                                  uint64_t stride, uint64_t n, int G, int slots, const SelectState* __restrict__ state,
                                                  double* __restrict__ partials, unsigned long long* __restrict__ hist) {
  __shared__ uint32_t h[2][MCP_SELECT_BINS];
  __shared__ double red[4];
  const int k = blockIdx.x / G, b = blockIdx.x % G;
  for (int i = threadIdx.x; i < 2 * MCP_SELECT_BINS; i += SB) (&h[0][0])[i] = 0u;
  __syncthreads();
  int shift, bits, pshift;
  pass_shape(pass, shift, bits, pshift);
  const uint32_t mask = (1u << bits) - 1u;
  const uint32_t pa = state[2 * k + 0].prefix, pb = state[2 * k + 1].prefix;
  const double v0d = (double)(float)prm.v0;
  const float* __restrict__ src = terminal + (size_t)k * stride;
  double below = 0.0;
  const uint64_t step = (uint64_t)G * SB;
  for (uint64_t i = (uint64_t)b * SB + threadIdx.x; i < n; i += step) {
    const float v = src[i];
    const uint32_t key = float_to_key(v);
    const uint32_t pre = key >> pshift, d = (key >> shift) & mask;
    if (pre == pa) atomicAdd(&h[0][d], 1u);
    if (pre == pb) atomicAdd(&h[1][d], 1u);
    // pass 1: digit-0 below the bucket's; pass 2: inside the digit-0 bucket, digit-1 below
    if (pre < pa && (pass == 1 || (pre >> 11) == (pa >> 11))) below += terminal_to_x(v, v0d, prm.compounding);
  }
  below = wave_sum(below);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = below;
  __syncthreads();
  if (threadIdx.x == 0)
    partials[((size_t)k * slots + b) * PARTIAL_DOUBLES + 5] = (red[0] + red[1]) + (red[2] + red[3]);
  unsigned long long* out = hist + (size_t)k * 2 * MCP_SELECT_BINS;
  for (int i = threadIdx.x; i < 2 * MCP_SELECT_BINS; i += SB) {
    const uint32_t cnt = (&h[0][0])[i];
    if (cnt) atomicAdd(&out[i], (unsigned long long)cnt);
  }
}

// ---- block-wide helpers of the scans --------------------------------------------------------------------------------
// exclusive prefix sum of one value per thread over the 256-thread block (wave scans by shuffle, wave totals in LDS)
__device__ __forceinline__ unsigned long long block_exclusive_scan(unsigned long long v, unsigned long long* wtot /* [4] LDS */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();                        // wtot may still be read from a previous call
  if (lane == 63) wtot[wv] = inc;
  __syncthreads();
  unsigned long long base = 0;
  for (int w = 0; w < wv; w++) base += wtot[w];
  return base + inc - v;
}

// fixed-order sum of one double per thread over the block -> every thread
__device__ __forceinline__ double block_sum(double v, double* scratch /* [4] LDS */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// One descent step for target w of portfolio k: c[] = this thread's PER consecutive bins.  The thread owning the rank
// updates state; returns nothing (the final kernel uses its own variant that also needs the counts).
__device__ __forceinline__ void descend(int pass, const unsigned long long (&c)[PER], unsigned long long tot,
                                        unsigned long long before, SelectState& st, SelectState* dst) {
  int shift, bits, pshift;
  pass_shape(pass, shift, bits, pshift);
  if (st.rank >= before && st.rank < before + tot) {   // exactly one thread owns the rank
#pragma unroll
    for (int i = 0; i < PER; i++) {
      if (st.rank < before + c[i]) {
        st.prefix = (pass == 0 ? 0u : (st.prefix << bits)) | (uint32_t)(threadIdx.x * PER + i);
        st.rank -= before;
        *dst = st;
        break;
      }
      before += c[i];
    }
  }
}

// ---- scan pass 0 / 1: grid = K ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SB) scan_kernel(int pass, uint64_t rank_lo, uint64_t rank_hi, int G, int slots,
                                                  const double* __restrict__ partials, unsigned long long* __restrict__ hist,
                                                  SelectState* __restrict__ state, mcp_record* __restrict__ record) {
  __shared__ unsigned long long wtot[4];
  __shared__ double red[4][5];
  const int k = blockIdx.x;
  // 1. reduce the partials of the streaming pass that produced this histogram (fixed order)
  const double* pp = partials + (size_t)k * slots * PARTIAL_DOUBLES;
  if (pass == 0) {
    double n = 0, s1 = 0, s2 = 0, mn = __builtin_inf(), mx = -__builtin_inf();
    for (int b = threadIdx.x; b < G; b += SB) {
      const double* p = pp + (size_t)b * PARTIAL_DOUBLES;
      n += p[0]; s1 += p[1]; s2 += p[2]; mn = fmin(mn, p[3]); mx = fmax(mx, p[4]);
    }
    n = wave_sum(n); s1 = wave_sum(s1); s2 = wave_sum(s2); mn = wave_min(mn); mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = n; red[wv][1] = s1; red[wv][2] = s2; red[wv][3] = mn; red[wv][4] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      mcp_record r;
      r.n = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
      r.sum = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
      r.sumsq = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
      r.min = fmin(fmin(red[0][3], red[1][3]), fmin(red[2][3], red[3][3]));
      r.max = fmax(fmax(red[0][4], red[1][4]), fmax(red[2][4], red[3][4]));
      r.below = 0.0; r.pad[0] = r.pad[1] = 0.0;
      record[k] = r;
    }
  } else {
    double bl = 0.0;
    for (int b = threadIdx.x; b < G; b += SB) bl += pp[(size_t)b * PARTIAL_DOUBLES + 5];
    bl = block_sum(bl, &red[0][0]);
    if (threadIdx.x == 0) record[k].below = bl;
  }
  // 2. descend; pass 0 has one histogram (no prefix yet) for both targets
#pragma unroll 1
  for (int w = 0; w < 2; w++) {
    unsigned long long* hh = hist + ((size_t)k * 2 + (pass == 0 ? 0 : w)) * MCP_SELECT_BINS;
    unsigned long long c[PER], tot = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) { c[i] = hh[threadIdx.x * PER + i]; tot += c[i]; }
    // read the state BEFORE the barriers of the scan: the owning thread rewrites it right after them, and a wave that
    // read it late would see the new prefix and descend a second time
    SelectState st = pass == 0 ? SelectState{0u, 0u, w ? rank_hi : rank_lo} : state[2 * k + w];
    const unsigned long long before = block_exclusive_scan(tot, wtot);
    descend(pass, c, tot, before, st, &state[2 * k + w]);
    if (pass != 0 || w == 1) {                      // consumed: clear for the next pass (read-and-clear protocol)
#pragma unroll
      for (int i = 0; i < PER; i++) hh[threadIdx.x * PER + i] = 0ull;
    }
  }
}

// x of the key `key` (a terminal value's order-preserving image)
__device__ __forceinline__ double key_to_x(uint32_t key, double v0d, int compounding) {
  return terminal_to_x(key_to_float(key), v0d, compounding);
}

__device__ __forceinline__ void finish_stats(const mcp_params& prm, const mcp_record& m, const Quantile& q, mcp_stats* out) {
  mcp_stats s;
  s.n = (uint64_t)m.n;
  s.mean = m.n > 0 ? m.sum / m.n : 0.0;
  double m2 = m.sumsq - m.sum * s.mean;
  if (m2 < 0.0) m2 = 0.0;
  s.m2 = m2;
  s.std = m.n > 1 ? sqrt(m2 / (m.n - 1.0)) : 0.0;                    // ddof = 1, app.py:234
  s.sharpe = s.std > 0.0 ? (s.mean - prm.rf) / s.std : 0.0;         // app.py:711
  s.var = q.var; s.x_lo = q.x_lo; s.x_hi = q.x_hi;
  s.n_tail = q.n_tail;
  s.sum_tail = m.below + q.level2;
  s.cvar = s.n_tail > 0 ? s.sum_tail / (double)s.n_tail : s.var;    // app.py:263
  s.min = m.min; s.max = m.max;
  *out = s;
}

// ---- final: last descent, quantile, tail; grid = K --------------------------------------------------------------------
// Tail {x <= var} (app.py:261-263).  x is non-decreasing in the key, so the tail is a prefix of the key order: every key
// below the bucket of the low order statistic (counted by the ranks, summed by the hist passes), plus the digits of that
// bucket -- and of the high statistic's bucket when it is another one -- whose x is <= var.  Walking the digits rather
// than assuming "keys <= key_lo" keeps `x <= var` literal where neighbouring terminal values collapse onto one double
// (V/v0 below ~2e-9: a portfolio that lost everything); only a collapse that runs past the end of a 1024-key bucket is
// resolved on key order instead (SPEC.md section 5).
__global__ void __launch_bounds__(SB) final_kernel(const mcp_params prm, double gamma, uint64_t rank_lo, uint64_t rank_hi, int G, int slots,
                                                   const double* __restrict__ partials, unsigned long long* __restrict__ hist,
                                                   const SelectState* __restrict__ state, mcp_record* __restrict__ record,
                                                   Quantile* __restrict__ quant, mcp_stats* __restrict__ stats) {
  __shared__ unsigned long long wtot[4];
  __shared__ double red[4];
  __shared__ uint32_t s_key[2];
  __shared__ double s_q[3];
  const int k = blockIdx.x;
  const double v0d = (double)(float)prm.v0;
  // below partials of hist pass 2
  const double* pp = partials + (size_t)k * slots * PARTIAL_DOUBLES;
  double bl = 0.0;
  for (int b = threadIdx.x; b < G; b += SB) bl += pp[(size_t)b * PARTIAL_DOUBLES + 5];
  bl = block_sum(bl, red);

  unsigned long long c[2][PER];
  const SelectState st0 = state[2 * k + 0], st1 = state[2 * k + 1];
#pragma unroll
  for (int w = 0; w < 2; w++) {
    unsigned long long* hh = hist + ((size_t)k * 2 + w) * MCP_SELECT_BINS;
    unsigned long long tot = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) { c[w][i] = hh[threadIdx.x * PER + i]; tot += c[w][i]; hh[threadIdx.x * PER + i] = 0ull; }
    const unsigned long long before = block_exclusive_scan(tot, wtot);
    const SelectState st = w ? st1 : st0;
    if (st.rank >= before && st.rank < before + tot) {       // exactly one thread owns the rank
      unsigned long long bf = before;
#pragma unroll
      for (int i = 0; i < PER; i++) {
        if (st.rank < bf + c[w][i]) { s_key[w] = (st.prefix << 10) | (uint32_t)(threadIdx.x * PER + i); break; }
        bf += c[w][i];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a = key_to_x(s_key[0], v0d, prm.compounding), b = key_to_x(s_key[1], v0d, prm.compounding);
    const double diff = b - a;                       // numpy 2.2 _lerp (method 'linear')
    double r = a + diff * gamma;
    if (gamma >= 0.5) r = b - diff * (1.0 - gamma);
    s_q[0] = a; s_q[1] = b; s_q[2] = r;
  }
  __syncthreads();
  const double var = s_q[2];
  double cnt = 0.0, sum = 0.0;                       // counts < 2^53: exact in double
#pragma unroll
  for (int w = 0; w < 2; w++) {
    if (w == 1 && st1.prefix == st0.prefix) break;   // same bucket: already walked
    const uint32_t pre = (w ? st1.prefix : st0.prefix) << 10;
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const uint32_t d = (uint32_t)(threadIdx.x * PER + i);
      if (c[w][i] && d < 1024u) {
        const double x = key_to_x(pre | d, v0d, prm.compounding);
        if (x <= var) { cnt += (double)c[w][i]; sum += (double)c[w][i] * x; }
      }
    }
  }
  cnt = block_sum(cnt, red);
  sum = block_sum(sum, red);
  if (threadIdx.x == 0) {
    // keys below the low bucket: rank_lo minus the rank the target still has inside its bucket
    const unsigned long long n_tail = (rank_lo - st0.rank) + (unsigned long long)cnt;
    Quantile q = {s_q[0], s_q[1], var, sum, n_tail, 0ull};
    quant[k] = q;
    mcp_record m = record[k];
    m.below += bl;
    record[k] = m;
    if (stats) finish_stats(prm, m, q, &stats[k]);
  }
  (void)rank_hi;
}

// ---- several ranks: merge the gathered records [world][K] in rank order and finish ------------------------------------
__global__ void stats_kernel(const mcp_params prm, int K, int world, const mcp_record* __restrict__ gathered,
                             const Quantile* __restrict__ quant, mcp_stats* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  mcp_record m = gathered[k];
  for (int r = 1; r < world; r++) {
    const mcp_record g = gathered[(size_t)r * K + k];
    m.n += g.n; m.sum += g.sum; m.sumsq += g.sumsq; m.min = fmin(m.min, g.min); m.max = fmax(m.max, g.max);
    m.below += g.below;
  }
  finish_stats(prm, m, quant[k], &out[k]);
}

// ---- same-device exchange: every buffer <- element-wise sum of all buffers ----------------------------------------------
struct PtrList { unsigned long long* p[8]; int n; };
__global__ void __launch_bounds__(SB) sum_u64_kernel(PtrList l, size_t words) {
  for (size_t i = (size_t)blockIdx.x * SB + threadIdx.x; i < words; i += (size_t)gridDim.x * SB) {
    unsigned long long t = 0;
    for (int j = 0; j < l.n; j++) t += l.p[j][i];
    for (int j = 0; j < l.n; j++) l.p[j][i] = t;
  }
}

// The normal generator on its own: z[i] = inverse-CDF normal of word x[i] (SPEC.md section 3); also what tests use
// to hit edge inputs (both ends of every octave, u == 1/2, the deepest tail).
__global__ void __launch_bounds__(256) normals_kernel(const uint32_t* __restrict__ x, uint64_t n,
                                                      const float4* __restrict__ table, float* __restrict__ z) {
  __shared__ float4 s_tab[ICDF_LDS_ENTRIES];
  for (int i = threadIdx.x; i < ICDF_ENTRIES; i += 256) s_tab[ICDF_PAD + i] = table[i];
  __syncthreads();
  const IcdfConsts kc = icdf_consts();
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) z[i] = normal_icdf(x[i], s_tab, kc);
}

hipError_t launch_normals(const uint32_t* x, uint64_t n, const float4* table, float* z, hipStream_t s) {
  uint64_t g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  normals_kernel<<<(unsigned)g, 256, 0, s>>>(x, n, table, z);
  return hipGetLastError();
}

// ---- launch wrappers (enqueue only) -------------------------------------------------------------
static bool grid_ok(int K, int G) { return (uint64_t)K * (uint64_t)G <= 0x7fffffffull; }

hipError_t launch_pass0(const mcp_params& prm, int K, const float* terminal, uint64_t stride, uint64_t n, double* partials,
                        unsigned long long* hist, hipStream_t s) {
  const int G = stream_grid(n, K);
  if (!grid_ok(K, G)) return hipErrorInvalidValue;
  pass0_kernel<<<(unsigned)(K * G), SB, 0, s>>>(prm, terminal, stride, n, G, stream_slots(K), partials, hist);
  return hipGetLastError();
}

hipError_t launch_scan(int K, int pass, uint64_t n, uint64_t rank_lo, uint64_t rank_hi, const double* partials,
                       unsigned long long* hist, SelectState* state, mcp_record* record, hipStream_t s) {
  scan_kernel<<<(unsigned)K, SB, 0, s>>>(pass, rank_lo, rank_hi, stream_grid(n, K), stream_slots(K), partials, hist, state, record);
  return hipGetLastError();
}

hipError_t launch_hist(const mcp_params& prm, int K, int pass, const float* terminal, uint64_t stride, uint64_t n,
                       const SelectState* state, double* partials, unsigned long long* hist, hipStream_t s) {
  const int G = stream_grid(n, K);
  if (!grid_ok(K, G)) return hipErrorInvalidValue;
  hist_kernel<<<(unsigned)(K * G), SB, 0, s>>>(prm, pass, terminal, stride, n, G, stream_slots(K), state, partials, hist);
  return hipGetLastError();
}

hipError_t launch_final(const mcp_params& prm, int K, uint64_t n, double gamma, uint64_t rank_lo, uint64_t rank_hi,
                        const double* partials, unsigned long long* hist, const SelectState* state, mcp_record* record,
                        Quantile* quant, mcp_stats* stats_or_null, hipStream_t s) {
  final_kernel<<<(unsigned)K, SB, 0, s>>>(prm, gamma, rank_lo, rank_hi, stream_grid(n, K), stream_slots(K), partials, hist, state,
                                          record, quant, stats_or_null);
  return hipGetLastError();
}

hipError_t launch_stats(const mcp_params& prm, int K, int world, const mcp_record* gathered, const Quantile* quant,
                        mcp_stats* out, hipStream_t s) {
  stats_kernel<<<(K + 63) / 64, 64, 0, s>>>(prm, K, world, gathered, quant, out);
  return hipGetLastError();
}

__global__ void __launch_bounds__(SB) zero_u64_kernel(unsigned long long* __restrict__ p, size_t words) {
  for (size_t i = (size_t)blockIdx.x * SB + threadIdx.x; i < words; i += (size_t)gridDim.x * SB) p[i] = 0ull;
}

// A plain kernel instead of hipMemsetAsync: ordered like any other launch on the stream, and safe inside a hipGraph
// (a captured memset node replayed wrongly on the ROCm 7.0 runtime bundled with torch).  bytes is a multiple of 8.
hipError_t launch_zero(void* p, size_t bytes, hipStream_t s) {
  const size_t words = bytes / 8;
  size_t g = (words + SB - 1) / SB;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  zero_u64_kernel<<<(unsigned)g, SB, 0, s>>>((unsigned long long*)p, words);
  return hipGetLastError();
}

hipError_t launch_sum_u64(unsigned long long* const* bufs, int nsrc, size_t words, hipStream_t s) {
  if (nsrc < 1 || nsrc > 8) return hipErrorInvalidValue;
  PtrList l;
  for (int j = 0; j < 8; j++) l.p[j] = j < nsrc ? bufs[j] : nullptr;
  l.n = nsrc;
  size_t g = (words + SB - 1) / SB;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  sum_u64_kernel<<<(unsigned)g, SB, 0, s>>>(l, words);
  return hipGetLastError();
}

}  // namespace mcp
