// mcp_stats_kernels.hip -- reductions behind the path kernels (gfx950).
//
// The moments and (for K <= 16) the digit-0 histogram of the radix select are produced by the path kernels' own epilogue
// (mcp_paths.h, mcp_sweep_paths.hip: V still in registers).  What remains here:
//
//   hist_kernel<0> digit-0 histogram (key bits 31..21) behind the MFMA sweep kernels, whose workgroups hold 512 portfolios
//                  and cannot keep 512 histograms in LDS: one lean read of V_T, no floating point.
//   pass0_kernel   standalone pass 0 over CALLER-SUPPLIED terminal values: the same MomentPartial records and digit-0
//                  histogram the fused epilogue leaves (mcp_launch_pass0).
//   scan_kernel    pass 0: MomentPartial records -> moment record, ranks -> select state, descend into the digit holding each
//                  rank; pass 1: the same descent plus the "below" partials of hist pass 1.  Clears the histogram it consumed.
//   hist_kernel<1|2> one read of V_T: digit histograms of the keys that match the prefixes found so far (bits 20..10, then
//                  9..0) AND the fp64 sum over the elements that sort strictly below the bucket of the low order statistic
//                  (what the CVaR tail mean needs; no separate tail pass).
//   final_kernel   last descent -> the two order statistics -> np.percentile's linear interpolation (app.py:258-259,
//                  numpy 2.2 _lerp); tail {x <= var} = keys below the low bucket (count from the ranks, sum from the hist
//                  passes) + the digits of the last bucket(s) whose x <= var (counts x values) (app.py:261-263); one rank:
//                  mean, std(ddof=1), Sharpe (app.py:711), CVaR.
//   stats_kernel   several ranks: merges the all-gathered records in rank order and finishes the same way.
//
// Moments are SHIFTED sums: every rank accumulates sum (x - c) and sum (x - c)^2 around the same
// per-portfolio pivot c (the analytic mean, mcp_pivots), so the merged variance (S2 - S1^2/n)/(n-1) does not cancel when
// sigma << |mean| (np.std(ddof=1) of app.py:234 is two-pass).  The "below" sums are kept as sum (V - v0) (simple
// compounding: x = (V - v0)/v0, no fp64 divide per element) or sum expm1(S) (log).
//
// Exchanges of a multi-GPU host (include/mcport.h): all-reduce(SUM, u64) of the histogram after paths / hist(1) /
// hist(2); all-gather of the [K] records after final.  Everything is fixed-order fp64 or integer: run-to-run
// deterministic.  HBM-bound streaming passes over V_T (4 B/path each) or trivially small.
#include <cstdlib>

#include "mcp_paths.h"
#include "mcp_stats_kernels.h"

#ifdef MCP_DIAG_CLOCK
namespace mcp { __device__ unsigned long long mcp_diag_stamps[2 * 8192]; }
// diagnostic build only: copies the stamps of the last mc_paths_kernel launch to the host (2 x n words)
extern "C" int mcp_diag_read(unsigned long long* out, int n_blocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcp::mcp_diag_stamps), (size_t)n_blocks * 2 * sizeof(unsigned long long));
}
#endif

namespace mcp {

constexpr int SB = 256;                               // threads per block of every kernel here
constexpr int PER = MCP_SELECT_BINS / SB;             // 8 consecutive bins per thread in the scans

__device__ __forceinline__ void pass_shape(int pass, int& shift, int& bits, int& pshift) {
  // pass 0: key[31:21]; pass 1: key[20:10]; pass 2: key[9:0]
  shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
  bits = pass == 2 ? 10 : 11;
  pshift = pass == 0 ? 32 : (pass == 1 ? 21 : 10);   // prefix = key >> pshift (pass 0: no prefix)
}

// blocks per portfolio of a streaming pass over n values: >= `per` values per 256-thread block, at most stream_slots(K).
static int stream_grid(uint64_t n, int K) {
  static const uint64_t per = [] { const char* e = getenv("MCP_STREAM_ELEMS"); const long v = e ? atol(e) : 0; return (uint64_t)(v > 0 ? v : 8192); }();
  uint64_t g = (n + per - 1) / per;
  if (g < 1) g = 1;
  const uint64_t cap = (uint64_t)stream_slots(K);
  return (int)(g > cap ? cap : g);
}

// "below" contribution of one terminal value (see the header comment)
template <bool LOGC>
__device__ __forceinline__ double below_term(float v, double v0d) {
  if constexpr (LOGC) return expm1((double)v);
  else return (double)v - v0d;
}

// ---- standalone pass 0: moments + digit-0 histogram of caller-supplied terminal values ------------------------------
// grid = K * G (block b of portfolio k at blockIdx.x = k*G + b: no 65,535 limit on K).  hist[k][0][*] must be zero on
// entry (the scans clear what they consume; buffers start zeroed).  Fills slot b < G of the portfolio's MomentPartial
// records and pads the remaining slots (up to `slots`, what a fused path launch of the same shape fills) with empty ones.
__global__ void __launch_bounds__(SB) pass0_kernel(const mcp_params prm, const float* __restrict__ terminal, uint64_t stride,
                                                   uint64_t n, int G, uint64_t slots, const double* __restrict__ pivot,
                                                   MomentPartial* __restrict__ partials, unsigned long long* __restrict__ hist) {
  __shared__ uint32_t h[MCP_SELECT_BINS];
  __shared__ double red[4][2];
  __shared__ float ext[4][2];
  __shared__ unsigned long long cnt[4];
  const int k = blockIdx.x / G, b = blockIdx.x % G;
  for (int i = threadIdx.x; i < MCP_SELECT_BINS; i += SB) h[i] = 0u;
  __syncthreads();
  const double v0d = (double)(float)prm.v0;
  const double c = pivot ? pivot[k] : 0.0;
  const float* __restrict__ src = terminal + (size_t)k * stride;
  double s1 = 0.0, s2 = 0.0;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  unsigned long long m = 0;
  const uint64_t step = (uint64_t)G * SB;
  for (uint64_t i0 = (uint64_t)b * SB; i0 < n; i0 += step) {     // uniform trip count: ballots inside see whole waves
    const uint64_t i = i0 + threadIdx.x;
    const bool live = i < n;
    const float v = live ? src[i] : 0.0f;
    if (live) {
      const double d = terminal_to_x(v, v0d, prm.compounding) - c;
      m += 1; s1 += d; s2 = __builtin_fma(d, d, s2); mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    lds_hist_add(h, float_to_key(v) >> 21, live);
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2); mn = wave_minf(mn); mx = wave_maxf(mx);
  m = (unsigned long long)wave_sum((double)m);                   // < 2^53: exact
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv][0] = s1; red[wv][1] = s2; ext[wv][0] = mn; ext[wv][1] = mx; cnt[wv] = m; }
  __syncthreads();
  MomentPartial* row = partials + (size_t)k * slots;
  if (threadIdx.x == 0) {
    MomentPartial o;
    o.s1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    o.s2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    o.vmin = fminf(fminf(ext[0][0], ext[1][0]), fminf(ext[2][0], ext[3][0]));
    o.vmax = fmaxf(fmaxf(ext[0][1], ext[1][1]), fmaxf(ext[2][1], ext[3][1]));
    o.n = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]);
    row[b] = o;
  }
  for (uint64_t j = (uint64_t)G + b * (uint64_t)SB + threadIdx.x; j < slots; j += step) {   // empty records behind the G real ones
    MomentPartial e;
    e.s1 = 0.0; e.s2 = 0.0; e.vmin = __builtin_inff(); e.vmax = -__builtin_inff(); e.n = 0ull;
    row[j] = e;
  }
  unsigned long long* out = hist + (size_t)k * 2 * MCP_SELECT_BINS;
  for (int i = threadIdx.x; i < MCP_SELECT_BINS; i += SB) {
    const uint32_t v = h[i];
    if (v) atomicAdd(&out[i], (unsigned long long)v);
  }
}

// ---- streaming select passes ----------------------------------------------------------------------------------------
// PASS 0 (behind the sweep kernels): hist[k][0][key >> 21] += 1, integer only.  Terminal values cluster on a dozen digit-0
// bins, where 64 lanes adding to one LDS word serialise; so every lane counts into its own column of a 16-bin window
// [dlo, dlo + 16) placed around the digit of the portfolio's pivot value (conflict-free ds_add_u32; the four waves of the
// block share the 64 columns, the adds are atomic), and only digits outside the window go to the full histogram.
// PASS 1 / 2: hist[k][w][digit] += #{keys of portfolio k matching state[k][w].prefix}; when both targets share a prefix (the
// usual case) only hist[k][0] is filled and the scans read it for both.  partial "below" = sum over the elements that
// sort below state[k][0]'s bucket and were not already summed by the previous pass.  grid = K * G.
constexpr int WIN = 16;                                // digit-0 window bins
template <int PASS, bool LOGC>
__global__ void __launch_bounds__(SB) hist_kernel(const mcp_params prm, const float* __restrict__ terminal, uint64_t stride, uint64_t n,
                                                  int G, int slots, const SelectState* __restrict__ state, const double* __restrict__ pivot,
                                                  double* __restrict__ below_out, unsigned long long* __restrict__ hist) {
  constexpr int NH = PASS == 0 ? 1 : 2;
  __shared__ uint32_t h[NH][MCP_SELECT_BINS];
  __shared__ uint32_t win[PASS == 0 ? WIN : 1][64];
  __shared__ double red[4];
  const int k = blockIdx.x / G, b = blockIdx.x % G;
  for (int i = threadIdx.x; i < NH * MCP_SELECT_BINS; i += SB) (&h[0][0])[i] = 0u;
  if constexpr (PASS == 0)
    for (int i = threadIdx.x; i < WIN * 64; i += SB) (&win[0][0])[i] = 0u;
  __syncthreads();
  int shift, bits, pshift;
  pass_shape(PASS, shift, bits, pshift);
  const uint32_t mask = (1u << bits) - 1u;
  const double v0d = (double)(float)prm.v0;
  uint32_t pa = 0, pb = 0, dlo = 0;
  if constexpr (PASS == 0) {
    // window around the terminal value that corresponds to the pivot: V = v0 (1 + c) (simple) or S = log1p(c) (log)
    const double c = pivot ? pivot[k] : 0.0;
    const float vc = LOGC ? (float)log1p(c > -1.0 ? c : 0.0) : (float)(v0d * (1.0 + c));
    const uint32_t dc = float_to_key(vc) >> 21;
    dlo = dc >= WIN / 2 ? dc - WIN / 2 : 0u;
    if (dlo > (uint32_t)(MCP_SELECT_BINS - WIN)) dlo = MCP_SELECT_BINS - WIN;
  } else {
    pa = state[2 * k + 0].prefix; pb = state[2 * k + 1].prefix;
  }
  const bool two = pa != pb;                              // block-uniform
  const int lane = threadIdx.x & 63;
  const float* __restrict__ src = terminal + (size_t)k * stride;
  double below = 0.0;

  auto one = [&](float v) {
    const uint32_t key = float_to_key(v);
    if constexpr (PASS == 0) {
      const uint32_t d = key >> 21, t = d - dlo;
      if (t < (uint32_t)WIN) atomicAdd(&win[t][lane], 1u);
      else atomicAdd(&h[0][d], 1u);
    } else {
      const uint32_t pre = key >> pshift, d = (key >> shift) & mask;
      if (pre == pa) atomicAdd(&h[0][d], 1u);
      if (two && pre == pb) atomicAdd(&h[1][d], 1u);
      // pass 1: digit-0 below the bucket's; pass 2: inside the digit-0 bucket, digit-1 below
      if (pre < pa && (PASS == 1 || (pre >> 11) == (pa >> 11))) below += below_term<LOGC>(v, v0d);
    }
  };

  // rows start at arbitrary element offsets (stride is the caller's): scalar head up to 16-byte alignment, float4 body, scalar tail
  const uint64_t head = n ? (uint64_t)((4 - (((uintptr_t)src >> 2) & 3)) & 3) : 0;
  const uint64_t hd = head < n ? head : n;
  const uint64_t nvec = (n - hd) / 4;
  const float4* __restrict__ vsrc = (const float4*)(src + hd);
  const uint64_t vstep = (uint64_t)G * SB;
  uint64_t j = (uint64_t)b * SB + threadIdx.x;
  for (; j + vstep < nvec; j += 2 * vstep) {              // two 16-byte loads in flight per lane
    const float4 q0 = vsrc[j], q1 = vsrc[j + vstep];
    one(q0.x); one(q0.y); one(q0.z); one(q0.w);
    one(q1.x); one(q1.y); one(q1.z); one(q1.w);
  }
  for (; j < nvec; j += vstep) {
    const float4 q0 = vsrc[j];
    one(q0.x); one(q0.y); one(q0.z); one(q0.w);
  }
  if (b == 0) {                                           // head and tail scalars: at most 6 elements of the row
    if (threadIdx.x < hd) one(src[threadIdx.x]);
    const uint64_t t0 = hd + 4 * nvec;
    if (t0 + threadIdx.x < n) one(src[t0 + threadIdx.x]);
  }

  if constexpr (PASS != 0) {
    below = wave_sum(below);
    const int wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = below;
  }
  __syncthreads();
  if constexpr (PASS != 0) {
    if (threadIdx.x == 0) below_out[(size_t)k * slots + b] = (red[0] + red[1]) + (red[2] + red[3]);
  } else {
    if (threadIdx.x < WIN) {                              // fold the 64 columns of window bin t into the histogram
      uint32_t t = 0;
      for (int l = 0; l < 64; l++) t += win[threadIdx.x][(l + threadIdx.x) & 63];
      if (t) atomicAdd(&h[0][dlo + threadIdx.x], t);
    }
    __syncthreads();
  }
  unsigned long long* out = hist + (size_t)k * 2 * MCP_SELECT_BINS;
  const int nh = (PASS == 0 || !two) ? 1 : 2;
  for (int i = threadIdx.x; i < nh * MCP_SELECT_BINS; i += SB) {
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
__global__ void __launch_bounds__(SB) scan_kernel(const mcp_params prm, int pass, uint64_t rank_lo, uint64_t rank_hi, int G, int bslots,
                                                  uint64_t mslots, const MomentPartial* __restrict__ partials,
                                                  const double* __restrict__ below, const double* __restrict__ pivot,
                                                  unsigned long long* __restrict__ hist, SelectState* __restrict__ state,
                                                  mcp_record* __restrict__ record) {
  __shared__ unsigned long long wtot[4];
  __shared__ double red[4][2];
  __shared__ float ext[4][2];
  __shared__ unsigned long long cnt[4];
  const int k = blockIdx.x;
  // 1. reduce the partials of the pass that produced this histogram (fixed order)
  if (pass == 0) {
    const MomentPartial* pp = partials + (size_t)k * mslots;
    double s1 = 0.0, s2 = 0.0;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    unsigned long long m = 0;
    for (uint64_t b = threadIdx.x; b < mslots; b += SB) {
      const MomentPartial p = pp[b];
      m += p.n; s1 += p.s1; s2 += p.s2; mn = fminf(mn, p.vmin); mx = fmaxf(mx, p.vmax);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2); mn = wave_minf(mn); mx = wave_maxf(mx);
    m = (unsigned long long)wave_sum((double)m);                 // < 2^53: exact
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = s1; red[wv][1] = s2; ext[wv][0] = mn; ext[wv][1] = mx; cnt[wv] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double v0d = (double)(float)prm.v0;
      mcp_record r;
      r.n = (double)((cnt[0] + cnt[1]) + (cnt[2] + cnt[3]));
      r.sum = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
      r.sumsq = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
      const float vmn = fminf(fminf(ext[0][0], ext[1][0]), fminf(ext[2][0], ext[3][0]));
      const float vmx = fmaxf(fmaxf(ext[0][1], ext[1][1]), fmaxf(ext[2][1], ext[3][1]));
      r.min = r.n > 0 ? terminal_to_x(vmn, v0d, prm.compounding) : __builtin_inf();    // x is monotone in the terminal value
      r.max = r.n > 0 ? terminal_to_x(vmx, v0d, prm.compounding) : -__builtin_inf();
      r.below = 0.0;
      r.pivot = pivot ? pivot[k] : 0.0;
      r.pad = 0.0;
      record[k] = r;
    }
  } else {
    const double* pp = below + (size_t)k * bslots;
    double bl = 0.0;
    for (int b = threadIdx.x; b < G; b += SB) bl += pp[b];
    bl = block_sum(bl, &red[0][0]);
    if (threadIdx.x == 0) record[k].below = bl;
  }
  // 2. descend.  Pass 0 has one histogram (no prefix yet) for both targets; so has pass 1 when both targets went into
  //    the same digit-0 bucket (hist_kernel fills only [k][0] then).  Both states are read BEFORE the barriers of the
  //    scans: the owning thread rewrites them right after, and a wave that read late would descend a second time.
  const SelectState in0 = pass == 0 ? SelectState{0u, 0u, rank_lo} : state[2 * k + 0];
  const SelectState in1 = pass == 0 ? SelectState{0u, 0u, rank_hi} : state[2 * k + 1];
  const bool shared_hist = pass == 0 || in0.prefix == in1.prefix;
#pragma unroll 1
  for (int w = 0; w < 2; w++) {
    unsigned long long* hh = hist + ((size_t)k * 2 + (shared_hist ? 0 : w)) * MCP_SELECT_BINS;
    unsigned long long c[PER], tot = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) { c[i] = hh[threadIdx.x * PER + i]; tot += c[i]; }
    SelectState st = w ? in1 : in0;
    const unsigned long long before = block_exclusive_scan(tot, wtot);
    descend(pass, c, tot, before, st, &state[2 * k + w]);
    if (!shared_hist || w == 1) {                   // consumed: clear for the next pass (read-and-clear protocol)
#pragma unroll
      for (int i = 0; i < PER; i++) hh[threadIdx.x * PER + i] = 0ull;
    }
  }
}

// x of the key `key` (a terminal value's order-preserving image)
__device__ __forceinline__ double key_to_x(uint32_t key, double v0d, int compounding) {
  return terminal_to_x(key_to_float(key), v0d, compounding);
}

// moments from the shifted sums: mean = c + S1/n, sum (x - mean)^2 = S2 - S1^2/n  (no cancellation: c is the analytic mean)
__device__ __forceinline__ void finish_stats(const mcp_params& prm, const mcp_record& m, const Quantile& q, mcp_stats* out) {
  const double v0d = (double)(float)prm.v0;
  mcp_stats s;
  s.n = (uint64_t)m.n;
  const double dm = m.n > 0 ? m.sum / m.n : 0.0;
  s.mean = m.pivot + dm;
  double m2 = m.sumsq - m.sum * dm;
  if (m2 < 0.0) m2 = 0.0;
  s.m2 = m2;
  s.std = m.n > 1 ? sqrt(m2 / (m.n - 1.0)) : 0.0;                    // ddof = 1, app.py:234
  s.sharpe = s.std > 0.0 ? (s.mean - prm.rf) / s.std : 0.0;         // app.py:711
  s.var = q.var; s.x_lo = q.x_lo; s.x_hi = q.x_hi;
  s.n_tail = q.n_tail;
  const double below_x = prm.compounding == MCP_COMPOUND_LOG ? m.below : m.below / v0d;   // sum (V - v0) / v0 = sum x
  s.sum_tail = below_x + q.level2;
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
__global__ void __launch_bounds__(SB) final_kernel(const mcp_params prm, double gamma, uint64_t rank_lo, uint64_t rank_hi, int G, int bslots,
                                                   const double* __restrict__ below, unsigned long long* __restrict__ hist,
                                                   const SelectState* __restrict__ state, mcp_record* __restrict__ record,
                                                   Quantile* __restrict__ quant, mcp_stats* __restrict__ stats) {
  __shared__ unsigned long long wtot[4];
  __shared__ double red[4];
  __shared__ uint32_t s_key[2];
  __shared__ double s_q[3];
  const int k = blockIdx.x;
  const double v0d = (double)(float)prm.v0;
  // below partials of hist pass 2
  const double* pp = below + (size_t)k * bslots;
  double bl = 0.0;
  for (int b = threadIdx.x; b < G; b += SB) bl += pp[b];
  bl = block_sum(bl, red);

  unsigned long long c[2][PER];
  const SelectState st0 = state[2 * k + 0], st1 = state[2 * k + 1];
  const bool shared_hist = st0.prefix == st1.prefix;        // hist pass 2 filled only [k][0]
#pragma unroll
  for (int w = 0; w < 2; w++) {
    unsigned long long* hh = hist + ((size_t)k * 2 + (shared_hist ? 0 : w)) * MCP_SELECT_BINS;
    unsigned long long tot = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) { c[w][i] = hh[threadIdx.x * PER + i]; tot += c[w][i]; }
    if (!shared_hist || w == 1) {
#pragma unroll
      for (int i = 0; i < PER; i++) hh[threadIdx.x * PER + i] = 0ull;
    }
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
    if (w == 1 && shared_hist) break;                // same bucket: already walked
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
// Every rank used the same pivot (mcp_pivots is a function of the inputs), so the shifted sums simply add.
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

hipError_t launch_pass0(const mcp_params& prm, int K, const float* terminal, uint64_t stride, uint64_t n, const double* pivot,
                        uint64_t slots, MomentPartial* partials, unsigned long long* hist, hipStream_t s) {
  int G = stream_grid(n, K);
  if ((uint64_t)G > slots) G = (int)slots;
  if (G < 1 || !grid_ok(K, G)) return hipErrorInvalidValue;
  pass0_kernel<<<(unsigned)(K * G), SB, 0, s>>>(prm, terminal, stride, n, G, slots, pivot, partials, hist);
  return hipGetLastError();
}

hipError_t launch_scan(const mcp_params& prm, int K, int pass, uint64_t n, uint64_t rank_lo, uint64_t rank_hi, uint64_t slots,
                       const MomentPartial* partials, const double* below, const double* pivot, unsigned long long* hist,
                       SelectState* state, mcp_record* record, hipStream_t s) {
  scan_kernel<<<(unsigned)K, SB, 0, s>>>(prm, pass, rank_lo, rank_hi, stream_grid(n, K), stream_slots(K), slots, partials, below, pivot,
                                         hist, state, record);
  return hipGetLastError();
}

hipError_t launch_hist(const mcp_params& prm, int K, int pass, const float* terminal, uint64_t stride, uint64_t n,
                       const SelectState* state, const double* pivot, double* below, unsigned long long* hist, hipStream_t s) {
  const int G = stream_grid(n, K);
  if (!grid_ok(K, G)) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)(K * G);
  const int bs = stream_slots(K);
  const bool lg = prm.compounding == MCP_COMPOUND_LOG;
#define MCP_HIST(P)                                                                                              \
  do {                                                                                                           \
    if (lg) hist_kernel<P, true><<<grid, SB, 0, s>>>(prm, terminal, stride, n, G, bs, state, pivot, below, hist); \
    else hist_kernel<P, false><<<grid, SB, 0, s>>>(prm, terminal, stride, n, G, bs, state, pivot, below, hist);   \
  } while (0)
  if (pass == 0) MCP_HIST(0);
  else if (pass == 1) MCP_HIST(1);
  else if (pass == 2) MCP_HIST(2);
  else return hipErrorInvalidValue;
#undef MCP_HIST
  return hipGetLastError();
}

hipError_t launch_final(const mcp_params& prm, int K, uint64_t n, double gamma, uint64_t rank_lo, uint64_t rank_hi,
                        const double* below, unsigned long long* hist, const SelectState* state, mcp_record* record,
                        Quantile* quant, mcp_stats* stats_or_null, hipStream_t s) {
  final_kernel<<<(unsigned)K, SB, 0, s>>>(prm, gamma, rank_lo, rank_hi, stream_grid(n, K), stream_slots(K), below, hist, state,
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
