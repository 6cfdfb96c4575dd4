// mcp_api.cpp -- the C ABI of libmcport.so (include/mcport.h): argument checking, parameter packing,
// enqueue-only launch entry points, and the host-level mcp_simulate() that strings them together on
// one device.  No torch types, no exceptions across the boundary.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/mcport.h"
#include "mcp_device.h"
#include "mcp_paths.h"
#include "mcp_stats_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(MCP_E_NODEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

constexpr int KT_WIDE = 8;           // portfolios per pass of the KT=8 kernel
constexpr int K_PAD = 512;           // W rows are zero-padded to a multiple of this (MFMA sweep tile: 32*MT)
constexpr int SWEEP_MIN_K = 17;      // from this many portfolios on (and N <= 16) the MFMA sweep kernel runs
constexpr int GRID_CAP = 8192;       // path-kernel blocks; tiles beyond are grid-strided

inline int n4_of(int n) { return 4 * ((n + 3) / 4); }
inline int kpad_of(int k) { return K_PAD * ((k + K_PAD - 1) / K_PAD); }

int check_params(const mcp_params* p) {
  if (!p) return fail(MCP_E_ARG, "params is NULL");
  if (p->n_assets < 1 || p->n_assets > MCP_MAX_ASSETS)
    return fail(MCP_E_ARG, "n_assets=%d outside [1,%d]", p->n_assets, MCP_MAX_ASSETS);
  if (p->n_steps < 0) return fail(MCP_E_ARG, "n_steps=%d < 0", p->n_steps);
  if (p->n_portfolios < 1) return fail(MCP_E_ARG, "n_portfolios=%d < 1", p->n_portfolios);
  if (p->compounding != MCP_COMPOUND_SIMPLE && p->compounding != MCP_COMPOUND_LOG)
    return fail(MCP_E_ARG, "compounding=%d unknown", p->compounding);
  if (!(p->alpha > 0.0 && p->alpha < 1.0)) return fail(MCP_E_ARG, "alpha=%g outside (0,1)", p->alpha);
  if (!(p->v0 > 0.0) || !std::isfinite(p->v0)) return fail(MCP_E_ARG, "v0=%g must be positive", p->v0);
  return MCP_OK;
}

int paths_per_thread(const mcp_params* p) {
#ifdef MCP_EXP_PPT2
  static const int env_ppt = [] {
    const char* e = getenv("MCP_PPT");
    return e ? atoi(e) : 1;
  }();
  const int nb = (p->n_assets + 3) / 4;
  return (env_ppt == 2 && nb <= 4 && p->n_portfolios == 1) ? 2 : 1;
#else
  (void)p;
  return 1;
#endif
}

const mcp::launch_paths_fn k_launch[16] = {
    mcp::launch_paths_nb1,  mcp::launch_paths_nb2,  mcp::launch_paths_nb3,  mcp::launch_paths_nb4,
    mcp::launch_paths_nb5,  mcp::launch_paths_nb6,  mcp::launch_paths_nb7,  mcp::launch_paths_nb8,
    mcp::launch_paths_nb9,  mcp::launch_paths_nb10, mcp::launch_paths_nb11, mcp::launch_paths_nb12,
    mcp::launch_paths_nb13, mcp::launch_paths_nb14, mcp::launch_paths_nb15, mcp::launch_paths_nb16};

// Box-Muller tables, one device-resident copy per device, built on first use by tables_init_kernel on
// the caller's stream (so later launches on that stream are ordered after it).  The first call on a
// device allocates: do it once before capturing launches into a hipGraph.
constexpr int MAX_DEVICES = 64;
std::mutex g_tab_mu;
float2* g_tables[MAX_DEVICES] = {nullptr};

int device_tables(hipStream_t stream, const float2** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return fail(MCP_E_UNSUPPORTED, "device index %d", dev);
  std::lock_guard<std::mutex> lock(g_tab_mu);
  if (!g_tables[dev]) {
    float2* t = nullptr;
    if (hipMalloc((void**)&t, 2 * mcp::BM_TAB * sizeof(float2)) != hipSuccess)
      return fail(MCP_E_NOMEM, "hipMalloc of the Box-Muller tables failed");
    hipError_t e = mcp::launch_tables_init(t, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);     // other streams may use the tables next
    if (e != hipSuccess) { (void)hipFree(t); return fail(MCP_E_NODEVICE, "tables_init_kernel: %s", hipGetErrorString(e)); }
    g_tables[dev] = t;
  }
  *out = g_tables[dev];
  return MCP_OK;
}

}  // namespace

struct mcp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  // device buffers, grown on demand (never on the steady-state path)
  void* d[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float* d_packed = nullptr;
  size_t packed_cap = 0;
  float* d_terminal = nullptr;
  size_t terminal_cap = 0;
  float* h_packed = nullptr;   // pinned staging
  size_t h_packed_cap = 0;
  mcp_stats* h_stats = nullptr;
  size_t h_stats_cap = 0;
  double* d_sweep = nullptr;   // inputs then outputs of mcp_sweep_historical
  size_t sweep_cap = 0;
};

extern "C" {

int mcp_abi_version(void) { return MCP_ABI_VERSION; }

int mcp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* mcp_last_error(void) { return g_err.c_str(); }

size_t mcp_packed_len(int n_assets, int n_portfolios) {
  if (n_assets < 1 || n_assets > MCP_MAX_ASSETS || n_portfolios < 1) return 0;
  const size_t n4 = (size_t)n4_of(n_assets);
  return n4 + n4 * (n4 / 2 + 1) + (size_t)kpad_of(n_portfolios) * n4;
}

int mcp_pack_params(int n_assets, int n_portfolios, const float* mu, const float* chol, const float* W,
                    float* out, size_t out_len) {
  const size_t need = mcp_packed_len(n_assets, n_portfolios);
  if (need == 0) return fail(MCP_E_ARG, "bad shape N=%d K=%d", n_assets, n_portfolios);
  if (!mu || !chol || !W || !out) return fail(MCP_E_ARG, "NULL pointer");
  if (out_len < need) return fail(MCP_E_ARG, "packed buffer too small: %zu < %zu", out_len, need);
  const int N = n_assets, n4 = n4_of(N);
  memset(out, 0, need * sizeof(float));
  for (int i = 0; i < N; i++) out[i] = mu[i] + 0.0f;   // -0 -> +0 (SPEC.md section 4)
  // lower triangle in row pairs: pair m = rows (2m, 2m+1), columns j = 0..2m+1 interleaved as
  // (L[2m][j], L[2m+1][j]); L[2m][2m+1] is a structural zero.  Offset of pair m: 2m(m+1).
  float* L = out + n4;
  for (int i = 0; i < N; i++)
    for (int j = 0; j <= i; j++) L[2 * (i / 2) * (i / 2 + 1) + 2 * j + (i & 1)] = chol[(size_t)i * N + j];
  float* Wp = L + (size_t)n4 * (n4 / 2 + 1);
  for (int k = 0; k < n_portfolios; k++)
    for (int i = 0; i < N; i++) Wp[(size_t)k * n4 + i] = W[(size_t)k * N + i];
  return MCP_OK;
}

size_t mcp_ws_bytes(int which, int K) {
  if (K < 1) return 0;
  switch (which) {
    case MCP_WS_PARTIALS: return (size_t)K * mcp::MOMENTS_GRID * sizeof(mcp_moments);
    case MCP_WS_MOMENTS: return (size_t)K * sizeof(mcp_moments);
    case MCP_WS_STATE: return (size_t)K * 2 * sizeof(mcp::SelectState);
    case MCP_WS_HIST: return (size_t)K * 2 * MCP_SELECT_BINS * sizeof(unsigned long long);
    case MCP_WS_QUANT: return (size_t)K * sizeof(mcp::Quantile);
    case MCP_WS_TAIL_PARTIAL: return (size_t)K * mcp::TAIL_GRID * 2 * sizeof(double);
    case MCP_WS_TAIL: return (size_t)K * 2 * sizeof(double);
    case MCP_WS_STATS: return (size_t)K * sizeof(mcp_stats);
    default: return 0;
  }
}

static int paths_grid(const mcp_params* prm, uint64_t n_paths) {
  const uint64_t tile = (uint64_t)mcp::PATH_BLOCK * paths_per_thread(prm);
  uint64_t tiles = (n_paths + tile - 1) / tile;
  if (tiles < 1) tiles = 1;
  return (int)(tiles < (uint64_t)GRID_CAP ? tiles : (uint64_t)GRID_CAP);
}

int mcp_launch_paths(const mcp_params* prm, const float* d_packed, uint64_t seed, uint64_t path_begin,
                     uint64_t n_paths, float* d_terminal, uint64_t stride, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_packed || !d_terminal) return fail(MCP_E_ARG, "NULL device pointer");
  if (n_paths < 1) return fail(MCP_E_ARG, "n_paths must be >= 1");
  if (stride < n_paths) return fail(MCP_E_ARG, "terminal_stride %llu < n_paths %llu",
                                    (unsigned long long)stride, (unsigned long long)n_paths);
  const int grid = paths_grid(prm, n_paths);
  if ((uint64_t)prm->n_steps * (uint64_t)((prm->n_assets + 3) / 4) > 0xFFFFFFFFull)
    return fail(MCP_E_UNSUPPORTED, "n_steps * ceil(N/4) exceeds the 32-bit Philox block counter");
  const int nb = (prm->n_assets + 3) / 4;
  const int K = prm->n_portfolios;
  int variant = 0;
  if (K > 1) variant |= mcp::VAR_KT8;
  if (prm->flags & MCP_FLAG_NATIVE_MATH) variant |= mcp::VAR_NATIVE;
  if (paths_per_thread(prm) == 2) variant |= mcp::VAR_PPT2;
  const int kt = (variant & mcp::VAR_KT8) ? KT_WIDE : 1;
  mcp::PathArgs a;
  const float2* tables = nullptr;
  if (int rc = device_tables((hipStream_t)stream, &tables)) return rc;
  a.tables = tables;
  a.packed = d_packed;
  a.terminal = d_terminal;
  a.seed = seed;
  a.path_begin = path_begin;
  a.n_paths = n_paths;
  a.stride = stride;
  a.n_steps = prm->n_steps;
  a.n_portfolios = K;
  a.compounding = prm->compounding;
  a.v0 = (float)prm->v0;
  static const int env_mt = [] { const char* e = getenv("MCP_SWEEP_MT"); return e ? atoi(e) : 0; }();   // 0: auto, -1: off
  if (nb <= 4 && K >= SWEEP_MIN_K && env_mt >= 0 && paths_per_thread(prm) == 1) {
    const int mt = env_mt ? env_mt : (K > 64 ? 4 : (K > 32 ? 2 : 1));
    const bool native = (prm->flags & MCP_FLAG_NATIVE_MATH) != 0;
    static const int env_shared = [] { const char* e = getenv("MCP_SWEEP_SHARED"); return e ? atoi(e) : 1; }();
    a.k_begin = 0;
    // >= 384 portfolios: four waves share one draw per 64 paths (512-portfolio workgroups)
    hipError_t e = (K >= 384 && env_shared && !env_mt) ? mcp::launch_sweep_shared(nb, native, a, (hipStream_t)stream)
                                                      : mcp::launch_sweep_paths(nb, mt, native, a, (hipStream_t)stream);
    if (e != hipSuccess) return fail(MCP_E_NODEVICE, "mc_sweep_kernel launch: %s", hipGetErrorString(e));
    return MCP_OK;
  }
  for (int kb = 0; kb < K; kb += kt) {
    a.k_begin = kb;
    hipError_t e = k_launch[nb - 1](variant, a, grid, (hipStream_t)stream);
    if (e != hipSuccess) return fail(MCP_E_NODEVICE, "mc_paths_kernel launch: %s", hipGetErrorString(e));
  }
  return MCP_OK;
}

int mcp_launch_moments(const mcp_params* prm, const float* d_terminal, uint64_t stride, uint64_t n,
                       void* d_partials, void* d_moments, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_terminal || !d_partials || !d_moments || stride < n) return fail(MCP_E_ARG, "bad argument");
  HIP_TRY(mcp::launch_moments(*prm, prm->n_portfolios, d_terminal, stride, n, (mcp_moments*)d_partials,
                              (mcp_moments*)d_moments, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_percentile_rank(uint64_t n, double alpha, uint64_t* rank_lo, uint64_t* rank_hi, double* gamma) {
  if (n < 1 || !rank_lo || !rank_hi || !gamma) return fail(MCP_E_ARG, "bad argument");
  // app.py:259  np.percentile(returns, (1-alpha)*100); numpy divides by 100 again, then method
  // 'linear' takes virtual_index = (n - 1) * q  (numpy 2.2 _QuantileMethods['linear']).
  const double pct = (1.0 - alpha) * 100.0;
  const double q = pct / 100.0;
  const double vi = (double)(n - 1) * q;
  if (vi >= (double)(n - 1)) { *rank_lo = *rank_hi = n - 1; *gamma = 0.0; return MCP_OK; }
  if (vi < 0.0) { *rank_lo = *rank_hi = 0; *gamma = 0.0; return MCP_OK; }
  const double fl = std::floor(vi);
  *rank_lo = (uint64_t)fl;
  *rank_hi = *rank_lo + 1;
  *gamma = vi - fl;
  return MCP_OK;
}

int mcp_launch_select_init(int K, uint64_t rank_lo, uint64_t rank_hi, void* d_state, void* stream) {
  if (K < 1 || !d_state) return fail(MCP_E_ARG, "bad argument");
  HIP_TRY(mcp::launch_select_init(K, rank_lo, rank_hi, (mcp::SelectState*)d_state, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_select_hist(int K, const float* d_terminal, uint64_t stride, uint64_t n, int pass,
                           const void* d_state, void* d_hist, void* stream) {
  if (K < 1 || !d_terminal || !d_state || !d_hist || pass < 0 || pass > 2 || stride < n)
    return fail(MCP_E_ARG, "bad argument");
  HIP_TRY(mcp::launch_select_hist(K, d_terminal, stride, n, pass, (const mcp::SelectState*)d_state,
                                  (unsigned long long*)d_hist, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_select_scan(int K, int pass, const void* d_hist, void* d_state, void* stream) {
  if (K < 1 || !d_hist || !d_state || pass < 0 || pass > 2) return fail(MCP_E_ARG, "bad argument");
  HIP_TRY(mcp::launch_select_scan(K, pass, (const unsigned long long*)d_hist, (mcp::SelectState*)d_state,
                                  (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_quantile(const mcp_params* prm, double gamma, const void* d_state, void* d_quant, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_state || !d_quant) return fail(MCP_E_ARG, "NULL device pointer");
  HIP_TRY(mcp::launch_quantile(*prm, prm->n_portfolios, gamma, (const mcp::SelectState*)d_state,
                               (mcp::Quantile*)d_quant, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_tail(const mcp_params* prm, const float* d_terminal, uint64_t stride, uint64_t n,
                    const void* d_quant, void* d_tail_partial, void* d_tail, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_terminal || !d_quant || !d_tail_partial || !d_tail || stride < n) return fail(MCP_E_ARG, "bad argument");
  HIP_TRY(mcp::launch_tail(*prm, prm->n_portfolios, d_terminal, stride, n, (const mcp::Quantile*)d_quant,
                           (double*)d_tail_partial, (double*)d_tail, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_stats(const mcp_params* prm, const void* d_moments, const void* d_quant, const void* d_tail,
                     void* d_stats, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_moments || !d_quant || !d_tail || !d_stats) return fail(MCP_E_ARG, "NULL device pointer");
  HIP_TRY(mcp::launch_stats(*prm, prm->n_portfolios, (const mcp_moments*)d_moments, (const mcp::Quantile*)d_quant,
                            (const double*)d_tail, (mcp_stats*)d_stats, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_box_muller(const uint32_t* d_xa, const uint32_t* d_xb, uint64_t n, float* d_zs, float* d_zc, int flags,
                          void* stream) {
  if (!d_xa || !d_xb || !d_zs || !d_zc) return fail(MCP_E_ARG, "NULL device pointer");
  const float2* tables = nullptr;
  if (int rc = device_tables((hipStream_t)stream, &tables)) return rc;
  HIP_TRY(mcp::launch_box_muller(d_xa, d_xb, n, tables, d_zs, d_zc, (flags & MCP_FLAG_NATIVE_MATH) != 0, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_sqrt(const float* d_in, float* d_out, uint64_t n, void* stream) {
  if (!d_in || !d_out) return fail(MCP_E_ARG, "NULL device pointer");
  HIP_TRY(mcp::launch_sqrt(d_in, d_out, n, (hipStream_t)stream));
  return MCP_OK;
}

uint32_t mcp_float_to_key(float v) { return mcp::float_to_key(v); }
float mcp_key_to_float(uint32_t key) { return mcp::key_to_float(key); }

double mcp_terminal_to_x(const mcp_params* prm, float terminal) {
  if (prm->compounding == MCP_COMPOUND_LOG) return std::expm1((double)terminal);
  return (double)terminal / (double)(float)prm->v0 - 1.0;
}

// ---- host-level context -----------------------------------------------------------------------------

int mcp_ctx_create(int device, mcp_ctx** out) {
  if (!out) return fail(MCP_E_ARG, "out is NULL");
  *out = nullptr;
  const int n = mcp_device_count();
  if (n <= 0) return fail(MCP_E_NODEVICE, "no HIP device visible (the product path has no CPU fallback)");
  if (device < 0 || device >= n) return fail(MCP_E_ARG, "device %d outside [0,%d)", device, n);
  mcp_ctx* c = new (std::nothrow) mcp_ctx;
  if (!c) return fail(MCP_E_NOMEM, "out of host memory");
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return fail(MCP_E_NODEVICE, "cannot create a stream on device %d", device);
  }
  *out = c;
  return MCP_OK;
}

void mcp_ctx_destroy(mcp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  for (int i = 0; i < 8; i++)
    if (c->d[i]) (void)hipFree(c->d[i]);
  if (c->d_packed) (void)hipFree(c->d_packed);
  if (c->d_terminal) (void)hipFree(c->d_terminal);
  if (c->h_packed) (void)hipHostFree(c->h_packed);
  if (c->h_stats) (void)hipHostFree(c->h_stats);
  if (c->d_sweep) (void)hipFree(c->d_sweep);
  delete c;
}

static int grow_dev(void** p, size_t* cap, size_t need) {
  if (need <= *cap) return MCP_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  if (hipMalloc(p, need) != hipSuccess) return fail(MCP_E_NOMEM, "hipMalloc(%zu) failed", need);
  *cap = need;
  return MCP_OK;
}

static int grow_host(void** p, size_t* cap, size_t need) {
  if (need <= *cap) return MCP_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  if (hipHostMalloc(p, need, hipHostMallocDefault) != hipSuccess) return fail(MCP_E_NOMEM, "hipHostMalloc(%zu) failed", need);
  *cap = need;
  return MCP_OK;
}

int mcp_simulate(mcp_ctx* c, const mcp_params* prm, const float* mu, const float* chol, const float* W,
                 uint64_t seed, uint64_t path_begin, uint64_t n_paths, float* terminal_out, mcp_stats* stats_out) {
  if (!c) return fail(MCP_E_ARG, "ctx is NULL");
  if (int rc = check_params(prm)) return rc;
  if (!mu || !chol || !W || !stats_out) return fail(MCP_E_ARG, "NULL pointer");
  if (n_paths < 1) return fail(MCP_E_ARG, "n_paths must be >= 1");
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  const int K = prm->n_portfolios;
  const size_t plen = mcp_packed_len(prm->n_assets, K);

  int rc;
  for (int w = 0; w < 8; w++)
    if ((rc = grow_dev(&c->d[w], &c->cap[w], mcp_ws_bytes(w, K)))) return rc;
  if ((rc = grow_dev((void**)&c->d_packed, &c->packed_cap, plen * sizeof(float)))) return rc;
  if ((rc = grow_dev((void**)&c->d_terminal, &c->terminal_cap, (size_t)K * n_paths * sizeof(float)))) return rc;
  if ((rc = grow_host((void**)&c->h_packed, &c->h_packed_cap, plen * sizeof(float)))) return rc;
  if ((rc = grow_host((void**)&c->h_stats, &c->h_stats_cap, (size_t)K * sizeof(mcp_stats)))) return rc;

  if ((rc = mcp_pack_params(prm->n_assets, K, mu, chol, W, c->h_packed, plen))) return rc;
  hipStream_t s = c->stream;
  HIP_TRY(hipMemcpyAsync(c->d_packed, c->h_packed, plen * sizeof(float), hipMemcpyHostToDevice, s));

  uint64_t lo, hi;
  double gamma;
  if ((rc = mcp_percentile_rank(n_paths, prm->alpha, &lo, &hi, &gamma))) return rc;

  if ((rc = mcp_launch_paths(prm, c->d_packed, seed, path_begin, n_paths, c->d_terminal, n_paths, s))) return rc;
  if ((rc = mcp_launch_moments(prm, c->d_terminal, n_paths, n_paths, c->d[MCP_WS_PARTIALS], c->d[MCP_WS_MOMENTS], s))) return rc;
  if ((rc = mcp_launch_select_init(K, lo, hi, c->d[MCP_WS_STATE], s))) return rc;
  for (int pass = 0; pass < 3; pass++) {
    if ((rc = mcp_launch_select_hist(K, c->d_terminal, n_paths, n_paths, pass, c->d[MCP_WS_STATE], c->d[MCP_WS_HIST], s))) return rc;
    if ((rc = mcp_launch_select_scan(K, pass, c->d[MCP_WS_HIST], c->d[MCP_WS_STATE], s))) return rc;
  }
  if ((rc = mcp_launch_quantile(prm, gamma, c->d[MCP_WS_STATE], c->d[MCP_WS_QUANT], s))) return rc;
  if ((rc = mcp_launch_tail(prm, c->d_terminal, n_paths, n_paths, c->d[MCP_WS_QUANT], c->d[MCP_WS_TAIL_PARTIAL],
                            c->d[MCP_WS_TAIL], s))) return rc;
  if ((rc = mcp_launch_stats(prm, c->d[MCP_WS_MOMENTS], c->d[MCP_WS_QUANT], c->d[MCP_WS_TAIL], c->d[MCP_WS_STATS], s))) return rc;
  HIP_TRY(hipMemcpyAsync(c->h_stats, c->d[MCP_WS_STATS], (size_t)K * sizeof(mcp_stats), hipMemcpyDeviceToHost, s));
  if (terminal_out)
    HIP_TRY(hipMemcpyAsync(terminal_out, c->d_terminal, (size_t)K * n_paths * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(stats_out, c->h_stats, (size_t)K * sizeof(mcp_stats));
  return MCP_OK;
}

int mcp_sweep_historical(mcp_ctx* c, int N, int R, int P, const double* returns, const double* mean, const double* cov,
                         const double* W, double rf, double alpha, double* o_ret, double* o_std, double* o_sharpe,
                         double* o_var, double* o_cvar) {
  if (!c) return fail(MCP_E_ARG, "ctx is NULL");
  if (N < 1 || N > MCP_MAX_ASSETS) return fail(MCP_E_ARG, "n_assets=%d outside [1,%d]", N, MCP_MAX_ASSETS);
  if (R < 1 || R > MCP_SWEEP_MAX_ROWS) return fail(MCP_E_ARG, "n_rows=%d outside [1,%d]", R, MCP_SWEEP_MAX_ROWS);
  if (P < 1) return fail(MCP_E_ARG, "n_portfolios=%d < 1", P);
  if (!(alpha > 0.0 && alpha < 1.0)) return fail(MCP_E_ARG, "alpha=%g outside (0,1)", alpha);
  if (!returns || !mean || !cov || !W || !o_ret || !o_std || !o_sharpe || !o_var || !o_cvar)
    return fail(MCP_E_ARG, "NULL pointer");
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  uint64_t lo, hi;
  double gamma;
  if (int rc = mcp_percentile_rank((uint64_t)R, alpha, &lo, &hi, &gamma)) return rc;
  const size_t n_ret = (size_t)R * N, n_cov = (size_t)N * N, n_w = (size_t)P * N, n_out = 5 * (size_t)P;
  const size_t total = n_ret + (size_t)N + n_cov + n_w + n_out;
  if (int rc = grow_dev((void**)&c->d_sweep, &c->sweep_cap, total * sizeof(double))) return rc;
  double* d_ret = c->d_sweep;
  double* d_mean = d_ret + n_ret;
  double* d_cov = d_mean + N;
  double* d_w = d_cov + n_cov;
  double* d_out = d_w + n_w;
  hipStream_t s = c->stream;
  HIP_TRY(hipMemcpyAsync(d_ret, returns, n_ret * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_mean, mean, (size_t)N * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_cov, cov, n_cov * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_w, W, n_w * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_TRY(mcp::launch_sweep_hist(N, R, P, d_ret, d_mean, d_cov, d_w, rf, lo, hi, gamma, d_out, s));
  double* outs[5] = {o_ret, o_std, o_sharpe, o_var, o_cvar};
  for (int i = 0; i < 5; i++)
    HIP_TRY(hipMemcpyAsync(outs[i], d_out + (size_t)i * P, (size_t)P * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return MCP_OK;
}

}  // extern "C"
