// mcp_api.cpp -- the C ABI of libmcport.so (include/mcport.h): argument checking, parameter packing,
// enqueue-only launch entry points, and the host-level mcp_simulate() that strings them together on
// one device.  No torch types, no exceptions across the boundary.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

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
    if (e_ != hipSuccess) return fail(MCP_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

constexpr int KT_WIDE = 8;           // portfolios per pass of the KT=8 kernel
constexpr int K_PAD = 512;           // W rows are zero-padded to a multiple of this (MFMA sweep tile: 32*MT)
using mcp::SWEEP_MIN_K;              // from this many portfolios on the MFMA sweep kernels run

// MCP_SWEEP_MT: 0 (default) = automatic tile count, 1/2/4 = force the per-wave kernel with that many 32-portfolio tiles,
// -1 = no MFMA kernels at all (K > 16 then runs as passes of the 8-portfolio kernel)
int env_sweep_mt() {
  static const int v = [] { const char* e = getenv("MCP_SWEEP_MT"); return e ? atoi(e) : 0; }();
  return v;
}
// does a launch for K portfolios go to the MFMA sweep kernels?  (decides the MomentPartial slot count)
bool uses_sweep(int K) { return K >= SWEEP_MIN_K && env_sweep_mt() >= 0; }

// How a sweep of K portfolios is cut into launches.  Two kernel families, both bit-identical to the oracle:
//   shared-draw (mc_sweep_shared_kernel): the four waves of a workgroup share one draw and own 128 MT portfolios;
//                MT = 4 | 2 for N <= 16 (512 | 256 portfolios), MT = 2 | 1 for 16 < N <= 64 (256 | 128)
//   per-wave    (mc_sweep_kernel, N <= 16): every wave draws for itself and owns 32 MT portfolios, MT = 1 | 2 | 4
// Whole big shared workgroups first; the remainder r goes to the smallest tiling that covers it:
//   N <= 16:  r <= 128 per-wave (32 / 64 / 128)   r <= 256 shared MT 2   r <= 384 shared MT 2 + per-wave   else shared MT 4
//   N  > 16:  r <= 128 shared MT 1                 else shared MT 2
// (K = 1,250 per GPU when configs[4] is sharded over 8: 2 x 512 + 226 -> one 256-portfolio workgroup row instead of a third 512.)
struct SweepSeg { bool shared; int mt, k_begin, k_count; };
int sweep_plan(int K, int nb, SweepSeg* out) {
  int n = 0;
  const int env_mt = env_sweep_mt();
  if (nb <= 4 && env_mt > 0) {                       // forced: per-wave kernel with that many tiles for everything
    out[n++] = {false, env_mt, 0, K};
    return n;
  }
  const int big = nb <= 4 ? 4 : 2, W = 128 * big;
  const int full = (K / W) * W;
  if (full) out[n++] = {true, big, 0, full};
  const int r = K - full;
  if (r == 0) return n;
  if (nb <= 4) {
    const auto per_wave = [](int k) { return k > 64 ? 4 : (k > 32 ? 2 : 1); };
    if (r <= 128) out[n++] = {false, per_wave(r), full, r};
    else if (r <= 256) out[n++] = {true, 2, full, r};
    else if (r <= 384) { out[n++] = {true, 2, full, 256}; out[n++] = {false, per_wave(r - 256), full + 256, r - 256}; }
    else out[n++] = {true, 4, full, r};
  } else {
    out[n++] = {true, r <= 128 ? 1 : 2, full, r};
  }
  return n;
}

inline int n4_of(int n) { return 4 * ((n + 3) / 4); }
// W rows are zero-padded to whole MFMA workgroups for a sweep, to whole KT_WIDE passes otherwise
inline int kpad_of(int k) { return k >= SWEEP_MIN_K ? K_PAD * ((k + K_PAD - 1) / K_PAD) : KT_WIDE * ((k + KT_WIDE - 1) / KT_WIDE); }

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

const mcp::launch_paths_fn k_launch[16] = {
    mcp::launch_paths_nb1,  mcp::launch_paths_nb2,  mcp::launch_paths_nb3,  mcp::launch_paths_nb4,
    mcp::launch_paths_nb5,  mcp::launch_paths_nb6,  mcp::launch_paths_nb7,  mcp::launch_paths_nb8,
    mcp::launch_paths_nb9,  mcp::launch_paths_nb10, mcp::launch_paths_nb11, mcp::launch_paths_nb12,
    mcp::launch_paths_nb13, mcp::launch_paths_nb14, mcp::launch_paths_nb15, mcp::launch_paths_nb16};

// Inverse-CDF coefficient table (SPEC.md section 3; DATA of the spec, generated by tools/fit_icdf_table.py): one
// device-resident copy per device, uploaded on first use on the caller's stream.  The first call on a device
// allocates and synchronises: do it once before capturing launches into a hipGraph.
const float k_icdf_table[mcp::ICDF_ENTRIES][4] = {
#include "mcp_icdf_table.inc"
};
constexpr int MAX_DEVICES = 64;
std::mutex g_tab_mu;
float4* g_tables[MAX_DEVICES] = {nullptr};

// Device that owns `stream` (the current device for the NULL stream).
int stream_device(hipStream_t stream, int* dev) {
  if (stream) {
    hipDevice_t d = 0;
    if (hipStreamGetDevice(stream, &d) == hipSuccess) { *dev = (int)d; return MCP_OK; }
  }
  HIP_TRY(hipGetDevice(dev));
  return MCP_OK;
}

// Makes `dev` current for the lifetime of the object (launches go to the current device; a caller may hand over a
// stream of another device than the thread's current one).
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); changed = err == hipSuccess; }
  }
  ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};

int device_tables(int dev, hipStream_t stream, const float4** out) {
  if (dev < 0 || dev >= MAX_DEVICES) return fail(MCP_E_UNSUPPORTED, "device index %d", dev);
  std::lock_guard<std::mutex> lock(g_tab_mu);
  if (!g_tables[dev]) {
    float4* t = nullptr;
    if (hipMalloc((void**)&t, sizeof k_icdf_table) != hipSuccess) return fail(MCP_E_NOMEM, "hipMalloc of the inverse-CDF table failed");
    hipError_t e = hipMemcpyAsync(t, k_icdf_table, sizeof k_icdf_table, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);     // other streams may use the table next
    if (e != hipSuccess) { (void)hipFree(t); return fail(MCP_E_HIP, "inverse-CDF table upload: %s", hipGetErrorString(e)); }
    g_tables[dev] = t;
  }
  *out = g_tables[dev];
  return MCP_OK;
}

// One shard of a context: a device, its stream and every buffer a pass needs there.
struct Shard {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;                 // same-device exchange: "my step is enqueued"
  void* ws[MCP_WS_COUNT] = {nullptr};
  size_t ws_cap[MCP_WS_COUNT] = {0};
  float* d_packed = nullptr;
  size_t packed_cap = 0;
  float* d_terminal = nullptr;
  size_t terminal_cap = 0;
  mcp_record* d_gather = nullptr;          // [shards][K tile] records of all shards (shard 0 finishes the statistics)
  size_t gather_cap = 0;
  float* h_packed = nullptr;               // pinned staging
  size_t h_packed_cap = 0;
  double* h_pivot = nullptr;               // pinned staging of the [K tile] pivots
  size_t h_pivot_cap = 0;
  mcp_stats* h_stats = nullptr;            // pinned AND mapped: ws[MCP_WS_STATS] is its device address, so the last kernel
  size_t h_stats_cap = 0;                  // of a pass writes the [K] records straight into host memory (no copy-back)
};

// RCCL entry points, resolved at run time (no link-time dependency: a single-device user never loads librccl).
// Prototypes as in /opt/rocm/include/rccl/rccl.h:236 (ncclCommInitAll), :260 (ncclCommDestroy), :339
// (ncclGetErrorString), :611 (ncclAllReduce), :678 (ncclAllGather), group calls; enum values :448 (ncclSum = 0),
// :459-467 (ncclUint64 = 5, ncclFloat64 = 8).
struct Rccl {
  void* handle = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t s) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
};
constexpr int NCCL_SUM = 0, NCCL_UINT64 = 5, NCCL_FLOAT64 = 8;
std::mutex g_rccl_mu;
Rccl g_rccl;

int load_rccl(const Rccl** out) {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (!g_rccl.handle) {
    const char* cands[3] = {getenv("MCP_RCCL_LIB"), "librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* c : cands)
      if (c && *c && (h = dlopen(c, RTLD_NOW | RTLD_LOCAL))) break;   // LOCAL: every entry point is taken with dlsym; a GLOBAL librccl
                                                                      // interposes symbols of a torch imported later (abort at exit)
    if (!h) return fail(MCP_E_COMM, "cannot load librccl (set MCP_RCCL_LIB): %s", dlerror());
    Rccl r;
    r.handle = h;
#define MCP_SYM(field, name) \
    *(void**)(&r.field) = dlsym(h, name); \
    if (!r.field) { dlclose(h); return fail(MCP_E_COMM, "librccl lacks %s", name); }
    MCP_SYM(CommInitAll, "ncclCommInitAll") MCP_SYM(CommDestroy, "ncclCommDestroy") MCP_SYM(GetErrorString, "ncclGetErrorString")
    MCP_SYM(AllReduce, "ncclAllReduce") MCP_SYM(AllGather, "ncclAllGather") MCP_SYM(GroupStart, "ncclGroupStart")
    MCP_SYM(GroupEnd, "ncclGroupEnd")
#undef MCP_SYM
    g_rccl = r;
  }
  *out = &g_rccl;
  return MCP_OK;
}

}  // namespace

struct mcp_ctx {
  std::vector<Shard> sh;
  std::mutex mu;
  bool all_same = false;             // every shard on ONE device (logical shards)
  bool same_device = false;          // exchange through a kernel of shard 0 that reads / writes every shard's buffer: shards
                                     // of ONE device, or distinct devices with peer access (the fallback when RCCL is unavailable)
  bool exchange_always = false;      // MCP_FORCE_RCCL=1: run the collectives with one device too (a 1-rank communicator)
  int exchange_mode = MCP_EXCHANGE_UNSET;
  std::string exchange_note;         // why RCCL was not used, when the peer-access kernel stands in for it
  const Rccl* rccl = nullptr;
  std::vector<void*> comms;          // one ncclComm_t per shard (distinct devices only)
  size_t terminal_budget = (size_t)8 << 30;
  double* d_sweep = nullptr;         // inputs then outputs of mcp_sweep_historical (shard 0)
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
  return n4 + n4 * (n4 / 2 + 1) + (size_t)kpad_of(n_portfolios) * n4 + 4 + n4;   // ... + fold block [c, v] (+3 pad)
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
  // fold block of portfolio 0 (SPEC.md section 4.1): c = w.mu, v = L^T w, accumulated in binary64 (i ascending)
  // from the binary32 inputs, rounded once to binary32
  float* F = Wp + (size_t)kpad_of(n_portfolios) * n4;
  double c = 0.0;
  for (int i = 0; i < N; i++) c += (double)W[i] * (double)out[i];
  F[0] = (float)c;
  for (int j = 0; j < N; j++) {
    double v = 0.0;
    for (int i = j; i < N; i++) v += (double)W[i] * (double)chol[(size_t)i * N + j];
    F[1 + j] = (float)v;
  }
  return MCP_OK;
}

uint64_t mcp_moment_slots(int K, uint64_t n_paths) { return K < 1 ? 0 : mcp::moment_slots(uses_sweep(K), n_paths); }

size_t mcp_ws_bytes(int which, int K, uint64_t n_paths) {
  if (K < 1) return 0;
  switch (which) {
    case MCP_WS_PARTIALS: return (size_t)K * (size_t)mcp_moment_slots(K, n_paths) * sizeof(mcp::MomentPartial);
    case MCP_WS_RECORD: return (size_t)K * sizeof(mcp_record);
    case MCP_WS_STATE: return (size_t)K * 2 * sizeof(mcp::SelectState);
    case MCP_WS_HIST: return (size_t)K * 2 * MCP_SELECT_BINS * sizeof(unsigned long long);
    case MCP_WS_QUANT: return (size_t)K * sizeof(mcp::Quantile);
    case MCP_WS_STATS: return (size_t)K * sizeof(mcp_stats);
    case MCP_WS_BELOW: return (size_t)K * mcp::stream_slots(K) * sizeof(double);
    case MCP_WS_PIVOT: return (size_t)K * sizeof(double);
    default: return 0;
  }
}

int mcp_pivots(const mcp_params* prm, const float* mu, const float* chol, const float* W, double* out) {
  if (int rc = check_params(prm)) return rc;
  if (!mu || !chol || !W || !out) return fail(MCP_E_ARG, "NULL pointer");
  const int N = prm->n_assets, K = prm->n_portfolios;
  const double T = (double)prm->n_steps;
  for (int k = 0; k < K; k++) {
    const float* w = W + (size_t)k * N;
    double m = 0.0;
    for (int i = 0; i < N; i++) m += (double)w[i] * (double)(mu[i] + 0.0f);
    double c;
    if (prm->compounding == MCP_COMPOUND_LOG) {
      double s2 = 0.0;                                   // |L^T w|^2 = w' Sigma w
      for (int j = 0; j < N; j++) {
        double v = 0.0;
        for (int i = j; i < N; i++) v += (double)w[i] * (double)chol[(size_t)i * N + j];
        s2 += v * v;
      }
      c = std::expm1(T * (m + 0.5 * s2));
    } else {
      c = m > -1.0 ? std::expm1(T * std::log1p(m)) : 0.0;
    }
    out[k] = std::isfinite(c) ? c : 0.0;
  }
  return MCP_OK;
}

int mcp_launch_paths(const mcp_params* prm, const float* d_packed, const double* d_pivot, uint64_t seed, uint64_t path_begin,
                     uint64_t n_paths, float* d_terminal, uint64_t stride, void* d_partials, void* d_hist, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_packed || !d_terminal) return fail(MCP_E_ARG, "NULL device pointer");
  if ((d_partials == nullptr) != (d_hist == nullptr)) return fail(MCP_E_ARG, "d_partials and d_hist go together (both or neither)");
  if (n_paths < 1) return fail(MCP_E_ARG, "n_paths must be >= 1");
  if (stride < n_paths) return fail(MCP_E_ARG, "terminal_stride %llu < n_paths %llu",
                                    (unsigned long long)stride, (unsigned long long)n_paths);
  const int grid = mcp::path_grid(n_paths);
  if ((uint64_t)prm->n_steps * (uint64_t)((prm->n_assets + 3) / 4) > 0xFFFFFFFFull)
    return fail(MCP_E_UNSUPPORTED, "n_steps * ceil(N/4) exceeds the 32-bit Philox block counter");
  const int nb = (prm->n_assets + 3) / 4;
  const int K = prm->n_portfolios;
  int variant = 0;
  if (prm->flags & MCP_FLAG_FOLD) {
    if (K != 1 || (prm->flags & MCP_FLAG_NATIVE_MATH)) return fail(MCP_E_UNSUPPORTED, "MCP_FLAG_FOLD needs one portfolio and the spec's normals");
    variant |= mcp::VAR_FOLD;
  }
  if (K > 1) variant |= mcp::VAR_KT8;
  if (prm->flags & MCP_FLAG_NATIVE_MATH) variant |= mcp::VAR_NATIVE;
  const int kt = (variant & mcp::VAR_KT8) ? KT_WIDE : 1;
  mcp::PathArgs a;
  const float4* tables = nullptr;
  int dev = 0;
  if (int rc = stream_device((hipStream_t)stream, &dev)) return rc;
  DeviceGuard guard(dev);                 // the stream may belong to another device than the thread's current one
  if (guard.err != hipSuccess) return fail(MCP_E_HIP, "hipSetDevice(%d): %s", dev, hipGetErrorString(guard.err));
  if (int rc = device_tables(dev, (hipStream_t)stream, &tables)) return rc;
  const bool sweep = uses_sweep(K);
  a.tables = tables;
  a.packed = d_packed;
  a.terminal = d_terminal;
  a.pivot = d_pivot;
  a.partials = (mcp::MomentPartial*)d_partials;
  a.hist = sweep ? nullptr : (unsigned long long*)d_hist;     // the sweep kernels leave digit 0 to hist(0) below
  a.slots = mcp::moment_slots(sweep, n_paths);
  a.v0d = (double)(float)prm->v0;
  a.inv_v0d = 1.0 / a.v0d;
  { int e = 0; a.v0_pow2 = std::frexp(a.v0d, &e) == 0.5; }
  a.seed = seed;
  a.path_begin = path_begin;
  a.n_paths = n_paths;
  a.stride = stride;
  a.n_steps = prm->n_steps;
  a.n_portfolios = K;
  a.compounding = prm->compounding;
  a.v0 = (float)prm->v0;
  a.k_count = K;
  a.fold_offset = (uint32_t)(n4_of(prm->n_assets) + n4_of(prm->n_assets) * (n4_of(prm->n_assets) / 2 + 1) + kpad_of(K) * n4_of(prm->n_assets));
  if (sweep) {
    const bool native = (prm->flags & MCP_FLAG_NATIVE_MATH) != 0;
    SweepSeg segs[4];
    const int n_seg = sweep_plan(K, nb, segs);
    hipError_t e = hipSuccess;
    for (int i = 0; i < n_seg && e == hipSuccess; i++) {
      a.k_begin = segs[i].k_begin;
      a.k_count = segs[i].k_count;
      e = segs[i].shared ? mcp::launch_sweep_shared(nb, segs[i].mt, native, a, (hipStream_t)stream)
                         : mcp::launch_sweep_paths(nb, segs[i].mt, native, a, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail(MCP_E_HIP, "mc_sweep_kernel launch: %s", hipGetErrorString(e));
    if (d_hist) {                              // digit 0 of the select: one lean read of the terminal values just written
      e = mcp::launch_hist(*prm, K, 0, d_terminal, stride, n_paths, nullptr, d_pivot, nullptr, (unsigned long long*)d_hist, (hipStream_t)stream);
      if (e != hipSuccess) return fail(MCP_E_HIP, "hist_kernel<0> launch: %s", hipGetErrorString(e));
    }
    return MCP_OK;
  }
  for (int kb = 0; kb < K; kb += kt) {
    a.k_begin = kb;
    hipError_t e = k_launch[nb - 1](variant, a, grid, (hipStream_t)stream);
    if (e != hipSuccess) return fail(MCP_E_HIP, "mc_paths_kernel launch: %s", hipGetErrorString(e));
  }
  return MCP_OK;
}

// Launches below go to the device that owns the stream.
#define MCP_ON_STREAM_DEVICE(stream)                                                                     \
  int dev_ = 0;                                                                                          \
  if (int rc_ = stream_device((hipStream_t)(stream), &dev_)) return rc_;                                 \
  DeviceGuard guard_(dev_);                                                                              \
  if (guard_.err != hipSuccess) return fail(MCP_E_HIP, "hipSetDevice(%d): %s", dev_, hipGetErrorString(guard_.err))

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

int mcp_launch_pass0(const mcp_params* prm, const float* d_terminal, uint64_t stride, uint64_t n, const double* d_pivot,
                     void* d_partials, void* d_hist, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_terminal || !d_partials || !d_hist || stride < n) return fail(MCP_E_ARG, "bad argument");
  MCP_ON_STREAM_DEVICE(stream);
  HIP_TRY(mcp::launch_pass0(*prm, prm->n_portfolios, d_terminal, stride, n, d_pivot, mcp_moment_slots(prm->n_portfolios, n),
                            (mcp::MomentPartial*)d_partials, (unsigned long long*)d_hist, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_scan(const mcp_params* prm, int pass, uint64_t n, uint64_t rank_lo, uint64_t rank_hi, const void* d_partials,
                    const void* d_below, const double* d_pivot, void* d_hist, void* d_state, void* d_record, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_partials || !d_below || !d_hist || !d_state || !d_record || pass < 0 || pass > 1) return fail(MCP_E_ARG, "bad argument");
  MCP_ON_STREAM_DEVICE(stream);
  HIP_TRY(mcp::launch_scan(*prm, prm->n_portfolios, pass, n, rank_lo, rank_hi, mcp_moment_slots(prm->n_portfolios, n),
                           (const mcp::MomentPartial*)d_partials, (const double*)d_below, d_pivot, (unsigned long long*)d_hist,
                           (mcp::SelectState*)d_state, (mcp_record*)d_record, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_hist(const mcp_params* prm, int pass, const float* d_terminal, uint64_t stride, uint64_t n,
                    const void* d_state, const double* d_pivot, void* d_below, void* d_hist, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_terminal || !d_hist || pass < 0 || pass > 2 || stride < n || (pass > 0 && (!d_state || !d_below)))
    return fail(MCP_E_ARG, "bad argument");
  MCP_ON_STREAM_DEVICE(stream);
  HIP_TRY(mcp::launch_hist(*prm, prm->n_portfolios, pass, d_terminal, stride, n, (const mcp::SelectState*)d_state, d_pivot,
                           (double*)d_below, (unsigned long long*)d_hist, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_final(const mcp_params* prm, uint64_t n, double gamma, uint64_t rank_lo, uint64_t rank_hi,
                     const void* d_below, void* d_hist, const void* d_state, void* d_record, void* d_quant,
                     void* d_stats, void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (!d_below || !d_hist || !d_state || !d_record || !d_quant) return fail(MCP_E_ARG, "NULL device pointer");
  MCP_ON_STREAM_DEVICE(stream);
  HIP_TRY(mcp::launch_final(*prm, prm->n_portfolios, n, gamma, rank_lo, rank_hi, (const double*)d_below,
                            (unsigned long long*)d_hist, (const mcp::SelectState*)d_state, (mcp_record*)d_record,
                            (mcp::Quantile*)d_quant, (mcp_stats*)d_stats, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_stats(const mcp_params* prm, int world, const void* d_gathered, const void* d_quant, void* d_stats,
                     void* stream) {
  if (int rc = check_params(prm)) return rc;
  if (world < 1 || !d_gathered || !d_quant || !d_stats) return fail(MCP_E_ARG, "bad argument");
  MCP_ON_STREAM_DEVICE(stream);
  HIP_TRY(mcp::launch_stats(*prm, prm->n_portfolios, world, (const mcp_record*)d_gathered, (const mcp::Quantile*)d_quant,
                            (mcp_stats*)d_stats, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_sum_u64(void* const* d_bufs, int n_bufs, size_t words, void* stream) {
  if (!d_bufs || n_bufs < 1 || n_bufs > 8) return fail(MCP_E_ARG, "n_bufs=%d outside [1,8]", n_bufs);
  for (int i = 0; i < n_bufs; i++)
    if (!d_bufs[i]) return fail(MCP_E_ARG, "NULL device pointer");
  MCP_ON_STREAM_DEVICE(stream);
  HIP_TRY(mcp::launch_sum_u64((unsigned long long* const*)d_bufs, n_bufs, words, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_stream_create(int device, int reserve_cus, void** out) {
  if (!out) return fail(MCP_E_ARG, "stream_out is NULL");
  *out = nullptr;
  const int n = mcp_device_count();
  if (n <= 0) return fail(MCP_E_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= n) return fail(MCP_E_ARG, "device %d outside [0,%d)", device, n);
  DeviceGuard guard(device);
  if (guard.err != hipSuccess) return fail(MCP_E_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(guard.err));
  hipStream_t s = nullptr;
  if (reserve_cus <= 0) {
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  } else {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount;
    if (reserve_cus >= cus) return fail(MCP_E_ARG, "reserve_cus=%d leaves none of the %d compute units", reserve_cus, cus);
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
    for (int i = 0; i < cus - reserve_cus; i++) mask[(size_t)i / 32] |= 1u << (i % 32);
    HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  }
  *out = (void*)s;
  return MCP_OK;
}

int mcp_stream_destroy(void* stream) {
  if (!stream) return fail(MCP_E_ARG, "stream is NULL");
  HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return MCP_OK;
}

int mcp_launch_normals(const uint32_t* d_x, uint64_t n, float* d_z, void* stream) {
  if (!d_x || !d_z) return fail(MCP_E_ARG, "NULL device pointer");
  MCP_ON_STREAM_DEVICE(stream);
  const float4* tables = nullptr;
  if (int rc = device_tables(dev_, (hipStream_t)stream, &tables)) return rc;
  HIP_TRY(mcp::launch_normals(d_x, n, tables, d_z, (hipStream_t)stream));
  return MCP_OK;
}

int mcp_icdf_table(float* out, size_t out_len) {
  if (!out || out_len < (size_t)mcp::ICDF_ENTRIES * 4) return fail(MCP_E_ARG, "buffer too small for %d x 4 floats", mcp::ICDF_ENTRIES);
  memcpy(out, k_icdf_table, sizeof k_icdf_table);
  return MCP_OK;
}

uint32_t mcp_float_to_key(float v) { return mcp::float_to_key(v); }
float mcp_key_to_float(uint32_t key) { return mcp::key_to_float(key); }

double mcp_terminal_to_x(const mcp_params* prm, float terminal) {
  if (prm->compounding == MCP_COMPOUND_LOG) return std::expm1((double)terminal);
  return (double)terminal / (double)(float)prm->v0 - 1.0;
}

// ---- host-level context -----------------------------------------------------------------------------

static void free_shard(Shard& sh) {
  (void)hipSetDevice(sh.device);
  if (sh.stream) { (void)hipStreamSynchronize(sh.stream); (void)hipStreamDestroy(sh.stream); }
  if (sh.ev) (void)hipEventDestroy(sh.ev);
  for (int i = 0; i < MCP_WS_COUNT; i++)
    if (sh.ws[i] && i != MCP_WS_STATS) (void)hipFree(sh.ws[i]);
  if (sh.d_packed) (void)hipFree(sh.d_packed);
  if (sh.d_terminal) (void)hipFree(sh.d_terminal);
  if (sh.d_gather) (void)hipFree(sh.d_gather);
  if (sh.h_packed) (void)hipHostFree(sh.h_packed);
  if (sh.h_pivot) (void)hipHostFree(sh.h_pivot);
  if (sh.h_stats) (void)hipHostFree(sh.h_stats);
}

int mcp_ctx_create_multi(const int* devices, int ndev, mcp_ctx** out) {
  if (!out) return fail(MCP_E_ARG, "out is NULL");
  *out = nullptr;
  if (!devices || ndev < 1 || ndev > MAX_DEVICES) return fail(MCP_E_ARG, "ndev=%d outside [1,%d]", ndev, MAX_DEVICES);
  const int n = mcp_device_count();
  if (n <= 0) return fail(MCP_E_NODEVICE, "no HIP device visible (the product path has no CPU fallback)");
  bool all_same = true, all_distinct = true;
  for (int i = 0; i < ndev; i++) {
    if (devices[i] < 0 || devices[i] >= n) return fail(MCP_E_ARG, "device %d outside [0,%d)", devices[i], n);
    if (devices[i] != devices[0]) all_same = false;
    for (int j = 0; j < i; j++)
      if (devices[j] == devices[i]) all_distinct = false;
  }
  if (ndev > 1 && !all_same && !all_distinct)
    return fail(MCP_E_UNSUPPORTED, "devices must be all distinct (RCCL) or all the same (logical shards of one GPU)");
  if (ndev > 1 && all_same && ndev > 8) return fail(MCP_E_UNSUPPORTED, "at most 8 logical shards on one device");
  mcp_ctx* c = new (std::nothrow) mcp_ctx;
  if (!c) return fail(MCP_E_NOMEM, "out of host memory");
  struct Restore { int dev = 0; Restore() { (void)hipGetDevice(&dev); } ~Restore() { (void)hipSetDevice(dev); } } restore;   // leave the caller's current device as it was
  c->all_same = ndev > 1 && all_same;
  c->sh.resize((size_t)ndev);
  for (int i = 0; i < ndev; i++) {
    Shard& sh = c->sh[(size_t)i];
    sh.device = devices[i];
    if (hipSetDevice(sh.device) != hipSuccess || hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&sh.ev, hipEventDisableTiming) != hipSuccess) {
      mcp_ctx_destroy(c);
      return fail(MCP_E_HIP, "cannot create a stream on device %d", devices[i]);
    }
  }
  const char* force = getenv("MCP_FORCE_RCCL");      // exercise the RCCL path on a one-GPU box
  c->exchange_always = ndev == 1 && force && *force == '1';
  // The communicator / peer mapping is set up by the first path-sharded mcp_simulate (ensure_exchange): a context that
  // only ever shards the PORTFOLIOS needs neither RCCL nor peer access.
  *out = c;
  return MCP_OK;
}

// First path-sharded call of a context with several shards (or MCP_FORCE_RCCL=1): decide how they exchange.
static int ensure_exchange(mcp_ctx* c) {
  if (c->exchange_mode != MCP_EXCHANGE_UNSET) return MCP_OK;
  const int ndev = (int)c->sh.size();
  if (ndev == 1 && !c->exchange_always) { c->exchange_mode = MCP_EXCHANGE_NONE; return MCP_OK; }
  if (c->all_same) { c->same_device = true; c->exchange_mode = MCP_EXCHANGE_KERNEL; return MCP_OK; }
  // RCCL (north_star: the sufficient statistics travel over xGMI through RCCL).  MCP_EXCHANGE=p2p, or a librccl that
  // cannot be loaded / initialised, falls back to the kernel exchange over peer access: device 0 reads and writes the
  // (<= 32 KiB per portfolio) buffers of its peers directly -- the choreography the one-device shards exercise.  The
  // fallback is reported: mcp_ctx_exchange_mode() / mcp_ctx_exchange_note().
  std::vector<int> devices((size_t)ndev);
  for (int i = 0; i < ndev; i++) devices[(size_t)i] = c->sh[(size_t)i].device;
  const char* mode = getenv("MCP_EXCHANGE");
  int rc = (mode && !strcmp(mode, "p2p") && !c->exchange_always) ? fail(MCP_E_COMM, "MCP_EXCHANGE=p2p") : load_rccl(&c->rccl);
  if (rc == MCP_OK) {
    c->comms.assign((size_t)ndev, nullptr);
    const int r = c->rccl->CommInitAll(c->comms.data(), ndev, devices.data());
    if (r != 0) {
      c->comms.clear();
      rc = fail(MCP_E_COMM, "ncclCommInitAll over %d devices: %s", ndev, c->rccl->GetErrorString(r));
      c->rccl = nullptr;
    }
  }
  if (rc == MCP_OK) { c->exchange_mode = MCP_EXCHANGE_RCCL; return MCP_OK; }
  const std::string why = g_err;
  bool peers = ndev > 1 && ndev <= 8 && hipSetDevice(devices[0]) == hipSuccess;
  for (int i = 1; peers && i < ndev; i++) {
    int can = 0;
    peers = hipDeviceCanAccessPeer(&can, devices[0], devices[(size_t)i]) == hipSuccess && can;
    if (peers) {
      const hipError_t e = hipDeviceEnablePeerAccess(devices[(size_t)i], 0);
      peers = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
      (void)hipGetLastError();
    }
  }
  if (!peers) return fail(MCP_E_COMM, "%s; and no peer access from device %d to the others for the kernel exchange", why.c_str(), devices[0]);
  c->rccl = nullptr;
  c->same_device = true;                               // kernel exchange
  c->exchange_mode = MCP_EXCHANGE_P2P;
  c->exchange_note = why;
  return MCP_OK;
}

int mcp_ctx_exchange_mode(const mcp_ctx* c) { return c ? c->exchange_mode : MCP_EXCHANGE_UNSET; }
const char* mcp_ctx_exchange_note(const mcp_ctx* c) { return c ? c->exchange_note.c_str() : ""; }

int mcp_ctx_create(int device, mcp_ctx** out) { return mcp_ctx_create_multi(&device, 1, out); }

int mcp_ctx_device_count(const mcp_ctx* c) { return c ? (int)c->sh.size() : 0; }

int mcp_ctx_set_terminal_budget(mcp_ctx* c, size_t bytes) {
  if (!c || bytes < 4096) return fail(MCP_E_ARG, "bad argument");
  std::lock_guard<std::mutex> lock(c->mu);
  c->terminal_budget = bytes;
  return MCP_OK;
}

void mcp_ctx_destroy(mcp_ctx* c) {
  if (!c) return;
  int prev_dev = 0;
  const bool have_prev = hipGetDevice(&prev_dev) == hipSuccess;
  for (Shard& sh : c->sh)                                  // drain every stream, then the communicators, then the streams
    if (sh.stream) { (void)hipSetDevice(sh.device); (void)hipStreamSynchronize(sh.stream); }
  if (c->rccl)
    for (void* comm : c->comms)
      if (comm) (void)c->rccl->CommDestroy(comm);
  for (Shard& sh : c->sh) free_shard(sh);
  if (c->d_sweep && !c->sh.empty()) { (void)hipSetDevice(c->sh[0].device); (void)hipFree(c->d_sweep); }
  if (have_prev) (void)hipSetDevice(prev_dev);
  delete c;
}

static int grow_dev(void** p, size_t* cap, size_t need, hipStream_t zero_on = nullptr, bool zero = false) {
  if (need <= *cap) return MCP_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t bytes = (need + 7) & ~(size_t)7;
  if (hipMalloc(p, bytes) != hipSuccess) return fail(MCP_E_NOMEM, "hipMalloc(%zu) failed", bytes);
  *cap = need;
  if (zero) HIP_TRY(mcp::launch_zero(*p, bytes, zero_on));
  return MCP_OK;
}

static int grow_host(void** p, size_t* cap, size_t need, bool mapped = false) {
  if (need <= *cap) return MCP_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  if (hipHostMalloc(p, need, mapped ? (hipHostMallocMapped | hipHostMallocPortable) : hipHostMallocPortable) != hipSuccess)   // portable: every device of a multi-device context copies from / writes to it
    return fail(MCP_E_NOMEM, "hipHostMalloc(%zu) failed", need);
  *cap = need;
  return MCP_OK;
}

namespace {

// What one shard does in one tile of portfolios.
struct Job {
  int k0 = 0, kt = 0;            // portfolios [k0, k0+kt) of the caller's W
  uint64_t p0 = 0, pn = 0;       // paths [p0, p0+pn) relative to path_begin
  bool active = false;
};

#define RCCL_TRY(c, expr)                                                                            \
  do {                                                                                               \
    int r_ = (expr);                                                                                 \
    if (r_ != 0) return fail(MCP_E_COMM, "%s: %s", #expr, (c)->rccl->GetErrorString(r_));            \
  } while (0)

// histogram all-reduce (SUM, u64) over the shards of a path-sharded tile
int exchange_hist(mcp_ctx* c, int kt) {
  const size_t words = (size_t)kt * 2 * MCP_SELECT_BINS;
  const size_t S = c->sh.size();
  if (c->same_device) {
    Shard& s0 = c->sh[0];
    HIP_TRY(hipSetDevice(s0.device));
    unsigned long long* bufs[8];
    for (size_t s = 0; s < S; s++) {
      bufs[s] = (unsigned long long*)c->sh[s].ws[MCP_WS_HIST];
      if (s) { HIP_TRY(hipEventRecord(c->sh[s].ev, c->sh[s].stream)); HIP_TRY(hipStreamWaitEvent(s0.stream, c->sh[s].ev, 0)); }
    }
    HIP_TRY(mcp::launch_sum_u64(bufs, (int)S, words, s0.stream));
    HIP_TRY(hipEventRecord(s0.ev, s0.stream));
    for (size_t s = 1; s < S; s++) HIP_TRY(hipStreamWaitEvent(c->sh[s].stream, s0.ev, 0));
    return MCP_OK;
  }
  RCCL_TRY(c, c->rccl->GroupStart());
  for (size_t s = 0; s < S; s++) {
    void* h = c->sh[s].ws[MCP_WS_HIST];
    const int r = c->rccl->AllReduce(h, h, words, NCCL_UINT64, NCCL_SUM, c->comms[s], c->sh[s].stream);
    if (r != 0) { (void)c->rccl->GroupEnd(); return fail(MCP_E_COMM, "ncclAllReduce (shard %zu): %s", s, c->rccl->GetErrorString(r)); }   // never leave the group open
  }
  RCCL_TRY(c, c->rccl->GroupEnd());
  return MCP_OK;
}

// all-gather of the [kt] records into every shard's d_gather [S][kt] (same device: only shard 0 needs them)
int exchange_records(mcp_ctx* c, int kt) {
  const size_t S = c->sh.size();
  const size_t bytes = (size_t)kt * sizeof(mcp_record);
  if (c->same_device) {
    Shard& s0 = c->sh[0];
    HIP_TRY(hipSetDevice(s0.device));
    for (size_t s = 0; s < S; s++) {
      if (s) { HIP_TRY(hipEventRecord(c->sh[s].ev, c->sh[s].stream)); HIP_TRY(hipStreamWaitEvent(s0.stream, c->sh[s].ev, 0)); }
      HIP_TRY(hipMemcpyAsync((char*)s0.d_gather + s * bytes, c->sh[s].ws[MCP_WS_RECORD], bytes, hipMemcpyDeviceToDevice, s0.stream));
    }
    return MCP_OK;
  }
  RCCL_TRY(c, c->rccl->GroupStart());
  for (size_t s = 0; s < S; s++) {
    const int r = c->rccl->AllGather(c->sh[s].ws[MCP_WS_RECORD], c->sh[s].d_gather, (size_t)kt * (sizeof(mcp_record) / sizeof(double)),
                                     NCCL_FLOAT64, c->comms[s], c->sh[s].stream);
    if (r != 0) { (void)c->rccl->GroupEnd(); return fail(MCP_E_COMM, "ncclAllGather (shard %zu): %s", s, c->rccl->GetErrorString(r)); }
  }
  RCCL_TRY(c, c->rccl->GroupEnd());
  return MCP_OK;
}

// One tile: every active shard simulates its (portfolios x paths) block -- the kernels' epilogue leaves the moment partials
// and the digit-0 histogram -- and the rest of the statistics pipeline runs, with the exchanges between the steps when the
// tile is path-sharded (`exchange`).
int run_tile(mcp_ctx* c, const mcp_params* prm, const float* mu, const float* chol, const float* W, uint64_t seed,
             uint64_t path_begin, uint64_t n_total, const std::vector<Job>& jobs, bool exchange, float* terminal_out,
             mcp_stats* stats_out) {
  const size_t S = c->sh.size();
  uint64_t lo, hi;
  double gamma;
  if (int rc = mcp_percentile_rank(n_total, prm->alpha, &lo, &hi, &gamma)) return rc;
  std::vector<mcp_params> tp(S, *prm);
  int rc;
  // 1a. buffers (steady state: nothing to do)
  for (size_t s = 0; s < S; s++) {
    const Job& j = jobs[s];
    if (!j.active) continue;
    Shard& sh = c->sh[s];
    HIP_TRY(hipSetDevice(sh.device));
    tp[s].n_portfolios = j.kt;
    const size_t plen = mcp_packed_len(prm->n_assets, j.kt);
    for (int w = 0; w < MCP_WS_COUNT; w++)      // only the histogram and the select state must start zeroed
      if (w != MCP_WS_STATS && (rc = grow_dev(&sh.ws[w], &sh.ws_cap[w], mcp_ws_bytes(w, j.kt, j.pn ? j.pn : 1), sh.stream,
                                              w == MCP_WS_HIST || w == MCP_WS_STATE))) return rc;
    if ((rc = grow_dev((void**)&sh.d_packed, &sh.packed_cap, plen * sizeof(float)))) return rc;
    if ((rc = grow_dev((void**)&sh.d_terminal, &sh.terminal_cap, (size_t)j.kt * (j.pn ? j.pn : 1) * sizeof(float)))) return rc;
    if (exchange && (rc = grow_dev((void**)&sh.d_gather, &sh.gather_cap, S * (size_t)j.kt * sizeof(mcp_record)))) return rc;
    if ((rc = grow_host((void**)&sh.h_packed, &sh.h_packed_cap, plen * sizeof(float)))) return rc;
    if ((rc = grow_host((void**)&sh.h_pivot, &sh.h_pivot_cap, (size_t)j.kt * sizeof(double)))) return rc;
    if ((size_t)j.kt * sizeof(mcp_stats) > sh.h_stats_cap) {
      if ((rc = grow_host((void**)&sh.h_stats, &sh.h_stats_cap, (size_t)j.kt * sizeof(mcp_stats), true))) return rc;
      HIP_TRY(hipHostGetDevicePointer(&sh.ws[MCP_WS_STATS], sh.h_stats, 0));
    }
  }
  // 1b. parameters up and the path kernels out, device after device with nothing else in between: every GPU should be
  //     simulating as early as possible.  Shards of a path-sharded tile share one packed block and one pivot vector
  //     (packed once; the pivot is a function of the inputs, identical on every shard by construction).
  const float* shared_packed = nullptr;
  const double* shared_pivot = nullptr;
  for (size_t s = 0; s < S; s++) {
    const Job& j = jobs[s];
    if (!j.active) continue;
    Shard& sh = c->sh[s];
    HIP_TRY(hipSetDevice(sh.device));
    const size_t plen = mcp_packed_len(prm->n_assets, j.kt);
    const float* src = sh.h_packed;
    const double* psrc = sh.h_pivot;
    if (exchange && shared_packed) {
      src = shared_packed;                                   // same portfolios on every shard: pinned + portable
      psrc = shared_pivot;
    } else {
      if ((rc = mcp_pack_params(prm->n_assets, j.kt, mu, chol, W + (size_t)j.k0 * prm->n_assets, sh.h_packed, plen))) return rc;
      if ((rc = mcp_pivots(&tp[s], mu, chol, W + (size_t)j.k0 * prm->n_assets, sh.h_pivot))) return rc;
      if (exchange) { shared_packed = sh.h_packed; shared_pivot = sh.h_pivot; }
    }
    HIP_TRY(hipMemcpyAsync(sh.d_packed, src, plen * sizeof(float), hipMemcpyHostToDevice, sh.stream));
    HIP_TRY(hipMemcpyAsync(sh.ws[MCP_WS_PIVOT], psrc, (size_t)j.kt * sizeof(double), hipMemcpyHostToDevice, sh.stream));
    if (j.pn) {
      if ((rc = mcp_launch_paths(&tp[s], sh.d_packed, (const double*)sh.ws[MCP_WS_PIVOT], seed, path_begin + j.p0, j.pn, sh.d_terminal,
                                 j.pn, sh.ws[MCP_WS_PARTIALS], sh.ws[MCP_WS_HIST], sh.stream))) return rc;
    } else {
      // a shard without paths (fewer paths than shards): empty moment partials, nothing in the histogram
      if ((rc = mcp_launch_pass0(&tp[s], sh.d_terminal, 1, 0, (const double*)sh.ws[MCP_WS_PIVOT], sh.ws[MCP_WS_PARTIALS],
                                 sh.ws[MCP_WS_HIST], sh.stream))) return rc;
    }
  }
  // 2. three descents of the radix select, exchanges between the steps
  const int kt_x = jobs[0].kt;                 // path-sharded tiles: the same portfolios on every shard
  for (int pass = 0; pass < 3; pass++) {
    if (exchange && (rc = exchange_hist(c, kt_x))) return rc;
    for (size_t s = 0; s < S; s++) {
      const Job& j = jobs[s];
      if (!j.active) continue;
      Shard& sh = c->sh[s];
      HIP_TRY(hipSetDevice(sh.device));
      const uint64_t stride = j.pn ? j.pn : 1;
      const double* piv = (const double*)sh.ws[MCP_WS_PIVOT];
      if (pass < 2) {
        if ((rc = mcp_launch_scan(&tp[s], pass, j.pn, lo, hi, sh.ws[MCP_WS_PARTIALS], sh.ws[MCP_WS_BELOW], piv, sh.ws[MCP_WS_HIST],
                                  sh.ws[MCP_WS_STATE], sh.ws[MCP_WS_RECORD], sh.stream))) return rc;
        if ((rc = mcp_launch_hist(&tp[s], pass + 1, sh.d_terminal, stride, j.pn, sh.ws[MCP_WS_STATE], piv, sh.ws[MCP_WS_BELOW],
                                  sh.ws[MCP_WS_HIST], sh.stream))) return rc;
      } else {
        if ((rc = mcp_launch_final(&tp[s], j.pn, gamma, lo, hi, sh.ws[MCP_WS_BELOW], sh.ws[MCP_WS_HIST], sh.ws[MCP_WS_STATE],
                                   sh.ws[MCP_WS_RECORD], sh.ws[MCP_WS_QUANT], exchange ? nullptr : sh.ws[MCP_WS_STATS], sh.stream)))
          return rc;
      }
    }
  }
  // 3. finish: merged records on shard 0 (path-sharded) or each shard's own records (portfolio-sharded / one shard)
  if (exchange) {
    if ((rc = exchange_records(c, kt_x))) return rc;
    Shard& s0 = c->sh[0];
    HIP_TRY(hipSetDevice(s0.device));
    if ((rc = mcp_launch_stats(&tp[0], (int)S, s0.d_gather, s0.ws[MCP_WS_QUANT], s0.ws[MCP_WS_STATS], s0.stream))) return rc;
  }
  for (size_t s = 0; s < S; s++) {
    const Job& j = jobs[s];
    if (!j.active) continue;
    Shard& sh = c->sh[s];
    HIP_TRY(hipSetDevice(sh.device));
    // the records are already in host memory when the stream drains: ws[MCP_WS_STATS] is the mapped h_stats
    if (terminal_out && j.pn)
      HIP_TRY(hipMemcpy2DAsync(terminal_out + (size_t)j.k0 * n_total + j.p0, n_total * sizeof(float), sh.d_terminal,
                               j.pn * sizeof(float), j.pn * sizeof(float), (size_t)j.kt, hipMemcpyDeviceToHost, sh.stream));
  }
  for (size_t s = 0; s < S; s++)
    if (jobs[s].active) { HIP_TRY(hipSetDevice(c->sh[s].device)); HIP_TRY(hipStreamSynchronize(c->sh[s].stream)); }
  for (size_t s = 0; s < S; s++)
    if (jobs[s].active && (!exchange || s == 0))
      memcpy(stats_out + jobs[s].k0, c->sh[s].h_stats, (size_t)jobs[s].kt * sizeof(mcp_stats));
  return MCP_OK;
}

// Portfolios per tile so that kt * n_paths * 4 B fits the budget: whole 512-portfolio workgroups of the MFMA sweep
// kernel when K is tiled at all.
int tile_portfolios(size_t budget, uint64_t n_paths, int K) {
  const uint64_t fit = budget / (sizeof(float) * (n_paths ? n_paths : 1));
  if (fit >= (uint64_t)K) return K;
  if (fit >= (uint64_t)K_PAD) return (int)(fit / K_PAD) * K_PAD;
  return fit >= 1 ? (int)fit : 1;
}

}  // namespace

int mcp_simulate(mcp_ctx* c, const mcp_params* prm, const float* mu, const float* chol, const float* W,
                 uint64_t seed, uint64_t path_begin, uint64_t n_paths, float* terminal_out, mcp_stats* stats_out) {
  if (!c) return fail(MCP_E_ARG, "ctx is NULL");
  if (int rc = check_params(prm)) return rc;
  if (!mu || !chol || !W || !stats_out) return fail(MCP_E_ARG, "NULL pointer");
  if (n_paths < 1) return fail(MCP_E_ARG, "n_paths must be >= 1");
  std::lock_guard<std::mutex> lock(c->mu);
  const size_t S = c->sh.size();
  const int K = prm->n_portfolios;
  int prev_dev = 0;
  (void)hipGetDevice(&prev_dev);
  int rc = MCP_OK;
  std::vector<Job> jobs(S);
  if (S > 1 && (prm->flags & MCP_FLAG_SHARD_PORTFOLIOS)) {
    // every shard walks all paths for its slice of W; slices are tiled independently; no exchange
    std::vector<int> kb(S + 1);
    for (size_t s = 0; s <= S; s++) kb[s] = (int)((int64_t)K * (int64_t)s / (int64_t)S);
    std::vector<int> done(S, 0);
    for (bool more = true; more && rc == MCP_OK;) {
      more = false;
      for (size_t s = 0; s < S; s++) {
        const int left = kb[s + 1] - kb[s] - done[s];
        jobs[s] = Job();
        if (left <= 0) continue;
        const int kt = std::min(left, tile_portfolios(c->terminal_budget, n_paths, left));
        jobs[s].k0 = kb[s] + done[s]; jobs[s].kt = kt; jobs[s].p0 = 0; jobs[s].pn = n_paths; jobs[s].active = true;
        done[s] += kt;
        more = true;
      }
      if (more) rc = run_tile(c, prm, mu, chol, W, seed, path_begin, n_paths, jobs, false, terminal_out, stats_out);
    }
  } else {
    // the path range is sharded; all shards see the same tile of portfolios
    uint64_t pn_max = 0;
    for (size_t s = 0; s < S; s++) {
      jobs[s].p0 = n_paths / S * s + std::min<uint64_t>(s, n_paths % S);
      jobs[s].pn = n_paths / S + (s < n_paths % S ? 1 : 0);
      jobs[s].active = true;
      pn_max = std::max(pn_max, jobs[s].pn);
    }
    const int kt_max = tile_portfolios(c->terminal_budget, pn_max, K);
    rc = ensure_exchange(c);
    for (int k0 = 0; k0 < K && rc == MCP_OK; k0 += kt_max) {
      for (size_t s = 0; s < S; s++) { jobs[s].k0 = k0; jobs[s].kt = std::min(kt_max, K - k0); }
      rc = run_tile(c, prm, mu, chol, W, seed, path_begin, n_paths, jobs, S > 1 || c->exchange_always, terminal_out, stats_out);
    }
  }
  if (rc != MCP_OK) {
    // Leave no work in flight behind a failed call, and restore the invariant of the read-and-clear protocol: a pass that
    // stopped half way may have left counts in the histograms, and the next call would add to them.
    const std::string why = g_err;
    for (Shard& sh : c->sh) { (void)hipSetDevice(sh.device); (void)hipStreamSynchronize(sh.stream); }
    for (Shard& sh : c->sh) {
      (void)hipSetDevice(sh.device);
      for (int w : {MCP_WS_HIST, MCP_WS_STATE})
        if (sh.ws[w]) (void)mcp::launch_zero(sh.ws[w], (sh.ws_cap[w] + 7) & ~(size_t)7, sh.stream);
      (void)hipStreamSynchronize(sh.stream);
    }
    (void)hipGetLastError();
    g_err = why;
  }
  (void)hipSetDevice(prev_dev);
  return rc;
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
  Shard& sh = c->sh[0];
  DeviceGuard guard(sh.device);
  if (guard.err != hipSuccess) return fail(MCP_E_HIP, "hipSetDevice(%d): %s", sh.device, hipGetErrorString(guard.err));
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
  hipStream_t s = sh.stream;
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
