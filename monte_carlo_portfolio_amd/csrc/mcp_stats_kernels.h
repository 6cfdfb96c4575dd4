// mcp_stats_kernels.h -- internal launch interface of mcp_stats_kernels.hip / mcp_paths_inst.hip / mcp_sweep_paths.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mcport.h"

namespace mcp {

struct SelectState {   // per (portfolio, which order statistic)
  uint32_t prefix;     // key bits decided so far (right-aligned)
  uint32_t pad;
  uint64_t rank;       // rank of the target among the keys sharing the prefix
};

struct Quantile {      // per portfolio, identical on every rank (derived from all-reduced histograms only)
  double x_lo, x_hi, var;
  double level2;       // sum of x over the tail elements inside the bucket of the low order statistic (+ ties of x_hi)
  uint64_t n_tail;     // #{x <= var} over all ranks
  uint64_t pad;
};

// What one workgroup of a path kernel (K <= 16) or one 64-path wave tile (MFMA sweep kernels) contributes to the moments
// of ONE portfolio: shifted sums around the portfolio's pivot c (include/mcport.h: mcp_pivots), the extreme TERMINAL
// values (x is monotone in them) and the number of paths.  32 bytes; [K][moment_slots] of them per pass.
struct MomentPartial {
  double s1, s2;             // sum (x - c), sum (x - c)^2
  float vmin, vmax;
  unsigned long long n;
};
static_assert(sizeof(MomentPartial) == 32, "MomentPartial is 32 bytes");

constexpr int PATH_GRID_CAP = 8192;      // path-kernel blocks (K <= 16); tiles beyond are grid-strided
constexpr int SWEEP_MIN_K = 17;          // from this many portfolios on the MFMA sweep kernels run
constexpr int SWEEP_TILE_PATHS = 64;     // the sweep kernels contribute one MomentPartial per portfolio and 64-path wave tile

// blocks of the one-lane-per-path kernels for n paths
__host__ __device__ inline int path_grid(uint64_t n) {
  uint64_t tiles = (n + 255) / 256;
  if (tiles < 1) tiles = 1;
  return (int)(tiles < (uint64_t)PATH_GRID_CAP ? tiles : (uint64_t)PATH_GRID_CAP);
}
// MomentPartial slots per portfolio that a fused path launch -- and the standalone pass 0, which pads to the same count --
// fills for n paths.  `sweep`: the launch goes to the MFMA sweep kernels (mcp_api.cpp: uses_sweep).
__host__ __device__ inline uint64_t moment_slots(bool sweep, uint64_t n) {
  return sweep ? (n + SWEEP_TILE_PATHS - 1) / SWEEP_TILE_PATHS + (n == 0) : (uint64_t)path_grid(n);
}

// "below" partial slots per portfolio = the most blocks a streaming select pass may use per portfolio.  One portfolio gets
// 2,048 blocks; many portfolios share 16,384 slots, at least 1 each.
__host__ __device__ inline int stream_slots(int K) {
  const int s = 16384 / (K < 1 ? 1 : K);
  return s > 2048 ? 2048 : (s < 1 ? 1 : s);
}

struct PathArgs;

// variant bits for launch_paths
enum { VAR_KT8 = 1, VAR_NATIVE = 4, VAR_FOLD = 8 };

// mcp_paths_inst.hip (one translation unit per NB): returns hipErrorInvalidValue for a variant that
// is not instantiated.
typedef hipError_t (*launch_paths_fn)(int variant, const PathArgs& args, int grid, hipStream_t stream);
#define MCP_DECL_NB(n) hipError_t launch_paths_nb##n(int variant, const PathArgs& args, int grid, hipStream_t stream);
MCP_DECL_NB(1) MCP_DECL_NB(2) MCP_DECL_NB(3) MCP_DECL_NB(4) MCP_DECL_NB(5) MCP_DECL_NB(6) MCP_DECL_NB(7) MCP_DECL_NB(8)
MCP_DECL_NB(9) MCP_DECL_NB(10) MCP_DECL_NB(11) MCP_DECL_NB(12) MCP_DECL_NB(13) MCP_DECL_NB(14) MCP_DECL_NB(15) MCP_DECL_NB(16)
#undef MCP_DECL_NB

// mcp_sweep_paths.hip: MFMA K-portfolio kernels; mt = 32-portfolio tiles per wave (1, 2 or 4)
hipError_t launch_sweep_shared(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream);   // mt: 4|2 (N <= 16), 2|1 (N > 16)
hipError_t launch_sweep_shared_p0(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_shared_p1(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_shared_p2(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_shared_p3(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_paths(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_normals(const uint32_t* x, uint64_t n, const float4* table, float* z, hipStream_t s);

// statistics pipeline (mcp_stats_kernels.hip):
//   fused:       paths (moment partials [+ digit-0 histogram for K <= 16]) [-> hist(0) for the sweep kernels]
//   standalone:  pass0 (moments + digit-0 histogram of caller-supplied terminal values)
//   then         scan(0) -> hist(1) -> scan(1) -> hist(2) -> final [-> stats]
hipError_t launch_pass0(const mcp_params& prm, int K, const float* terminal, uint64_t stride, uint64_t n, const double* pivot,
                        uint64_t slots, MomentPartial* partials, unsigned long long* hist, hipStream_t s);
hipError_t launch_scan(const mcp_params& prm, int K, int pass, uint64_t n, uint64_t rank_lo, uint64_t rank_hi, uint64_t slots,
                       const MomentPartial* partials, const double* below, const double* pivot, unsigned long long* hist,
                       SelectState* state, mcp_record* record, hipStream_t s);
hipError_t launch_hist(const mcp_params& prm, int K, int pass, const float* terminal, uint64_t stride, uint64_t n,
                       const SelectState* state, const double* pivot, double* below, unsigned long long* hist, hipStream_t s);
hipError_t launch_final(const mcp_params& prm, int K, uint64_t n, double gamma, uint64_t rank_lo, uint64_t rank_hi,
                        const double* below, unsigned long long* hist, const SelectState* state, mcp_record* record,
                        Quantile* quant, mcp_stats* stats_or_null, hipStream_t s);
hipError_t launch_stats(const mcp_params& prm, int K, int world, const mcp_record* gathered, const Quantile* quant,
                        mcp_stats* out, hipStream_t s);
hipError_t launch_zero(void* p, size_t bytes, hipStream_t s);
// every buffer <- element-wise sum of the `nsrc` buffers (u64 words): the exchange between logical shards of ONE device
// (or of devices with peer access)
hipError_t launch_sum_u64(unsigned long long* const* bufs, int nsrc, size_t words, hipStream_t s);

hipError_t launch_sweep_hist(int N, int R, int P, const double* returns, const double* mean, const double* cov,
                             const double* W, double rf, uint64_t rank_lo, uint64_t rank_hi, double gamma,
                             double* out5, hipStream_t s);

}  // namespace mcp
