// mcp_stats_kernels.h -- internal launch interface of mcp_stats_kernels.hip / mcp_paths_inst.hip.
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

constexpr int PARTIAL_DOUBLES = 6;    // {n, sum, sumsq, min, max} of pass 0, {below} of passes 1 and 2
// Partial-record slots per portfolio = the most blocks a streaming pass may use per portfolio.  One portfolio gets 2,048
// blocks (8 waves per SIMD: the passes are latency-bound until then); many portfolios share 16,384 slots, at least 8 each.
__host__ __device__ inline int stream_slots(int K) {
  const int s = 16384 / (K < 1 ? 1 : K);
  return s > 2048 ? 2048 : (s < 8 ? 8 : s);
}

struct PathArgs;

// variant bits for launch_paths
enum { VAR_KT8 = 1, VAR_PPT2 = 2, VAR_NATIVE = 4, VAR_FOLD = 8 };

// mcp_paths_inst.hip (one translation unit per NB): returns hipErrorInvalidValue for a variant that
// is not instantiated.
typedef hipError_t (*launch_paths_fn)(int variant, const PathArgs& args, int grid, hipStream_t stream);
#define MCP_DECL_NB(n) hipError_t launch_paths_nb##n(int variant, const PathArgs& args, int grid, hipStream_t stream);
MCP_DECL_NB(1) MCP_DECL_NB(2) MCP_DECL_NB(3) MCP_DECL_NB(4) MCP_DECL_NB(5) MCP_DECL_NB(6) MCP_DECL_NB(7) MCP_DECL_NB(8)
MCP_DECL_NB(9) MCP_DECL_NB(10) MCP_DECL_NB(11) MCP_DECL_NB(12) MCP_DECL_NB(13) MCP_DECL_NB(14) MCP_DECL_NB(15) MCP_DECL_NB(16)
#undef MCP_DECL_NB

// mcp_sweep_paths.hip: MFMA K-portfolio kernel (N <= 16); mt = 32-portfolio tiles per wave (1, 2 or 4)
hipError_t launch_sweep_shared(int nb, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_shared_p0(int nb, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_shared_p1(int nb, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_shared_p2(int nb, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_shared_p3(int nb, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_sweep_paths(int nb, int mt, bool native, const PathArgs& args, hipStream_t stream);
hipError_t launch_normals(const uint32_t* x, uint64_t n, const float4* table, float* z, hipStream_t s);

// statistics pipeline (mcp_stats_kernels.hip): pass0 -> scan(0) -> hist(1) -> scan(1) -> hist(2) -> final [-> stats]
hipError_t launch_pass0(const mcp_params& prm, int K, const float* terminal, uint64_t stride, uint64_t n, double* partials,
                        unsigned long long* hist, hipStream_t s);
hipError_t launch_scan(int K, int pass, uint64_t n, uint64_t rank_lo, uint64_t rank_hi, const double* partials,
                       unsigned long long* hist, SelectState* state, mcp_record* record, hipStream_t s);
hipError_t launch_hist(const mcp_params& prm, int K, int pass, const float* terminal, uint64_t stride, uint64_t n,
                       const SelectState* state, double* partials, unsigned long long* hist, hipStream_t s);
hipError_t launch_final(const mcp_params& prm, int K, uint64_t n, double gamma, uint64_t rank_lo, uint64_t rank_hi,
                        const double* partials, unsigned long long* hist, const SelectState* state, mcp_record* record,
                        Quantile* quant, mcp_stats* stats_or_null, hipStream_t s);
hipError_t launch_stats(const mcp_params& prm, int K, int world, const mcp_record* gathered, const Quantile* quant,
                        mcp_stats* out, hipStream_t s);
// out[i] = sum over the `nsrc` buffers src[0..nsrc) (u64 words), written to every buffer: the exchange between
// several logical shards resident on ONE device (mcp_ctx_create_multi with a repeated device)
hipError_t launch_zero(void* p, size_t bytes, hipStream_t s);
hipError_t launch_sum_u64(unsigned long long* const* bufs, int nsrc, size_t words, hipStream_t s);

hipError_t launch_sweep_hist(int N, int R, int P, const double* returns, const double* mean, const double* cov,
                             const double* W, double rf, uint64_t rank_lo, uint64_t rank_hi, double gamma,
                             double* out5, hipStream_t s);

}  // namespace mcp
