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

struct Quantile {
  double x_lo, x_hi, var;
};

constexpr int TAIL_GRID = 256;
constexpr int MOMENTS_GRID = 256;

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
hipError_t launch_moments(const mcp_params& prm, int K, const float* terminal, uint64_t stride, uint64_t n,
                          mcp_moments* partials, mcp_moments* out, hipStream_t s);
hipError_t launch_moments_merge(int K, int world, const mcp_moments* gathered, mcp_moments* out, hipStream_t s);
hipError_t launch_select_init(int K, uint64_t rank_lo, uint64_t rank_hi, SelectState* state, hipStream_t s);
hipError_t launch_select_hist(int K, const float* terminal, uint64_t stride, uint64_t n, int pass,
                              const SelectState* state, unsigned long long* hist, hipStream_t s);
hipError_t launch_select_scan(int K, int pass, const unsigned long long* hist, SelectState* state, hipStream_t s);
hipError_t launch_quantile(const mcp_params& prm, int K, double gamma, const SelectState* state, Quantile* out, hipStream_t s);
hipError_t launch_tail(const mcp_params& prm, int K, const float* terminal, uint64_t stride, uint64_t n,
                       const Quantile* quant, double* partial, double* tail, hipStream_t s);
hipError_t launch_stats(const mcp_params& prm, int K, const mcp_moments* mom, const Quantile* quant,
                        const double* tail, mcp_stats* out, hipStream_t s);

hipError_t launch_sweep_hist(int N, int R, int P, const double* returns, const double* mean, const double* cov,
                             const double* W, double rf, uint64_t rank_lo, uint64_t rank_hi, double gamma,
                             double* out5, hipStream_t s);

}  // namespace mcp
