// mcp_paths_inst.hip -- instantiates mc_paths_kernel for ONE value of NB (= ceil(N/4), -DMCP_NB=n).
// Built once per NB in 1..16 so the 16 translation units compile in parallel (see Makefile).
#include <cstdlib>

#include "mcp_paths.h"
#include "mcp_stats_kernels.h"

#ifndef MCP_NB
#error "compile with -DMCP_NB=<1..16>"
#endif

#define MCP_CAT_(a, b) a##b
#define MCP_CAT(a, b) MCP_CAT_(a, b)

namespace mcp {

// MCP_PATHS_LDS_PAD (bytes, env, experiment): extra dynamic LDS per workgroup.  4608 on top of the 16 KiB of tables
// makes 7 instead of 8 workgroups fit a CU, which leaves one wave slot per SIMD free for the small statistics
// kernels of the previous batch when batches are pipelined (engine.PathEngine).
static size_t lds_pad() {
  static const size_t pad = [] { const char* e = getenv("MCP_PATHS_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }();
  return pad;
}

template <int KT, int PPT, bool NATIVE, bool FOLD = false>
static hipError_t go(const PathArgs& args, int grid, hipStream_t stream) {
  if (args.compounding == MCP_COMPOUND_LOG)
    mc_paths_kernel<MCP_NB, KT, PPT, NATIVE, FOLD, true><<<grid, PATH_BLOCK, lds_pad(), stream>>>(args);
  else
    mc_paths_kernel<MCP_NB, KT, PPT, NATIVE, FOLD, false><<<grid, PATH_BLOCK, lds_pad(), stream>>>(args);
  return hipGetLastError();
}

hipError_t MCP_CAT(launch_paths_nb, MCP_NB)(int variant, const PathArgs& args, int grid, hipStream_t stream) {
  switch (variant) {
    case 0: return go<1, 1, false>(args, grid, stream);
    case VAR_NATIVE: return go<1, 1, true>(args, grid, stream);
    case VAR_FOLD: return go<1, 1, false, true>(args, grid, stream);
    case VAR_KT8: return go<8, 1, false>(args, grid, stream);
    case VAR_KT8 | VAR_NATIVE: return go<8, 1, true>(args, grid, stream);
#if MCP_NB <= 4 && defined(MCP_EXP_PPT2)      // two paths per lane: measured 3 % slower (129 VGPRs), kept behind a build flag
    case VAR_PPT2: return go<1, 2, false>(args, grid, stream);
    case VAR_PPT2 | VAR_NATIVE: return go<1, 2, true>(args, grid, stream);
#endif
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mcp
