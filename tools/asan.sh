#!/bin/bash
# Host-side sanitizer run: builds libmcport_asan.so (host code of every translation unit under
# AddressSanitizer + UBSan; device code untouched) and oracle/libmcoracle_asan.so, then runs the CPU test suite against them
# with the sanitizer runtime preloaded.  No GPU needed.   tools/asan.sh [pytest args...]   (log: stdout)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
make -C "$R/monte_carlo_portfolio_amd/csrc" -j8 asan > /dev/null
make -C "$R/oracle" libmcoracle_asan.so > /dev/null
cd "$R"
export MCP_LIB_PATH="$R/monte_carlo_portfolio_amd/libmcport_asan.so" MCO_LIB_PATH="$R/oracle/libmcoracle_asan.so"
# detect_leaks=0: CPython itself "leaks" at exit; verify_asan_link_order=0: the interpreter is not instrumented, the runtime is
# preloaded instead; abort on the first report so that a finding fails the run
export ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1:halt_on_error=1" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
echo "sanitizer runtime: $RT"
echo "libraries: $MCP_LIB_PATH $MCO_LIB_PATH"
echo "instrumentation: $(nm -D "$MCP_LIB_PATH" | grep -c '__asan\|__ubsan') __asan/__ubsan symbols referenced by libmcport_asan.so, $(nm -D "$MCO_LIB_PATH" | grep -c '__asan\|__ubsan') by libmcoracle_asan.so"
set +e
LD_PRELOAD="$RT" python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
rc=$?
# the instrumented libraries are big and must not travel to the GPU box with the tree
rm -rf "$R/monte_carlo_portfolio_amd/csrc/build_asan" "$MCP_LIB_PATH" "$MCO_LIB_PATH"
exit $rc
