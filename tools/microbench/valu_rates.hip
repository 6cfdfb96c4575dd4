// Instruction-issue microbenchmark for gfx950 (MI355X): measures cycles per wave-instruction per
// SIMD for the VALU / transcendental / integer-multiply / MFMA instructions the Monte Carlo path
// kernel is made of, at 1, 2, 4 and 8 waves per SIMD. Calibrates the VALU issue roofline used in
// DESIGN.md and bench.py.  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;
constexpr int PER_ITER = 32;   // instructions of the tested kind per loop iteration

// 8 independent chains x 4 repeats
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) bench(float* out, unsigned long long* cyc, float seedf, unsigned seedu) {
  float v[8]; unsigned u[8]; unsigned long long w[8]; f32x2 p[8]; f32x4 acc[8];
  for (int i = 0; i < 8; i++) {
    v[i] = seedf + 0.001f * (threadIdx.x + i); u[i] = seedu * (threadIdx.x + 7 * i + 1);
    w[i] = u[i]; p[i] = f32x2{v[i], v[i] + 1.f}; acc[i] = f32x4{v[i], 0.f, 1.f, 2.f};
  }
  float ca = seedf * 0.5f, cb = 0.25f; unsigned cu = seedu | 1u;
  __shared__ f32x4 ldsbuf[64]; if (threadIdx.x < 64) ldsbuf[threadIdx.x] = f32x4{seedf, ca, cb, 1.f}; __syncthreads();
  unsigned ldsaddr = (unsigned)(size_t)(&ldsbuf[0]) + (blockIdx.x & 1) * 16; f32x2 spv = {seedf, seedf}; unsigned long long sp; __builtin_memcpy(&sp, &spv, 8); sp = __builtin_amdgcn_readfirstlane((unsigned)sp) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(sp >> 32)) << 32);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
    if constexpr (KIND == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(ca), "v"(cb));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 1) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 2) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 3) {
#define OP(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 4) {
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(u[i]), "v"(cu) : "vcc"); \
              asm volatile("" : "+v"(u[i]) : "v"(w[i]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 5) {
#define OP(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 6) {
#define OP(i) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 7) {
#define OP(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 8) {
#define OP(i) asm volatile("v_sin_f32 %0, %0" : "+v"(v[i]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 9) {
#define OP(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(v[i]) : "v"(u[i]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 10) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(cu) : );
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 11) {
#define OP(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 12) {
#define OP(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 13) {   // MFMA 16x16x4 f32 alone
#define OP(i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i], ca, acc[i], 0, 0, 0);
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 14) {   // 32 FMA + 8 MFMA interleaved in ONE wave
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(ca), "v"(cb));
#define OM(i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca, cb, acc[i], 0, 0, 0);
      OM(0) REP8(OP) OM(1) REP8(OP) OM(2) REP8(OP) OM(3) REP8(OP)
      OM(4) OM(5) OM(6) OM(7)
#undef OP
#undef OM
    } else if constexpr (KIND == 15) {   // v_add_u32
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 16) {   // fma with SGPR operand
#define OP(i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(v[i]) : "s"(seedf), "v"(cb));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 17) {   // v_mul_f32
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(ca));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 18) {   // v_mad_u64_u32 with 64-bit addend chain (true dependent use)
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(cu) : "vcc");
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 19) {   // 24 fma + 8 v_log interleaved
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(ca), "v"(cb));
#define OT(i) asm volatile("v_log_f32 %0, %0" : "+v"(p[i].x));
      OT(0) OP(0) OP(1) OP(2) OT(1) OP(3) OP(4) OP(5) OT(2) OP(6) OP(7) OP(0) OT(3) OP(1) OP(2) OP(3)
      OT(4) OP(4) OP(5) OP(6) OT(5) OP(7) OP(0) OP(1) OT(6) OP(2) OP(3) OP(4) OT(7) OP(5) OP(6) OP(7)
#undef OP
#undef OT
    } else if constexpr (KIND == 20) {   // 24 fma + 8 v_mul_hi interleaved
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(ca), "v"(cb));
#define OT(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(cu));
      OT(0) OP(0) OP(1) OP(2) OT(1) OP(3) OP(4) OP(5) OT(2) OP(6) OP(7) OP(0) OT(3) OP(1) OP(2) OP(3)
      OT(4) OP(4) OP(5) OP(6) OT(5) OP(7) OP(0) OP(1) OT(6) OP(2) OP(3) OP(4) OT(7) OP(5) OP(6) OP(7)
#undef OP
#undef OT
    } else if constexpr (KIND == 21) {   // VOP2 v_fmac_f32 with SGPR src0
#define OP(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i]) : "s"(seedf), "v"(cb));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 22) {   // VOP2 v_fmac_f32 all VGPR
#define OP(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i]) : "v"(ca), "v"(cb));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 23) {   // v_fmaak_f32 literal
#define OP(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000344" : "+v"(v[i]) : "v"(ca));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 24) {   // v_pk_fma_f32 with SGPR pair operand
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(sp));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 25) {   // v_cndmask_b32 with vcc set once
#define OP(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(cu));
      asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1" :: "v"(u[0]), "v"(cu) : "vcc");
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 26) {   // v_mul_f32 with SGPR
#define OP(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(v[i]) : "s"(seedf));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 27) {   // ds_read_b128 broadcast + 4 fmac (VGPR)
#define OP(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(acc[i]) : "v"(ldsaddr)); 
#define OF(i) asm volatile("s_waitcnt lgkmcnt(0)\n v_fmac_f32_e32 %0, %1, %2\n v_fmac_f32_e32 %0, %3, %2\n v_fmac_f32_e32 %0, %4, %2\n v_fmac_f32_e32 %0, %5, %2" : "+v"(v[i]) : "v"(acc[i].x), "v"(cb), "v"(acc[i].y), "v"(acc[i].z), "v"(acc[i].w));
      REP8(OP) REP8(OF)
#undef OP
#undef OF
    } else if constexpr (KIND == 28) {   // v_xor3 / bitop3 with sgpr
#define OP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u[i]) : "s"(seedu), "v"(cu));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 29) {   // v_xor_b32 with sgpr (VOP2)
#define OP(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(u[i]) : "s"(seedu));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 30) {   // v_cvt_f32_i32
#define OP(i) asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(v[i]) : "v"(u[i]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 31) {   // v_pk_mul_f32
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 32) {   // v_bfe_i32
#define OP(i) asm volatile("v_bfe_i32 %0, %0, 0, 30" : "+v"(u[i]));
      REP32(OP)
#undef OP
    } else if constexpr (KIND == 33) {   // v_cmp_lt_f32 to sgpr pair + cndmask e64
#define OP(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, %0, %3, s[20:21]" : "+v"(u[i]) : "v"(ca), "v"(v[i]), "v"(cu) : "s20", "s21");
      REP8(OP) REP8(OP)
#undef OP
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f; unsigned su = 0;
  for (int i = 0; i < 8; i++) { s += v[i] + p[i].x + p[i].y + acc[i].x + acc[i].y + acc[i].z + acc[i].w; su ^= u[i] ^ (unsigned)w[i] ^ (unsigned)(w[i] >> 32); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)su;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Kind { int id; const char* name; int n_per_iter; };

template <int KIND>
void run(const char* name, int n_per_iter, float* d_out, unsigned long long* d_cyc, int num_cu) {
  for (int wps : {1, 2, 4, 8}) {
    int blocks = num_cu * wps;   // 256 threads = 4 waves -> one per SIMD per block
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bench<KIND><<<blocks, 256>>>(d_out, d_cyc, 1.0001f, 0x9E3779B9u);   // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    bench<KIND><<<blocks, 256>>>(d_out, d_cyc, 1.0001f, 0x9E3779B9u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), d_cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double avg = 0; for (auto c : h) avg += (double)c; avg /= blocks;
    double n_instr = (double)ITERS * n_per_iter;            // per wave
    // s_memtime runs at a fixed 100 MHz on gfx9 (REFCLK); report wall-derived cycles at 2.4 GHz too
    double wall_cyc_per_instr_per_simd = (ms * 1e-3 * 2.4e9) / (n_instr * wps);
    printf("%-28s waves/SIMD=%d  wall=%8.3f ms  cyc@2.4GHz/instr/SIMD=%6.2f  memtime_ticks/wave=%10.0f\n",
           name, wps, ms, wall_cyc_per_instr_per_simd, avg);
  }
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int num_cu = prop.multiProcessorCount;
  printf("device %s  CUs=%d  clockRate=%d kHz\n", prop.gcnArchName, num_cu, prop.clockRate);
  float* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, (size_t)num_cu * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&d_cyc, (size_t)num_cu * 8 * sizeof(unsigned long long)));
  run<0>("v_fma_f32", 32, d_out, d_cyc, num_cu);
  run<16>("v_fma_f32 (sgpr operand)", 32, d_out, d_cyc, num_cu);
  run<17>("v_mul_f32", 32, d_out, d_cyc, num_cu);
  run<1>("v_pk_fma_f32", 32, d_out, d_cyc, num_cu);
  run<5>("v_xor_b32", 32, d_out, d_cyc, num_cu);
  run<15>("v_add_u32", 32, d_out, d_cyc, num_cu);
  run<10>("v_cndmask_b32", 32, d_out, d_cyc, num_cu);
  run<9>("v_cvt_f32_u32", 32, d_out, d_cyc, num_cu);
  run<2>("v_mul_lo_u32", 32, d_out, d_cyc, num_cu);
  run<3>("v_mul_hi_u32", 32, d_out, d_cyc, num_cu);
  run<4>("v_mad_u64_u32", 32, d_out, d_cyc, num_cu);
  run<18>("v_mad_u64_u32 (acc chain)", 32, d_out, d_cyc, num_cu);
  run<11>("v_mul_u32_u24", 32, d_out, d_cyc, num_cu);
  run<12>("v_mul_hi_u32_u24", 32, d_out, d_cyc, num_cu);
  run<6>("v_log_f32", 32, d_out, d_cyc, num_cu);
  run<7>("v_sqrt_f32", 32, d_out, d_cyc, num_cu);
  run<8>("v_sin_f32", 32, d_out, d_cyc, num_cu);
  run<13>("v_mfma_f32_16x16x4_f32", 32, d_out, d_cyc, num_cu);
  run<14>("32 fma + 8 mfma16x16x4 (1 wave)", 40, d_out, d_cyc, num_cu);
  run<19>("24 fma + 8 v_log", 32, d_out, d_cyc, num_cu);
  run<20>("24 fma + 8 v_mul_hi", 32, d_out, d_cyc, num_cu);
  run<21>("v_fmac_f32_e32 sgpr src0", 32, d_out, d_cyc, num_cu);
  run<22>("v_fmac_f32_e32 vgpr", 32, d_out, d_cyc, num_cu);
  run<23>("v_fmaak_f32 literal", 32, d_out, d_cyc, num_cu);
  run<24>("v_pk_fma_f32 sgpr pair", 32, d_out, d_cyc, num_cu);
  run<25>("v_cndmask_b32_e32 (vcc)", 32, d_out, d_cyc, num_cu);
  run<26>("v_mul_f32_e32 sgpr", 32, d_out, d_cyc, num_cu);
  run<27>("8 ds_read_b128 + 32 fmac", 40, d_out, d_cyc, num_cu);
  run<28>("v_xor3_b32 sgpr", 32, d_out, d_cyc, num_cu);
  run<29>("v_xor_b32_e32 sgpr", 32, d_out, d_cyc, num_cu);
  run<30>("v_cvt_f32_i32", 32, d_out, d_cyc, num_cu);
  run<31>("v_pk_mul_f32", 32, d_out, d_cyc, num_cu);
  run<32>("v_bfe_i32", 32, d_out, d_cyc, num_cu);
  run<33>("v_cmp_e64+v_cndmask_e64 (16 pairs)", 32, d_out, d_cyc, num_cu);
  return 0;
}
