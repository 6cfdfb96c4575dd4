// configs[3] (64 assets): the per-step triangular product r = mu + L z for 64 paths, timed in isolation under the two
// formulations BASELINE.json / VERDICT r1 item 6 ask to settle by measurement:
//
//   VALU   what mc_paths_kernel<16,...> runs: one lane = one path, z[64] in VGPRs, L as SGPR row pairs, two rows per
//          v_pk_fma_f32 (1,056 packed FMAs per wave-step = the 2,080 FMAs of the lower triangle + padding zeros).
//   MFMA   "Cholesky-GEMV cast as fp32 MFMA GEMM": R[64 x 64 paths] = L[64 x 64] Z[64 x 64 paths] as
//          v_mfma_f32_16x16x4_f32 tiles that skip the empty upper blocks (160 MFMAs per wave-step).  Optimistic for MFMA:
//          the 40 A operands (tiles of L) stay resident in VGPRs for the whole walk; z is written to LDS once per step in
//          lane-per-path form and read back in B-operand layout (64 ds_read_b32), exactly what the product would have to
//          do because the normals are generated one path per lane; the accumulators are consumed by a weight dot.
//
// Both kernels run T steps with the same (synthetic) z stream and print cycles per wave-step per SIMD.  Exact fp32 in both
// (MFMA fp32 multiplies and accumulates in binary32; the k-order inside a 16x16x4 MFMA is fixed by the hardware).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/microbench/gemv64_engines.hip -o gemv64_engines
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 64;

// cheap per-step pseudo-normal so the compiler cannot fold the products (not a statistical generator)
__device__ __forceinline__ float fake_z(uint32_t s, int j) { return __uint_as_float(0x3f000000u | ((s * 2654435761u + (uint32_t)j * 40503u) & 0x007fffffu)) - 0.75f; }

// ---- VALU: row pairs from SGPRs, as mcp_paths.h ----------------------------------------------------------------------
__global__ void __launch_bounds__(256) gemv_valu(const float* __restrict__ packed, int T, float* __restrict__ out) {
  typedef const __attribute__((address_space(4))) float* cfloat_p;
  cfloat_p mu = (cfloat_p)packed;
  cfloat_p Lp = mu + N;
  cfloat_p Wk = mu + N + N * (N / 2 + 1);
  float V = 1.0f;
  const uint32_t lane_seed = blockIdx.x * 256 + threadIdx.x;
  for (int t = 0; t < T; t++) {
    asm volatile("" : "+s"(mu), "+s"(Lp), "+s"(Wk));
    float z[N];
#pragma unroll
    for (int j = 0; j < N; j++) z[j] = fake_z(lane_seed + t, j);
    float rho = 0.0f;
#pragma unroll
    for (int m = 0; m < N / 2; m++) {
      f32x2 acc = {mu[2 * m], mu[2 * m + 1]};
#pragma unroll
      for (int j = 0; j <= 2 * m + 1; j++) {
        const f32x2 l2 = {Lp[2 * m * (m + 1) + 2 * j], Lp[2 * m * (m + 1) + 2 * j + 1]};
        acc = __builtin_elementwise_fma(l2, (f32x2){z[j], z[j]}, acc);
      }
      rho = __builtin_fmaf(Wk[2 * m], acc.x, rho);
      rho = __builtin_fmaf(Wk[2 * m + 1], acc.y, rho);
    }
    V = __builtin_fmaf(V, rho, V);
  }
  out[blockIdx.x * 256 + threadIdx.x] = V;
}

// ---- VALU, quad rows: four rows per SGPR quad (L[4g..4g+3][j]), two independent v_pk_fma_f32 chains per column ----------
// Same FMAs plus 2 packed zero-FMAs per quad (rows 4g, 4g+1 run to column 4g+3); the two accumulators of a column are
// independent, so consecutive instructions never depend on each other.
__global__ void __launch_bounds__(256) gemv_valu_quad(const float* __restrict__ packed, const float* __restrict__ quad, int T, float* __restrict__ out) {
  typedef const __attribute__((address_space(4))) float* cfloat_p;
  cfloat_p mu = (cfloat_p)packed;
  cfloat_p Lq = (cfloat_p)quad;
  cfloat_p Wk = mu + N + N * (N / 2 + 1);
  float V = 1.0f;
  const uint32_t lane_seed = blockIdx.x * 256 + threadIdx.x;
  for (int t = 0; t < T; t++) {
    asm volatile("" : "+s"(mu), "+s"(Lq), "+s"(Wk));
    float z[N];
#pragma unroll
    for (int j = 0; j < N; j++) z[j] = fake_z(lane_seed + t, j);
    float rho = 0.0f;
#pragma unroll
    for (int g = 0; g < N / 4; g++) {
      f32x2 a0 = {mu[4 * g], mu[4 * g + 1]}, a1 = {mu[4 * g + 2], mu[4 * g + 3]};
      const int base = 8 * g * (g + 1);                      // 4 * sum_{h<g} (4h + 4)
#pragma unroll
      for (int j = 0; j <= 4 * g + 3; j++) {
        const f32x2 l0 = {Lq[base + 4 * j], Lq[base + 4 * j + 1]}, l1 = {Lq[base + 4 * j + 2], Lq[base + 4 * j + 3]};
        a0 = __builtin_elementwise_fma(l0, (f32x2){z[j], z[j]}, a0);
        a1 = __builtin_elementwise_fma(l1, (f32x2){z[j], z[j]}, a1);
      }
      rho = __builtin_fmaf(Wk[4 * g], a0.x, rho);
      rho = __builtin_fmaf(Wk[4 * g + 1], a0.y, rho);
      rho = __builtin_fmaf(Wk[4 * g + 2], a1.x, rho);
      rho = __builtin_fmaf(Wk[4 * g + 3], a1.y, rho);
    }
    V = __builtin_fmaf(V, rho, V);
  }
  out[blockIdx.x * 256 + threadIdx.x] = V;
}

// ---- MFMA: 16x16x4 tiles of the lower triangle -----------------------------------------------------------------------
// One wave = 64 paths.  LDS per wave: z image [64 assets][64 paths] floats (16 KiB), r image the same (reused buffer).
__global__ void __launch_bounds__(256) gemv_mfma(const float* __restrict__ Ldense /* [64][64] row-major */, const float* __restrict__ mu,
                                                 const float* __restrict__ w, int T, float* __restrict__ out) {
  __shared__ float s_z[4][N][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float (*zz)[64] = s_z[wave];
  // A operands: tile (mt, kk) of L, lane l holds L[16 mt + l%16][4 kk + l/16]; 4+8+12+16 = 40 resident VGPRs
  float a[4][16];
#pragma unroll
  for (int mt = 0; mt < 4; mt++)
#pragma unroll
    for (int kk = 0; kk < 16; kk++) a[mt][kk] = kk < 4 * (mt + 1) ? Ldense[(16 * mt + (lane & 15)) * N + 4 * kk + (lane >> 4)] : 0.0f;
  // accumulator init: D[m = 4 (l/16) + i][n = l%16] starts at mu[16 mt + m]
  float mu_d[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; mt++)
#pragma unroll
    for (int i = 0; i < 4; i++) mu_d[mt][i] = mu[16 * mt + 4 * (lane >> 4) + i];
  float V = 1.0f;
  const uint32_t lane_seed = blockIdx.x * 256 + threadIdx.x;
  for (int t = 0; t < T; t++) {
    // the draw is lane-per-path: write the z image, then read it in B layout (lane l: asset 4 kk + l/16, path 16 nt + l%16)
#pragma unroll
    for (int j = 0; j < N; j++) zz[j][lane] = fake_z(lane_seed + t, j);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same-wave LDS write -> read ordering
    f32x4 acc[4][4];                                   // [nt][mt]
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      float b[16];
#pragma unroll
      for (int kk = 0; kk < 16; kk++) b[kk] = zz[4 * kk + (lane >> 4)][16 * nt + (lane & 15)];
#pragma unroll
      for (int mt = 0; mt < 4; mt++) {
        f32x4 d = {mu_d[mt][0], mu_d[mt][1], mu_d[mt][2], mu_d[mt][3]};
#pragma unroll
        for (int kk = 0; kk < 4 * (mt + 1); kk++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][kk], b[kk], d, 0, 0, 0);
        acc[nt][mt] = d;
      }
    }
    // r image back to lane-per-path through LDS (D[m][n]: asset 16 mt + 4 (l/16) + i, path 16 nt + l%16)
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int i = 0; i < 4; i++) zz[16 * mt + 4 * (lane >> 4) + i][16 * nt + (lane & 15)] = acc[nt][mt][i];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float rho = 0.0f;
#pragma unroll
    for (int i = 0; i < N; i++) rho = __builtin_fmaf(w[i], zz[i][lane], rho);   // w[i]: uniform -> scalar load
    V = __builtin_fmaf(V, rho, V);
  }
  out[blockIdx.x * 256 + threadIdx.x] = V;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 200;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<float> Ld(N * N, 0.0f), mu(N), w(N, 1.0f / N), packed(N + N * (N / 2 + 1) + N, 0.0f);
  for (int i = 0; i < N; i++) { mu[i] = 1e-4f * (i + 1); for (int j = 0; j <= i; j++) Ld[i * N + j] = 0.01f / (1 + i - j) * (j == i ? 1.0f : 0.3f); }
  for (int i = 0; i < N; i++) packed[i] = mu[i];
  for (int i = 0; i < N; i++) for (int j = 0; j <= i; j++) packed[N + 2 * (i / 2) * (i / 2 + 1) + 2 * j + (i & 1)] = Ld[i * N + j];
  for (int i = 0; i < N; i++) packed[N + N * (N / 2 + 1) + i] = w[i];
  std::vector<float> quad;
  for (int g = 0; g < N / 4; g++)
    for (int j = 0; j <= 4 * g + 3; j++)
      for (int h = 0; h < 4; h++) quad.push_back(j <= 4 * g + h ? Ld[(4 * g + h) * N + j] : 0.0f);
  float *d_packed, *d_L, *d_mu, *d_w, *d_out, *d_quad;
  CHECK(hipMalloc(&d_quad, quad.size() * 4));
  CHECK(hipMemcpy(d_quad, quad.data(), quad.size() * 4, hipMemcpyHostToDevice));
  const int blocks_per_cu = 2, grid = cus * blocks_per_cu * 4;   // several rounds of workgroups
  CHECK(hipMalloc(&d_packed, packed.size() * 4)); CHECK(hipMalloc(&d_L, Ld.size() * 4)); CHECK(hipMalloc(&d_mu, N * 4));
  CHECK(hipMalloc(&d_w, N * 4)); CHECK(hipMalloc(&d_out, (size_t)grid * 256 * 4));
  CHECK(hipMemcpy(d_packed, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_L, Ld.data(), Ld.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_mu, mu.data(), N * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_w, w.data(), N * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> ref((size_t)grid * 256), got((size_t)grid * 256);
  std::vector<float> got2((size_t)grid * 256);
  for (int which = 0; which < 3; which++) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CHECK(hipEventRecord(e0));
      if (which == 0) gemv_valu<<<grid, 256>>>(d_packed, T, d_out);
      else if (which == 1) gemv_mfma<<<grid, 256>>>(d_L, d_mu, d_w, T, d_out);
      else gemv_valu_quad<<<grid, 256>>>(d_packed, d_quad, T, d_out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CHECK(hipMemcpy(which == 0 ? ref.data() : (which == 1 ? got.data() : got2.data()), d_out, (size_t)grid * 256 * 4, hipMemcpyDeviceToHost));
    const double wave_steps = (double)grid * 4 * T;
    const double cyc = best * 1e-3 * 2.4e9 * (cus * 4) / wave_steps;     // SIMD-cycles at the nominal 2.4 GHz per wave-step
    printf("%-5s  %8.3f ms for %d workgroups x %d steps -> %7.0f SIMD-cycles@2.4GHz per wave-step (64 paths x 1 step), %.2f us per wave-step-round\n",
           which == 0 ? "VALU" : (which == 1 ? "MFMA" : "VALU4"), best, grid, T, cyc, best * 1e3 / T);
  }
  double maxrel = 0;
  for (size_t i = 0; i < ref.size(); i++) { const double r = fabs((double)got[i] / (double)ref[i] - 1.0); if (r > maxrel) maxrel = r; }
  printf("max relative difference of the final values MFMA vs VALU: %.2e (different summation order inside a 16x16x4 tile)\n", maxrel);
  size_t diff = 0;
  for (size_t i = 0; i < ref.size(); i++) diff += memcmp(&ref[i], &got2[i], 4) != 0;
  printf("VALU4 (quad rows) vs VALU (row pairs): %zu of %zu values differ\n", diff, ref.size());
  return 0;
}
