/*
 * mc_oracle.c -- CPU restatement (ORACLE, test infrastructure only) of the Monte Carlo path
 * simulator frozen in SPEC.md ("MC-A").
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it.  The product (libmcport.so) never links it.
 *
 * PARITY STATUS: "parity unpinned" against reference *code* for the path simulator itself: the
 * reference (/root/reference/app.py) contains no Cholesky / normal-draw / path loop at all
 * (SURVEY.md section 0.2).  What the reference does pin, and what this file follows, are the
 * conventions:
 *   - portfolio return of a fixed-weight portfolio  rho = returns @ w      app.py:710
 *   - compounding  prod(1 + r) / cumprod(1 + r)                            app.py:249, app.py:253
 *   - mu / Sigma parameterisation  returns.mean(), returns.cov()           app.py:679-680
 * The integer RNG stream (Philox4x32-10) IS pinned: against the Random123 known-answer vectors
 * and against rocRAND's host-callable engine (/opt/rocm/include/rocrand/rocrand_philox4x32_10.h:
 * 270-296), see tests/test_oracle_rng.py.  The statistics (VaR/CVaR/Sharpe) are computed by
 * oracle/ref_stats.py, which follows app.py:258-263 and app.py:711 and is pinned by goldens
 * generated from the reference itself (tests/golden/).
 *
 * Everything here is IEEE-754 binary32 with round-to-nearest-even and explicit fmaf(); compile
 * with -ffp-contract=off so the compiler adds no fusions of its own.  The HIP kernel executes the same operations in the same order, so terminal values
 * are compared BIT-EXACTLY.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MCO_MAX_ASSETS 64

#if defined(__x86_64__) && defined(__GNUC__) && !defined(MCO_NO_CLONES)
#define MCO_CLONES __attribute__((target_clones("avx2,fma", "default")))
#else
#define MCO_CLONES
#endif
#define MCO_INLINE static inline __attribute__((always_inline))

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123).  Same constants / round function as
 *      rocRAND rocrand_philox4x32_10.h:62-65, 287-296. ---------------------------------------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

MCO_INLINE void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

MCO_INLINE float u32_as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
MCO_INLINE uint32_t f32_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ---- normal transform, SPEC.md section 3: inverse CDF, table-driven, exact arithmetic -------------------------
 * One 32-bit word -> one N(0,1) draw.  bit 31 = sign, low 31 bits v -> u = (v + 1/2) 2^-32 in [2^-33, 1/2];
 * u's binary32 exponent E (94..126) and top 5 mantissa bits select one of 33 x 32 table entries holding a cubic
 * in the remaining 18 mantissa bits; the cubic evaluates a(u) = -Phi^-1(u) >= 0 (Horner, fma).  On the GPU the
 * table (16.5 KiB) lives in LDS.  The coefficients are DATA of the spec (icdf_table.inc, generated once by
 * tools/fit_icdf_table.py from scipy.special.ndtri); the kernels carry their own identical copy.  */
#define MCO_ICDF_ENTRIES 1056
#define MCO_ICDF_E_LO 94
static const float g_icdf[MCO_ICDF_ENTRIES][4] = {
#include "icdf_table.inc"
};

MCO_INLINE float normal_icdf(uint32_t x)
{
    float u = fmaf((float)(x & 0x7fffffffu), 0x1p-32f, 0x1p-33f);
    uint32_t b = f32_as_u32(u) - ((uint32_t)MCO_ICDF_E_LO << 23);
    const float *c = g_icdf[b >> 18];
    float dc = u32_as_f32((b & 0x0003ffffu) | 0x3f800000u) - 0x1.04p+0f;      /* delta - 1/64, delta in [0, 1/32) */
    float a = fmaf(c[3], dc, c[2]);
    a = fmaf(a, dc, c[1]);
    a = fmaf(a, dc, c[0]);
    return u32_as_f32((f32_as_u32(a) & 0x7fffffffu) | (x & 0x80000000u));     /* |a| with the sign of bit 31 */
}

void mco_icdf_table(float *out /* [1056*4] */) { memcpy(out, g_icdf, sizeof g_icdf); }

/* normals of one path-step: z[m*nb + q] = normal m of Philox block q   (SPEC.md section 2) */
MCO_INLINE void step_normals(uint32_t k0, uint32_t k1, uint64_t path, uint32_t step, int nb, float *z)
{
    for (int q = 0; q < nb; q++) {
        uint64_t blk = (uint64_t)step * (uint32_t)nb + (uint32_t)q;
        uint32_t x[4];
        philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)path, (uint32_t)(path >> 32), k0, k1, x);
        for (int m = 0; m < 4; m++) z[m * nb + q] = normal_icdf(x[m]);
    }
}

/* ---- exported small pieces (unit-tested against known answers) ------------------------------ */
void mco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

/* n words at once (vector form for the accuracy tests) */
MCO_CLONES
void mco_normals_n(const uint32_t *x, float *z, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) z[i] = normal_icdf(x[i]);
}

MCO_CLONES
void mco_step_normals(uint64_t seed, uint64_t path, uint32_t step, int n_assets, float *z /* [4*ceil(N/4)] */)
{
    int nb = (n_assets + 3) / 4;
    step_normals((uint32_t)seed, (uint32_t)(seed >> 32), path, step, nb, z);
}

/* ---- the path loop --------------------------------------------------------------------------- */
typedef struct {
    int n_assets, n_steps, n_portfolios, compounding;   /* compounding: 0 simple, 1 log-sum; +2: folded (SPEC.md 4.1) */
    float v0;
    const float *mu, *chol, *W;
    uint64_t seed, path_begin, n_paths, p_lo, p_hi;
    float *terminal;                                     /* [K][n_paths] */
} mco_job;

MCO_CLONES
static void simulate_range(const mco_job *j)
{
    const int N = j->n_assets, K = j->n_portfolios, T = j->n_steps;
    const int nb = (N + 3) / 4, N4 = 4 * nb;
    const uint32_t k0 = (uint32_t)j->seed, k1 = (uint32_t)(j->seed >> 32);
    float L[MCO_MAX_ASSETS * MCO_MAX_ASSETS], mu[MCO_MAX_ASSETS];
    float z[MCO_MAX_ASSETS], r[MCO_MAX_ASSETS];
    float *Wp = (float *)calloc((size_t)K * N4, sizeof(float));
    float *V = (float *)malloc((size_t)K * sizeof(float));
    memset(L, 0, sizeof(float) * N4 * N4);
    memset(mu, 0, sizeof(float) * N4);
    for (int i = 0; i < N; i++) {
        mu[i] = j->mu[i] + 0.0f;                           /* -0 -> +0 */
        for (int c = 0; c <= i; c++) L[i * N4 + c] = j->chol[i * N + c];
    }
    for (int k = 0; k < K; k++)
        for (int i = 0; i < N; i++) Wp[k * N4 + i] = j->W[k * N + i];

    const int fold = (j->compounding & 2) != 0, logc = (j->compounding & 1) != 0;
    float fc = 0.0f, fv[MCO_MAX_ASSETS];
    memset(fv, 0, sizeof fv);
    if (fold) {                                            /* c = w.mu, v = L^T w in binary64, rounded once */
        double c = 0.0;
        for (int i = 0; i < N; i++) c += (double)j->W[i] * (double)mu[i];
        fc = (float)c;
        for (int col = 0; col < N; col++) {
            double v = 0.0;
            for (int i = col; i < N; i++) v += (double)j->W[i] * (double)j->chol[i * N + col];
            fv[col] = (float)v;
        }
    }
    for (uint64_t p = j->p_lo; p < j->p_hi; p++) {
        const uint64_t path = j->path_begin + p;
        for (int k = 0; k < K; k++) V[k] = logc ? 0.0f : j->v0;
        for (int t = 0; t < T; t++) {
            step_normals(k0, k1, path, (uint32_t)t, nb, z);
            if (fold) {                                     /* rho = c + v.z, j ascending, fma (one portfolio) */
                float rho = fc;
                for (int c = 0; c < N4; c++) rho = fmaf(fv[c], z[c], rho);
                if (logc) V[0] = V[0] + rho; else V[0] = fmaf(V[0], rho, V[0]);
                continue;
            }
            for (int i = 0; i < N4; i++) {                  /* r = mu + L z, j ascending, fma */
                float acc = mu[i];
                for (int c = 0; c <= i; c++) acc = fmaf(L[i * N4 + c], z[c], acc);
                r[i] = acc;
            }
            for (int k = 0; k < K; k++) {                   /* rho = w . r, i ascending, fma */
                float rho = 0.0f;
                for (int i = 0; i < N4; i++) rho = fmaf(Wp[k * N4 + i], r[i], rho);
                if (logc) V[k] = V[k] + rho;                /* S += rho */
                else V[k] = fmaf(V[k], rho, V[k]);          /* V *= (1 + rho) */
            }
        }
        for (int k = 0; k < K; k++) j->terminal[(size_t)k * j->n_paths + p] = V[k];
    }
    free(Wp); free(V);
}

static void *worker(void *arg) { simulate_range((const mco_job *)arg); return NULL; }

/* returns 0 on success, <0 on bad arguments */
int mco_simulate(int n_assets, int n_steps, int n_portfolios, int compounding, float v0,
                 const float *mu, const float *chol /* [N*N] row-major lower */,
                 const float *W /* [K*N] */, uint64_t seed, uint64_t path_begin, uint64_t n_paths,
                 float *terminal /* [K*n_paths] */, int n_threads)
{
    if (n_assets < 1 || n_assets > MCO_MAX_ASSETS || n_steps < 0 || n_portfolios < 1) return -1;
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n_paths) n_threads = n_paths ? (int)n_paths : 1;
    mco_job *jobs = (mco_job *)malloc(sizeof(mco_job) * n_threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    for (int i = 0; i < n_threads; i++) {
        mco_job jb = {n_assets, n_steps, n_portfolios, compounding, v0, mu, chol, W, seed, path_begin,
                      n_paths, n_paths * i / n_threads, n_paths * (i + 1) / n_threads, terminal};
        jobs[i] = jb;
    }
    for (int i = 1; i < n_threads; i++) pthread_create(&th[i], NULL, worker, &jobs[i]);
    simulate_range(&jobs[0]);
    for (int i = 1; i < n_threads; i++) pthread_join(th[i], NULL);
    free(jobs); free(th);
    return 0;
}

/* ---- float64 evaluation of the same spec ------------------------------------------------------------------
 * The reference's own arithmetic is binary64 throughout (NumPy/pandas defaults: app.py:258-263, 708-713).  This
 * mode answers "how far do T fp32 steps drift from what float64 NumPy would compute on IDENTICAL normals":
 * the Philox words and the fp32 inverse-CDF normals are exactly those of SPEC.md sections 2-3 (promoted to
 * double), the inputs mu / L / W are the same binary32 values (promoted), and the recurrence of SPEC.md
 * section 4 -- r = mu + L z, rho = w.r, V <- V(1+rho) -- runs in binary64 in the same order with fma().
 * Terminal values are doubles; they are NOT what the kernel is compared with bit for bit, they bound its
 * rounding error (tests/test_f64_drift.py, bench.py var_abs_err_f64). */
typedef struct {
    mco_job j;
    double *terminal64;
} mco_job64;

static void simulate_range_f64(const mco_job64 *jj)
{
    const mco_job *j = &jj->j;
    const int N = j->n_assets, K = j->n_portfolios, T = j->n_steps;
    const int nb = (N + 3) / 4, N4 = 4 * nb;
    const uint32_t k0 = (uint32_t)j->seed, k1 = (uint32_t)(j->seed >> 32);
    const int logc = (j->compounding & 1) != 0;
    float z[MCO_MAX_ASSETS];
    double r[MCO_MAX_ASSETS];
    double *V = (double *)malloc((size_t)K * sizeof(double));
    for (uint64_t p = j->p_lo; p < j->p_hi; p++) {
        const uint64_t path = j->path_begin + p;
        for (int k = 0; k < K; k++) V[k] = logc ? 0.0 : (double)j->v0;
        for (int t = 0; t < T; t++) {
            step_normals(k0, k1, path, (uint32_t)t, nb, z);
            for (int i = 0; i < N; i++) {
                double acc = (double)(j->mu[i] + 0.0f);
                for (int c = 0; c <= i; c++) acc = fma((double)j->chol[i * N + c], (double)z[c], acc);
                r[i] = acc;
            }
            for (int k = 0; k < K; k++) {
                double rho = 0.0;
                for (int i = 0; i < N; i++) rho = fma((double)j->W[k * N + i], r[i], rho);
                if (logc) V[k] = V[k] + rho;
                else V[k] = fma(V[k], rho, V[k]);
            }
        }
        for (int k = 0; k < K; k++) jj->terminal64[(size_t)k * j->n_paths + p] = V[k];
    }
    (void)N4;
    free(V);
}

static void *worker64(void *arg) { simulate_range_f64((const mco_job64 *)arg); return NULL; }

int mco_simulate_f64(int n_assets, int n_steps, int n_portfolios, int compounding, float v0,
                     const float *mu, const float *chol, const float *W, uint64_t seed, uint64_t path_begin,
                     uint64_t n_paths, double *terminal /* [K*n_paths] */, int n_threads)
{
    if (n_assets < 1 || n_assets > MCO_MAX_ASSETS || n_steps < 0 || n_portfolios < 1 || (compounding & ~1)) return -1;
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n_paths) n_threads = n_paths ? (int)n_paths : 1;
    mco_job64 *jobs = (mco_job64 *)malloc(sizeof(mco_job64) * n_threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    for (int i = 0; i < n_threads; i++) {
        mco_job jb = {n_assets, n_steps, n_portfolios, compounding, v0, mu, chol, W, seed, path_begin,
                      n_paths, n_paths * i / n_threads, n_paths * (i + 1) / n_threads, NULL};
        jobs[i].j = jb;
        jobs[i].terminal64 = terminal;
    }
    for (int i = 1; i < n_threads; i++) pthread_create(&th[i], NULL, worker64, &jobs[i]);
    simulate_range_f64(&jobs[0]);
    for (int i = 1; i < n_threads; i++) pthread_join(th[i], NULL);
    free(jobs); free(th);
    return 0;
}
