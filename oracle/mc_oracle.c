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
 * Everything here is IEEE-754 binary32 with round-to-nearest-even, explicit fmaf(), and
 * correctly rounded sqrtf(); compile with -ffp-contract=off so the compiler adds no fusions of
 * its own.  The HIP kernel executes the same operations in the same order, so terminal values
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

/* ---- Box-Muller pair, SPEC.md section 3: table-driven, exact arithmetic ------------------------
 * Two 1024-entry tables (on the GPU they live in LDS):
 *   SC[i] = (sin, cos)(2 pi (i + 1/2) / 1024)    built with the fixed fp32 polynomial below (bin midpoints)
 *   LG[j] = (inv_c, -2 ln(1/inv_c))               c = midpoint of mantissa bin j of [sqrt(.5), sqrt(2)),
 *                                                 the bin holding 1.0 uses c = 1 exactly
 * Table construction uses only IEEE +,*,/,fma (binary32 and binary64), so it is reproducible
 * bit for bit on any IEEE machine; the HIP side builds the same tables in a device init kernel.  */
#define NEG_2LN2 -0x1.62e43p+0f          /* -2 ln 2 rounded to binary32 */
#define TWO_PI_2M32 0x1.921fb6p-30f      /* 2 pi / 2^32 rounded to binary32 */
#define PI_1024 0x1.921fb6p-9f           /* 2^21 * TWO_PI_2M32 = pi/1024 (same significand) */
/* sin(a) = a + a^3 S(a^2), cos(a) = 1 - a^2/2 + a^4 C(a^2), |a| <= pi/4 (tools/fit_coeffs.py) */
#define SS0 -0x1.55554p-3f
#define SS1  0x1.1105b4p-7f
#define SS2 -0x1.98da62p-13f
#define CC0  0x1.55554ap-5f
#define CC1 -0x1.6c0c8cp-10f
#define CC2  0x1.9a0256p-16f

#define MCO_TAB 1024
static float g_sc[MCO_TAB][2];
static float g_lg[MCO_TAB][2];
static pthread_once_t g_tab_once = PTHREAD_ONCE_INIT;

/* (sin, cos)(2 pi xb / 2^32): exact integer quadrant reduction + degree-7/8 polynomials */
static void sincos_poly(uint32_t xb, float *sn_out, float *cs_out)
{
    uint32_t y = xb + 0x20000000u;
    int32_t r = (int32_t)(xb << 2) >> 2;                     /* xb - kq*2^30, in [-2^29, 2^29) */
    float a = (float)r * TWO_PI_2M32;
    float a2 = a * a;
    float ps = fmaf(a2, SS2, SS1); ps = fmaf(a2, ps, SS0);
    float sn = fmaf(a * a2, ps, a);
    float pc = fmaf(a2, CC2, CC1); pc = fmaf(a2, pc, CC0);
    float cs = fmaf(a2 * a2, pc, fmaf(a2, -0.5f, 1.0f));
    uint32_t kq = y >> 30;                                   /* theta = kq*pi/2 + a */
    float vs = (kq & 1u) ? cs : sn;
    float vc = (kq & 1u) ? sn : cs;
    if (kq & 2u) vs = -vs;                                   /* kq in {2,3} */
    if (kq == 1u || kq == 2u) vc = -vc;
    *sn_out = vs + 0.0f; *cs_out = vc + 0.0f;                /* -0 -> +0 */
}

/* ln(x) for x in [0.7, 1.42], binary64, atanh series: only IEEE +,*,/ (no libm) */
static double ln_series(double x)
{
    double y = (x - 1.0) / (x + 1.0), y2 = y * y, s = 0.0;
    for (int n = 17; n >= 0; n--) s = s * y2 + 1.0 / (double)(2 * n + 1);
    return 2.0 * y * s;
}

static void build_tables(void)
{
    for (uint32_t i = 0; i < MCO_TAB; i++) sincos_poly((i << 22) + 0x00200000u, &g_sc[i][0], &g_sc[i][1]);
    for (uint32_t j = 0; j < MCO_TAB; j++) {
        uint32_t lo = 0x3f3504f3u + (j << 13);
        float c = u32_as_f32(lo + 0x1000u);
        if (lo <= 0x3f800000u && 0x3f800000u < lo + 0x2000u) c = 1.0f;
        float inv_c = 1.0f / c;
        g_lg[j][0] = inv_c;
        g_lg[j][1] = (c == 1.0f) ? 0.0f : (float)(-2.0 * ln_series(1.0 / (double)inv_c));
    }
}

void mco_tables(float *sc /* [1024*2] */, float *lg /* [1024*2] */)
{
    pthread_once(&g_tab_once, build_tables);
    memcpy(sc, g_sc, sizeof g_sc);
    memcpy(lg, g_lg, sizeof g_lg);
}

/* sqrt: IEEE correctly rounded (sqrtss). */
MCO_INLINE void box_muller(uint32_t xa, uint32_t xb, float *z_sin, float *z_cos)
{
    /* radius: u in [2^-32, 1], u = 2^k m, m in [sqrt(.5), sqrt(2)), -2 ln u = k(-2 ln 2) + LG[j] - 2 log1p(r) */
    float u = fmaf((float)xa, 0x1p-32f, 0x1p-32f);
    uint32_t ib = f32_as_u32(u) - 0x3f3504f3u;
    int32_t k = (int32_t)ib >> 23;
    uint32_t mant = ib & 0x007fffffu;
    float m = u32_as_f32(mant + 0x3f3504f3u);
    uint32_t j = mant >> 13;
    float r = fmaf(m, g_lg[j][0], -1.0f);                    /* (m - c')/c', c' = 1/inv_c */
    float w = r * (r - 2.0f);                                /* -2 log1p(r) to O(r^3) */
    float t = fmaf((float)k, NEG_2LN2, g_lg[j][1]);
    t = t + w;
    float s = sqrtf(t);
    /* angle: theta = 2 pi xb / 2^32 = theta_i + d, theta_i = midpoint of table bin i = xb >> 22, |d| <= pi/1024 */
    uint32_t i = xb >> 22;
    float d = fmaf((float)(xb & 0x003fffffu), TWO_PI_2M32, -PI_1024);   /* (xb mod 2^22 - 2^21) 2 pi / 2^32 */
    float sc = g_sc[i][0], cc = g_sc[i][1];
    float cd = fmaf(d * -0.5f, d, 1.0f);                     /* cos d */
    float sn = fmaf(cc, d, sc * cd);                         /* sin(theta_i + d), sin d ~ d */
    float cs = fmaf(-sc, d, cc * cd);
    *z_sin = s * sn;
    *z_cos = s * cs;
}

/* normals of one path-step: z[m*nb + q] = normal m of Philox block q   (SPEC.md section 2) */
MCO_INLINE void step_normals(uint32_t k0, uint32_t k1, uint64_t path, uint32_t step, int nb, float *z)
{
    for (int q = 0; q < nb; q++) {
        uint64_t blk = (uint64_t)step * (uint32_t)nb + (uint32_t)q;
        uint32_t x[4];
        philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)path, (uint32_t)(path >> 32), k0, k1, x);
        box_muller(x[0], x[1], &z[0 * nb + q], &z[1 * nb + q]);
        box_muller(x[2], x[3], &z[2 * nb + q], &z[3 * nb + q]);
    }
}

/* ---- exported small pieces (unit-tested against known answers) ------------------------------ */
void mco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

void mco_box_muller(uint32_t xa, uint32_t xb, float *z_sin, float *z_cos)
{
    pthread_once(&g_tab_once, build_tables);
    box_muller(xa, xb, z_sin, z_cos);
}

/* n pairs at once (vector form for the accuracy tests) */
MCO_CLONES
void mco_box_muller_n(const uint32_t *xa, const uint32_t *xb, float *z_sin, float *z_cos, uint64_t n)
{
    pthread_once(&g_tab_once, build_tables);
    for (uint64_t i = 0; i < n; i++) box_muller(xa[i], xb[i], &z_sin[i], &z_cos[i]);
}

MCO_CLONES
void mco_step_normals(uint64_t seed, uint64_t path, uint32_t step, int n_assets, float *z /* [4*ceil(N/4)] */)
{
    int nb = (n_assets + 3) / 4;
    pthread_once(&g_tab_once, build_tables);
    step_normals((uint32_t)seed, (uint32_t)(seed >> 32), path, step, nb, z);
}

/* ---- the path loop --------------------------------------------------------------------------- */
typedef struct {
    int n_assets, n_steps, n_portfolios, compounding;   /* compounding: 0 simple, 1 log-sum */
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

    for (uint64_t p = j->p_lo; p < j->p_hi; p++) {
        const uint64_t path = j->path_begin + p;
        for (int k = 0; k < K; k++) V[k] = j->compounding ? 0.0f : j->v0;
        for (int t = 0; t < T; t++) {
            step_normals(k0, k1, path, (uint32_t)t, nb, z);
            for (int i = 0; i < N4; i++) {                  /* r = mu + L z, j ascending, fma */
                float acc = mu[i];
                for (int c = 0; c <= i; c++) acc = fmaf(L[i * N4 + c], z[c], acc);
                r[i] = acc;
            }
            for (int k = 0; k < K; k++) {                   /* rho = w . r, i ascending, fma */
                float rho = 0.0f;
                for (int i = 0; i < N4; i++) rho = fmaf(Wp[k * N4 + i], r[i], rho);
                if (j->compounding) V[k] = V[k] + rho;      /* S += rho */
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
    pthread_once(&g_tab_once, build_tables);
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
