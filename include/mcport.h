/*
 * mcport.h -- C ABI of libmcport.so, the MI355X (gfx950) Monte Carlo portfolio path engine.
 *
 * The reference (mohammadmarghzari/monte-carlo-portfolio, app.py) has no FFI / plugin interface: it
 * is a flat Streamlit script.  This ABI is therefore the boundary SURVEY.md section 8(b) defines; each
 * entry point names the reference lines whose role it takes over.  Bound from Python with ctypes
 * (monte_carlo_portfolio_amd/_ffi.py); INTEGRATION.md shows the stub a maintainer of app.py would add.
 *
 * Conventions: every function returns 0 on success or a negative MCP_E_* code and never throws;
 * mcp_last_error() returns a thread-local message.  Host pointers are caller-owned and only need to
 * stay valid for the call.  "d_" pointers are DEVICE pointers (hipMalloc / torch CUDA tensors) and
 * "stream" is a hipStream_t passed as void* (NULL = the default stream); the *_launch_* functions only
 * enqueue work on that stream, they never allocate, synchronise or copy to the host, so they may be
 * captured in a hipGraph.
 */
#ifndef MCPORT_H
#define MCPORT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCP_ABI_VERSION 1
#define MCP_MAX_ASSETS 64        /* thread-per-path kernels are instantiated for N4 = 4..64 */
#define MCP_SELECT_BINS 2048     /* radix-select digit: 11 + 11 + 10 bits */

enum {
    MCP_OK = 0,
    MCP_E_ARG = -1,        /* bad argument (shape, NULL, range) */
    MCP_E_NODEVICE = -2,   /* no HIP device / HIP runtime error; message has the HIP error string */
    MCP_E_NOMEM = -3,
    MCP_E_UNSUPPORTED = -4
};

enum {
    MCP_COMPOUND_SIMPLE = 0,   /* V <- V*(1+rho): np.cumprod(1+r) idiom, app.py:249, app.py:253 */
    MCP_COMPOUND_LOG = 1       /* S <- S+rho, x = expm1(S) */
};

enum {
    MCP_FLAG_NATIVE_MATH = 1,  /* normals by Box-Muller on the v_log/v_sqrt/v_sin/v_cos hardware approximations instead
                                  of the spec's inverse-CDF table: statistically equivalent draws from the same Philox
                                  stream, NOT comparable to the oracle value by value */
    MCP_FLAG_FOLD = 2          /* one portfolio only: rho = w.mu + (L^T w).z with L^T w folded on the host (SPEC.md 4.1)
                                  instead of the triangular GEMV + weight dot.  A separately reported fast path: same
                                  normals, other rounding than the unfolded recurrence (agrees to ~1e-7 relative) */
};

typedef struct mcp_ctx mcp_ctx;

/* Problem description.  mu/Sigma play the role of app.py:679-680 (per STEP, i.e. already divided by
 * the annualisation factor), W rows are portfolios as drawn at app.py:702. */
typedef struct {
    int32_t n_assets;       /* N in [1, MCP_MAX_ASSETS] */
    int32_t n_steps;        /* T >= 0 */
    int32_t n_portfolios;   /* K >= 1; all K portfolios see the same normals (common random numbers) */
    int32_t compounding;    /* MCP_COMPOUND_* */
    int32_t flags;          /* MCP_FLAG_* */
    int32_t reserved;
    double v0;              /* initial value; rounded to binary32 on the device */
    double alpha;           /* VaR/CVaR confidence, reference default 0.95 (app.py:258, app.py:684) */
    double rf;              /* risk-free rate over the horizon, subtracted as at app.py:711 */
} mcp_params;

/* Per-portfolio result on x = V_T/V0 - 1 (or expm1(S_T)), semantics of app.py:258-263, app.py:711. */
typedef struct {
    uint64_t n;             /* number of paths */
    uint64_t n_tail;        /* #{x <= VaR} */
    double mean;
    double m2;              /* sum (x - mean)^2 */
    double std;             /* ddof = 1, as np.std(ddof=1) at app.py:234 */
    double sharpe;          /* (mean - rf)/std, 0 if std == 0 (app.py:711) */
    double var;             /* np.percentile(x, (1-alpha)*100), linear interpolation (app.py:259) */
    double cvar;            /* mean of x[x <= var], var if empty (app.py:263) */
    double min, max;
    double sum_tail;        /* sum of x[x <= var] */
    double x_lo, x_hi;      /* the two order statistics np.percentile interpolates between */
} mcp_stats;

/* Raw sufficient statistics of one portfolio on one device: what ranks all-reduce (SUM on the first
 * three, MIN / MAX on the last two). */
typedef struct {
    double n, sum, sumsq, min, max;
} mcp_moments;

int mcp_abi_version(void);
int mcp_device_count(void);                 /* 0 when no GPU is visible; never fails */
const char *mcp_last_error(void);

/* ---- host-level API: NumPy in, NumPy out (replaces the script lines app.py:699-717 for a simulated
 *      terminal distribution).  Owns its device buffers; calls on one ctx are serialised. ---------- */
int mcp_ctx_create(int device, mcp_ctx **out);
void mcp_ctx_destroy(mcp_ctx *ctx);

int mcp_simulate(mcp_ctx *ctx, const mcp_params *prm,
                 const float *mu,      /* [N] */
                 const float *chol,    /* [N*N] row-major, lower triangular (upper ignored) */
                 const float *W,       /* [K*N] */
                 uint64_t seed, uint64_t path_begin, uint64_t n_paths,
                 float *terminal_out,  /* NULL or host [K*n_paths] */
                 mcp_stats *stats_out  /* [K] */);

/* The reference's own sweep (app.py:699-717) over HISTORICAL returns, loop body app.py:708-713 for P weight
 * vectors at once, binary64 like the reference.  returns: [R*N] row-major (returns_df.values, app.py:667),
 * mean/cov: the annualised mean_returns / cov_matrix of app.py:679-680, W: [P*N] (rows as drawn at
 * app.py:702), rf in the reference's units (user_rf, app.py:711), alpha = cvar_alpha (app.py:684).
 * Outputs are [P] each.  Limits: N <= MCP_MAX_ASSETS, R <= MCP_SWEEP_MAX_ROWS. */
#define MCP_SWEEP_MAX_ROWS 4096
int mcp_sweep_historical(mcp_ctx *ctx, int n_assets, int n_rows, int n_portfolios, const double *returns,
                         const double *mean, const double *cov, const double *W, double rf, double alpha,
                         double *port_return, double *port_std, double *sharpe, double *var, double *cvar);

/* ---- device-level API: the same kernels as separate enqueue-only steps, for a host that owns the
 *      buffers and the collectives (one process per GPU, torch.distributed over RCCL).  Work buffers
 *      are opaque device memory of the byte sizes given by mcp_ws_bytes().  Between a *_hist step and
 *      its *_scan step (and after mcp_launch_moments / mcp_launch_tail) a multi-GPU host all-reduces
 *      the buffer in place; a single-GPU host just runs the steps back to back. -------------------- */

enum {
    MCP_WS_PARTIALS = 0,   /* [K][256] mcp_moments: per-block partials of the moments pass          */
    MCP_WS_MOMENTS = 1,    /* [K] mcp_moments: all-reduce SUM on {n,sum,sumsq}, MIN on min, MAX on max */
    MCP_WS_STATE = 2,      /* [K][2] select state {u32 prefix, u32 pad, u64 rank}                   */
    MCP_WS_HIST = 3,       /* [K][2][MCP_SELECT_BINS] uint64: all-reduce SUM                        */
    MCP_WS_QUANT = 4,      /* [K] {double x_lo, x_hi, var}                                          */
    MCP_WS_TAIL_PARTIAL = 5,
    MCP_WS_TAIL = 6,       /* [K] {double count, double sum}: all-reduce SUM                        */
    MCP_WS_STATS = 7       /* [K] mcp_stats                                                         */
};
size_t mcp_ws_bytes(int which, int n_portfolios);

/* Number of floats of the packed parameter block for (N, K). */
size_t mcp_packed_len(int n_assets, int n_portfolios);
/* Pack mu, lower(chol), W (and portfolio 0's fold block) into the padded device layout (host side, no GPU needed). */
int mcp_pack_params(int n_assets, int n_portfolios, const float *mu, const float *chol, const float *W,
                    float *packed_out, size_t packed_len);

/* Simulate paths [path_begin, path_begin+n_paths) of all K portfolios and store the terminal values:
 * d_terminal is [K][terminal_stride] floats (terminal_stride >= n_paths), 4 B per path and portfolio. */
int mcp_launch_paths(const mcp_params *prm, const float *d_packed, uint64_t seed, uint64_t path_begin,
                     uint64_t n_paths, float *d_terminal, uint64_t terminal_stride, void *stream);

/* {n, sum x, sum x^2, min, max} of this device's n terminal values per portfolio -> d_moments [K]
 * (fixed-order two-stage fp64 reduction: run-to-run deterministic). */
int mcp_launch_moments(const mcp_params *prm, const float *d_terminal, uint64_t terminal_stride, uint64_t n,
                       void *d_partials, void *d_moments, void *stream);

/* Multi-GPU: merge the all-gathered moment records of `world` ranks, d_gathered [world][K] mcp_moments in rank order
 * (SUM on n, sum, sumsq; MIN on min; MAX on max) into d_moments [K]. */
int mcp_launch_moments_merge(int n_portfolios, int world, const void *d_gathered, void *d_moments, void *stream);

/* np.percentile(x, (1-alpha)*100) bookkeeping (numpy 2.2 `_compute_virtual_index`/`_get_indexes`,
 * method 'linear'; the q of app.py:259): ranks of the two order statistics and the weight. */
int mcp_percentile_rank(uint64_t n_total, double alpha, uint64_t *rank_lo, uint64_t *rank_hi, double *gamma);

/* Exact order statistics by radix select on the order-preserving key of the float bits.
 * pass 0: key[31:21], pass 1: key[20:10], pass 2: key[9:0].  Per pass: hist (zeroes d_hist, then
 * counts this device's keys that match the prefix found so far) -> [all-reduce] -> scan (descends). */
int mcp_launch_select_init(int n_portfolios, uint64_t rank_lo, uint64_t rank_hi, void *d_state, void *stream);
int mcp_launch_select_hist(int n_portfolios, const float *d_terminal, uint64_t terminal_stride, uint64_t n,
                           int pass, const void *d_state, void *d_hist, void *stream);
int mcp_launch_select_scan(int n_portfolios, int pass, const void *d_hist, void *d_state, void *stream);

/* VaR from the selected order statistics (numpy `_lerp`). */
int mcp_launch_quantile(const mcp_params *prm, double gamma, const void *d_state, void *d_quant, void *stream);

/* count and sum of x over this device's paths with x <= VaR (compared in double as app.py:263). */
int mcp_launch_tail(const mcp_params *prm, const float *d_terminal, uint64_t terminal_stride, uint64_t n,
                    const void *d_quant, void *d_tail_partial, void *d_tail, void *stream);

/* mean, std (ddof=1), Sharpe, VaR, CVaR -> d_stats [K] mcp_stats. */
int mcp_launch_stats(const mcp_params *prm, const void *d_moments, const void *d_quant, const void *d_tail,
                     void *d_stats, void *stream);

/* The normal generator on its own: d_z[i] = inverse-CDF normal (SPEC.md section 3) of the 32-bit word d_x[i]. */
int mcp_launch_normals(const uint32_t *d_x, uint64_t n, float *d_z, void *stream);

/* The inverse-CDF coefficient table the kernels use (1056 x 4 floats; pure CPU): lets a test compare it with the
 * oracle's copy. */
int mcp_icdf_table(float *out, size_t out_len);

/* Host helpers shared by both levels (pure CPU). */
uint32_t mcp_float_to_key(float v);
float mcp_key_to_float(uint32_t key);
/* x from a terminal value, in double: V/fl32(v0) - 1, or expm1(S). */
double mcp_terminal_to_x(const mcp_params *prm, float terminal);

#ifdef __cplusplus
}
#endif
#endif /* MCPORT_H */
