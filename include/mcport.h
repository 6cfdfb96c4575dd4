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

#define MCP_ABI_VERSION 3
#define MCP_MAX_ASSETS 64        /* thread-per-path kernels are instantiated for N4 = 4..64 */
#define MCP_SELECT_BINS 2048     /* radix-select digit: 11 + 11 + 10 bits */

enum {
    MCP_OK = 0,
    MCP_E_ARG = -1,        /* bad argument (shape, NULL, range) */
    MCP_E_NODEVICE = -2,   /* no HIP device visible (the product has no CPU fallback) */
    MCP_E_NOMEM = -3,
    MCP_E_UNSUPPORTED = -4,
    MCP_E_HIP = -5,        /* a HIP runtime call or kernel launch failed; mcp_last_error() has the call and the HIP error string */
    MCP_E_COMM = -6        /* RCCL could not be loaded or a collective failed (multi-device contexts only) */
};

enum {
    MCP_COMPOUND_SIMPLE = 0,   /* V <- V*(1+rho): np.cumprod(1+r) idiom, app.py:249, app.py:253 */
    MCP_COMPOUND_LOG = 1       /* S <- S+rho, x = expm1(S) */
};

enum {
    MCP_FLAG_NATIVE_MATH = 1,  /* normals by Box-Muller on the v_log/v_sqrt/v_sin/v_cos hardware approximations instead
                                  of the spec's inverse-CDF table: statistically equivalent draws from the same Philox
                                  stream, NOT comparable to the oracle value by value */
    MCP_FLAG_FOLD = 2,         /* one portfolio only: rho = w.mu + (L^T w).z with L^T w folded on the host (SPEC.md 4.1)
                                  instead of the triangular GEMV + weight dot.  A separately reported fast path: same
                                  normals, other rounding than the unfolded recurrence (agrees to ~1e-7 relative) */
    MCP_FLAG_SHARD_PORTFOLIOS = 4  /* multi-device contexts: every device walks ALL paths for its slice of the K weight
                                  vectors (BASELINE configs[4]; no collective at all) instead of sharding the path range */
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

/* Raw sufficient statistics of one portfolio on one device: what ranks exchange (one all-gather, merged in rank
 * order: SUM on n, sum, sumsq, below; MIN on min; MAX on max).  SHIFTED sums: sum = sum (x - pivot),
 * sumsq = sum (x - pivot)^2 with the same pivot on every rank (mcp_pivots: the analytic mean of x), so that
 * mean = pivot + sum/n and sum (x - mean)^2 = sumsq - sum^2/n do not cancel for a low-volatility portfolio
 * (np.std(ddof=1), app.py:234, is two-pass).  `below` = this device's paths that sort strictly below the bucket of the low
 * order statistic of the radix select (the CVaR tail, app.py:261-263), as sum (V - v0) (simple compounding;
 * sum x = below / v0) or sum x (log). */
typedef struct {
    double n, sum, sumsq, min, max, below;
    double pivot;
    double pad;
} mcp_record;

int mcp_abi_version(void);
int mcp_device_count(void);                 /* 0 when no GPU is visible; never fails */
const char *mcp_last_error(void);

/* ---- host-level API: NumPy in, NumPy out (replaces the script lines app.py:699-717 for a simulated
 *      terminal distribution).  Owns its device buffers; calls on one ctx are serialised. ---------- */
int mcp_ctx_create(int device, mcp_ctx **out);
/* one context over `ndev` devices, one stream per device, the path range (or, with
 * MCP_FLAG_SHARD_PORTFOLIOS, the weight matrix) sharded over them.  Distinct devices exchange through RCCL
 * (ncclCommInitAll; librccl is loaded at run time).  If librccl cannot be loaded or initialised -- or MCP_EXCHANGE=p2p is
 * set -- and the first device has peer access to the others (at most 8 devices), the exchange runs as a kernel of the
 * first device over peer access instead; MCP_E_COMM if neither is possible.  A device listed more than once holds
 * several logical shards that exchange through that same kernel (what a one-GPU box can exercise).  ndev = 1 is
 * mcp_ctx_create.  Results equal the one-device results: order statistics, counts and argmax exactly, fp64 sums up to
 * association. */
int mcp_ctx_create_multi(const int *devices, int ndev, mcp_ctx **out);
int mcp_ctx_device_count(const mcp_ctx *ctx);   /* number of shards of the context */
/* How the shards of the context exchange histograms and records.  The communicator (or the peer mapping) is set up on the
 * first path-sharded mcp_simulate, not at creation: a portfolio-sharded call (MCP_FLAG_SHARD_PORTFOLIOS) needs neither.
 * MCP_EXCHANGE_UNSET until then.  When RCCL could not be used and the context fell back to the peer-access kernel,
 * mcp_ctx_exchange_note() says why (empty string otherwise). */
enum {
    MCP_EXCHANGE_UNSET = 0,    /* nothing exchanged yet */
    MCP_EXCHANGE_NONE = 1,     /* one shard */
    MCP_EXCHANGE_RCCL = 2,     /* distinct devices, ncclAllReduce / ncclAllGather over xGMI */
    MCP_EXCHANGE_KERNEL = 3,   /* logical shards of ONE device: a kernel sums the shards' buffers */
    MCP_EXCHANGE_P2P = 4       /* distinct devices, RCCL unavailable (or MCP_EXCHANGE=p2p): the same kernel over peer access */
};
int mcp_ctx_exchange_mode(const mcp_ctx *ctx);
const char *mcp_ctx_exchange_note(const mcp_ctx *ctx);
void mcp_ctx_destroy(mcp_ctx *ctx);

/* Large K: the terminal values are produced and reduced in tiles of portfolios so that at most about
 * `bytes` of V_T are resident per device (default 8 GiB; SURVEY.md section 8a N2: V_T[K x paths] is never
 * materialised whole unless terminal_out asks for it). */
int mcp_ctx_set_terminal_budget(mcp_ctx *ctx, size_t bytes);

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
 *      are opaque device memory of the byte sizes given by mcp_ws_bytes(); MCP_WS_HIST must be ZERO when first
 *      used (the steps clear what they consume).  One pass, in this order:
 *          paths (fused: V_T, moment partials, digit-0 histogram)
 *                -> [all-reduce HIST] -> scan(0) -> hist(1) -> [all-reduce HIST] -> scan(1)
 *                -> hist(2) -> [all-reduce HIST] -> final -> [all-gather RECORD -> stats]
 *      A single-GPU host skips the bracketed exchanges and passes d_stats to mcp_launch_final.  A host that has terminal
 *      values of its own replaces `paths` by mcp_launch_pass0. -------- */

enum {
    MCP_WS_PARTIALS = 0,   /* [K][mcp_moment_slots] x 32 B {sum (x-c), sum (x-c)^2, float min V, max V, u64 n}: one per
                              workgroup (K <= 16) or per 64-path wave tile (K >= 17) of the path kernels          */
    MCP_WS_RECORD = 1,     /* [K] mcp_record: this device's sufficient statistics (all-gathered)                  */
    MCP_WS_STATE = 2,      /* [K][2] select state {u32 prefix, u32 pad, u64 rank}                                 */
    MCP_WS_HIST = 3,       /* [K][2][MCP_SELECT_BINS] uint64: all-reduce SUM after paths / hist                   */
    MCP_WS_QUANT = 4,      /* [K] {double x_lo, x_hi, var, level2; u64 n_tail, pad}: identical on all ranks       */
    MCP_WS_STATS = 5,      /* [K] mcp_stats                                                                       */
    MCP_WS_BELOW = 6,      /* [K][slots(K)] double: per-block tail partials of hist(1) / hist(2)                  */
    MCP_WS_PIVOT = 7,      /* [K] double: the shift of the moments (host: mcp_pivots, then copy to the device)    */
    MCP_WS_COUNT = 8
};
/* bytes of work buffer `which` for K portfolios and n_paths paths on this device (only MCP_WS_PARTIALS depends on n_paths) */
size_t mcp_ws_bytes(int which, int n_portfolios, uint64_t n_paths);
/* MomentPartial slots per portfolio a pass over n_paths fills (informative; mcp_ws_bytes uses it) */
uint64_t mcp_moment_slots(int n_portfolios, uint64_t n_paths);

/* Number of floats of the packed parameter block for (N, K). */
size_t mcp_packed_len(int n_assets, int n_portfolios);
/* Pack mu, lower(chol), W (and portfolio 0's fold block) into the padded device layout (host side, no GPU needed). */
int mcp_pack_params(int n_assets, int n_portfolios, const float *mu, const float *chol, const float *W,
                    float *packed_out, size_t packed_len);
/* The shift of the moments, one per portfolio (host side, binary64 from the binary32 inputs): the analytic mean of x --
 * w_k.mu is the per-step `port_return` of app.py:708, |L^T w_k|^2 the per-step `port_std`^2 of app.py:709 --
 *   simple: c_k = (1 + w_k.mu)^T - 1          log: c_k = expm1(T (w_k.mu + |L^T w_k|^2 / 2)).
 * A function of the inputs only, hence identical on every rank; 0 where it is not finite. */
int mcp_pivots(const mcp_params *prm, const float *mu, const float *chol, const float *W, double *pivots_out /* [K] */);

/* Simulate paths [path_begin, path_begin+n_paths) of all K portfolios and store the terminal values:
 * d_terminal is [K][terminal_stride] floats (terminal_stride >= n_paths), 4 B per path and portfolio.
 * With d_partials and d_hist (both or neither) the kernels' epilogue also reduces the paths while V is in registers:
 * moment partials around d_pivot ([K] doubles, NULL = 0) into d_partials and the digit-0 histogram of the radix select
 * (key bits 31..21) into d_hist -- what mcp_launch_scan(pass 0) consumes.  (K >= 17: the histogram is one lean read of
 * V_T enqueued behind the MFMA kernel, whose workgroups hold 512 portfolios.) */
int mcp_launch_paths(const mcp_params *prm, const float *d_packed, const double *d_pivot, uint64_t seed, uint64_t path_begin,
                     uint64_t n_paths, float *d_terminal, uint64_t terminal_stride, void *d_partials, void *d_hist,
                     void *stream);

/* np.percentile(x, (1-alpha)*100) bookkeeping (numpy 2.2 `_compute_virtual_index`/`_get_indexes`,
 * method 'linear'; the q of app.py:259): ranks of the two order statistics and the weight. */
int mcp_percentile_rank(uint64_t n_total, double alpha, uint64_t *rank_lo, uint64_t *rank_hi, double *gamma);

/* Standalone pass 0 over CALLER-SUPPLIED terminal values (n per portfolio): the same moment partials and digit-0
 * histogram the fused epilogue of mcp_launch_paths leaves.  d_pivot: [K] doubles or NULL (= 0: raw sums). */
int mcp_launch_pass0(const mcp_params *prm, const float *d_terminal, uint64_t terminal_stride, uint64_t n,
                     const double *d_pivot, void *d_partials, void *d_hist, void *stream);
/* pass 0: partials -> d_record (with d_pivot, NULL = 0), (rank_lo, rank_hi) of the GLOBAL n -> d_state, descend into the
 * digit holding each rank; pass 1: descend again (and fold the tail partials of hist pass 1 into d_record).  Clears d_hist. */
int mcp_launch_scan(const mcp_params *prm, int pass, uint64_t n, uint64_t rank_lo, uint64_t rank_hi, const void *d_partials,
                    const void *d_below, const double *d_pivot, void *d_hist, void *d_state, void *d_record, void *stream);
/* pass 1 (key bits 20..10) / pass 2 (bits 9..0): digit histograms of the keys matching the prefixes in d_state, and the
 * tail sum below the low bucket (CVaR tail, app.py:261-263) into d_below.  pass 0: the digit-0 histogram alone (d_state
 * unused; d_pivot, if given, centres the kernel's counting window). */
int mcp_launch_hist(const mcp_params *prm, int pass, const float *d_terminal, uint64_t terminal_stride, uint64_t n,
                    const void *d_state, const double *d_pivot, void *d_below, void *d_hist, void *stream);
/* Last descent -> order statistics -> VaR (numpy `_lerp`) -> d_quant; tail count / in-bucket tail sum from the
 * (global) histogram; local `below` into d_record.  d_stats != NULL (single device): also mean, std (ddof=1), Sharpe,
 * CVaR -> d_stats [K] mcp_stats.  Clears d_hist. */
int mcp_launch_final(const mcp_params *prm, uint64_t n, double gamma, uint64_t rank_lo, uint64_t rank_hi,
                     const void *d_below, void *d_hist, const void *d_state, void *d_record, void *d_quant,
                     void *d_stats, void *stream);
/* Multi-GPU: merge the all-gathered records of `world` ranks, d_gathered [world][K] mcp_record in rank order, and
 * finish -> d_stats [K] mcp_stats. */
int mcp_launch_stats(const mcp_params *prm, int world, const void *d_gathered, const void *d_quant, void *d_stats,
                     void *stream);
/* Exchange between logical shards that live in ONE process (several shards of one device, or devices with peer access):
 * every one of the `n_bufs` (<= 8) device buffers <- their element-wise sum (u64 words).  The kernel form of the histogram
 * all-reduce; the caller orders it after the producers and before the consumers of every buffer (events). */
int mcp_launch_sum_u64(void *const *d_bufs, int n_bufs, size_t words, void *stream);

/* A stream of `device` whose kernels may run on all but `reserve_cus` compute units (hipExtStreamCreateWithCUMask; the
 * reserved ones are the highest-numbered CUs of the mask).  For hosts that pipeline batches: path kernels on such streams
 * leave a few CUs to the small statistics / exchange kernels of the batch before, which otherwise queue for wave slots
 * beside a kernel that fills the chip.  reserve_cus = 0: an ordinary non-blocking stream. */
int mcp_stream_create(int device, int reserve_cus, void **stream_out);
int mcp_stream_destroy(void *stream);

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
