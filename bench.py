#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: simulated paths/s at 16 assets x 252 steps (+ VaR
abs-err vs the oracle), one process per GPU.

  python bench.py [--gpus N --steps K --warmup W]        N > 1: this process touches no GPU; it starts N rank processes
                                                         (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their env), relays rank 0's
                                                         JSON line and exits non-zero if any rank does
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          the same ranks under an external launcher

A "step" is one full pass of the hot path over one batch resident in HBM: fused path kernel
(Philox -> inverse-CDF normals -> Cholesky GEMV -> compounding) for BASELINE configs[1] per GPU (16 synthetic
assets, 1,000,000 paths, 252 steps, fp32), then moments, exact VaR (3-pass radix select) and CVaR,
with the cross-rank exchanges of SURVEY.md section 8(e) when N > 1.  Weak scaling: every rank simulates its
own 1M-path shard of one global path range.  Steps are independent batches; PathEngine double-buffers them
over HIP streams (the statistics passes and collectives of batch i overlap the path kernel of batch i+1), and
the timed region ends with a barrier + device synchronise, so every batch is complete inside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL; must be set before HIP initialises

import numpy as np  # noqa: E402

N_ASSETS, N_STEPS, PATHS_PER_GPU = 16, 252, 1_000_000
MODEL_FLOPS_PER_PATH = N_STEPS * (N_ASSETS * N_ASSETS + 3 * N_ASSETS + 2)     # 77,112
HBM_BYTES_PER_PATH = 4                                                          # V_T store
# VALU-issue ceiling: the cheapest instruction mix that implements north_star's
# algorithm (Philox4x32-10 -> Box-Muller on hardware transcendentals -> triangular GEMV -> weight dot ->
# compounding), per wave and path-step at N = 16, priced at the issue costs measured on this chip at 8
# waves/SIMD (profiles/r01_valu_rates.txt):
#   Philox   80 v_mad_u64_u32 x 4.7 + 80 three-input xor x 2.6                            =  584
#   8 pairs  2 cvt x 4.4 + 4 transcendentals x 8.2 + 5 packed fp ops x 2.15               =  419
#   GEMV     136 FMA as 68 v_pk_fma_f32 x 4.3; weight dot 8 v_pk_fma_f32; compound 1 x 2.6 =  329
ISSUE_MODEL = json.load(open(os.path.join(ROOT, "profiles", "issue_model.json")))     # the ceiling's instruction table, as data
ISSUE_CYCLES_PER_WAVE_STEP = sum(r["count"] * r["cycles"] for r in ISSUE_MODEL["rows"])
VALU_CEILING_PATHS_PER_S = 1024 * 2.4e9 * 64 / (ISSUE_CYCLES_PER_WAVE_STEP * N_STEPS)
HBM_PEAK_GBS = 8000.0
FP32_VECTOR_PEAK_TFLOPS = 157.3


def physical_cores():
    """Physical cores this process may run on (lscpu-style: unique (package, core) pairs of the allowed CPUs)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            seen.add(("?", str(c)))
    return len(seen), len(cpus)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` typed as such: the parent makes NO GPU call (it never imports torch); it starts N fresh
    rank processes, relays rank 0's stdout (the JSON line) and returns non-zero if any rank failed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCP_BENCH_RANK_PROCESS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    alive = set(range(n))
    while alive:
        time.sleep(0.1)
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:                         # a dead rank leaves the others waiting in a collective
                    procs[q].terminate()                # exact PIDs this function started
    reader.join(timeout=10)
    sys.stdout.write(b"".join(buf).decode("utf-8", "replace"))
    sys.stdout.flush()
    return rc


def cpu_baseline(mu32, L, W32, seed):
    """The oracle (CPU port of the same spec) on this box's host cores, bounded to ~10-20 s."""
    from oracle import mc_oracle
    cores, logical = physical_cores()
    threads = min(logical, 64)
    t0 = time.perf_counter()
    mc_oracle.simulate(mu32, L, W32, N_STEPS, 4096, seed, n_threads=threads)
    rate = 4096 / (time.perf_counter() - t0)
    n = int(min(max(rate * 12.0, 8192), 2_000_000))
    t0 = time.perf_counter()
    term = mc_oracle.simulate(mu32, L, W32, N_STEPS, n, seed, n_threads=threads)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "paths/s", "cores": threads, "physical_cores": cores, "logical_cpus": logical, "kind": "port",
            "sample": f"{n} paths of the bench workload (16 assets x 252 steps, seed 0x5EED5EED), oracle/mc_oracle.c, "
                      f"{threads} threads on {cores} physical cores"}, term


def mc_oracle_f64(mu32, L, W32, seed, n):
    from oracle import mc_oracle
    return mc_oracle.simulate_f64(mu32, L, W32, N_STEPS, n, seed)[0]


def numpy_baseline(mu32, L, W32, seed, budget_s=8.0):
    """the NumPy CPU loop of the same model (oracle/np_oracle.py, float64 = the reference's
    arithmetic: Z @ L.T, returns @ w, running product), single process, bounded."""
    from oracle import np_oracle
    t0 = time.perf_counter()
    np_oracle.simulate(mu32, L, W32, N_STEPS, 2048, seed, dtype=np.float64)
    rate = 2048 / (time.perf_counter() - t0)
    n = int(min(max(rate * budget_s, 4096), 200_000))
    t0 = time.perf_counter()
    np_oracle.simulate(mu32, L, W32, N_STEPS, n, seed, dtype=np.float64)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "paths/s", "cores": 1, "kind": "port", "dtype": "f64",
            "sample": f"{n} paths of the bench workload, oracle/np_oracle.py (pure NumPy, float64, one process; BLAS threads as configured)"}



def rehearsal(args, world, rank):
    """CPU rehearsal of the multi-rank bench (launcher + PathEngine choreography over gloo with the NumPy/oracle test
    double).  Not a measurement: the line says so."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_kernels import FakeKernels
    from monte_carlo_portfolio_amd import synthetic
    from monte_carlo_portfolio_amd.engine import PathEngine
    from monte_carlo_portfolio_amd.simulate import prepare_inputs
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local = min(PATHS_PER_GPU, 2000)
    mu, cov = synthetic.synthetic_market(N_ASSETS)
    mu32, L, W32 = prepare_inputs(mu, cov, synthetic.equal_weights(N_ASSETS))
    eng = PathEngine(mu32, L, W32, 16, n_local, device="cpu", kernels=FakeKernels(mu32, L, W32),
                     group=dist.group.WORLD if world > 1 else None, world_size=world, rank=rank)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step(synthetic.BENCH_SEED)
    st = eng.stats()[0]
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL (gloo + CPU test double, not a measurement)", "value": n_local * world * args.steps / dt,
                          "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "data": "synthetic",
                          "stats": {"n": int(st["n"]), "n_tail": int(st["n_tail"]), "var95": float(st["var"])}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    global PATHS_PER_GPU
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--native-math", action="store_true", help="normals by hardware log/sqrt/sin/cos Box-Muller (not the spec's values)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="one stream, batches back to back without overlap (for profiling: every mc_paths_kernel row of a "
                         "--kernel-trace --stats summary is then a serial launch, comparable with roofline.kernel_ms)")
    ap.add_argument("--spawn", action="store_true", help="go through the rank-spawning parent also for --gpus 1")
    ap.add_argument("--backend", default=os.environ.get("MCP_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo", "gloo-fake"],
                    help="gloo-fake: CPU rehearsal of the launcher and the rank choreography (tests/fake_kernels.py over gloo); "
                         "gloo: the real kernels with the collectives over gloo, ranks sharing the visible GPUs round-robin "
                         "(rehearsal of N > 1 on a one-GPU box).  Both print a line marked rehearsal, never a measurement")
    ap.add_argument("--sweep", action="store_true",
                    help="side benchmark, NOT the BASELINE metric: configs[4] shape (10,000 Dirichlet portfolios, 16 assets, "
                         "252 steps, --sweep-paths paths), portfolio-sharded over the ranks, MFMA kernel")
    ap.add_argument("--sweep-paths", type=int, default=131072)
    ap.add_argument("--config3", action="store_true",
                    help="side benchmark, NOT the BASELINE metric: configs[3] shape (64 assets, 1260 steps, --config3-paths paths per GPU)")
    ap.add_argument("--config3-paths", type=int, default=10_000_000)
    ap.add_argument("--paths-per-gpu", type=int, default=PATHS_PER_GPU,
                    help="default 1,000,000 = BASELINE configs[1]; 12,500,000 is one GPU's shard of configs[2]")
    args = ap.parse_args()

    PATHS_PER_GPU = args.paths_per_gpu

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        # typed as `python bench.py --gpus N`: start the ranks from here, before anything touches a GPU
        sys.exit(spawn_ranks(args.gpus, [a for a in sys.argv[1:] if a != "--spawn"]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("MCP_BENCH_FAIL_RANK") == str(rank):        # test hook: a rank that dies must fail the whole run
        sys.exit(3)
    if args.backend == "gloo-fake":
        return rehearsal(args, world, rank)
    rehearse = args.backend == "gloo"
    if rehearse:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # under torch.distributed.run also with one rank (exercises RCCL)
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        group = dist.group.WORLD

    from monte_carlo_portfolio_amd import synthetic
    from monte_carlo_portfolio_amd.engine import PathEngine
    from monte_carlo_portfolio_amd.simulate import prepare_inputs

    mu, cov = synthetic.synthetic_market(N_ASSETS)
    seed = synthetic.BENCH_SEED
    if args.sweep:
        K = 10_000
        mu32, L, W32 = prepare_inputs(mu, cov, synthetic.dirichlet_weights(N_ASSETS, K))
        eng = PathEngine(mu32, L, W32, N_STEPS, args.sweep_paths, group=group, world_size=world, rank=rank,
                         native_math=args.native_math, shard="portfolios", pipeline=False)
        for _ in range(max(args.warmup, 1)):
            eng.step(seed)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.step(seed)
        st = eng.gathered_stats()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        if rank == 0:
            flops = 2.0 * K * N_ASSETS * args.sweep_paths * N_STEPS
            print(json.dumps({"metric": "portfolio-paths/s (side benchmark, configs[4] shape)", "value": K * args.sweep_paths / dt,
                              "unit": "portfolio-paths/s", "n_gpus": world, "steps": args.steps, "ms_per_step": dt * 1e3,
                              "config": {"workload": f"10,000 portfolios x {args.sweep_paths} paths x 252 steps, 16 assets, portfolio-sharded"},
                              "wr_product_tflops": flops / dt / 1e12, "opt_idx_max_sharpe": int(np.argmax(st["sharpe"])),
                              "configs4_seconds_extrapolated": dt * 1e6 / args.sweep_paths}))
        if world > 1:
            dist.destroy_process_group()
        return
    if args.config3:
        N3, T3, P3 = 64, 1260, args.config3_paths
        mu3, cov3 = synthetic.synthetic_market(N3)
        mu32, L, W32 = prepare_inputs(mu3, cov3, synthetic.equal_weights(N3))
        eng = PathEngine(mu32, L, W32, T3, P3, group=group, world_size=world, rank=rank, native_math=args.native_math, pipeline=False)
        for _ in range(max(args.warmup, 1)):
            eng.step(seed)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.step(seed)
        st = eng.stats()[0]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        if rank == 0:
            flops = T3 * (N3 * N3 + 3 * N3 + 2)                    # 5,405,400 per path
            tf = flops * P3 * world / dt / 1e12
            print(json.dumps({"metric": "paths/s (side benchmark, configs[3] shape)", "value": P3 * world / dt, "unit": "paths/s",
                              "n_gpus": world, "steps": args.steps, "ms_per_step": dt * 1e3, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"configs[3]: 64 assets, {P3:,} paths per GPU, 1260 steps; triangular GEMV as "
                                                     "SGPR-fed v_pk_fma_f32 (DESIGN.md section 4: fp32 MFMA shares the FMA pipe)"},
                              "model_tflops": tf, "frac_of_fp32_vector_peak": tf / world / FP32_VECTOR_PEAK_TFLOPS,
                              "stats": {"mean": float(st["mean"]), "std": float(st["std"]), "var95": float(st["var"]), "n": int(st["n"])}}))
        if world > 1:
            dist.destroy_process_group()
        return
    w = synthetic.equal_weights(N_ASSETS)
    mu32, L, W32 = prepare_inputs(mu, cov, w)
    eng = PathEngine(mu32, L, W32, N_STEPS, PATHS_PER_GPU, group=group, world_size=world, rank=rank,
                     native_math=args.native_math, n_buffers=int(os.environ.get("MCP_BENCH_NBUF", "0")) or None,
                     pipeline=not args.serial)

    def sync():
        torch.cuda.synchronize()          # all three pipeline streams drained before the cross-rank barrier is issued
        if group is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        eng.step(seed, path_base=0)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.step(seed, path_base=0)
    sync()
    elapsed = time.perf_counter() - t0
    if group is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = eng.stats()[0]

    # a sustained leg after the timed region (not `value`): the same pipelined steps for about two seconds, so that the
    # figure above is corroborated by a run long enough for an outside utilisation sampler to see
    n_sus = max(args.steps, int(2.0 / max(elapsed / args.steps, 1e-4)))
    sync()
    t0 = time.perf_counter()
    for i in range(n_sus):
        eng.step(seed, path_base=0)
    sync()
    sus = time.perf_counter() - t0
    if group is not None:
        t = torch.tensor([sus], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sus = float(t.item())

    # dominant kernel alone, HIP events on the launch stream
    n_k = 10
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.launch_paths_only(seed)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n_k):
        eng.launch_paths_only(seed)
    ev1.record()
    torch.cuda.synchronize()
    k_ms = ev0.elapsed_time(ev1) / n_k

    # the hardware-transcendental variant of the same kernel (tolerance parity), for the record
    native = None
    if world == 1 and not args.native_math:
        eng_n = PathEngine(mu32, L, W32, N_STEPS, PATHS_PER_GPU, native_math=True)
        eng_n.launch_paths_only(seed)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(n_k):
            eng_n.launch_paths_only(seed)
        ev1.record()
        torch.cuda.synchronize()
        nk_ms = ev0.elapsed_time(ev1) / n_k
        native = {"kernel_ms": nk_ms, "kernel_paths_per_s": PATHS_PER_GPU / (nk_ms * 1e-3),
                  "frac_of_issue_model": PATHS_PER_GPU / (nk_ms * 1e-3) / VALU_CEILING_PATHS_PER_S,
                  "note": "MCP_FLAG_NATIVE_MATH: normals by v_log/v_sqrt/v_sin/v_cos Box-Muller (the mix the ceiling is priced on); "
                          "same distribution, other values than the spec"}

    # the folded fast path (SPEC.md 4.1), reported separately and never as `value`
    fold = None
    if world == 1 and not args.native_math:
        eng_f = PathEngine(mu32, L, W32, N_STEPS, PATHS_PER_GPU, fold=True)
        eng_f.launch_paths_only(seed)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(n_k):
            eng_f.launch_paths_only(seed)
        ev1.record()
        torch.cuda.synchronize()
        fk_ms = ev0.elapsed_time(ev1) / n_k
        fold = {"kernel_ms": fk_ms, "kernel_paths_per_s": PATHS_PER_GPU / (fk_ms * 1e-3),
                "note": "MCP_FLAG_FOLD: rho = w.mu + (L^T w).z folded on the host, 16 instead of 152 FMAs per step; same normals, "
                        "other rounding; separately reported fast path, not the headline (SURVEY 7.7)"}

    if rank == 0:
        total_paths = PATHS_PER_GPU * world * args.steps
        value = total_paths / elapsed
        k_paths_s = PATHS_PER_GPU / (k_ms * 1e-3)
        traffic, traffic_src = None, None
        for name in ("r02_pmc_summary.json", "r01_pmc_summary.json"):
            prof = os.path.join(ROOT, "profiles", name)
            if os.path.exists(prof):
                try:
                    traffic = json.load(open(prof)).get("mc_paths_kernel_hbm_bytes_per_launch")
                    traffic_src = f"replayed from profiles/{name} (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this kernel at this shape; not measured in this run)"
                except Exception:
                    traffic = None
                break
        achieved_tf = MODEL_FLOPS_PER_PATH * k_paths_s / 1e12
        roofline = {
            "bound": "valu",
            "kernel": "mc_paths_kernel<4,1,1,false>",
            "achieved": achieved_tf,
            "peak": FP32_VECTOR_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved_tf / FP32_VECTOR_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel_ms": k_ms,
            "kernel_paths_per_s": k_paths_s,
            "pipelined_ms_per_step": elapsed / args.steps * 1e3,
            "frac_of_issue_model": k_paths_s / VALU_CEILING_PATHS_PER_S,
            "issue_model": {"cycles_per_wave_step": ISSUE_CYCLES_PER_WAVE_STEP, "ceiling_paths_per_s": VALU_CEILING_PATHS_PER_S,
                            "table": "profiles/issue_model.json"},
            "note": "VALU-issue bound (SURVEY 0.4/8d): neither HBM (4 B/path) nor MFMA binds.  achieved = model fp32 FLOPs "
                    "(77,112 per path, SURVEY 8d) x paths/s of ONE serial launch (kernel_ms, HIP events on the launch stream); "
                    "peak = the 157.3 TFLOP/s fp32 vector peak of MI355X_MICROARCH.md.  The model FLOPs are 9 % of the "
                    "instructions the algorithm needs (Philox and the normal transform carry no model FLOPs), so frac_of_issue_model "
                    "prices the same launch against a builder-authored issue-cycle table (profiles/issue_model.json).  "
                    "pipelined_ms_per_step < kernel_ms because PathEngine alternates two path streams: batch i+1's first "
                    "waves fill the CUs the partial last round of batch i's 15,625 waves leaves idle (profiles/r02_overlap.txt)",
            "native_math_kernel": native,
            "folded_kernel": fold,
            "hbm": {"achieved": HBM_BYTES_PER_PATH * k_paths_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": HBM_BYTES_PER_PATH * k_paths_s / 1e9 / HBM_PEAK_GBS},
        }
        out = {
            "metric": ("REHEARSAL (ranks share the GPU, collectives over gloo; not a measurement)" if rehearse else
                       "simulated paths/sec (16 assets x 252 steps) + VaR abs-err vs NumPy ref"),
            "value": value, "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: 16 synthetic assets, {PATHS_PER_GPU:,} paths per GPU, 252 steps, fp32, "
                                   "full pass = paths + moments + exact VaR/CVaR",
                       "n_assets": N_ASSETS, "n_steps": N_STEPS, "paths_per_gpu": PATHS_PER_GPU,
                       "global_paths": PATHS_PER_GPU * world, "parallelism": f"path-sharded x{world}",
                       "math": "native" if args.native_math else "exact",
                       "pipeline": ("serial: one stream, no overlap (profiling mode)" if args.serial else
                                    "double-buffered batches: path kernel i+1 overlaps statistics + collectives of batch i")},
            "stats": {"mean": float(stats["mean"]), "std": float(stats["std"]), "sharpe": float(stats["sharpe"]),
                      "var95": float(stats["var"]), "cvar95": float(stats["cvar"]), "n": int(stats["n"]),
                      "n_tail": int(stats["n_tail"])},
            "roofline": roofline,
            "sustained": {"steps": n_sus, "seconds": sus, "paths_per_s": PATHS_PER_GPU * world * n_sus / sus,
                          "note": "the same pipelined steps run for ~2 s after the timed region; corroborates value, is not value"},
        }
        if world == 1 and not args.no_cpu_baseline:
            base, term = cpu_baseline(mu32, L, W32, seed)
            out["cpu_baseline"] = base
            # VaR abs-err vs the NumPy reference on identical seeds: same paths on both sides
            from oracle import ref_stats
            n = term.shape[1]
            from monte_carlo_portfolio_amd import simulate_paths
            g = simulate_paths(mu, cov, w, n_steps=N_STEPS, n_paths=n, seed=seed, native_math=args.native_math)
            want = ref_stats.path_stats(term[0])
            out["var_abs_err"] = abs(g["var"] - want["var"])
            out["sharpe_rel_err"] = abs(g["sharpe"] - want["sharpe"]) / abs(want["sharpe"])
            out["var_check_paths"] = n
            # the same draws evaluated in float64 (the reference's arithmetic, app.py:258-263/708-713): what the fp32
            # recurrence costs in accuracy.  north_star's bar: 1e-6.
            x64 = mc_oracle_f64(mu32, L, W32, seed, n) - 1.0
            v64 = ref_stats.var(x64)
            s64 = x64.mean() / x64.std(ddof=1)
            out["var_abs_err_f64"] = abs(g["var"] - v64)
            out["sharpe_rel_err_f64"] = abs(g["sharpe"] - s64) / abs(s64)
            out["cvar_abs_err_f64"] = abs(g["cvar"] - ref_stats.cvar(x64))
            out["mean_abs_err_f64"] = abs(g["mean"] - x64.mean())
            out["cpu_baseline_numpy"] = numpy_baseline(mu32, L, W32, seed)
            # side figure: the reference's own loop, app.py:699-717, on historical rows
            from monte_carlo_portfolio_amd import sweep
            Rm = np.random.default_rng(0).normal(0.0005, 0.02, (252, N_ASSETS))
            Wm = synthetic.dirichlet_weights(N_ASSETS, 10_000)
            Rc, mean_h, cov_h = sweep.sweep_inputs(Rm, 252)
            sweep.score_portfolios(Rc, mean_h, cov_h, Wm[:16], 0.03)
            t0 = time.perf_counter()
            sg = sweep.score_portfolios(Rc, mean_h, cov_h, Wm, 0.03)
            t_gpu = time.perf_counter() - t0
            t0 = time.perf_counter()
            series = Rc @ Wm.T
            v_np = np.percentile(series, (1 - 0.95) * 100, axis=0)
            np.where(series <= v_np, series, 0.0).sum(axis=0) / (series <= v_np).sum(axis=0)
            (Wm @ mean_h - 0.03) / np.sqrt(np.einsum("pi,ij,pj->p", Wm, cov_h, Wm))
            t_np = time.perf_counter() - t0
            Rs = np.random.default_rng(1).normal(0.01, 0.15, (13, 3))          # the shape of the reference's own run: 13 monthly rows x 3 assets
            sweep.run_all_methods(Rs, user_rf=3.0, annual_factor=12, seed=0)
            t0 = time.perf_counter()
            for i in range(5):
                sweep.run_all_methods(Rs, user_rf=3.0, annual_factor=12, seed=i)
            t_tab2 = (time.perf_counter() - t0) / 5
            out["historical_sweep"] = {
                "tab2_loop": {"workload": "the whole loop of app.py:682-722: 5 methods, 4 x 2,500 Dirichlet portfolios + equal weights, 13 x 3 returns; "
                                          "weights from NumPy's legacy stream exactly as the reference draws them (block-drawn), scoring on the GPU",
                              "seconds": t_tab2, "portfolios_per_s": 10_001 / t_tab2,
                              "reference_portfolios_per_s": 1850, "reference_note": "app.py:699-717 as written at 13 x 3, 1 core, survey container (SURVEY.md section 6)"},
                "workload": "10,000 Dirichlet portfolios x 252 rows x 16 assets, loop body app.py:708-713",
                "gpu_portfolios_per_s_incl_pcie": 10_000 / t_gpu, "numpy_vectorised_portfolios_per_s": 10_000 / t_np,
                "reference_loop_portfolios_per_s": 1940, "reference_loop_note": "app.py:699-717 as written, 1 core, survey container (BASELINE.md section 2)",
                "var_max_abs_diff_vs_numpy": float(np.max(np.abs(sg["var_95"] - v_np)))}
        print(json.dumps(out))
    if group is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
