"""PathEngine: the device-level pipeline, one process per GPU, buffers resident in HBM.

PyTorch is plumbing only here: it allocates the device buffers, provides the HIP streams the kernels
are enqueued on, and carries the RCCL collectives (torch.distributed backend "nccl" on ROCm).  Every
kernel is launched through libmcport.so's C ABI (include/mcport.h, mcp_launch_*) by `HipKernels`;
nothing on the path is a torch op.

One pass = the fused path kernel (Philox -> normals -> Cholesky GEMV -> compounding -> terminal values, and in its
epilogue the shifted moment partials and the digit-0 histogram of the radix select) followed by four short stages:

    stage 1   [exchange HIST] scan(0)  hist(1)
    stage 2   [exchange HIST] scan(1)  hist(2)
    stage 3   [exchange HIST] final
    stage 4   [exchange RECORD -> stats]          (only when there is something to exchange)

Sharding: rank g simulates the global path range [g*P, (g+1)*P); the Philox
counter carries the *global* path id, so any partition yields the same terminal values.  The exchanges are
3 all_reduce(SUM) of the [K][2][2048] uint64 digit histograms and one all_gather of the [K] mcp_record
{n, sum (x-c), sum (x-c)^2, min, max, below, pivot} (64 B per portfolio), merged in rank order by the statistics kernel.
All are latency-bound; xGMI bandwidth is irrelevant at these sizes.

Skewed schedule (opt-in, `skew=True`).  Every stage is a handful of tiny kernels that must find a free wave slot beside the path kernel of
the NEXT batch, which saturates the chip: run back to back, each of them (and each collective, which is a kernel too)
waits for a slot, and the chain of one batch can take longer than the path kernel it hides behind.  So with exchanges
in the chain the engine enqueues stage s of batch t-s at step t (a software pipeline, `depth` batches deep): by the time
a stage is issued, the stage it depends on was issued a whole step earlier and is long done; no stream ever stalls in
front of a collective, and every small kernel has a full step to get its slot.  torch's NCCL process group runs all
collectives on one internal stream in issue order: the skew also keeps that stream from blocking on a starved kernel.

Logical shards (`logical_shards=S`, one process, one GPU): the rank's path range is split over S shards with their own
buffers and streams that exchange through a KERNEL (mcp_launch_sum_u64; records by device copies) -- the same
choreography and the same kind of work in the tail (exchange kernels beside a saturating path kernel) that several GPUs
have, executable on a one-GPU box.

The kernel launches sit behind the small `HipKernels` interface so that the choreography (the only multi-rank logic
there is) can be exercised on CPU with world_size 2 over gloo by a test double (tests/fake_kernels.py); the product
always uses HipKernels and fails without a GPU.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _ffi


class HipKernels:
    """Enqueue-only launches of libmcport.so's kernels on torch's current HIP stream."""

    device_type = "cuda"

    def __init__(self, torch, device):
        self.torch, self.device = torch, device
        self.lib = _ffi.lib()
        if self.lib.mcp_device_count() < 1:
            raise _ffi.McpError("no HIP device visible (the product path has no CPU fallback)")

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def paths(self, prm, packed, pivot, seed, path_begin, n, terminal, partials, hist):
        _ffi.check(self.lib.mcp_launch_paths(ctypes.byref(prm), self._p(packed), self._p(pivot), seed, path_begin, n,
                                             self._p(terminal), terminal.shape[1], self._p(partials), self._p(hist),
                                             self._stream()))

    def pass0(self, prm, terminal, n, pivot, partials, hist):
        _ffi.check(self.lib.mcp_launch_pass0(ctypes.byref(prm), self._p(terminal), terminal.shape[1], n, self._p(pivot),
                                             self._p(partials), self._p(hist), self._stream()))

    def scan(self, prm, p, n, lo, hi, partials, below, pivot, hist, state, record):
        _ffi.check(self.lib.mcp_launch_scan(ctypes.byref(prm), p, n, lo, hi, self._p(partials), self._p(below), self._p(pivot),
                                            self._p(hist), self._p(state), self._p(record), self._stream()))

    def hist(self, prm, p, terminal, n, state, pivot, below, hist):
        _ffi.check(self.lib.mcp_launch_hist(ctypes.byref(prm), p, self._p(terminal), terminal.shape[1], n, self._p(state),
                                            self._p(pivot), self._p(below), self._p(hist), self._stream()))

    def final(self, prm, n, gamma, lo, hi, below, hist, state, record, quant, stats):
        _ffi.check(self.lib.mcp_launch_final(ctypes.byref(prm), n, gamma, lo, hi, self._p(below), self._p(hist),
                                             self._p(state), self._p(record), self._p(quant), self._p(stats), self._stream()))

    def stats(self, prm, world, gathered, quant, stats):
        _ffi.check(self.lib.mcp_launch_stats(ctypes.byref(prm), world, self._p(gathered), self._p(quant), self._p(stats),
                                             self._stream()))

    def sum_u64(self, bufs, words):
        arr = (ctypes.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
        _ffi.check(self.lib.mcp_launch_sum_u64(arr, len(bufs), words, self._stream()))


class PathEngine:
    """One rank's pipeline.  `step()` enqueues one full pass; with `pipeline=True` (default on a GPU) batches are
    multi-buffered over HIP streams: alternating path streams, and a small pool of statistics streams (`n_stats_streams` per
    logical shard, shared by the buffers modulo), so the tails of different batches overlap each other and the next path kernel."""

    N_STAGES = 4

    def __init__(self, mu32, chol32, W32, n_steps, n_paths_local, *, compounding="simple", v0=1.0, alpha=0.95,
                 rf=0.0, native_math=False, device=None, group=None, world_size=1, rank=0, kernels=None,
                 pipeline=True, shard="paths", n_buffers=None, fold=False, logical_shards=1, skew=None, fused=True, cu_reserve=0,
                 n_stats_streams=None):
        import torch

        self.torch = torch
        if kernels is None:
            device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            kernels = HipKernels(torch, device)
        self.k = kernels
        self.device = torch.device(device if device is not None else kernels.device_type)
        if shard not in ("paths", "portfolios"):
            raise ValueError("shard must be 'paths' or 'portfolios'")
        self.shard = shard
        self.gather_group, self.gather_world, self.gather_rank = group, int(world_size), int(rank)
        if shard == "portfolios":
            # BASELINE configs[4]: every rank walks ALL paths for its own slice of the weight matrix (same seed ->
            # common random numbers across ranks); no data-path collective at all, one all_gather of the records.
            k_all = W32.shape[0]
            kc = -(-k_all // int(world_size))
            self.k_slice = (min(int(rank) * kc, k_all), min((int(rank) + 1) * kc, k_all))
            self.k_chunk, self.k_all = kc, k_all
            W_local = np.zeros((kc, W32.shape[1]), np.float32)           # ragged last rank: zero-weight padding rows
            W_local[:self.k_slice[1] - self.k_slice[0]] = W32[self.k_slice[0]:self.k_slice[1]]
            W32, group, world_size, rank = W_local, None, 1, 0
        self.group, self.world, self.rank = group, int(world_size), int(rank)
        self.S = int(logical_shards)
        if self.S < 1 or self.S > 8:
            raise ValueError("logical_shards must be in 1..8")
        if self.S > 1 and self.group is not None:
            raise ValueError("logical shards and a process group are alternatives")
        self.n_local = int(n_paths_local)
        self.n_total = self.n_local * self.world
        # logical shard j simulates paths [off_j, off_j + n_j) of this rank's range
        self.shard_n = [self.n_local // self.S + (1 if j < self.n_local % self.S else 0) for j in range(self.S)]
        self.shard_off = [sum(self.shard_n[:j]) for j in range(self.S)]
        K = W32.shape[0]
        self.K = K
        self.fused = bool(fused)
        self.prm = _ffi.make_params(mu32.shape[0], n_steps, K, compounding, v0, alpha, rf, native_math, fold)
        self.rank_lo, self.rank_hi, self.gamma = _ffi.percentile_rank(self.n_total, alpha)
        self.pipeline = bool(pipeline) and self.device.type == "cuda"
        self.exchanging = self.group is not None or self.S > 1
        # skewed schedule (module docstring): default where a torch process group carries the exchanges (its collectives run on
        # ONE internal stream in issue order; the skew keeps that stream from waiting on a starved kernel).  It is a property of
        # the enqueue ORDER (it also runs without streams, which is how the CPU tests cover it).  Streams are kept FEW: the HIP
        # runtime multiplexes more than 4 streams onto 4 hardware queues (GPU_MAX_HW_QUEUES), where a kernel waiting for an
        # event blocks the streams behind it -- with one statistics stream per buffer in flight (14 streams at 2 logical shards)
        # the step took +12 %, and raising GPU_MAX_HW_QUEUES is no way out: oversubscribed hardware queues are time-sliced (one
        # configuration dropped to 300 ms per step; profiles/r03_tail_ab_queues.txt).  So batches share `n_stats_streams`
        # statistics streams (per shard) by buffer index modulo, whatever the number of buffers: 2 by default, 1 under a
        # process group (2 path streams + 1 statistics stream + the process group's own stream = one per hardware queue).
        # Which streams end up sharing a hardware queue is the runtime's choice, so A/B figures of stream layouts are only
        # meaningful between fresh processes (bench.py --logical-shards / --skew), not between engines of one process.
        self.skew = (self.group is not None and self.pipeline) if skew is None else bool(skew)
        if n_stats_streams is None:
            n_stats_streams = 1 if self.group is not None else 2
        self.depth = self.N_STAGES if self.skew else 0
        if n_buffers:
            self.n_buf = int(n_buffers) if (self.pipeline or self.skew) else 1
        else:
            self.n_buf = self.depth + 2 if self.skew else (2 if self.pipeline else 1)
        if self.skew and self.n_buf < self.depth + 1:
            raise ValueError(f"the skewed schedule keeps {self.depth + 1} batches in flight: n_buffers >= {self.depth + 1}")

        lib = _ffi.lib()
        packed = _ffi.pack_params(mu32, chol32, W32)
        self.d_packed = torch.from_numpy(packed).to(self.device)
        self.d_pivot = torch.from_numpy(_ffi.pivots(self.prm, mu32, chol32, W32)).to(self.device)
        self.bufs = []                                     # [buffer][shard] -> dict
        for _ in range(self.n_buf):
            row = []
            for j in range(self.S):
                nj = max(self.shard_n[j], 1)
                b = {"terminal": torch.empty((K, nj), dtype=torch.float32, device=self.device), "ws": {}, "n": self.shard_n[j],
                     "off": self.shard_off[j]}
                for which in range(_ffi.WS_COUNT):            # zero-initialised: the steps clear what they consume
                    nbytes = lib.mcp_ws_bytes(which, K, nj)
                    b["ws"][which] = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=self.device)
                b["ws"][_ffi.WS_PIVOT] = self.d_pivot.view(torch.int64)
                # typed views for the collectives
                b["record"] = b["ws"][_ffi.WS_RECORD].view(torch.float64).view(K, _ffi.RECORD_DOUBLES)
                b["hist"] = b["ws"][_ffi.WS_HIST]                                  # int64 counts [K][2][2048]
                if self.group is not None:
                    b["gather"] = torch.empty((self.world * K, _ffi.RECORD_DOUBLES), dtype=torch.float64, device=self.device)
                elif self.S > 1 and j == 0:
                    b["gather"] = torch.empty((self.S * K, _ffi.RECORD_DOUBLES), dtype=torch.float64, device=self.device)
                row.append(b)
            self.bufs.append(row)
        self.t = 0                                     # batches enqueued so far (paths)
        self.stage_done = {}                           # batch id -> stages already enqueued (skewed schedule)
        self.last = 0                                  # buffer of the most recent COMPLETE batch
        if self.pipeline:
            # alternating path streams: the next batch's path kernel fills the CUs that the previous one's last
            # (partial) round of waves leaves idle; statistics streams from a pool of n_stats_streams per shard (see above)
            n_ps = max(1, min(int(os.environ.get("MCP_ENGINE_PATH_STREAMS", "2")), self.n_buf * self.S))
            self.cu_reserve = int(cu_reserve)
            self._raw_streams = []
            if self.cu_reserve > 0:
                # path kernels leave `cu_reserve` compute units to the statistics / exchange kernels of the batches before
                # (include/mcport.h: mcp_stream_create); torch only wraps the handle
                self.s_paths = []
                for _ in range(n_ps):
                    h = ctypes.c_void_p()
                    _ffi.check(_ffi.lib().mcp_stream_create(self.device.index or 0, self.cu_reserve, ctypes.byref(h)))
                    self._raw_streams.append(h)
                    self.s_paths.append(torch.cuda.ExternalStream(h.value, device=self.device))
            else:
                self.s_paths = [torch.cuda.Stream(self.device) for _ in range(n_ps)]
            ns = max(1, min(int(n_stats_streams), self.n_buf))
            pool = [[torch.cuda.Stream(self.device, priority=-1) for _ in range(self.S)] for _ in range(ns)]
            self._stat_pool = pool
            self.s_stats = [pool[i % ns] for i in range(self.n_buf)]          # buffer i -> statistics stream i % ns (per shard)
            self.ev_paths = [[torch.cuda.Event() for _ in range(self.S)] for _ in range(self.n_buf)]
            self.ev_stats = [[torch.cuda.Event() for _ in range(self.S)] for _ in range(self.n_buf)]
            self.ev_x = [[torch.cuda.Event() for _ in range(self.S)] for _ in range(self.n_buf)]      # exchange hand-shakes
            cur = torch.cuda.current_stream(self.device)
            for row in self.ev_stats:
                for e in row:
                    e.record(cur)
            # buffers were filled on the current stream: order every pipeline stream after it
            for sp in self.s_paths:
                sp.wait_stream(cur)
            for row in pool:
                for ss in row:
                    ss.wait_stream(cur)

    # convenience views of the most recent complete batch's buffers (shard 0 holds the merged statistics)
    @property
    def d_terminal(self):
        return self.bufs[self.last][0]["terminal"]

    @property
    def ws(self):
        return self.bufs[self.last][0]["ws"]

    # ---- stream plumbing -------------------------------------------------------------------------------------------
    def _on_stats(self, i, j):
        if not self.pipeline:
            import contextlib
            return contextlib.nullcontext()
        return self.torch.cuda.stream(self.s_stats[i][j])

    # ---- the pieces of one pass ---------------------------------------------------------------------------------------
    def _enqueue_paths(self, b, seed, path_base, with_stats=True):
        n, ws = b["n"], b["ws"]
        if n < 1:
            return
        stats = with_stats and self.fused
        self.k.paths(self.prm, self.d_packed, self.d_pivot, seed, path_base + self.rank * self.n_local + b["off"], n, b["terminal"],
                     ws[_ffi.WS_PARTIALS] if stats else None, ws[_ffi.WS_HIST] if stats else None)

    def _exchange_hist(self, i):
        row = self.bufs[i]
        if self.group is not None:
            with self._on_stats(i, 0):
                self.torch.distributed.all_reduce(row[0]["hist"], group=self.group)
        elif self.S > 1:
            words = row[0]["hist"].numel()
            if self.pipeline:
                s0 = self.s_stats[i][0]
                for j in range(1, self.S):
                    self.ev_x[i][j].record(self.s_stats[i][j])
                    s0.wait_event(self.ev_x[i][j])
            with self._on_stats(i, 0):
                self.k.sum_u64([b["hist"] for b in row], words)
            if self.pipeline:
                self.ev_x[i][0].record(self.s_stats[i][0])
                for j in range(1, self.S):
                    self.s_stats[i][j].wait_event(self.ev_x[i][0])

    def _stage(self, i, s):
        """Enqueue stage s (1..4) of the batch in buffer i."""
        k, prm, lo, hi = self.k, self.prm, self.rank_lo, self.rank_hi
        row = self.bufs[i]
        if s <= 3:
            if self.exchanging:
                self._exchange_hist(i)
            for j, b in enumerate(row):
                ws, n = b["ws"], b["n"]
                with self._on_stats(i, j):
                    if s <= 2:
                        k.scan(prm, s - 1, n, lo, hi, ws[_ffi.WS_PARTIALS], ws[_ffi.WS_BELOW], self.d_pivot, ws[_ffi.WS_HIST],
                               ws[_ffi.WS_STATE], ws[_ffi.WS_RECORD])
                        k.hist(prm, s, b["terminal"], n, ws[_ffi.WS_STATE], self.d_pivot, ws[_ffi.WS_BELOW], ws[_ffi.WS_HIST])
                    else:
                        k.final(prm, n, self.gamma, lo, hi, ws[_ffi.WS_BELOW], ws[_ffi.WS_HIST], ws[_ffi.WS_STATE], ws[_ffi.WS_RECORD],
                                ws[_ffi.WS_QUANT], None if self.exchanging else ws[_ffi.WS_STATS])
        else:
            b0 = row[0]
            if self.group is not None:
                with self._on_stats(i, 0):
                    self.torch.distributed.all_gather_into_tensor(b0["gather"], b0["record"], group=self.group)
                    k.stats(prm, self.world, b0["gather"], b0["ws"][_ffi.WS_QUANT], b0["ws"][_ffi.WS_STATS])
            elif self.S > 1:
                if self.pipeline:
                    for j in range(1, self.S):
                        self.ev_x[i][j].record(self.s_stats[i][j])
                        self.s_stats[i][0].wait_event(self.ev_x[i][j])
                with self._on_stats(i, 0):
                    for j, b in enumerate(row):
                        b0["gather"][j * self.K:(j + 1) * self.K].copy_(b["record"], non_blocking=True)
                    k.stats(prm, self.S, b0["gather"], b0["ws"][_ffi.WS_QUANT], b0["ws"][_ffi.WS_STATS])
            if self.pipeline:
                self.ev_stats[i][0].record(self.s_stats[i][0])
                for j in range(1, self.S):                           # shard j's buffers are free when ITS last kernel is done
                    self.ev_stats[i][j].record(self.s_stats[i][j])

    def _first_stage_empty_shards(self, i):
        """A logical shard without paths (fewer paths than shards) contributes empty moment partials."""
        for j, b in enumerate(self.bufs[i]):
            if b["n"] < 1:
                with self._on_stats(i, j):
                    self.k.pass0(self.prm, b["terminal"], 0, self.d_pivot, b["ws"][_ffi.WS_PARTIALS], b["ws"][_ffi.WS_HIST])

    def _begin_batch(self, i, seed, path_base):
        row = self.bufs[i]
        if not self.pipeline:
            for b in row:
                self._enqueue_paths(b, seed, path_base)
                if not self.fused and b["n"] > 0:
                    self.k.pass0(self.prm, b["terminal"], b["n"], self.d_pivot, b["ws"][_ffi.WS_PARTIALS], b["ws"][_ffi.WS_HIST])
            self._first_stage_empty_shards(i)
            return
        torch = self.torch
        for j, b in enumerate(row):
            sp = self.s_paths[(i * self.S + j) % len(self.s_paths)]
            with torch.cuda.stream(sp):
                sp.wait_event(self.ev_stats[i][0])                 # the pass that last used this buffer is done (merged on shard 0)
                if j:
                    sp.wait_event(self.ev_stats[i][j])
                self._enqueue_paths(b, seed, path_base)
                self.ev_paths[i][j].record(sp)
            self.s_stats[i][j].wait_event(self.ev_paths[i][j])
            if not self.fused and b["n"] > 0:
                with self._on_stats(i, j):
                    self.k.pass0(self.prm, b["terminal"], b["n"], self.d_pivot, b["ws"][_ffi.WS_PARTIALS], b["ws"][_ffi.WS_HIST])
        self._first_stage_empty_shards(i)

    def step(self, seed: int, path_base: int = 0):
        """Enqueue one full pass (paths -> statistics).  No host sync.  Skewed schedule: the statistics stages of this batch
        are enqueued by the next `depth` calls (or by synchronize())."""
        t = self.t
        i = t % self.n_buf
        self._begin_batch(i, seed, path_base)
        self.stage_done[t] = 0
        self.t = t + 1
        if self.skew:
            for s in range(1, self.N_STAGES + 1):          # stage s of batch t - s: oldest batch first = same order on every rank
                bt = t - s
                if bt in self.stage_done and self.stage_done[bt] == s - 1:
                    self._advance(bt)
        else:
            while t in self.stage_done:
                self._advance(t)

    def _advance(self, bt):
        s = self.stage_done[bt] + 1
        last_stage = self.N_STAGES if self.exchanging else self.N_STAGES - 1
        i = bt % self.n_buf
        if s <= last_stage:
            self._stage(i, s)
        if s >= last_stage:
            if not self.exchanging and self.pipeline:       # stage 3 was the last: buffers free when final is done
                for j in range(self.S):
                    self.ev_stats[i][j].record(self.s_stats[i][j])
            del self.stage_done[bt]
            self.last = i
        else:
            self.stage_done[bt] = s

    def flush(self):
        """Enqueue every stage still owed by the skewed schedule (oldest batch first)."""
        while self.stage_done:
            self._advance(min(self.stage_done))

    def launch_paths_only(self, seed: int, path_base: int = 0, with_stats: bool = True, clear_hist: bool = True):
        """The dominant kernel alone (with its fused epilogue), on the CURRENT stream into the most recent buffer
        (roofline timing).  Call synchronize() first if batches enqueued by step() may still be in flight.  The digit-0
        counts the epilogue leaves in the histogram are cleared again (nothing consumes them here); a timing loop passes
        clear_hist=False and clears once at the end, so that no fill kernel sits between the timed launches."""
        b = self.bufs[self.last][0]
        self._enqueue_paths(b, seed, path_base, with_stats)
        if with_stats and self.fused and clear_hist:
            b["hist"].zero_()

    def close(self):
        """Destroy the streams this engine created through the C ABI (CU-masked path streams)."""
        if getattr(self, "_raw_streams", None):
            self.synchronize()
            for h in self._raw_streams:
                _ffi.lib().mcp_stream_destroy(h)
            self._raw_streams = []

    def synchronize(self):
        self.flush()
        if self.pipeline:
            for sp in self.s_paths:
                sp.synchronize()
            for row in self._stat_pool:
                for ss in row:
                    ss.synchronize()
        elif self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def stats(self) -> np.ndarray:
        """Synchronise and fetch the [K] mcp_stats records of the last step()."""
        self.synchronize()
        nbytes = self.K * _ffi.STATS_DTYPE.itemsize
        raw = self.ws[_ffi.WS_STATS].cpu().numpy().view(np.uint8)[:nbytes]
        return raw.view(_ffi.STATS_DTYPE).copy()

    def terminal(self) -> np.ndarray:
        self.synchronize()
        return np.concatenate([b["terminal"][:, :b["n"]].cpu().numpy() for b in self.bufs[self.last]], axis=1)

    def gathered_stats(self) -> np.ndarray:
        """shard='portfolios': all ranks' records in portfolio order ([K_all] mcp_stats), one all_gather."""
        if self.shard != "portfolios":
            return self.stats()
        torch = self.torch
        self.synchronize()
        words = self.k_chunk * _ffi.STATS_DTYPE.itemsize // 8
        mine = self.ws[_ffi.WS_STATS][:words].contiguous()
        if self.gather_world > 1:
            out = torch.empty(self.gather_world * words, dtype=torch.int64, device=self.device)
            torch.distributed.all_gather_into_tensor(out, mine, group=self.gather_group)
        else:
            out = mine
        rec = out.cpu().numpy().view(np.uint8).view(_ffi.STATS_DTYPE)
        return rec[:self.k_all].copy()          # drop the zero-weight padding rows of the last rank
