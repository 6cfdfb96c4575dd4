"""PathEngine: the device-level pipeline, one process per GPU, buffers resident in HBM.

PyTorch is plumbing only here: it allocates the device buffers, provides the HIP stream the kernels
are enqueued on, and carries the RCCL collectives (torch.distributed backend "nccl" on ROCm).  Every
kernel is launched through libmcport.so's C ABI (include/mcport.h, mcp_launch_*) by `HipKernels`;
nothing on the path is a torch op.

Sharding: rank g simulates the global path range [g*P, (g+1)*P); the Philox
counter carries the *global* path id, so any partition yields the same terminal values.  Exchanges
per step:
  - VaR: 3 all_reduce(SUM) of the [K][2][2048] uint64 digit histograms of the radix select
  - moments + CVaR tail: one all_gather of the [K] mcp_record {n, sum, sumsq, min, max, below} (64 B per portfolio),
    merged in rank order by the statistics kernel
All are latency-bound; xGMI bandwidth is irrelevant at these sizes.

The kernel launches sit behind the small `HipKernels` interface so that the collective choreography
(the only multi-rank logic there is) can be exercised on CPU with world_size 2 over gloo by a test
double (tests/fake_kernels.py); the product always uses HipKernels and fails without a GPU.
"""
from __future__ import annotations

import ctypes

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
        return ctypes.c_void_p(t.data_ptr())

    def paths(self, prm, packed, seed, path_begin, n, terminal):
        _ffi.check(self.lib.mcp_launch_paths(ctypes.byref(prm), self._p(packed), seed, path_begin, n,
                                             self._p(terminal), terminal.shape[1], self._stream()))

    def pass0(self, prm, terminal, n, partials, hist):
        _ffi.check(self.lib.mcp_launch_pass0(ctypes.byref(prm), self._p(terminal), terminal.shape[1], n,
                                             self._p(partials), self._p(hist), self._stream()))

    def scan(self, prm, p, n, lo, hi, partials, hist, state, record):
        _ffi.check(self.lib.mcp_launch_scan(ctypes.byref(prm), p, n, lo, hi, self._p(partials), self._p(hist),
                                            self._p(state), self._p(record), self._stream()))

    def hist(self, prm, p, terminal, n, state, partials, hist):
        _ffi.check(self.lib.mcp_launch_hist(ctypes.byref(prm), p, self._p(terminal), terminal.shape[1], n, self._p(state),
                                            self._p(partials), self._p(hist), self._stream()))

    def final(self, prm, n, gamma, lo, hi, partials, hist, state, record, quant, stats):
        _ffi.check(self.lib.mcp_launch_final(ctypes.byref(prm), n, gamma, lo, hi, self._p(partials), self._p(hist),
                                             self._p(state), self._p(record), self._p(quant),
                                             self._p(stats) if stats is not None else None, self._stream()))

    def stats(self, prm, world, gathered, quant, stats):
        _ffi.check(self.lib.mcp_launch_stats(ctypes.byref(prm), world, self._p(gathered), self._p(quant), self._p(stats),
                                             self._stream()))


class PathEngine:
    """One rank's pipeline.  `step()` enqueues one full pass; with `pipeline=True` (default on a GPU) passes
    are double-buffered over two HIP streams: the statistics passes and the collectives of batch i run on
    the statistics stream while the path kernel of batch i+1 already runs on the path stream, so the
    latency-bound tail of a pass (6 small launches, 4 collectives) is hidden behind VALU-bound work."""

    def __init__(self, mu32, chol32, W32, n_steps, n_paths_local, *, compounding="simple", v0=1.0, alpha=0.95,
                 rf=0.0, native_math=False, device=None, group=None, world_size=1, rank=0, kernels=None,
                 pipeline=True, shard="paths", n_buffers=None, fold=False):
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
        self.n_local = int(n_paths_local)
        self.n_total = self.n_local * self.world
        K = W32.shape[0]
        self.K = K
        self.prm = _ffi.make_params(mu32.shape[0], n_steps, K, compounding, v0, alpha, rf, native_math, fold)
        self.rank_lo, self.rank_hi, self.gamma = _ffi.percentile_rank(self.n_total, alpha)
        self.pipeline = bool(pipeline) and self.device.type == "cuda"
        # in-flight batches: 2 hide the statistics tail on one GPU; with collectives in that tail (each one waits for
        # a free wave slot beside the running path kernel on EVERY rank) 4 keep the path streams from stalling
        self.n_buf = (int(n_buffers) if n_buffers else (4 if self.group is not None else 2)) if self.pipeline else 1

        lib = _ffi.lib()
        packed = _ffi.pack_params(mu32, chol32, W32)
        self.d_packed = torch.from_numpy(packed).to(self.device)
        self.bufs = []
        for _ in range(self.n_buf):
            b = {"terminal": torch.empty((K, self.n_local), dtype=torch.float32, device=self.device), "ws": {}}
            for which in range(_ffi.WS_COUNT):            # zero-initialised: the steps clear what they consume
                nbytes = lib.mcp_ws_bytes(which, K)
                b["ws"][which] = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=self.device)
            # typed views for the collectives
            b["record"] = b["ws"][_ffi.WS_RECORD].view(torch.float64).view(K, _ffi.RECORD_DOUBLES)
            b["hist"] = b["ws"][_ffi.WS_HIST]                                  # int64 counts [K][2][2048]
            if self.group is not None:
                b["gather"] = torch.empty((self.world * K, _ffi.RECORD_DOUBLES), dtype=torch.float64, device=self.device)
            self.bufs.append(b)
        self.cur = 0                                   # buffer the NEXT step writes
        self.last = 0                                  # buffer of the most recent step
        if self.pipeline:
            # one path stream per buffer: the next batch's path kernel fills the CUs that the previous one's
            # last (partial) round of waves leaves idle
            import os
            n_ps = max(1, min(int(os.environ.get("MCP_ENGINE_PATH_STREAMS", "2")), self.n_buf))
            self.s_paths = [torch.cuda.Stream(self.device) for _ in range(n_ps)]
            self.s_stats = torch.cuda.Stream(self.device, priority=-1)   # small kernels: dispatch ahead of path blocks
            self.ev_paths = [torch.cuda.Event() for _ in range(self.n_buf)]
            self.ev_stats = [torch.cuda.Event() for _ in range(self.n_buf)]
            for e in self.ev_stats:
                e.record(torch.cuda.current_stream(self.device))
            # buffers were filled on the current stream: order both pipeline streams after it
            for sp in self.s_paths:
                sp.wait_stream(torch.cuda.current_stream(self.device))
            self.s_stats.wait_stream(torch.cuda.current_stream(self.device))

    # convenience views of the most recent step's buffers
    @property
    def d_terminal(self):
        return self.bufs[self.last]["terminal"]

    @property
    def ws(self):
        return self.bufs[self.last]["ws"]

    def _enqueue_paths(self, b, seed, path_base):
        n = self.n_local
        self.k.paths(self.prm, self.d_packed, seed, path_base + self.rank * n, n, b["terminal"])

    def _enqueue_stats(self, b):
        k, n, ws, prm = self.k, self.n_local, b["ws"], self.prm
        dist = self.torch.distributed if self.group is not None else None      # also exercised with 1 rank
        lo, hi = self.rank_lo, self.rank_hi
        k.pass0(prm, b["terminal"], n, ws[_ffi.WS_PARTIALS], ws[_ffi.WS_HIST])
        for p in range(3):
            if dist is not None:
                dist.all_reduce(b["hist"], group=self.group)
            if p < 2:
                k.scan(prm, p, n, lo, hi, ws[_ffi.WS_PARTIALS], ws[_ffi.WS_HIST], ws[_ffi.WS_STATE], ws[_ffi.WS_RECORD])
                k.hist(prm, p + 1, b["terminal"], n, ws[_ffi.WS_STATE], ws[_ffi.WS_PARTIALS], ws[_ffi.WS_HIST])
            else:
                k.final(prm, n, self.gamma, lo, hi, ws[_ffi.WS_PARTIALS], ws[_ffi.WS_HIST], ws[_ffi.WS_STATE], ws[_ffi.WS_RECORD],
                        ws[_ffi.WS_QUANT], None if dist is not None else ws[_ffi.WS_STATS])
        if dist is not None:
            dist.all_gather_into_tensor(b["gather"], b["record"], group=self.group)
            k.stats(prm, self.world, b["gather"], ws[_ffi.WS_QUANT], ws[_ffi.WS_STATS])

    def step(self, seed: int, path_base: int = 0):
        """Enqueue one full pass (paths -> statistics).  No host sync."""
        i = self.cur
        b = self.bufs[i]
        if not self.pipeline:
            self._enqueue_paths(b, seed, path_base)
            self._enqueue_stats(b)
        else:
            torch = self.torch
            sp = self.s_paths[i % len(self.s_paths)]
            with torch.cuda.stream(sp):
                sp.wait_event(self.ev_stats[i])                 # the pass that last used this buffer is done
                self._enqueue_paths(b, seed, path_base)
                self.ev_paths[i].record(sp)
            with torch.cuda.stream(self.s_stats):
                self.s_stats.wait_event(self.ev_paths[i])
                self._enqueue_stats(b)
                self.ev_stats[i].record(self.s_stats)
        self.last = i
        self.cur = (i + 1) % self.n_buf

    def launch_paths_only(self, seed: int, path_base: int = 0):
        """The dominant kernel alone, on the CURRENT stream into the most recent buffer (roofline timing).  Call
        synchronize() first if batches enqueued by step() may still be in flight."""
        self._enqueue_paths(self.bufs[self.last], seed, path_base)

    def synchronize(self):
        if self.pipeline:
            for sp in self.s_paths:
                sp.synchronize()
            self.s_stats.synchronize()
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
        return self.d_terminal.cpu().numpy()

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
