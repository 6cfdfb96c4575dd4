"""Test double for monte_carlo_portfolio_amd.engine.HipKernels: the same step interface and the same
buffer layouts, computed on CPU torch tensors with NumPy and the C oracle.

TEST INFRASTRUCTURE ONLY.  It exists so the multi-rank choreography of PathEngine.step (shard
offsets, histogram all-reduce between the select passes, all-gather of the records, rank-ordered merge)
runs on CPU with world_size 2 over gloo.  The product never imports this module.
"""
import ctypes

import numpy as np

from monte_carlo_portfolio_amd import _ffi
from oracle import mc_oracle, ref_stats

BINS = _ffi.MCP_SELECT_BINS
SHAPE = {0: (21, 11, 32), 1: (10, 11, 21), 2: (0, 10, 10)}     # pass -> (shift, bits, prefix shift)


def _keys(v):
    b = v.view(np.uint32)
    return np.where(b >> 31 != 0, ~b, b | np.uint32(0x80000000)).astype(np.uint32)


def _key_to_float(k):
    k = np.uint32(k)
    b = np.uint32(k ^ np.uint32(0x80000000)) if (k >> 31) else np.uint32(~k)
    return np.array([b], np.uint32).view(np.float32)[0]


class FakeKernels:
    device_type = "cpu"

    def __init__(self, mu32, chol32, W32):
        self.mu, self.L, self.W = mu32, chol32, W32

    @staticmethod
    def _np(t, dtype):
        return t.numpy().view(dtype)

    def paths(self, prm, packed, seed, path_begin, n, terminal):
        comp = "log" if prm.compounding == 1 else "simple"
        terminal.numpy()[:, :n] = mc_oracle.simulate(self.mu, self.L, self.W, prm.n_steps, n, seed, path_begin=path_begin,
                                                     v0=prm.v0, compounding=comp, n_threads=2)

    # ---- statistics pipeline: same buffers, same read-and-clear protocol as mcp_stats_kernels.hip -----------------
    def _x(self, prm, v):
        return ref_stats.terminal_to_x(v, prm.v0, "log" if prm.compounding == 1 else "simple")

    def pass0(self, prm, terminal, n, partials, hist):
        K = prm.n_portfolios
        part = self._np(partials, np.float64).reshape(K, -1, 6)
        h = self._np(hist, np.int64).reshape(K, 2, BINS)
        for k in range(K):
            v = terminal.numpy()[k, :n]
            x = self._x(prm, v)
            part[k, 0, :5] = [x.size, x.sum(), (x * x).sum(), x.min() if n else np.inf, x.max() if n else -np.inf]
            h[k, 0] += np.bincount(_keys(v) >> np.uint32(21), minlength=BINS)

    def _descend(self, p, hh, prefix, rank):
        shift, bits, pshift = SHAPE[p]
        cum = np.cumsum(hh)
        d = int(np.searchsorted(cum, rank, side="right"))
        before = int(cum[d - 1]) if d > 0 else 0
        return (d if p == 0 else ((prefix << bits) | d) & 0xFFFFFFFF), rank - before, int(hh[d])

    def scan(self, prm, p, n, lo, hi, partials, hist, state, record):
        K = prm.n_portfolios
        part = self._np(partials, np.float64).reshape(K, -1, 6)
        h = self._np(hist, np.int64).reshape(K, 2, BINS)
        s = self._np(state, np.uint64).reshape(K, 2, 2)
        rec = self._np(record, np.float64).reshape(K, 8)
        for k in range(K):
            if p == 0:
                rec[k] = list(part[k, 0, :5]) + [0.0, 0.0, 0.0]
            else:
                rec[k, 5] = part[k, 0, 5]
            for w in range(2):
                prefix, rank = (0, (lo, hi)[w]) if p == 0 else (int(s[k, w, 0] & np.uint64(0xFFFFFFFF)), int(s[k, w, 1]))
                pre, r, _ = self._descend(p, h[k, 0 if p == 0 else w], prefix, rank)
                s[k, w] = [pre, r]
            h[k] = 0

    def hist(self, prm, p, terminal, n, state, partials, hist):
        shift, bits, pshift = SHAPE[p]
        K = prm.n_portfolios
        s = self._np(state, np.uint64).reshape(K, 2, 2)
        h = self._np(hist, np.int64).reshape(K, 2, BINS)
        part = self._np(partials, np.float64).reshape(K, -1, 6)
        for k in range(K):
            v = terminal.numpy()[k, :n]
            keys = _keys(v)
            digit = (keys >> np.uint32(shift)) & np.uint32((1 << bits) - 1)
            pre = keys >> np.uint32(pshift)
            for w in range(2):
                prefix = np.uint32(s[k, w, 0] & np.uint64(0xFFFFFFFF))
                h[k, w] += np.bincount(digit[pre == prefix], minlength=BINS)
            pa = np.uint32(s[k, 0, 0] & np.uint64(0xFFFFFFFF))
            sel = pre < pa
            if p == 2:
                sel &= (pre >> np.uint32(11)) == (pa >> np.uint32(11))
            part[k, 0, 5] = self._x(prm, v[sel]).sum()

    def final(self, prm, n, gamma, lo, hi, partials, hist, state, record, quant, stats):
        K = prm.n_portfolios
        part = self._np(partials, np.float64).reshape(K, -1, 6)
        h = self._np(hist, np.int64).reshape(K, 2, BINS)
        s = self._np(state, np.uint64).reshape(K, 2, 2)
        rec = self._np(record, np.float64).reshape(K, 8)
        q = quant.numpy().view(np.uint8)[:K * _ffi.QUANT_DTYPE.itemsize].view(_ffi.QUANT_DTYPE)
        for k in range(K):
            pres = [int(s[k, w, 0] & np.uint64(0xFFFFFFFF)) for w in range(2)]
            keys = [self._descend(2, h[k, w], pres[w], int(s[k, w, 1]))[0] for w in range(2)]
            a, b = [self._x(prm, np.array([_key_to_float(key)], np.float32))[0] for key in keys]
            d = b - a
            var = a + d * gamma if gamma < 0.5 else b - d * (1 - gamma)
            cnt, level2 = 0, 0.0
            for w in range(2 if pres[1] != pres[0] else 1):          # walk the digits of the last bucket(s): x <= var literally
                for dd in np.nonzero(h[k, w, :1024])[0]:
                    x = self._x(prm, np.array([_key_to_float((pres[w] << 10) | int(dd))], np.float32))[0]
                    if x <= var:
                        cnt += int(h[k, w, dd])
                        level2 += float(h[k, w, dd]) * x
            n_tail = (lo - int(s[k, 0, 1])) + cnt
            q[k] = (a, b, var, level2, n_tail, 0)
            rec[k, 5] += part[k, 0, 5]
            h[k] = 0
        if stats is not None:
            self._finish(prm, rec, q, stats)

    def stats(self, prm, world, gathered, quant, stats):
        K = prm.n_portfolios
        g = gathered.numpy().reshape(world, K, 8)
        q = quant.numpy().view(np.uint8)[:K * _ffi.QUANT_DTYPE.itemsize].view(_ffi.QUANT_DTYPE)
        m = np.zeros((K, 8))
        m[:, [0, 1, 2, 5]] = g[:, :, [0, 1, 2, 5]].sum(axis=0)
        m[:, 3] = g[:, :, 3].min(axis=0)
        m[:, 4] = g[:, :, 4].max(axis=0)
        self._finish(prm, m, q, stats)

    @staticmethod
    def _finish(prm, m, q, stats):
        K = prm.n_portfolios
        out = stats.numpy().view(np.uint8)[:K * _ffi.STATS_DTYPE.itemsize].view(_ffi.STATS_DTYPE)
        for k in range(K):
            n, s1, s2 = m[k, 0], m[k, 1], m[k, 2]
            mean = s1 / n
            m2 = max(s2 - s1 * mean, 0.0)
            std = np.sqrt(m2 / (n - 1)) if n > 1 else 0.0
            n_tail, sum_tail = int(q[k]["n_tail"]), m[k, 5] + q[k]["level2"]
            out[k] = (int(n), n_tail, mean, m2, std, (mean - prm.rf) / std if std > 0 else 0.0, q[k]["var"],
                      sum_tail / n_tail if n_tail > 0 else q[k]["var"], m[k, 3], m[k, 4], sum_tail, q[k]["x_lo"], q[k]["x_hi"])
