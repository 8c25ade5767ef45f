"""Token-range bookkeeping for the variable-length attention kernels (host side)."""
from __future__ import annotations

import numpy as np
import torch


class Segments:
    """Per-image token ranges of a packed batch, plus the 128-row block tables the varlen kernels walk."""
    BLOCK = 128

    def __init__(self, q_lens, k_lens, device):
        self.q_lens, self.k_lens = list(map(int, q_lens)), list(map(int, k_lens))
        assert len(self.q_lens) == len(self.k_lens)
        self.nseg = len(self.q_lens)
        cu_q = np.concatenate([[0], np.cumsum(self.q_lens)]).astype(np.int32)
        cu_k = np.concatenate([[0], np.cumsum(self.k_lens)]).astype(np.int32)
        self.cu_q_host, self.cu_k_host = cu_q, cu_k
        self.tq, self.tk = int(cu_q[-1]), int(cu_k[-1])

        def blocks(lens):
            seg, r0 = [], []
            for s, n in enumerate(lens):
                for b in range(0, max(n, 1), self.BLOCK):
                    seg.append(s); r0.append(b)
            return np.asarray(seg, np.int32), np.asarray(r0, np.int32)

        qs, qr = blocks(self.q_lens)
        ks, kr = blocks(self.k_lens)
        up = lambda a: torch.from_numpy(a).to(device)
        self.cu_q, self.cu_k = up(cu_q), up(cu_k)
        self.qblk_seg, self.qblk_r0, self.kblk_seg, self.kblk_r0 = up(qs), up(qr), up(ks), up(kr)
        self.nqblk, self.nkblk = len(qs), len(ks)



_uniform_cache = {}


def uniform_segments(B: int, N: int, device) -> "Segments":
    """B segments of N tokens each (a fixed-length batch seen as a packed one); cached per (B, N, device)."""
    key = (B, N, str(device))
    s = _uniform_cache.get(key)
    if s is None:
        if len(_uniform_cache) > 16:
            _uniform_cache.clear()
        s = _uniform_cache[key] = Segments([N] * B, [N] * B, device)
    return s
