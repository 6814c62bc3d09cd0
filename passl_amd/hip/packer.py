"""WeightPacker: one-launch repacking of the fp32 master weights (flat buffer, each conv weight
physically [K][R][S][C]) into the compute-dtype operand buffers the implicit-GEMM kernels read
(forward [K][R][S][C], data-gradient [C][TR][TS][K] per residue class, stem padding)."""
import numpy as np
import torch

from . import lib as L
from . import ops

_BLOCK_ELEMS = 1024


class WeightPacker:
    def __init__(self):
        self._jobs = []          # (src_off, K, R, S, C, Pack)
        self._size = 0
        self._built = False

    def add(self, src_off, K, R, S, Cc, pack):
        """Register a job; returns its element offset in the packed buffer (16-byte aligned for
        both dtypes)."""
        assert not self._built
        off = self._size
        pack.dst_off = off
        self._jobs.append((int(src_off), K, R, S, Cc, pack))
        self._size += (pack.size + 7) // 8 * 8
        return off

    @property
    def size(self):
        return self._size

    def build(self, device, dtype):
        n = len(self._jobs)
        arr = (L.PackJob * max(n, 1))()
        bj, bs = [], []
        for i, (src_off, K, R, S, Cc, p) in enumerate(self._jobs):
            j = arr[i]
            j.src_off, j.dst_off = src_off, p.dst_off
            j.K, j.R, j.S, j.C = K, R, S, Cc
            j.TR, j.TS = p.TR, p.TS
            j.r_base, j.r_step, j.s_base, j.s_step = p.r_base, p.r_step, p.s_base, p.s_step
            j.transpose, j.c_pad = p.transpose, p.c_pad
            # every tap kept, in order (the step is irrelevant for a single tap)
            full = (p.TR == R and p.TS == S and p.r_base == 0 and (R == 1 or p.r_step == 1) and
                    p.s_base == 0 and (S == 1 or p.s_step == 1))
            if p.transpose and R == 1 and S == 1 and full and p.c_pad in (0, K) and p.size == K * Cc:
                j.transpose = 2            # tiled 2-D transpose: blocks enumerate 32 x 32 tiles
                for t in range(((K + 31) // 32) * ((Cc + 31) // 32)):
                    bj.append(i)
                    bs.append(t)
                continue
            if not p.transpose and full and p.c_pad in (0, Cc) and p.size == K * R * S * Cc and src_off % 4 == 0:
                j.transpose = 3            # contiguous cast
            for start in range(0, p.size, _BLOCK_ELEMS):
                bj.append(i)
                bs.append(start)
        raw = np.frombuffer(memoryview(arr), dtype=np.uint8).copy()
        self.jobs_dev = torch.from_numpy(raw).to(device)
        self.block_job = torch.tensor(bj, dtype=torch.int32, device=device)
        self.block_start = torch.tensor(bs, dtype=torch.int32, device=device)
        self.n_blocks = len(bj)
        self.buffer = torch.zeros(max(self._size, 8), dtype=dtype, device=device)
        self._built = True
        return self

    def view(self, pack, rows):
        """2-D view [rows, size/rows] of one packed operand."""
        return self.buffer[pack.dst_off:pack.dst_off + pack.size].view(rows, -1)

    def run(self, src_flat):
        assert self._built
        if self.n_blocks:
            ops.pack_weights(src_flat, self.buffer, self.jobs_dev, self.block_job,
                             self.block_start, self.n_blocks)
