"""Frame sharding across GPUs: frames are independent (no state crosses frames in
qvrcnn::forward_blu, inference/qvrcnn.cu:168-242), so frame i goes to rank i mod G and the data
path needs no collective.  The only exchange is the reduction of per-rank integer squared-error
sums for the whole-clip PSNR (inference/yuv_data.cpp:87-97) and the gather of per-frame PSNRs for
the psnr_pf log (inference/yuv_data.cpp:98-112)."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Tuple

import numpy as np

from .formats import psnr_from_sse


def frames_for_rank(nframes: int, rank: int, world: int) -> range:
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return range(rank, nframes, world)


def sse_u8(a: np.ndarray, b: np.ndarray) -> int:
    d = a.astype(np.int64).ravel() - b.astype(np.int64).ravel()
    return int((d * d).sum())


def run_sharded(anchor: Callable[[int], np.ndarray], original: Callable[[int], np.ndarray], nframes: int,
                process: Callable[[np.ndarray], np.ndarray], rank: int = 0, world: int = 1,
                dist=None) -> Tuple[Dict[int, np.ndarray], List[float], float, float]:
    """Processes this rank's frames with `process` (one Y frame in, one reconstructed frame out) and
    returns (recons of this rank, per-frame after-net PSNR of ALL frames, clip PSNR before, clip PSNR
    after).  With dist (torch.distributed, initialised) the PSNR statistics are reduced over ranks."""
    recs: Dict[int, np.ndarray] = {}
    per_frame = np.zeros((nframes, 3), np.float64)     # sse_before, sse_after, pixels  (exact below 2^53)
    for i in frames_for_rank(nframes, rank, world):
        y, o = anchor(i), original(i)
        r = process(y)
        recs[i] = r
        per_frame[i] = (sse_u8(y, o), sse_u8(r, o), o.size)
    if dist is not None and world > 1:
        import torch
        t = torch.from_numpy(per_frame)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)       # every frame is owned by exactly one rank
        per_frame = t.numpy()
    pf = [psnr_from_sse(int(s), int(n)) for _, s, n in per_frame]
    tot = per_frame.sum(0)
    return recs, pf, psnr_from_sse(int(tot[0]), int(tot[2])), psnr_from_sse(int(tot[1]), int(tot[2]))
