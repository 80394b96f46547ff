"""Multi-rank path on CPU: world_size-2 gloo processes shard frames i mod G, reduce the integer
PSNR statistics, and together reproduce the single-process result exactly.  The frame processor
here is the oracle (no GPU in this test); on the GPU box the same function runs with the engine."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from qcnn_gpu_amd import sharding, synth  # noqa: E402

H, W, QP, NF = 24, 40, 27, 5


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle
    mb = synth.synth_model(QP).to_nchw_vect_c_bytes()
    ori, anc = synth.synth_frames(H, W, QP, frames=NF)
    recs, pf, p1, p2 = sharding.run_sharded(lambda i: anc[i], lambda i: ori[i], NF,
                                            lambda y: c_oracle.forward_blu(mb, y), rank, world, dist)
    q.put((rank, {k: v.tobytes() for k, v in recs.items()}, pf, p1, p2))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    from oracle import c_oracle
    mb = synth.synth_model(QP).to_nchw_vect_c_bytes()
    ori, anc = synth.synth_frames(H, W, QP, frames=NF)
    recs1, pf1, p1a, p1b = sharding.run_sharded(lambda i: anc[i], lambda i: ori[i], NF,
                                                lambda y: c_oracle.forward_blu(mb, y))
    assert sorted(recs1) == list(range(NF))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    owned = {}
    for rank, recs, pf, pa, pb in got:
        assert sorted(recs) == list(sharding.frames_for_rank(NF, rank, 2))      # frame i -> rank i mod 2
        owned.update(recs)
        assert pf == pf1 and pa == p1a and pb == p1b        # reduced statistics identical on every rank
    for i in range(NF):
        assert owned[i] == recs1[i].tobytes()                # bytewise equal to the 1-rank run
    assert p1b == c_oracle.psnr(np.stack([recs1[i] for i in range(NF)]), ori)


def test_partition_properties():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            parts = [list(sharding.frames_for_rank(n, r, world)) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        sharding.frames_for_rank(4, 2, 2)
