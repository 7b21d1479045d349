"""World-size 2, 3 and 8 checks of the multi-GPU path on CPU (gloo): sample-range split + one reduce of the float4
accumulation buffers. The per-rank render is done by the oracle here (no GPU in this process); on the GPU
box bench.py runs the same split and reduce with the HIP renderer and the nccl (RCCL) backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH = 32, 18, 7, 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    import oracle
    from ray_tracing_amd.distributed import reduce_accum, sample_range, weak_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    begin, count = sample_range(SPP, world, rank) if mode == "strong" else weak_range(SPP, rank)
    r = oracle.OracleRenderer("cornell", W, H)
    sums = np.zeros((H, W, 3), dtype=np.float32)
    if count:
        sums, _ = r.render_sums(count, DEPTH, seed=4, spp_begin=begin, threads=1, total_spp=SPP if mode == "strong" else world * SPP)
    acc = np.concatenate([sums, np.full((H, W, 1), count, dtype=np.float32)], axis=2)
    t = torch.from_numpy(acc.reshape(-1).copy())
    reduce_accum(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy().reshape(H, W, 4))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_range_partition():
    from ray_tracing_amd.distributed import sample_range, weak_range
    for total in (1, 7, 1024, 16384):
        for world in (1, 2, 3, 8):
            parts = [sample_range(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    assert weak_range(1024, 3) == (3072, 1024)
    with pytest.raises(ValueError):
        sample_range(4, 2, 2)


@pytest.mark.parametrize("mode,world", [("strong", 2), ("weak", 2), ("strong", 3), ("weak", 3), ("strong", 8)])
def test_ranks_reduce_equals_single_render(tmp_path, mode, world):
    """World sizes 2, 3 (uneven split: 7 samples over 3 ranks) and 8 (more ranks than samples: one rank renders
    nothing and still takes part in the reduce)."""
    sys.path.insert(0, ROOT)
    import oracle
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    got = np.load(out)
    total = SPP if mode == "strong" else world * SPP
    want, _ = oracle.OracleRenderer("cornell", W, H).render_sums(total, DEPTH, seed=4)
    assert (got[..., 3] == total).all()
    np.testing.assert_allclose(got[..., :3], want, rtol=1e-5, atol=1e-6)
