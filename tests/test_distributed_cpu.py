"""world_size-2 gloo test of the multi-GPU leg: contiguous image sharding + the final detection gather."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crowdsam_amd.distributed import detections_to_rows, gather_rows, shard_range
    n_images = 7
    s, e = shard_range(n_images, rank, world)
    rows = [detections_to_rows(i, np.full((i % 3, 4), i, np.float32), np.full(i % 3, 0.1 * i, np.float32))
            for i in range(s, e)]
    rows = np.concatenate(rows) if rows else np.zeros((0, 6), np.float32)
    allr = gather_rows(rows)
    q.put((rank, (s, e), allr))
    dist.destroy_process_group()


def test_shard_and_gather_two_ranks():
    from crowdsam_amd.distributed import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert res[0][1] == (0, 3) and res[1][1] == (3, 7)
    expect = np.concatenate([np.concatenate([np.full((i % 3, 1), i), np.full((i % 3, 4), i), np.full((i % 3, 1), 0.1 * i)], 1)
                             for i in range(7)]).astype(np.float32)
    for _, _, allr in res:
        np.testing.assert_allclose(allr, expect)
