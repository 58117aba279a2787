"""world_size-2 and world_size-8 gloo tests of the multi-GPU leg: contiguous image sharding (incl. the remainder shard of the last
rank and a rank without a single detection) + the final detection gather."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crowdsam_amd.distributed import detections_to_rows, gather_rows, shard_range
    n_images = 7 if world == 2 else 19
    s, e = shard_range(n_images, rank, world)
    nd = lambda i: 0 if (world == 8 and rank == 3) else i % 3        # rank 3 of 8: no detection on any of its images
    rows = [detections_to_rows(i, np.full((nd(i), 4), i, np.float32), np.full(nd(i), 0.1 * i, np.float32))
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


def test_shard_and_gather_eight_ranks_remainder_and_empty_rank():
    """tools/batch_eval.py:80-89 at -n 8: 19 images -> 2 per rank, the last rank takes 5 (the remainder shard); rank 3 gathers
    zero rows.  Every rank must end up with all rows in image order."""
    from crowdsam_amd.distributed import shard_range
    assert shard_range(19, 7, 8) == (14, 19) and shard_range(19, 0, 8) == (0, 2)
    assert shard_range(5, 7, 8) == (0, 5) and shard_range(5, 0, 8) == (0, 0)      # fewer images than ranks: all on the last
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(8)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert [r[1] for r in res] == [(2 * r, 2 * r + 2) for r in range(7)] + [(14, 19)]
    nd = lambda i: 0 if i in (6, 7) else i % 3
    expect = np.concatenate([np.concatenate([np.full((nd(i), 1), i), np.full((nd(i), 4), i), np.full((nd(i), 1), 0.1 * i)], 1)
                             for i in range(19)]).astype(np.float32)
    for _, _, allr in res:
        np.testing.assert_allclose(allr, expect)
