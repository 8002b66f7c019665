"""N>1 path on CPU: world_size-2 gloo run of the end-of-run throughput reduction (the only collective of the path;
on the GPUs the same code runs over RCCL/xGMI) and of the sequence sharding."""
import os
import socket

import torch.multiprocessing as mp

from ov2slam_amd import dist_util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    assert dist_util.init_from_env("gloo")
    el, cnt = dist_util.aggregate(1.0 + rank, [100 * (rank + 1), 7, rank], device="cpu")
    q.put((rank, el, cnt, dist_util.shard_sequences(8, rank, world)))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, el, cnt, shard in res:
        assert el == 2.0                      # MAX over ranks
        assert cnt == [300.0, 14.0, 1.0]      # SUM over ranks
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5, 7]


def test_single_rank_is_identity():
    el, cnt = dist_util.aggregate(0.5, [3, 4])
    assert el == 0.5 and cnt == [3.0, 4.0]
    assert dist_util.shard_sequences(8, 0, 1) == list(range(8))
