"""Multi-rank path on CPU: GOF sharding + gather of the re-encoded sub-bitstreams with world_size 2 over
gloo. The transcode itself is replaced by a tagging function here (no GPU in this container); the N-GPU path in bench.py
uses the same gather with backend nccl (RCCL)."""
import os
import sys
import torch.multiprocessing as mp
import rbt_lib


def _worker(rank, world, port, n_gofs, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gs = rbt_lib.module_file("gof_shard")
    mine = gs.gofs_of_rank(n_gofs, rank, world)
    local = []
    for g in mine:      # three sub-bitstreams per GOF, variable length, content identifies (gof, stream)
        for s in range(3):
            local.append(bytes([g, s]) * (10 + 7 * g + s))
    gathered = gs.gather_streams(local)
    if rank == 0:
        q.put(gs.stitch(gathered, n_gofs, 3))
    dist.barrier()
    dist.destroy_process_group()


def test_gof_sharding_and_gather_world2():
    gs = rbt_lib.module_file("gof_shard")
    assert gs.gofs_of_rank(10, 0, 8) == [0, 8] and gs.gofs_of_rank(10, 1, 8) == [1, 9] and gs.gofs_of_rank(10, 7, 8) == [7]
    assert sorted(sum((gs.gofs_of_rank(10, r, 3) for r in range(3)), [])) == list(range(10))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_gofs, world, port = 5, 2, 29517
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_gofs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for g in range(n_gofs):
        for s in range(3):
            assert res[g][s] == bytes([g, s]) * (10 + 7 * g + s)
