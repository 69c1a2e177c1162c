"""world_size-2 gloo tests of the data-parallel plumbing (the same classes drive RCCL on the GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from sed_crnn_amd.dist import BucketedAllReduce, broadcast_parameters, init_from_env, shard_batch
    import sed_crnn_amd as sed
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # different init per rank ...
    m = sed.TimePooledCRNN(conv_channels=8, dropout=0.0)
    broadcast_parameters(m)                             # ... made identical by the broadcast
    ref = [torch.zeros_like(m.flat_parameters()) for _ in range(world)]
    dist.all_gather(ref, m.flat_parameters())
    same = all(torch.equal(ref[0], t) for t in ref)
    # staged all-reduce of the flat gradient arena, slices in backward-completion order
    g = m.flat_grads()
    red = BucketedAllReduce(g, m.bucket_slices())
    expect = torch.zeros_like(g)
    for s, (a, b) in enumerate(m.bucket_slices()):
        g[a:b] = float(rank + 1) * (s + 1)              # "stage s finished on this rank"
        expect[a:b] = (s + 1) * sum(range(1, world + 1)) / world
        red.launch(s)
    red.wait_all()
    ok = torch.allclose(g, expect)
    # batch sharding
    x = torch.arange(8 * 3).reshape(8, 3)
    sh = shard_batch(x, rank, world)
    shard_ok = sh.shape[0] == 8 // world and int(sh[0, 0]) == rank * (8 // world) * 3
    try:
        shard_batch(torch.zeros(7, 1), rank, world)
        raised = False
    except ValueError:
        raised = True
    q.put((rank, same, ok, shard_ok, raised))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gloo_world2_bucketed_allreduce_and_broadcast():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in ps:
        p.join(30)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert r[1:] == (True, True, True, True), r
