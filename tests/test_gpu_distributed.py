"""Data-parallel fit step on the GPU with two ranks (gloo transport, both on cuda:0): the staged all-reduce of
FusedTrainStep must equal "every rank back-propagates its own shard, gradients are averaged, Adam is applied".
RCCL itself only runs in the driver's multi-GPU bench; the bucket / stage logic is the same code."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    import sed_crnn_amd as sed
    from oracle import crnn_ref
    from sed_crnn_amd.dist import broadcast_parameters, shard_batch
    from sed_crnn_amd.trainer import FusedTrainStep
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(5 + rank)
    m = sed.TimePooledCRNN(conv_channels=32, dropout=0.0, gru_hidden=32).cuda()
    broadcast_parameters(m)
    x, y = crnn_ref.synthetic_batch(8, 1, 40, 64, 8, seed=21)
    xs, ys = shard_batch(x, rank, world).cuda(), shard_batch(y, rank, world).cuda()
    step = FusedTrainStep(m, lr=1e-3, loss="bce", clip_norm=1.0)
    assert step.reducer is not None
    p0 = m.flat_parameters().clone()
    loss, _ = step.step(xs, ys)
    torch.cuda.synchronize()
    # reference: local gradients of every rank (recomputed here), averaged by hand
    grads = []
    for r in range(world):
        mm = sed.TimePooledCRNN(conv_channels=32, dropout=0.0, gru_hidden=32).cuda()
        mm.flat_parameters().copy_(p0)
        s1 = FusedTrainStep(mm, lr=0.0, loss="bce", distributed=False)
        s1.step(shard_batch(x, r, world).cuda(), shard_batch(y, r, world).cuda())
        grads.append(mm.flat_grads().clone())
    gavg = sum(grads) / world
    ok_grad = torch.allclose(m.flat_grads(), gavg, atol=1e-6, rtol=1e-5)
    ref = sed.TimePooledCRNN(conv_channels=32, dropout=0.0, gru_hidden=32).cuda()
    ref.flat_parameters().copy_(p0)
    ref.flat_grads().copy_(gavg)
    ref.bind_flat_grads()
    opt = sed.FusedAdam(ref.parameters(), lr=1e-3, max_grad_norm=1.0).attach(ref)
    opt.step()
    ok_param = torch.allclose(m.flat_parameters(), ref.flat_parameters(), atol=1e-7, rtol=1e-6)
    allp = [torch.zeros_like(p0) for _ in range(world)]
    dist.all_gather(allp, m.flat_parameters())
    in_sync = all(torch.equal(allp[0], t) for t in allp)
    q.put((rank, bool(ok_grad), bool(ok_param), bool(in_sync), float(loss.item())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_fused_step_equals_manual_gradient_average():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in ps:
        p.join(30)
    for r in res:
        assert r[1] and r[2] and r[3], r


def _worker_syncbn(rank, world, port, q, channels=32):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    import sed_crnn_amd as sed
    from oracle import crnn_ref
    from sed_crnn_amd.dist import broadcast_parameters, shard_batch
    from sed_crnn_amd.trainer import FusedTrainStep
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(31)
    kw = dict(conv_channels=channels, dropout=0.0, gru_hidden=16)      # 128: blocks 2 / 3 run as Winograd kernels, phase by phase
    m = sed.TimePooledCRNN(**kw).cuda().enable_sync_bn()
    m.plan_flags = int(os.environ.get("SED_TEST_PLAN_FLAGS", "0"))
    broadcast_parameters(m)
    x, y = crnn_ref.synthetic_batch(8, 1, 40, 32, 4, seed=77)
    p0 = m.flat_parameters().clone()
    step = FusedTrainStep(m, lr=1e-3, loss="bce")
    loss, probs = step.step(shard_batch(x, rank, world).cuda(), shard_batch(y, rank, world).cuda())
    torch.cuda.synchronize()
    # single-device run on the WHOLE batch (plain BatchNorm): must be what the synchronised 2-rank run computed
    ref = sed.TimePooledCRNN(**kw).cuda()
    ref.plan_flags = int(os.environ.get("SED_TEST_PLAN_FLAGS", "0"))
    ref.flat_parameters().copy_(p0)
    rstep = FusedTrainStep(ref, lr=1e-3, loss="bce", distributed=False)
    rloss, rprobs = rstep.step(x.cuda(), y.cuda())
    torch.cuda.synchronize()
    per = 8 // world
    ok_probs = torch.allclose(probs, rprobs[rank * per:(rank + 1) * per], atol=1e-5, rtol=1e-4)
    ok_grad = torch.allclose(m.flat_grads(), ref.flat_grads(), atol=2e-5, rtol=2e-3)
    if not ok_grad and channels == 128:
        # 1.3 M pre-pool elements per block at this width: the two runs' BatchNorm coefficients differ in the last bit (the statistics
        # are summed in a different order), and an element whose z = scale y + shift is within that bit of 0 takes the other side of
        # the ReLU gate — one element of one channel then moves that channel's sums (seen here: channel 121 of the first block,
        # d(beta) -9.79e-4 against -1.022e-3, every other channel equal to 1e-8).  Such a run must still agree in every bucket to
        # 2e-3 relative L2, with the per-entry misses confined to a handful of entries.
        g1, g0 = m.flat_grads(), ref.flat_grads()
        bad = ((g1 - g0).abs() > 2e-5 + 2e-3 * g0.abs())
        rels = [float((g1[lo:hi] - g0[lo:hi]).norm() / g0[lo:hi].norm()) for lo, hi in m.bucket_slices() if hi > lo]
        print(f"rank {rank}: {int(bad.sum())} entries beyond the per-entry bound, relative L2 per bucket {['%.1e' % r_ for r_ in rels]}", flush=True)
        ok_grad = int(bad.sum()) <= 24 and max(rels) < 2e-3
    # Adam's first step moves every weight by lr*sign(g): entries whose gradient is rounding noise (the conv biases in
    # front of BatchNorm, analytically zero) are excluded, everything else must land on the same value
    sig = ref.flat_grads().abs() > 1e-6
    ok_param = torch.allclose(m.flat_parameters()[sig], ref.flat_parameters()[sig], atol=2e-5, rtol=1e-3) and int(sig.sum()) > 1000
    sd, rsd = m.state_dict(), ref.state_dict()
    ok_running = all(torch.allclose(sd[k], rsd[k], atol=1e-6, rtol=1e-5) for k in sd if "running" in k)
    q.put((rank, bool(ok_probs), bool(ok_grad), bool(ok_param), bool(ok_running)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("channels", [32, 128])
def test_two_rank_sync_bn_equals_single_device_on_the_global_batch(channels):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_syncbn, args=(r, world, port, q, channels)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in ps:
        p.join(30)
    for r in res:
        assert all(r[1:]), r


def _worker_rccl_single(port, q):
    """the RCCL calls of the N>1 bench path (init, parameter broadcast, staged AVG all-reduce on the gradient arena with the
    aux-stream backward, device-id barrier, MAX all-reduce of the timing) on a world of one rank: AVG over one rank is the
    identity, so the run must be bit-identical to the non-distributed step"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import sed_crnn_amd as sed
    from oracle import crnn_ref
    from sed_crnn_amd.dist import broadcast_parameters
    from sed_crnn_amd.trainer import FusedTrainStep
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    kw = dict(conv_channels=32, dropout=0.5, gru_hidden=32)
    torch.manual_seed(3)
    m = sed.TimePooledCRNN(**kw).cuda()
    broadcast_parameters(m)
    ref = sed.TimePooledCRNN(**kw).cuda()
    ref.flat_parameters().copy_(m.flat_parameters())
    x, y = crnn_ref.synthetic_batch(8, 1, 40, 64, 8, seed=9)
    x, y = x.cuda(), y.cuda()
    a = FusedTrainStep(m, lr=1e-3, loss="bce", clip_norm=1.0, distributed=True)
    b = FusedTrainStep(ref, lr=1e-3, loss="bce", clip_norm=1.0, distributed=False)
    assert a.reducer is not None and b.reducer is None
    for _ in range(3):
        la, _ = a.step(x, y)
        lb, _ = b.step(x, y)
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    t = torch.tensor([1.25], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    same_p = torch.equal(m.flat_parameters(), ref.flat_parameters())
    same_g = torch.equal(m.flat_grads(), ref.flat_grads())
    q.put((bool(same_p), bool(same_g), float(la.item()), float(lb.item()), float(t.item())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_single_rank_staged_allreduce_is_identity():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl_single, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(30)
    assert res[0] and res[1] and res[2] == res[3] and res[4] == 1.25, res


@pytest.mark.timeout(600)
def test_bench_launches_two_ranks_and_runs_the_staged_step_on_the_hip_path():
    """`python bench.py --gpus 2` as the driver calls it (no torchrun): the launcher starts two rank processes that run the
    REAL fit step (HIP kernels, staged all-reduce of the arena between backward stages, max-over-ranks clock).  Only one GPU
    is available here, so both ranks share cuda:0 over gloo (`--share-gpu`, a rehearsal: the line says so); on a multi-GPU
    node the same command without those two flags runs one rank per GPU over RCCL."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 256 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp2-rehearsal-one-gpu"
    assert out["value"] > 0 and 0.2 < out["config"]["final_loss"] < 1.5
    assert out["roofline"]["launches"] == 3 * 4            # rank 0's conv forward / data-gradient launches in the timed steps
    # the self-diagnosis of a scaling run: every rank's own median step time and the part of the gradient all-reduce that its
    # backward did not hide (round-3 verdict item 9: the first hardware SCALE line must say where a loss comes from)
    rk = out["ranks"]
    assert len(rk["per_rank_ms_per_step_median"]) == 2 and all(v > 0 for v in rk["per_rank_ms_per_step_median"])
    assert rk["ms_per_step_median_min_rank"] <= rk["ms_per_step_median_max_rank"]
    assert 0.0 <= rk["allreduce_exposed_ms_avg"] <= rk["allreduce_exposed_ms_max_rank"] < 1e4
    assert "eval_forward" in out and out["eval_forward"]["ms"] > 0
