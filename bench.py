#!/usr/bin/env python3
"""bench.py — mel-frames/s of the SEDnet fit step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step is one full fit step of the hot path on one resident synthetic batch: forward (train-mode
BatchNorm, dropout 0.5) + BCEWithLogits + backward + (gradient all-reduce over RCCL) + Adam, i.e.
reference sed.py:134-137.  Workload = BASELINE config 2 per GPU (mono, B=128, 256 frames x 40 mel,
3x conv128 + BiGRU 2x128), weak scaling: every rank trains its own 128-sample shard of the global
batch, gradients averaged.  Prints ONE JSON line on rank 0.

Launching.  One process per GPU.  Three ways in, one code path (`run_rank`):
  * `python bench.py --gpus 1`              -> this process is rank 0 of 1;
  * under torchrun (RANK/WORLD_SIZE set)     -> this process is the rank torchrun says;
  * `python bench.py --gpus N`, N > 1, no torchrun environment -> this process is only a LAUNCHER: before any GPU
    call it starts N fresh children of this same script with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, forwards rank 0's
    JSON line and exits with the worst child code.  It never touches the GPU itself and never re-execs.
`n_gpus` in the JSON line is the world size the process group itself reports (all-reduce of ones), not the flag.

`--plumbing-only` (used by the CPU tests, gloo): launch, rendezvous, parameter broadcast, the staged all-reduce of the
flat gradient arena and the max-over-ranks clock — everything of the N-rank path except the HIP kernels.  Its JSON line
says `"metric": "plumbing-only"` and carries no throughput, so it can never be read as a bench result.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(B=128, Cin=1, F=40, T=256, C=128, H=128, gru_layers=2, dropout=0.5)
F32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)
PROFILE_ROUNDS = ("r4", "r3", "r2", "r1")   # newest committed PMC summary first

# the dominant kernel: conv2 / conv3 forward (<14, false>) and their data gradients (<14, true, *>: the same main loop with the
# BatchNorm-backward reduction of the block below in its epilogue); one in-library timer tag covers both.  Since round 4 it is the
# Winograd F(2x2,3x3) form (wino.hip): 16 MFMA products per 36 algorithmic multiply-adds.  --direct-conv runs the direct kernels.
DOMINANT_KERNEL = "conv3x3_wino_k"
DOMINANT_PREFIX = "void conv3x3_wino_k<14"             # every instantiation: forward, data gradient (+ fused sums)
DIRECT_KERNEL = "conv3x3_mfma_fwd2_k"
DIRECT_PREFIX = "void conv3x3_mfma_fwd2_k<4, 2"
WINO_EXECUTED = 16.0 / 36.0                            # executed / algorithmic multiply-adds of F(2x2, 3x3)


def pmc_traffic_bytes(prefix=None):
    """HBM traffic per launch of the dominant kernel, from the committed PMC summary (collected as the
    MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes; FETCH_SIZE counts half of a wide
    coalesced read on gfx950, verified here on the pure-streaming bn kernel), launch-weighted over the kernel's
    instantiations.  (None, None) when no summary is committed."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, "pmc_fetch_write_per_kernel.json")
        try:
            d = json.load(open(path))
        except Exception:
            continue
        tot, n = 0.0, 0
        for name, e in d.items():
            if name.startswith(prefix or DOMINANT_PREFIX) and "false, 0, true" not in name and "FETCH_SIZE_KB_avg" in e and "WRITE_SIZE_KB_avg" in e:      # (not the inference instantiation)
                k = int(e.get("launches", 1))
                tot += k * (2.0 * e["FETCH_SIZE_KB_avg"] + e["WRITE_SIZE_KB_avg"]) * 1024
                n += k
        if n:
            return int(tot / n), rnd
    return None, None


def forward_flops_per_batch(w):
    """algorithmic FLOPs (2 x MAC) of one forward pass of the workload's net over one batch, SURVEY 8(d): conv_l =
    2*9*Cin_l*C*F*T_l, GRU input projection 2*2*T'*3H*in_l, recurrence 2*2*T'*3H*H per layer, head 2*T'*2H"""
    B, F, T, C, H = w["B"], w["F"], w["T"], w["C"], w["H"]
    fl, cin, t = 0.0, w["Cin"], T
    for _ in range(3):
        fl += 2.0 * 9 * cin * C * F * t
        cin, t = C, t // 2
    tp, inp = t, C * F
    for _ in range(w["gru_layers"]):
        fl += 2.0 * 2 * tp * 3 * H * inp + 2.0 * 2 * tp * 3 * H * H
        inp = 2 * H
    fl += 2.0 * tp * 2 * H
    return fl * B


# ───────────────────────── CPU baseline (BASELINE.md §2 / SURVEY §8d) ─────────────────────────
def host_cpu_info():
    """What this process may actually use: scheduler affinity, cgroup CPU quota, physical cores among the allowed CPUs."""
    logical = os.cpu_count() or 1
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(logical))
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        pass
    model, cores = "unknown", set()
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = (s.strip() for s in line.split(":", 1))
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    model = cur.get("model name", model)
                cur = {}
    except Exception:
        pass
    physical = len(cores) or len(allowed)
    threads = max(1, min(len(allowed), physical, int(quota) if quota and quota >= 1 else len(allowed)))
    return dict(cpu_model=model, logical_cpus=logical, allowed_cpus=len(allowed), cgroup_quota_cpus=quota,
                physical_cores_allowed=physical, threads=threads)


def cpu_baseline(steps=5):
    """The oracle's fit step (torch.nn CPU restatement of sed.py, pinned by tests/golden) on the host cores: zero_grad +
    forward + BCEWithLogits + backward + Adam(1e-3), dropout active, BN in train mode; median of `steps` after 1 warm-up.
    Headline = c1 (the reference's own net, C=128 / GRU 2x32, B=16 x 256 frames: BASELINE.md §2); also the bench
    workload's net (c2: GRU 2x128) at the same B=16, so the GPU line has a like-for-like row."""
    import torch
    from oracle import crnn_ref
    info = host_cpu_info()
    torch.set_num_threads(info["threads"])

    def timed(H, B):
        torch.manual_seed(0)
        w = WORKLOAD
        net = crnn_ref.SedNetRef(conv_channels=w["C"], dropout=w["dropout"], in_channels=w["Cin"], n_mels=w["F"],
                                 gru_hidden=H, gru_layers=w["gru_layers"])
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        x, y = crnn_ref.synthetic_batch(B, w["Cin"], w["F"], w["T"], w["T"] // 8, seed=1234)
        crnn_ref.fit_step(net, opt, x, y)                       # warm-up
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            crnn_ref.fit_step(net, opt, x, y)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        return B * w["T"] / med, med

    c1, c1_med = timed(32, 16)
    c2, c2_med = timed(WORKLOAD["H"], 16)
    return {"value": round(c1, 1), "unit": "mel-frames/s", "cores": info["threads"], "kind": "port",
            "sample": f"c1 = reference net (C128, GRU 2x32) at B=16 x 256 frames: median of {steps} fit steps after 1 warm-up, "
                      f"{c1_med*1e3:.0f} ms/step, torch {torch.__version__} CPU with {info['threads']} threads",
            "c2_net": {"value": round(c2, 1), "unit": "mel-frames/s",
                       "sample": f"the bench workload's net (C128, GRU 2x128) at B=16 x 256 frames, same protocol, {c2_med*1e3:.0f} ms/step"},
            "host": info}


# ───────────────────────── launcher (N > 1 without torchrun) ─────────────────────────
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Start n fresh rank processes of this script and forward rank 0's output.  The launcher has not imported torch and
    makes no GPU call.  If one rank dies the others would wait in the rendezvous or in a collective until their timeout:
    the launcher ends exactly the processes it started (by handle, never by pattern) as soon as one of them has failed."""
    import tempfile
    port = _free_port()
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        failed = False
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                failed = True
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                for p in procs:
                    try:
                        p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                break
            time.sleep(0.05)
        codes = [p.wait() for p in procs]
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or failed:
        print(f"[bench] rank processes failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


# ───────────────────────── one rank ─────────────────────────
def run_rank(args):
    import torch
    import torch.distributed as dist
    import sed_crnn_amd as sed
    from sed_crnn_amd.dist import BucketedAllReduce, broadcast_parameters, init_from_env

    plumbing = args.plumbing_only
    backend = args.backend or ("gloo" if plumbing else "nccl")
    if not plumbing:
        want = int(os.environ.get("WORLD_SIZE", "1"))
        have = torch.cuda.device_count()                     # counting devices does not initialise the GPU
        if args.share_gpu:
            if backend != "gloo":
                raise SystemExit("[bench] --share-gpu is a rehearsal of the N-rank path on ONE GPU: it needs --backend gloo "
                                 "(RCCL does not put two ranks on one device) and its value is not a scaling number")
            os.environ["LOCAL_RANK"] = "0"
        elif have < want:
            raise SystemExit(f"[bench] {want} ranks requested but only {have} GPU(s) visible: refusing to share a GPU")
    rank, world, local = init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the process group has {world} ranks")
    on_gpu = not plumbing
    if on_gpu:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    w = dict(WORKLOAD)
    if plumbing:
        w.update(C=8, H=8)                                    # a small arena: this mode measures nothing
    torch.manual_seed(0 if not plumbing else rank)           # plumbing: prove the broadcast by starting different
    model = sed.TimePooledCRNN(conv_channels=w["C"], dropout=w["dropout"], in_channels=w["Cin"], n_mels=w["F"],
                               gru_hidden=w["H"], gru_layers=w["gru_layers"]).to(dev)
    if world > 1:
        broadcast_parameters(model)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                 # the world size as the collective library sees it
        world_seen = int(ones.item())
    else:
        world_seen = 1

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local]) if on_gpu else dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    base = {"n_gpus": world_seen, "steps": args.steps, "warmup": args.warmup}
    if plumbing:
        red = BucketedAllReduce(model.flat_grads(), model.bucket_slices()) if world > 1 else None
        g = model.flat_grads()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.fill_(float(rank + 1))
            if red is not None:
                for s in range(len(model.bucket_slices())):
                    red.launch(s)
                red.wait_all()
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        expect = sum(range(1, world + 1)) / world
        ok = bool(torch.allclose(g, torch.full_like(g, expect)))
        psum = torch.tensor([float(model.flat_parameters().double().sum())], dtype=torch.float64)
        lo, hi = psum.clone(), psum.clone()
        if world > 1:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps(dict(base, metric="plumbing-only", value=None, backend=backend, allreduce_avg_ok=ok,
                                  params_in_sync=bool(lo.item() == hi.item()), ms_per_step=round(dt.item() / args.steps * 1e3, 4),
                                  config={"parallelism": f"dp{world_seen}"})), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0 if ok else 1

    import ctypes as C
    from sed_crnn_amd import _lib
    from sed_crnn_amd.trainer import FusedTrainStep
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(w["B"], w["Cin"], w["F"], w["T"], generator=g).to(dev)
    y = (torch.rand(w["B"], w["T"] // 8, 1, generator=g) > 0.8).float().to(dev)
    if args.conv_bf16x3:
        model.set_conv_precision("bf16x3")
    if args.direct_conv:
        model.plan_flags = 0x4                               # SED_NET_DIRECT_CONV: the 36-product kernels (A/B against the Winograd default)
    step = FusedTrainStep(model, lr=1e-3, loss="bce")
    for _ in range(args.warmup):
        step.step(x, y)
    lib = _lib.lib()
    tag = 0                                                  # SED_K_CONV_MFMA_FWD: the dominant kernel
    # ── the timed region: exactly K steps between two barriers, NOTHING instrumented inside except one event record per
    # step on the compute stream (a host-side enqueue of ~1 us; it gives the per-step GPU times for the median) ──
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        loss, _ = step.step(x, y)
    marks[args.steps].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    med_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    # ── the same K steps once more, instrumented: hipEvent pairs around every launch of the dominant kernel on its launch
    # stream (sed_prof_*), and around the all-reduce wait of every step.  Kept OUT of the timed region (round-2 verdict,
    # item 9); `ms_per_step_instrumented` beside `ms_per_step` shows what the instrumentation costs. ──
    tag_dg = 12                                              # SED_K_CONV_MFMA_DGRAD: the same kernel's data-gradient instantiation
    lib.sed_prof_enable((1 << tag) | (1 << tag_dg))
    step.time_allreduce = world > 1
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step.step(x, y)
    barrier()
    dt_inst = time.perf_counter() - t1
    ms_f, n_f, units_f = C.c_double(), C.c_long(), C.c_double()
    ms_d, n_d, units_d = C.c_double(), C.c_long(), C.c_double()
    lib.sed_prof_read(tag, C.byref(ms_f), C.byref(n_f), C.byref(units_f))
    lib.sed_prof_read(tag_dg, C.byref(ms_d), C.byref(n_d), C.byref(units_d))
    lib.sed_prof_enable(0)
    ms, n, units = (C.c_double(ms_f.value + ms_d.value), C.c_long(n_f.value + n_d.value),
                    C.c_double(units_f.value + units_d.value))           # every launch of the kernel, both instantiations
    step.time_allreduce = False
    ar_wait = [a.elapsed_time(b) for a, b in step.allreduce_events]
    step.allreduce_events.clear()
    # the same kernel with the GPU to itself (outside the timed region): in the fit step one of its four launches per step
    # (the data gradient of conv2) shares the CUs with the top block's weight gradient on the auxiliary stream, so its
    # duration above covers part of that kernel's work too; the forward launches of a forward-only pass (same two shapes,
    # in the same proportion) run alone
    ms_x, n_x, units_x = C.c_double(), C.c_long(), C.c_double()
    if rank == 0:
        model.train()
        lib.sed_prof_enable(1 << tag)
        for _ in range(8):
            model._run_forward(x, training=True)
        torch.cuda.synchronize()
        lib.sed_prof_read(tag, C.byref(ms_x), C.byref(n_x), C.byref(units_x))
        lib.sed_prof_enable(0)
    # ── inference: the eval-mode forward of the same net on the same resident batch (run_epoch with optim=None, sed.py:128-141:
    # BatchNorm on running statistics folded into the conv weights, ReLU + pool in the conv epilogue, no dropout), after the
    # timed region; SURVEY 8(d) "also report eval-forward frames/s" ──
    eval_fwd = None
    if rank == 0:
        model.eval()
        with torch.no_grad():
            for _ in range(3):
                model(x)
            em = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            em[0].record()
            for i in range(args.steps):
                model(x)
                em[i + 1].record()
            torch.cuda.synchronize()
        ems = sorted(em[i].elapsed_time(em[i + 1]) for i in range(args.steps))
        e_med = ems[len(ems) // 2]
        fl = forward_flops_per_batch(w)
        peak_e = F32_MFMA_PEAK_TFLOPS
        eval_fwd = {"ms": round(e_med, 4), "ms_min": round(ems[0], 4), "value": round(w["B"] * w["T"] / (e_med * 1e-3), 1), "unit": "mel-frames/s",
                    "batches": args.steps,
                    "roofline": {"bound": "mfma", "achieved": round(fl / (e_med * 1e-3) / 1e12, 2), "peak": peak_e, "unit": "TFLOP/s",
                                 "frac": round(fl / (e_med * 1e-3) / 1e12 / peak_e, 4), "flops_per_batch": fl,
                                 "ceiling_ms": round(fl / (peak_e * 1e12) * 1e3, 4),
                                 "note": "conv2 / conv3 run as Winograd F(2x2,3x3) with BatchNorm folded and ReLU + pool in the epilogue (16/36 of the algorithmic multiply-adds are executed, hence frac > 1); "
                                         "whole eval forward (one packing launch, 3 conv blocks, 2 BiGRU layers, head) against the fp32 "
                                         "MFMA peak: 88 % of its FLOPs are conv2/conv3, exact fp32"}}
        model.train()
    rank_stats = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        # per-rank view, so that a slow scaling run can be diagnosed from the one line: every rank's median step time on its
        # own GPU timeline and the part of the all-reduce its backward did not hide
        mine = torch.tensor([med_ms, sum(ar_wait) / max(len(ar_wait), 1), dt_inst], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        rank_stats = {"ms_per_step_median_min_rank": round(float(allr[:, 0].min()), 4),
                      "ms_per_step_median_max_rank": round(float(allr[:, 0].max()), 4),
                      "allreduce_exposed_ms_avg": round(float(allr[:, 1].mean()), 4),
                      "allreduce_exposed_ms_max_rank": round(float(allr[:, 1].max()), 4),
                      "per_rank_ms_per_step_median": [round(float(v), 4) for v in allr[:, 0]]}
        dt_inst = float(allr[:, 2].max())
    final_loss = loss.item()

    if args.breakdown and rank == 0:
        lib.sed_prof_enable(0xFFFF)
        for _ in range(2):
            step.step(x, y)
        torch.cuda.synchronize()
        tot = 0.0
        for k in range(lib.sed_prof_tag_count()):
            lib.sed_prof_read(k, C.byref(ms2 := C.c_double()), C.byref(n2 := C.c_long()), C.byref(u2 := C.c_double()))
            if n2.value:
                rate = u2.value / (ms2.value * 1e-3) / 1e12
                tot += ms2.value / 2
                print(f"[bench] {lib.sed_prof_tag_name(k).decode():24s} {ms2.value/2:8.3f} ms/step  {n2.value//2:3d} launches/step  "
                      f"{rate:8.2f} T(units)/s", file=sys.stderr)
        print(f"[bench] sum of instrumented kernels {tot:.3f} ms/step", file=sys.stderr)
        lib.sed_prof_enable(0)

    if rank == 0:
        frames = w["B"] * w["T"] * world_seen * args.steps
        out = dict(base)
        out.update({
            "metric": "mel-frames/sec training throughput (seq=256, mel=40)",
            "value": round(frames / dt, 1),
            "unit": "mel-frames/s",
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_step_median": round(med_ms, 4),
            "ms_per_step_instrumented": round(dt_inst / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if not args.conv_bf16x3 else "f32 with conv2/conv3 forward+dgrad+wgrad on bf16x3-split MFMA (experiment, not the headline)",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: mono (B={w['B']},256,40,1) per GPU, 3xConv3x3(128)+BN+ReLU+pool(1,2)+dropout0.5, "
                                   f"BiGRU 2x128, Linear(256,1), BCEWithLogits, Adam lr 1e-3; full fit step (fwd+loss+bwd+allreduce+Adam)",
                       "global_batch": w["B"] * world_seen, "seq_len": w["T"], "n_mels": w["F"],
                       "parallelism": (f"dp{world_seen}" if world_seen > 1 else "single") + ("-rehearsal-one-gpu" if args.share_gpu else ""),
                       "final_loss": round(final_loss, 6)},
        })
        out = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median",
                                   "ms_per_step_instrumented", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")}
        if rank_stats is not None:
            out["ranks"] = rank_stats
        if n.value:
            avg_ms = ms.value / n.value
            tf = units.value / (ms.value * 1e-3) / 1e12
            wino = not args.conv_bf16x3 and not args.direct_conv
            traffic, rnd = pmc_traffic_bytes(None if wino else DIRECT_PREFIX)
            peak = F32_MFMA_PEAK_TFLOPS if not args.conv_bf16x3 else 2500.0 / 3.0      # 3 bf16 MFMA flops per algorithmic flop
            kname = (DOMINANT_KERNEL + "<14, *> (Winograd F(2x2,3x3), exact-fp32 MFMA)" if wino else
                     DIRECT_KERNEL + "<4, 2, *>" if not args.conv_bf16x3 else "conv3x3_mfma_fwd_bf16x3_k<4, 2>")
            if args.conv_bf16x3:
                traffic = None
            out["roofline"] = {"bound": "mfma", "kernel": kname + " (conv2/conv3 forward + their data gradients)",
                               "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(tf / peak, 4), "traffic": traffic,
                               "traffic_source": f"committed profile (profiles/{rnd}), not measured in this run",
                               "traffic_note": f"HBM bytes per launch from a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass of this "
                                               f"command (profiles/{rnd}): (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 half-count correction",
                               "avg_launch_ms": round(avg_ms, 4), "launches": n.value,
                               "flops_per_launch_avg": units.value / n.value,
                               "measured_over": f"hipEvent pairs on the launch stream around every launch of this kernel in {args.steps} "
                                                "fit steps identical to, and run right after, the timed region (ms_per_step_instrumented)"}
            if wino:
                out["roofline"]["executed"] = {
                    "achieved": round(tf * WINO_EXECUTED, 2), "frac": round(tf * WINO_EXECUTED / peak, 4), "unit": "TFLOP/s",
                    "note": "`achieved` / `frac` above are ALGORITHMIC flops (2 x 9 x Cin x Cout per output position, SURVEY 8(d)) per second "
                            "against the fp32 MFMA peak, as the contract defines them; the Winograd form executes 16/36 of them on the "
                            "matrix cores, so frac > 1 is the algorithm's gain and THIS object is the MFMA pipe's own utilisation"}
            if n_f.value and n_d.value:
                def _inst(ms_i, n_i, u_i):
                    tf_i = u_i.value / (ms_i.value * 1e-3) / 1e12
                    return {"achieved": round(tf_i, 2), "frac": round(tf_i / peak, 4), "avg_launch_ms": round(ms_i.value / n_i.value, 4),
                            "launches": n_i.value}
                out["roofline"]["in_step_by_instantiation"] = {
                    "forward": _inst(ms_f, n_f, units_f),
                    "data gradient + BatchNorm-backward sums": _inst(ms_d, n_d, units_d),
                    "note": ("every launch has the GPU to itself inside the step (the conv phase of the backward is serial on the main stream "
                             "since the Winograd kernels); the data gradient's epilogue also forms the BatchNorm-backward sums of the block "
                             "below and, for conv2, the first block's weight-gradient sums: `frac` above averages over all four launches per "
                             "step as the contract asks") if wino else
                            ("the forward launches have the GPU to themselves inside the step; the data gradient of conv2 runs while the "
                             "top block's weight-gradient kernel (auxiliary stream) holds part of the CUs, so its event pair spans the "
                             "work of both kernels: `frac` above averages over all four launches per step as the contract asks")}
            if n_x.value:
                tf_x = units_x.value / (ms_x.value * 1e-3) / 1e12
                out["roofline"]["alone"] = {
                    "achieved": round(tf_x, 2), "frac": round(tf_x / peak, 4), "avg_launch_ms": round(ms_x.value / n_x.value, 4),
                    "launches": n_x.value,
                    "note": "the forward instantiation, same two shapes, in 8 forward-only passes after the timed region (nothing else "
                            "on the GPU)"}
        if eval_fwd is not None:
            out["eval_forward"] = eval_fwd
        if not args.no_cpu_baseline and world == 1:              # rank 0 at N = 1 only (a reported baseline, not part of the step)
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--direct-conv", action="store_true", help="A/B: the direct 36-product conv kernels instead of the Winograd default")
    ap.add_argument("--breakdown", action="store_true", help="per-kernel-family times of 2 extra steps on stderr")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="exercise launch + rendezvous + staged all-reduce without the HIP kernels (CPU tests); not a benchmark")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo with --plumbing-only)")
    ap.add_argument("--conv-bf16x3", action="store_true",
                    help="EXPERIMENT, reported on its own line (dtype says so): conv2/conv3 forward, data and weight gradient on the 3-term "
                         "bf16-split MFMA path instead of exact fp32; the default line stays f32")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: all ranks on cuda:0 over gloo (the real N-rank code path on a one-GPU box)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("[bench] --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))       # launcher only: no torch import, no GPU call above this line
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
