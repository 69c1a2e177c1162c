#!/usr/bin/env python3
"""bench.py — mel-frames/s of the SEDnet fit step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step is one full fit step of the hot path on one resident synthetic batch: forward (train-mode
BatchNorm, dropout 0.5) + BCEWithLogits + backward + (gradient all-reduce over RCCL) + Adam, i.e.
reference sed.py:134-137.  Workload = BASELINE config 2 per GPU (mono, B=128, 256 frames x 40 mel,
3x conv128 + BiGRU 2x128), weak scaling: every rank trains its own 128-sample shard of the global
batch, gradients averaged.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(B=128, Cin=1, F=40, T=256, C=128, H=128, gru_layers=2, dropout=0.5)
F32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)


DOMINANT_KERNEL = "conv3x3_mfma_fwd2_k<4, 2>"


def pmc_traffic_bytes():
    """HBM traffic per launch of the dominant kernel, from the committed PMC summary (collected as the
    MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes; FETCH_SIZE counts half of a wide
    coalesced read on gfx950, verified here on the pure-streaming bn kernel).  None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "r1", "pmc_fetch_write_per_kernel.json")
    try:
        d = json.load(open(path))
        e = d["void " + DOMINANT_KERNEL]
        return int((2.0 * e["FETCH_SIZE_KB_avg"] + e["WRITE_SIZE_KB_avg"]) * 1024)
    except Exception:
        return None


def cpu_baseline(sample_B=8, steps=3):
    """The oracle's fit step (torch.nn CPU restatement of sed.py, pinned by tests/golden) on the host cores."""
    import torch
    from oracle import crnn_ref
    w = WORKLOAD
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = crnn_ref.SedNetRef(conv_channels=w["C"], dropout=w["dropout"], in_channels=w["Cin"], n_mels=w["F"],
                             gru_hidden=w["H"], gru_layers=w["gru_layers"])
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x, y = crnn_ref.synthetic_batch(sample_B, w["Cin"], w["F"], w["T"], w["T"] // 8, seed=1234)
    crnn_ref.fit_step(net, opt, x, y)                       # warm-up
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        crnn_ref.fit_step(net, opt, x, y)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(sample_B * w["T"] / med, 1), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"{steps} fit steps (median) of the same net at B={sample_B}, T={w['T']} (torch {torch.__version__} CPU, "
                      f"{cores} threads); {med*1e3:.0f} ms/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="per-kernel-family times of 2 extra steps on stderr")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sed_crnn_amd as sed
    from sed_crnn_amd import _lib
    from sed_crnn_amd.dist import broadcast_parameters, init_from_env
    from sed_crnn_amd.trainer import FusedTrainStep

    rank, world, local = init_from_env("nccl")
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    w = WORKLOAD
    torch.manual_seed(0)
    model = sed.TimePooledCRNN(conv_channels=w["C"], dropout=w["dropout"], in_channels=w["Cin"], n_mels=w["F"],
                               gru_hidden=w["H"], gru_layers=w["gru_layers"]).to(dev)
    if world > 1:
        broadcast_parameters(model)
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(w["B"], w["Cin"], w["F"], w["T"], generator=g).to(dev)
    y = (torch.rand(w["B"], w["T"] // 8, 1, generator=g) > 0.8).float().to(dev)
    step = FusedTrainStep(model, lr=1e-3, loss="bce")

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step.step(x, y)
    lib = _lib.lib()
    tag = 0                                                  # SED_K_CONV_MFMA_FWD: the dominant kernel
    barrier()
    lib.sed_prof_enable(1 << tag)                            # 8 event records per step; nothing else instrumented
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step.step(x, y)
    barrier()
    dt = time.perf_counter() - t0
    import ctypes as C
    ms, n, units = C.c_double(), C.c_long(), C.c_double()
    lib.sed_prof_read(tag, C.byref(ms), C.byref(n), C.byref(units))
    lib.sed_prof_enable(0)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()

    if args.breakdown and rank == 0:
        lib.sed_prof_enable(0xFFFF)
        for _ in range(2):
            step.step(x, y)
        torch.cuda.synchronize()
        tot = 0.0
        for k in range(11):
            lib.sed_prof_read(k, C.byref(ms2 := C.c_double()), C.byref(n2 := C.c_long()), C.byref(u2 := C.c_double()))
            if n2.value:
                rate = u2.value / (ms2.value * 1e-3) / 1e12
                tot += ms2.value / 2
                print(f"[bench] {lib.sed_prof_tag_name(k).decode():24s} {ms2.value/2:8.3f} ms/step  {n2.value//2:3d} launches/step  "
                      f"{rate:8.2f} T(units)/s", file=sys.stderr)
        print(f"[bench] sum of instrumented kernels {tot:.3f} ms/step", file=sys.stderr)
        lib.sed_prof_enable(0)

    if rank == 0:
        frames = w["B"] * w["T"] * world * args.steps
        out = {
            "metric": "mel-frames/sec training throughput (seq=256, mel=40)",
            "value": round(frames / dt, 1),
            "unit": "mel-frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: mono (B={w['B']},256,40,1) per GPU, 3xConv3x3(128)+BN+ReLU+pool(1,2)+dropout0.5, "
                                   f"BiGRU 2x128, Linear(256,1), BCEWithLogits, Adam lr 1e-3; full fit step (fwd+loss+bwd+allreduce+Adam)",
                       "global_batch": w["B"] * world, "seq_len": w["T"], "n_mels": w["F"],
                       "parallelism": f"dp{world}" if world > 1 else "single", "final_loss": round(final_loss, 6)},
        }
        if n.value:
            avg_ms = ms.value / n.value
            tf = units.value / (ms.value * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": DOMINANT_KERNEL + " (conv2/conv3 forward + their data gradients)",
                               "achieved": round(tf, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic_bytes(),
                               "traffic_note": "HBM bytes per launch from a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass of this "
                                               "command (profiles/r1): (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 half-count correction",
                               "avg_launch_ms": round(avg_ms, 4), "launches": n.value,
                               "flops_per_launch_avg": units.value / n.value}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
