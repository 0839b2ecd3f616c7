#!/usr/bin/env python3
"""Headline benchmark: embeddings/sec @ bs=256 on InceptionResnetV1 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the HIP embed path (pack -> 130 MFMA convolutions + pools -> L2 norm) over
one batch of 256 synthetic 160x160 crops (N(0,1), seed 0, bf16, already resident in HBM) with
random-init generator weights (seed 0; no pretrained weights exist offline).  With N > 1 every
rank embeds its own 256 crops (weak scaling, no data-path collective) and the ranks exchange the
resulting (256,512) fp32 embeddings with one RCCL all-gather per step, overlapped with the next
step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16/f16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
BATCH = 256


def host_cores():
    """Cores this process may really use: affinity mask, cgroup CPU quota, and the GPU box's
    per-GPU CPU share (16) -- os.cpu_count() reports the whole host and oversubscribes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(budget_s=12.0):
    """The oracle (CPU restatement of the reference's InceptionResnetV1 forward, fp32 torch-CPU
    eager) timed on this box's host cores on a bounded sample of the same workload."""
    import torch
    from oracle import irv1
    from vn_celeb_face_recognition_amd.weights import generate_state_dict
    torch.set_num_threads(host_cores())
    sd = generate_state_dict("irv1", 0, as_torch=True)
    bs = 32
    x = torch.randn((bs, 3, 160, 160), generator=torch.Generator().manual_seed(0))
    irv1.irv1_forward(sd, x[:4])  # warm up allocator / thread pool
    n = 0
    t0 = time.perf_counter()
    while True:
        irv1.irv1_forward(sd, x)
        n += bs
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= BATCH * 2:
            break
    return {"value": round(n / dt, 2), "unit": "embeddings/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d images (batches of %d of the bs=256 workload), fp32 torch-CPU oracle, %.1f s" % (n, bs, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vn_celeb_face_recognition_amd import dist as vdist
    from vn_celeb_face_recognition_amd.models import InceptionResnetV1
    from vn_celeb_face_recognition_amd.weights import IRV1_MACS_PER_IMAGE

    rank, world, local = vdist.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    model = InceptionResnetV1(pretrained=None, device=dev, compute_dtype=args.dtype, max_batch=BATCH).eval()
    g = torch.Generator().manual_seed(rank)
    x = torch.randn((BATCH, 3, 160, 160), generator=g).to(dev).to(tdt)
    gathered = [torch.empty((world * BATCH, 512), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None

    def step(i, pending):
        emb = model(x)
        if world > 1:
            if pending is not None:
                pending.wait()
            pending = vdist.all_gather_fixed(gathered[i & 1], emb, async_op=True)
        return emb, pending

    pending = None
    for i in range(args.warmup):
        emb, pending = step(i, pending)
    if pending is not None:
        pending.wait()
        pending = None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        emb, pending = step(i, pending)
    ev1.record()
    if pending is not None:
        pending.wait()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels were launched on
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    assert torch.isfinite(emb).all()

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = world * BATCH * args.steps / wall
        flop_per_step = 2.0 * IRV1_MACS_PER_IMAGE * BATCH          # 2.8353 GFLOP / image
        achieved = flop_per_step / (dev_ms / args.steps * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if args.dtype != "f32" else 157.3
        alg, executed = model.flops_per_image()
        out = {
            "metric": "embeddings/sec @ bs=256 (InceptionResnetV1, 160x160)",
            "value": round(value, 1), "unit": "embeddings/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: InceptionResnetV1 embedding only, synthetic 160x160 %s, "
                                   "bs=256 per GPU, generator weights seed 0" % args.dtype,
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "parallelism": "dp%d (frames sharded, all-gather of embeddings)" % world},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": None,
                         "kernel": "conv_igemm_kernel (all launches of one embed step; device time by HIP events)",
                         "flop_per_step_algorithmic": flop_per_step,
                         "flop_per_image_executed": executed, "flop_per_image_counted_by_engine": alg,
                         "device_ms_per_step": round(dev_ms / args.steps, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
