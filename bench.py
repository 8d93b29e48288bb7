#!/usr/bin/env python3
"""Headline benchmark: audio-seconds generated per second, 2^16-sample stereo, 100-step v-DDIM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|f16|fp32] [--batch 16] [--ddim-steps 100]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE pass of the hot path over one batch: ``sample(model, x[B,2,65536], 100, eta=0)`` on the
221M-parameter uncond DiffusionAttnUnet1D (BASELINE.json configs[1]; random-init weights seed 0, noise
seed 1, synthetic).  Multi-GPU is batch-parallel (weak scaling: B per GPU fixed), one process per GPU,
with one RCCL all-gather of the output audio inside the timed region.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     : the dominant kernel (k=5 implicit-GEMM conv on MFMA), algorithmic FLOPs / HIP-event time,
                 measured live on the launch stream after the timed region (same plan, same buffers).
  cpu_baseline : the repo's CPU oracle (plain PyTorch fp32, oracle/) timed on the host cores on a bounded
                 sample (B=1, a few denoiser calls), extrapolated to the 100-step metric.  Reported only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLE_RATE = 48000          # reference defaults.ini:41
T = 65536                    # reference defaults.ini:23
PEAK = {"bf16": 2500.0, "f16": 2500.0, "fp32": 157.3}     # dense MFMA TFLOP/s (MI355X_MICROARCH.md)
DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}


def build_model():
    import audio_diffusion_amd as A
    torch.manual_seed(0)
    return A.DiffusionAttnUnet1D(n_attn_layers=4).eval()


def cpu_baseline(model, calls, threads):
    """Time the CPU oracle (test infrastructure; never on the product path) on config 1's shape."""
    from oracle import unet_oracle as U, sampling_oracle as S
    torch.set_num_threads(threads)
    cfg = U.make_config(n_attn_layers=4)
    fn = U.make_model_fn(model.state_dict(), cfg)
    torch.manual_seed(1)
    x = torch.randn(1, 2, T)
    t, _, _ = S.schedule(100)
    fn(x, torch.ones(1) * t[0])                      # warm-up call (thread pool, allocator)
    t0 = time.perf_counter()
    for i in range(calls):
        fn(x, torch.ones(1) * t[i])
    dt = (time.perf_counter() - t0) / calls
    return {"value": (T / SAMPLE_RATE) / (dt * 100), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "sample": f"B=1 x 2x{T}, {calls} denoiser calls timed ({dt:.3f} s/call), x100 for the 100-step loop; "
                      "the DDIM update itself is <0.1% of a call"}


def conv_roofline(model, B, dtype, dev, reps=3):
    """HIP-event timing of every k=5 MFMA conv launch of one denoiser call, on the launch stream."""
    from audio_diffusion_amd import _lib
    plan = model.plan_for(B, T, dtype, dev)
    x = torch.randn(B, 2, T, device=dev)
    tt = torch.ones(B, device=dev)
    out = torch.empty_like(x)
    plan.bind(x.data_ptr(), tt.data_ptr(), out.data_ptr())
    stream = torch.cuda.current_stream(dev)
    groups = {}
    for i, m in enumerate(plan.meta):
        if m["kind"] == _lib.OP_CONV and m["taps"] == 5 and m["cotile"] >= 128:
            groups.setdefault(f'{m["cotile"]}co x {m["postile"]}pos', []).append(i)
    timed = set(i for g in groups.values() for i in g)
    ms = {i: 0.0 for i in timed}
    for r in range(reps + 1):
        evs = {}
        for i in range(plan.n_ops):
            if i in timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                _lib.run_ops(_offset(plan.ops, i), 1, stream.cuda_stream)
                e1.record(stream)
                evs[i] = (e0, e1)
            else:
                _lib.run_ops(_offset(plan.ops, i), 1, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        if r:                                        # first pass is a warm-up
            for i, (e0, e1) in evs.items():
                ms[i] += e0.elapsed_time(e1) / reps
    key = {v: k for k, v in DTYPES.items()}[dtype]
    per = {}
    for pt, idx in groups.items():
        fl = sum(plan.meta[i]["flops"] for i in idx)
        tm = sum(ms[i] for i in idx)
        per[pt] = {"launches_per_call": len(idx), "avg_launch_us": 1e3 * tm / len(idx), "flop_per_call": fl,
                   "tflops": fl / (tm * 1e-3) / 1e12, "ms_per_call": tm}
    dom = max(per, key=lambda k: per[k]["flop_per_call"])
    d = per[dom]
    alg_bytes = sum(plan.meta[i]["bytes"] for i in groups[dom]) / len(groups[dom])
    return {"bound": "mfma", "kernel": f"conv_mfma_kernel<{key}, taps=5, tile {dom}>",
            "achieved": d["tflops"], "peak": PEAK[key], "unit": "TFLOP/s", "frac": d["tflops"] / PEAK[key],
            "traffic": pmc_traffic(key, dom), "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, "
            "separate --pmc passes, profiles/)", "algorithmic_bytes_per_launch": alg_bytes,
            "launches_per_call": d["launches_per_call"], "avg_launch_us": d["avg_launch_us"],
            "flop_per_launch_avg": d["flop_per_call"] / d["launches_per_call"],
            "share_of_call_flop": d["flop_per_call"] / (6.899e11 * B),
            "other_tiles": {str(k): {"tflops": v["tflops"], "launches_per_call": v["launches_per_call"],
                                     "avg_launch_us": v["avg_launch_us"]} for k, v in per.items() if k != dom}}


def pmc_traffic(key, tile):
    """HBM bytes per launch of the dominant kernel from the committed counter summary (tools/run_pmc.sh ->
    profiles/rNN/pmc_summary.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction).
    bench.py cannot collect counters itself; null when no summary covers this kernel."""
    import glob
    import re
    m = re.match(r"(\d+)co x (\d+)pos", tile)
    if not m or key != "bf16":
        return None
    tpos = int(m.group(2)) // 64
    # conv_mfma_kernel<EXT=false, bf16, bf16, TAPS=5, WCO=2, WPOS=2, TCO=2, TPOS, DIL=1> (older summaries: no EXT / DIL)
    pat = re.compile(rf"_Z16conv_mfma_kernelI(Lb0E)?DF16bDF16bLi5ELi2ELi2ELi2ELi{tpos}E(Li1E)?Ev8ConvArgs$")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json")), reverse=True):
        try:
            summary = json.load(open(f))
        except (OSError, ValueError):
            continue
        for name, e in summary.items():
            if pat.match(name) and "hbm_bytes_per_launch" in e:
                return e["hbm_bytes_per_launch"]
    return None


def _offset(ops, i):
    import ctypes
    from audio_diffusion_amd import _lib
    return ctypes.cast(ctypes.byref(ops, i * ctypes.sizeof(_lib.AdkOp)), ctypes.POINTER(_lib.AdkOp))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "fp32"])
    ap.add_argument("--batch", type=int, default=16, help="samples per GPU")
    ap.add_argument("--ddim-steps", type=int, default=100)
    ap.add_argument("--cpu-calls", type=int, default=4, help="denoiser calls timed for cpu_baseline (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    import audio_diffusion_amd as A
    from audio_diffusion_amd import distributed as D

    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        D.init(rank, world)
    dtype = DTYPES[a.dtype]

    model = build_model()
    cpu = None
    if rank == 0 and world == 1 and a.cpu_calls > 0:
        cpu = cpu_baseline(model, a.cpu_calls, min(16, os.cpu_count() or 1))
    model = model.to(dev)
    model.compute_dtype = dtype

    # noise generated ONCE from a single CPU seed and sliced per rank: a sample's result does not depend on N
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(a.batch * world, 2, T, generator=g)
    x = noise[rank * a.batch:(rank + 1) * a.batch].to(dev)

    def one_pass():
        pred = A.sample(model, x, a.ddim_steps, 0)
        return D.all_gather_audio(pred) if world > 1 else pred

    for _ in range(a.warmup):
        one_pass()
    D.barrier(world)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = one_pass()
    D.barrier(world)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, world, dev)
    assert torch.isfinite(out).all()

    if rank == 0:
        ms = dt / a.steps * 1e3
        total_audio_s = world * a.batch * T / SAMPLE_RATE
        rec = {
            "metric": "audio-seconds generated/sec, 2^16-sample stereo, 100-step v-DDIM",
            "value": total_audio_s / (ms * 1e-3), "unit": "audio-s/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"configs[1]: DiffusionAttnUnet1D uncond (221M params, random init seed 0), "
                                   f"batch {a.batch}/GPU x 2ch x {T}, {a.ddim_steps}-step v-DDIM eta=0, "
                                   "hipGraph-captured step loop",
                       "batch_per_gpu": a.batch, "global_batch": a.batch * world, "ddim_steps": a.ddim_steps,
                       "parallelism": f"batch-parallel x{world}" + (" + RCCL all-gather of output" if world > 1 else "")},
            "ms_per_denoiser_call": ms / a.ddim_steps,
            "model_tflops": 6.899e11 * a.batch * world * a.ddim_steps / (ms * 1e-3) / 1e12,        # whole job
        }
        if not a.no_roofline:
            rec["roofline"] = conv_roofline(model, a.batch, dtype, dev)
        if cpu is not None:
            rec["cpu_baseline"] = cpu
        print(json.dumps(rec), flush=True)
    D.barrier(world)
    if world > 1:
        D.shutdown()


if __name__ == "__main__":
    main()
