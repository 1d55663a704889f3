"""CPU baseline at the metric's own shape: the torch-CPU oracle (restatement of ppdiffusers; Paddle cannot be
installed here) on the FULL batch of the workload, timed on the GPU box's host cores.

    python scripts/cpu_baseline.py [--workload sdxl-1024-bs8] [--steps 1] [--out profiles/r02_cpu_baseline_<workload>.json]

Mirrors the reference's timing method (ppdiffusers/deploy/sd3/text_to_image_generation-stable_diffusion_3.py:107-137: warm-up,
then wall time of whole calls); a bs-8 SDXL forward is minutes of CPU work, so the default is one small page-in call + `steps`
timed full forwards. bench.py attaches the committed JSON as cpu_baseline.full_batch_measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import psutil
    import torch

    import bench
    from oracle import unet_ref as U
    from paddlemix_amd.unet import synth_unet_params

    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sdxl-1024-bs8", choices=["sdxl-1024-bs8", "sd15-512-bs1"])
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--where", default="host of the GPU box", help="which machine this ran on (recorded in the JSON)")
    ap.add_argument("--chunk", type=int, default=0, help="prompts per oracle call (0: the whole batch when the host has > 96 GiB free, else 1)")
    a = ap.parse_args()
    wl = bench.WORKLOADS[a.workload]
    cfg, B, H, W, L = wl["cfg"], wl["B"], wl["H"], wl["W"], wl["L"]
    # physical cores (os.cpu_count() counts SMT siblings: 256 on the GPU box, and ran > 1.7x slower), never more than the
    # container's affinity mask / CPU quota grants
    threads = min(torch.get_num_threads(), bench.usable_cpus())
    torch.set_num_threads(threads)
    avail = psutil.virtual_memory().available / 2 ** 30
    # the reference's math attention materialises [B*h, S, S] fp32 scores (+ the softmax copy): ~11 GB at bs 8, S = 4096
    chunk = a.chunk or (B if avail > 96 else 1)
    print(f"cpu_baseline: {threads} threads, {avail:.0f} GiB free, {chunk} prompt(s) per call; drawing the weights ...", flush=True)
    P = synth_unet_params(cfg, seed=1234)
    print("cpu_baseline: weights drawn", flush=True)
    g = torch.Generator().manual_seed(0)
    s = torch.randn(B, 4, H, W, generator=g)
    e = torch.randn(B, L, cfg["cross_attention_dim"], generator=g)
    ad = None
    if cfg.get("addition_embed_type") == "text_time":
        td = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
        ad = dict(text_embeds=torch.randn(B, td, generator=g), time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(B, 1))
    times = []
    with torch.no_grad():
        U.unet_forward(P, cfg, s[:1, :, :8, :8], 500, e[:1], added_cond_kwargs=None if ad is None else {k: v[:1] for k, v in ad.items()})
        for _ in range(a.steps):
            t0 = time.perf_counter()
            for b0 in range(0, B, chunk):
                sl = slice(b0, b0 + chunk)
                U.unet_forward(P, cfg, s[sl], 500, e[sl], added_cond_kwargs=None if ad is None else {k: v[sl] for k, v in ad.items()})
            times.append(time.perf_counter() - t0)
            print(f"cpu_baseline: step {len(times)}/{a.steps}: {times[-1]:.1f} s", flush=True)
    sec = sum(times) / len(times)
    res = {"workload": a.workload, "value": 1.0 / sec, "unit": "steps/s", "seconds_per_step": sec, "timed_steps": a.steps,
           "cores": threads, "machine": a.where, "host_mem_available_gib": round(avail, 1), "batch_chunk": chunk, "kind": "port",
           "tflops_effective": wl["gflop_step"] / 1e3 / sec,
           "what": f"torch-CPU fp32 restatement of ppdiffusers (Paddle unavailable), full bs-{B} {H}x{W} UNet forward"}
    print(json.dumps(res))
    out = a.out or os.path.join(ROOT, "gpurun_out", f"cpu_baseline_{a.workload}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
