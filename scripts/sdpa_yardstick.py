"""Same-box yardstick for the d = 64 self-attention launches of the SDXL step: torch's scaled_dot_product_attention (the ROCm flash
back end torch ships -- bench infrastructure only, never linked by the product) against mi355x_sd_sdpa, sustained and interleaved
(scripts/blas_yardstick.py's protocol): random normal bf16 q / k / v in the [B, S, h, d] layout both sides read (torch gets the
[B, h, S, d] view of the same buffers), `rounds` x [ours >= `seconds` | torch >= `seconds`].

    python scripts/sdpa_yardstick.py [--seconds 1.0] [--rounds 3] [--out gpurun_out/sdpa_yardstick.txt]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

SHAPES = [(8, 10, 4096, 64, 10, "level-1 self-attention"), (8, 20, 1024, 64, 60, "level-2 self-attention"),
          (8, 24, 4250, 64, 24, "SD3 joint attention (4096 + 154 tokens)")]


def burst(fn, seconds):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    n = max(50, int(seconds / (e0.elapsed_time(e1) / 20 * 1e-3)))
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from paddlemix_amd import ops
    dev = torch.device("cuda:0")
    lines = [f"torch {torch.__version__} F.scaled_dot_product_attention vs mi355x_sd_sdpa, {a.rounds} x [{a.seconds} s ours | {a.seconds} s torch], "
             "interleaved, random normal bf16, d = 64, no mask",
             f"{'B x h x S x d':>20s} {'n/step':>6s} | {'ours us':>9s} {'TF':>7s} | {'torch us':>9s} {'TF':>7s} | ours/torch time"]
    for B, H, S, D, n_step, what in SHAPES:
        g = torch.Generator(device=dev).manual_seed(S + H)
        q, k, v = (torch.randn(B, S, H, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(3))
        qt, kt, vt = (t.permute(0, 2, 1, 3) for t in (q, k, v))
        ref = F.scaled_dot_product_attention(qt, kt, vt).permute(0, 2, 1, 3)
        got = ops.sdpa(q, k, v)
        err = ((got.float() - ref.float()).norm() / ref.float().norm()).item()
        assert err < 1e-2, (S, err)
        t_o, t_t = [], []
        for _ in range(a.rounds):
            t_o.append(burst(lambda: ops.sdpa(q, k, v), a.seconds))
            t_t.append(burst(lambda: F.scaled_dot_product_attention(qt, kt, vt), a.seconds))
        mo, mt = statistics.median(t_o), statistics.median(t_t)
        fl = 4.0 * B * H * S * S * D
        lines.append(f"{B:>3d} x {H:>2d} x {S:>4d} x {D:>2d}     {n_step:>6d} | {mo:9.1f} {fl / mo / 1e6:7.0f} | {mt:9.1f} {fl / mt / 1e6:7.0f} | {mo / mt:5.3f}   {what}")
        print(lines[-1], flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
