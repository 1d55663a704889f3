"""Same-box vendor yardstick, SUSTAINED: hipBLASLt (through torch.nn.functional.linear -- bench infrastructure only, the product never
links a vendor BLAS) against this library's GEMM on the SDXL step's linear shapes, interleaved in one process.

    python scripts/blas_yardstick.py [--seconds 1.0] [--rounds 3] [--out gpurun_out/blas_yardstick.txt]

Protocol ("measure sustained"): random bf16 operands (full-range normal data: zero or constant fills flatter
both sides by up to 20 %), per shape `rounds` x [ours for >= `seconds`, hipBLASLt for >= `seconds`], each arm timed with events
around the whole burst after a 50-launch lead-in; both arms compute out = a @ w^T + bias into a bf16 tensor (what F.linear
computes; the step's own epilogues -- GEGLU, residual -- are extra work on our side only and are reported in the per-shape table of
bench.py, not here). Reported: median microseconds per launch and TFLOP/s per arm, the ratio, the board's clock / power sampled by
rocm-smi during the last burst of each arm.
"""
import argparse
import os
import re
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# (M, N, K, launches per SDXL bs-8 step, what it is)
SHAPES = [
    (8192, 10240, 1280, 60, "FF1 (step: GEGLU epilogue)"),
    (8192, 1280, 1280, 192, "to_out / to_q / proj (step: + residual)"),
    (8192, 3840, 1280, 60, "fused QKV"),
    (8192, 1280, 5120, 60, "FF2 (step: + residual)"),
    (32768, 5120, 640, 10, "FF1 at 640 (step: GEGLU)"),
    (32768, 640, 640, 40, "to_out / to_q at 640"),
    (32768, 1920, 640, 10, "fused QKV at 640"),
    (32768, 640, 2560, 10, "FF2 at 640"),
]


# the batch-1 SD-1.5 step's most frequent linear shapes (--sd15): 0.84-GFLOP problems that take split-K slices here
SHAPES_SD15 = [
    (256, 1280, 1280, 25, "to_q / to_out / proj at 16x16"),
    (1024, 640, 640, 25, "the same at 32x32"),
    (4096, 320, 320, 25, "the same at 64x64"),
    (256, 10240, 1280, 5, "FF1 at 16x16 (step: GEGLU)"),
    (256, 1280, 5120, 5, "FF2 at 16x16"),
    (1024, 640, 2560, 5, "FF2 at 32x32"),
    (4096, 320, 1280, 5, "FF2 at 64x64"),
    (256, 3840, 1280, 5, "fused QKV at 16x16"),
    (64, 1280, 1280, 5, "to_q / to_out at 8x8"),
]


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        sclk = re.findall(r"sclk clock level:[^(]*\((\d+)Mhz\)", out)
        power = re.findall(r"Power \(W\):\s*([\d.]+)", out)
        return (int(sclk[0]) if sclk else None, float(power[0]) if power else None)
    except Exception:
        return (None, None)


def burst(fn, seconds):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    # size the burst from a short calibration
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 50 * 1e-3
    n = max(100, int(seconds / per))
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, n   # us per launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--sd15", action="store_true", help="the batch-1 SD-1.5 shapes instead of the SDXL bs-8 ones")
    a = ap.parse_args()
    shapes = SHAPES_SD15 if a.sd15 else SHAPES
    from paddlemix_amd import ops
    dev = torch.device("cuda:0")
    lines = [f"hipBLASLt (torch {torch.__version__} F.linear) vs libmi355x_sd on the SDXL step's linear shapes: {a.rounds} x [{a.seconds} s ours | "
             f"{a.seconds} s hipBLASLt], interleaved, random normal bf16 operands, out = a w^T + bias (bf16)",
             f"{'M x N x K':>22s} {'n/step':>6s} | {'ours us':>9s} {'TF':>7s} {'MHz':>5s} {'W':>5s} | {'hipBLASLt us':>12s} {'TF':>7s} {'MHz':>5s} {'W':>5s} | ours/hipBLASLt time"]
    tot = {"ours": 0.0, "blas": 0.0}
    for M, N, K, n_step, what in shapes:
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, generator=g, device=dev)
        b16 = b.to(torch.bfloat16)
        out_o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ref = torch.nn.functional.linear(x, w, b16)
        ops.linear(x, w, b, out=out_o)
        err = ((out_o.float() - ref.float()).norm() / ref.float().norm()).item()
        assert err < 1e-2, (M, N, K, err)
        f_ours = lambda: ops.linear(x, w, b, out=out_o)                  # noqa: E731
        f_blas = lambda: torch.nn.functional.linear(x, w, b16)            # noqa: E731
        t_o, t_b, s_o, s_b = [], [], (None, None), (None, None)
        for r in range(a.rounds):
            last = r == a.rounds - 1
            box = {}
            th = threading.Thread(target=lambda: (time.sleep(0.5 * a.seconds), box.update(v=smi()))) if last else None
            if th:
                th.start()
            t_o.append(burst(f_ours, a.seconds)[0])
            if th:
                th.join()
                s_o = box.get("v", (None, None))
            box = {}
            th = threading.Thread(target=lambda: (time.sleep(0.5 * a.seconds), box.update(v=smi()))) if last else None
            if th:
                th.start()
            t_b.append(burst(f_blas, a.seconds)[0])
            if th:
                th.join()
                s_b = box.get("v", (None, None))
        mo, mb = statistics.median(t_o), statistics.median(t_b)
        fl = 2.0 * M * N * K
        tot["ours"] += mo * n_step
        tot["blas"] += mb * n_step
        lines.append(f"{M:>7d} x {N:>5d} x {K:>4d} {n_step:>6d} | {mo:9.1f} {fl / mo / 1e6:7.0f} {str(s_o[0]):>5s} {str(s_o[1]):>5s} | {mb:12.1f} {fl / mb / 1e6:7.0f} "
                     f"{str(s_b[0]):>5s} {str(s_b[1]):>5s} | {mo / mb:5.3f}   {what}")
        print(lines[-1], flush=True)
        del x, w, out_o, ref
        torch.cuda.empty_cache()
    lines.append(f"launch-count-weighted (one SDXL step's linear class on these eight shapes): ours {tot['ours'] / 1e3:.2f} ms, hipBLASLt {tot['blas'] / 1e3:.2f} ms")
    print(lines[-1])
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
