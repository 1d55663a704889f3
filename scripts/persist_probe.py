"""Persistent-block streaming GEMM (csrc/gemm_persist.hip, MI355X_SD_GEMM_PERSIST=1) against the one-tile-per-block kernel:
bit-for-bit equality of the outputs and time per launch, per shape. Run on the GPU box:

    python scripts/persist_probe.py            # parent: runs itself twice (env off / on) and compares the dumps

The switch is read once per process, hence the two child processes. Shapes: the launches of the SDXL bs-8 step that put more than
one 256x320 tile on a CU (FF1 GEGLU, the 640-channel GEGLU, the 131072-row convs) plus ragged / one-K-tile / two-K-tile cases."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GEMMS = [  # M, N, K, geglu, residual
    (8192, 10240, 1280, True, False), (32768, 5120, 640, True, False), (32768, 2560, 640, False, True),
    (9000, 10240, 64, True, False), (8200, 1920, 128, False, True), (70000, 640, 1280, False, False)]
CONVS = [  # B, H, W, Cin, Cout
    (8, 128, 128, 320, 320), (8, 128, 128, 640, 320), (8, 64, 64, 640, 640), (3, 100, 90, 64, 320)]


def child():
    import torch
    from paddlemix_amd import ops
    ops.init(0)

    def timeit(fn, n=8):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    res = {}
    for M, N, K, geglu, resid in GEMMS:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda", generator=g)
        n_out = N // 2 if geglu else N
        r = torch.randn(M, n_out, device="cuda", generator=g).to(torch.bfloat16) if resid else None
        out = torch.empty(M, n_out, device="cuda", dtype=torch.bfloat16)
        fn = lambda: ops.linear(a, w, b, out=out, geglu=geglu, residual=r)   # noqa: E731
        us = timeit(fn)
        res[f"gemm {M}x{N}x{K}{'g' if geglu else ''}{'+R' if resid else ''}"] = dict(
            us=round(us, 1), tf=round(2.0 * M * N * K / us / 1e6), finite=bool(torch.isfinite(out.float()).all()),
            sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16])
    for B, H, W, Cin, Cout in CONVS:
        g = torch.Generator(device="cuda").manual_seed(B + H + Cin)
        x = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(Cout, 9 * Cin, device="cuda", generator=g) / (9 * Cin) ** 0.5).to(torch.bfloat16)
        b = torch.randn(Cout, device="cuda", generator=g)
        out = torch.empty(B * H * W, Cout, device="cuda", dtype=torch.bfloat16)
        fn = lambda: ops.conv3x3(x, w, b, out=out)   # noqa: E731
        us = timeit(fn)
        res[f"conv {B}x{H}x{W}x{Cin}->{Cout}"] = dict(
            us=round(us, 1), tf=round(2.0 * B * H * W * Cout * 9 * Cin / us / 1e6), finite=bool(torch.isfinite(out.float()).all()),
            sha=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16])
    print("PROBE_JSON " + json.dumps(res))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child()
    runs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, MI355X_SD_GEMM_PERSIST=mode)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("PROBE_JSON ")]
        if p.returncode != 0 or not line:
            print(f"child MI355X_SD_GEMM_PERSIST={mode} failed:\n{p.stderr[-2000:]}")
            sys.exit(1)
        runs[mode] = json.loads(line[-1][len("PROBE_JSON "):])
    bad = 0
    for k, base in runs["0"].items():
        pers = runs["1"][k]
        same = base["sha"] == pers["sha"] and pers["finite"]
        bad += not same
        print(f"{k:34s} one-tile {base['us']:8.1f} us {base['tf']:5d} TF | persistent {pers['us']:8.1f} us {pers['tf']:5d} TF | "
              f"{'bit-identical' if same else 'DIFFERENT'}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
