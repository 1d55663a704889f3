"""A/B of LayerNorm + linear vs row_stats + linear_ln on one MI355X (isolated, back-to-back chains)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from paddlemix_amd import ops  # noqa: E402

ops.init(0)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for M, C, N in [(8192, 1280, 1280), (8192, 1280, 3840), (32768, 640, 640), (32768, 640, 1920)]:
    x = (torch.randn(M, C, device="cuda") * 2 + 1).to(torch.bfloat16)
    xn = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, C, device="cuda") / C ** 0.5).to(torch.bfloat16)
    ws = w.float().sum(1).contiguous()
    b = torch.randn(N, device="cuda")
    g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    st = ops.row_stats(x)
    ln = torch.empty_like(x)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_lin_raw = timeit(lambda: ops.linear(x, w, b, out=out))
    t_lin_norm = timeit(lambda: ops.linear(xn, w, b, out=out))
    t_lnlin = timeit(lambda: ops.linear_ln(x, st, w, ws, b, out=out))
    t_ln = timeit(lambda: ops.layer_norm(x, g, be, out=ln))
    t_rs = timeit(lambda: ops.row_stats(x))
    t_chain_a = timeit(lambda: (ops.layer_norm(x, g, be, out=ln), ops.linear(ln, w, b, out=out)))
    t_chain_b = timeit(lambda: (ops.row_stats(x), ops.linear_ln(x, st, w, ws, b, out=out)))
    print(f"{M}x{N}x{C}: linear(raw x) {t_lin_raw:.1f} us  linear(N(0,1)) {t_lin_norm:.1f}  linear_ln {t_lnlin:.1f} | "
          f"LN {t_ln:.1f}  row_stats {t_rs:.1f} | chain LN+linear {t_chain_a:.1f}  chain stats+linear_ln {t_chain_b:.1f}")
