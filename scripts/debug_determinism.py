"""Find the first op whose output differs materially between two runs of the same program on identical inputs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params
from tests.configs import MINI_XL, TINY
from tests.test_gpu_unet import _inputs, _cuda

cfg = MINI_XL if len(sys.argv) < 2 or sys.argv[1] == "mini" else TINY
P = synth_unet_params(cfg, seed=1234)
sample, enc, added = _inputs(cfg, 2, 32, 32)
m = UNet2DConditionModel(cfg, P, use_graph=False)
plan = m._get_plan(2, 32, 32, 77)


def run_once(ref=None):
    outs = []
    with torch.cuda.stream(m._stream):
        for t in plan.keep:
            t.zero_()
        plan.in_scale.fill_(1.0)
        m.stage_inputs(plan, _cuda(sample), 501, _cuda(enc), _cuda(added))
        m._stream.synchronize()
        for j, (fn, args, kind, fl) in enumerate(plan.prog):
            if ref is None:
                before = [t.clone() for t in plan.keep]
            rc = fn(*args)
            assert rc == 0
            m._stream.synchronize()
            if ref is None:
                changed = [(i, t.clone()) for i, (t, b) in enumerate(zip(plan.keep, before)) if not torch.equal(t, b)]
            else:
                changed = [(i, plan.keep[i].clone()) for i, _ in ref[j][1]]
            outs.append((kind, changed))
    return outs


r1 = run_once()
r2 = run_once(r1)
shown = 0
for i, ((k1, c1), (k2, c2)) in enumerate(zip(r1, r2)):
    for (idx, a), (_, b) in zip(c1, c2):
        if torch.equal(a, b):
            continue
        if a.dtype == torch.uint8:
            fa, fb = a.view(torch.bfloat16).float(), b.view(torch.bfloat16).float()
            ga, gb = a.view(torch.float32), b.view(torch.float32)
            nd = (a != b).sum().item()
            print(f"op {i} {k1}: scratch {idx} ({a.numel()} B) differs in {nd} bytes; as-bf16 maxdiff "
                  f"{torch.nan_to_num(fa - fb).abs().max().item():.3e} (max {torch.nan_to_num(fa).abs().max().item():.3e}); "
                  f"as-f32 maxdiff {torch.nan_to_num(ga - gb).abs().max().item():.3e}")
        else:
            fa, fb = a.float(), b.float()
            print(f"op {i} {k1}: tensor {idx} {tuple(a.shape)} {a.dtype} maxdiff {(fa - fb).abs().max().item():.3e} (max {fa.abs().max().item():.3e})")
        shown += 1
    if shown > 12:
        break
print("done; prog len", len(r1))
for i in range(0, 0):
    print(i, plan.prog[i][2], plan.prog[i][1][:12])
