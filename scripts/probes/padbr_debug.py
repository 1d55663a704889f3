import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from paddlemix_amd import ops
ops.init(0)
g = torch.Generator().manual_seed(1)
B, H, W, C, Cout = 1, 16, 16, 32, 32
x = torch.randn(B, H, W, C, generator=g).to(torch.bfloat16)
w = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(torch.bfloat16)
wk = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
xn = x.float().permute(0, 3, 1, 2)
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
out = ops.conv3x3(x.cuda(), wk.cuda(), None, stride=2, pad_br=True).float().cpu().reshape(B, 8, 8, Cout).permute(0, 3, 1, 2)
sym = ops.conv3x3(x.cuda(), wk.cuda(), None, stride=2).float().cpu().reshape(B, 8, 8, Cout).permute(0, 3, 1, 2)
print("out vs br ", rel(out, F.conv2d(F.pad(xn, (0, 1, 0, 1)), w.float(), stride=2)))
print("out vs sym", rel(out, F.conv2d(xn, w.float(), stride=2, padding=1)))
print("out vs tl ", rel(out, F.conv2d(F.pad(xn, (1, 0, 1, 0)), w.float(), stride=2)))
print("out vs pad2tl", rel(out, F.conv2d(F.pad(xn, (2, 0, 2, 0)), w.float(), stride=2)[:, :, :8, :8]))
print("sym vs sym", rel(sym, F.conv2d(xn, w.float(), stride=2, padding=1)))
print("out == sym", torch.equal(out, sym))
