import torch
shapes = [(8192, 8192, 8192), (8192, 1280, 1280), (8192, 3840, 1280), (8192, 10240, 1280), (32768, 640, 640), (8192, 1280, 5120), (616, 2560, 2048), (32768, 5120, 640)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(5):
        c = torch.nn.functional.linear(a, w)
torch.cuda.synchronize()
