"""world_size-2 gloo tests (CPU) of the N>1 path: bucketed weight broadcast from rank 0 and prompt sharding.
The data path has no collective, so what must hold is (a) every rank ends up with rank 0's weights bit for bit and
(b) the union of the ranks' shards equals the unsharded batch result (batch rows are independent)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import unet_ref as U
    from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params, unet_param_shapes
    from tests.abi_emulator import Emulator
    from tests.configs import TINY as cfg
    torch.manual_seed(100 + rank)
    if rank == 0:
        P = synth_unet_params(cfg, seed=1234)
    else:
        P = {n: torch.randn(s) for n, s in unet_param_shapes(cfg).items()}  # garbage until the broadcast
    bench.broadcast_params(P, rank, world)
    ref_P = synth_unet_params(cfg, seed=1234)
    same = all(torch.equal(P[k], ref_P[k]) for k in ref_P)
    # prompt sharding: global batch 4 -> 2 per rank
    g = torch.Generator().manual_seed(0)
    sample = torch.randn(4, 4, 8, 8, generator=g)
    enc = torch.randn(4, 7, cfg["cross_attention_dim"], generator=g)
    sl = slice(2 * rank, 2 * rank + 2)
    model = UNet2DConditionModel(cfg, P, _test_backend=Emulator())
    out = model(sample[sl], 321, enc[sl]).sample
    gathered = [torch.empty_like(out) for _ in range(world)]
    dist.all_gather(gathered, out)
    if rank == 0:
        full = UNet2DConditionModel(cfg, ref_P, _test_backend=Emulator())(sample, 321, enc).sample
        q.put((same, torch.allclose(torch.cat(gathered), full, atol=1e-5, rtol=1e-5)))
    else:
        q.put((same, True))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_prompt_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(a and b for a, b in res), res
