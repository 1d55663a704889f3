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
    from paddlemix_amd import _lib
    from paddlemix_amd.dist import empty_wire_params, gather_latents, wire_params
    from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params, unet_param_shapes
    from tests.abi_emulator import Emulator, on_emulator
    from tests.configs import TINY as cfg
    ed = _lib.elem_dtype()
    torch.manual_seed(100 + rank)
    if rank == 0:
        P = wire_params(synth_unet_params(cfg, seed=1234), ed)
    else:
        P = empty_wire_params(unet_param_shapes(cfg), ed, "cpu")
        for t in P.values():
            t.normal_()   # garbage until the broadcast
    nbytes = bench.broadcast_params(P)
    ref_P = wire_params(synth_unet_params(cfg, seed=1234), ed)
    same = all(torch.equal(P[k], ref_P[k]) and P[k].dtype == ref_P[k].dtype for k in ref_P)
    # matrices travel as 16-bit elements, 1-D parameters as fp32
    expect = sum(v.numel() * (2 if v.dim() > 1 else 4) for v in ref_P.values())
    # prompt sharding with per-rank seeds: global batch 2 * world, every rank draws ITS prompts / latents from its own generator
    g = torch.Generator().manual_seed(1000 + rank)
    sample = torch.randn(2, 4, 8, 8, generator=g)
    enc = torch.randn(2, 7, cfg["cross_attention_dim"], generator=g)
    model = on_emulator(UNet2DConditionModel, cfg, P)
    out = model(sample, 321, enc).sample
    allz = gather_latents(out)                      # the job's only other collective
    ok = True
    if rank == 0:
        full_s, full_e = [], []
        for r in range(world):
            gr = torch.Generator().manual_seed(1000 + r)
            full_s.append(torch.randn(2, 4, 8, 8, generator=gr))
            full_e.append(torch.randn(2, 7, cfg["cross_attention_dim"], generator=gr))
        full = on_emulator(UNet2DConditionModel, cfg, ref_P)(torch.cat(full_s), 321, torch.cat(full_e)).sample
        ok = allz.shape == full.shape and torch.allclose(allz, full, atol=1e-5, rtol=1e-5)
        ok = ok and not torch.allclose(allz[:2], allz[2:4])      # the ranks really worked on different prompts
    q.put((same and nbytes == expect, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_broadcast_and_prompt_sharding(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 200) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(a and b for a, b in res), res


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` exactly as the driver invokes it for N > 1 without a launcher (no WORLD_SIZE in the
    environment): bench.py becomes torch.distributed.run with one process per rank, and rank 0 prints ONE JSON line carrying the
    whole-job value, the per-rank step times and the gathered latents. CPU ranks (gloo) + the C-ABI interpreter stand in for the
    GPUs (--selftest-cpu); the line says so."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    for attempt in range(2):   # (the rendezvous port is picked free, then handed to the launcher: a rare race with the OS re-using it)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--selftest-cpu"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        if p.returncode == 0 or "address already in use" not in p.stderr.lower():
            break
    assert p.returncode == 0, p.stderr[-2000:]
    # stdout is the JSON line and nothing else: bench.py points fd 1 at stderr for everything but that line (the gloo transport of
    # the CPU stand-in announces its connections on fd 1, RCCL its version banner -- after the JSON line, at process exit)
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1, p.stdout
    r = json.loads(out_lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and "selftest" in r
    assert len(r["per_rank_ms_per_step"]) == 2 and r["gathered_latents"] == [4, 4, 8, 8]
    assert r["config"]["global_batch"] == 2 * r["config"]["batch_per_gpu"]
    # value = the units all ranks processed / the max-over-ranks time of the K timed steps
    assert abs(r["value"] - 2 * 2 / (r["ms_per_step"] * 2 / 1e3)) < 1e-6 * r["value"]
    assert r["ms_per_step"] >= max(r["per_rank_ms_per_step"]) * 0.999


def test_bench_force_dist_runs_the_collective_path_with_one_rank():
    """`bench.py --gpus 1 --force-dist`: the multi-rank branch (process group, weight broadcast, barriers, all-gather, the extra JSON
    fields) with a group of one -- how the RCCL calls are exercised on the single GPU of a `gpurun` box (profiles/r04_*_force_dist*)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1",
                        "--selftest-cpu"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1, p.stdout   # (one line on stdout, whatever the communication library prints)
    r = json.loads(out_lines[0])
    assert r["n_gpus"] == 1 and r["weight_broadcast_gb"] > 0 and r["gathered_latents"] == [2, 4, 8, 8] and len(r["per_rank_ms_per_step"]) == 1


def test_bench_single_process_prints_one_line_with_the_contract_fields():
    """`python bench.py` with no launcher and one rank (what the driver runs for N = 1), on the CPU stand-in: stdout is ONE JSON line
    carrying the fields of the bench contract; everything else (library banners, progress) is on stderr."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--selftest-cpu"], env=env,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out_lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1, p.stdout
    r = json.loads(out_lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True and "workload" in r["config"]
    assert abs(r["value"] - 1e3 / r["ms_per_step"]) < 1e-6 * r["value"]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-cpu"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_receiver_tables_match_the_sender_for_every_broadcast_model():
    """The receivers allocate from the *shape tables*, rank 0 sends what the *generators* produce: the broadcast pairs tensors by
    position in the dict, so names, order and shapes must agree for every model bench.py broadcasts (UNet + both text encoders)."""
    sys.path.insert(0, ROOT)
    import bench
    from paddlemix_amd.clip import clip_param_shapes, synth_clip_params
    from paddlemix_amd.unet import synth_unet_params, unet_param_shapes
    from tests.configs import MINI_XL, TINY
    pairs = [(synth_unet_params(c, seed=1), unet_param_shapes(c)) for c in (TINY, MINI_XL)]
    pairs += [(synth_clip_params(c, seed=1), clip_param_shapes(c))
              for c in (dict(bench.CLIP_L, num_hidden_layers=2), dict(bench.CLIP_BIGG, num_hidden_layers=2))]
    for P, S in pairs:
        assert list(P) == list(S)
        assert all(tuple(P[k].shape) == tuple(S[k]) for k in S)
    # the SDXL configuration of the headline itself, names and order only (the shape table needs no memory)
    S = unet_param_shapes(bench.SDXL)
    assert len(S) > 1000 and sum(int(torch.tensor(s).prod()) for s in S.values()) == 2_567_463_684


def test_broadcast_sequence_does_not_depend_on_strides(monkeypatch):
    """Which tensors go out on their own and which in buckets is decided by size alone: a rank holding a non-contiguous view of a
    big tensor must issue the same collectives (count and sizes) as the ranks holding contiguous ones."""
    from paddlemix_amd import dist as D
    calls = []
    monkeypatch.setattr(dist, "broadcast", lambda t, src=0: calls.append((t.numel(), t.dtype, t.is_contiguous())))
    big = torch.zeros(2048, 2049)                      # >= BIG elements
    P_contig = {"a.weight": big.clone(), "a.bias": torch.zeros(7), "b.weight": torch.zeros(3, 5, dtype=torch.bfloat16)}
    P_view = dict(P_contig, **{"a.weight": torch.zeros(2049, 2048).t()})
    assert not P_view["a.weight"].is_contiguous() and P_view["a.weight"].numel() >= D.BIG
    n1 = D.broadcast_params(P_contig)
    seq1, calls[:] = list(calls), []
    n2 = D.broadcast_params(P_view)
    assert n1 == n2 and [c[:2] for c in calls] == [c[:2] for c in seq1] and all(c[2] for c in calls)
