"""Debug aid (round 4): the mini VAE decoder planned on the C-ABI emulator (host memory) and the same model planned on the device --
where do the two launch lists differ, and which scratch sizes does the emulator answer differently from the library?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd import _lib  # noqa: E402
from tests import export_cases as EC  # noqa: E402
from tests.abi_emulator import Emulator, on_emulator  # noqa: E402

lib = _lib.load()
emu = Emulator()
for args in ((2, 1024, 32), (2, 256, 64), (2, 64, 64), (8, 16384, 128)):
    print("groupnorm_workspace_floats", args, "library", lib.mi355x_sd_groupnorm_workspace_floats(*args), "emulator",
          emu.mi355x_sd_groupnorm_workspace_floats(*args))
m_cpu, run_cpu, outs = on_emulator(EC.build, "vae_decode", True)
want = run_cpu().sample.float()
m_dev, run_dev, _ = EC.build("vae_decode", False)
got = run_dev().sample.float().cpu()
print("device-planned model vs emulated model: rel", float((got - want).norm() / want.norm()))
pc, pd = EC.last_plan(m_cpu), EC.last_plan(m_dev)
print("launches: emulator-planned", len(pc.prog), "device-planned", len(pd.prog))
for i, (a, b) in enumerate(zip(pc.prog, pd.prog)):
    na, nb = getattr(a[0], "__name__", None) or getattr(a[0], "name", None), getattr(b[0], "__name__", None) or getattr(b[0], "name", None)
    ia = [x for x in a[1] if isinstance(x, (int, float)) and abs(x) < 1e6]
    ib = [x for x in b[1] if isinstance(x, (int, float)) and abs(x) < 1e6]
    if na != nb or ia != ib:
        print("launch", i, "differs:", na, ia, "|", nb, ib)
print("scratch (name: bytes) emulator-planned vs device-planned:")
sa, sb = getattr(pc, "scratch_bytes", None), getattr(pd, "scratch_bytes", None)
print(sa, sb)
