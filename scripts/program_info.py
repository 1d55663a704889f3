"""What is inside an exported step program (paddlemix_amd/export.py, *.mi3prg): header, regions by kind, named inputs / outputs and the
launch histogram. Runs anywhere the library loads (no GPU needed: `mi355x_sd_program_load` parses and type-checks on the host).

    python scripts/program_info.py model.mi3prg
"""
import collections
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from paddlemix_amd.export import MAGIC, ExportedProgram  # noqa: E402

path = sys.argv[1]
raw = open(path, "rb").read()
assert raw[:8] == MAGIC, "not a program file"
version, abi, elem, n_regions, n_ops, n_io, ws = struct.unpack_from("<IIIIIIQ", raw, 8)
print(f"{path}: format {version}, ABI {abi}, {'fp16' if elem else 'bf16'} elements, split-K workspace {ws / 2 ** 20:.0f} MiB")
pos = 8 + struct.calcsize("<IIIIIIQ")
kinds = collections.Counter()
size = collections.Counter()
for _ in range(n_regions):
    kind, nbytes, _off = struct.unpack_from("<IQQ", raw, pos)
    pos += 20
    (n,) = struct.unpack_from("<I", raw, pos)
    pos += 4 + n
    kinds[kind] += 1
    size[kind] += nbytes
for k, name in enumerate(("weight", "const", "scratch", "io")):
    print(f"   {name:8s} {kinds[k]:5d} regions  {size[k] / 2 ** 20:10.2f} MiB")
prog = ExportedProgram(path)     # host-side load: every launch checked against the library's entry points
print(f"   device buffer the caller provides: {prog.device_bytes() / 2 ** 20:.2f} MiB; {prog.num_launches} launches per run")
for io in prog.info():
    dt = ("fp32", "elem16", "int32", "uint8")[io["dtype"]]
    print(f"   {'output' if io['is_output'] else 'input ':6s} {io['name']:16s} {dt:7s} {io['shape']}")
for _ in range(n_io):      # skip the I/O table to reach the launch list
    pos += struct.calcsize("<IIII4q")
    (n,) = struct.unpack_from("<I", raw, pos)
    pos += 4 + n
ops = collections.Counter()
for _ in range(n_ops):
    (n,) = struct.unpack_from("<I", raw, pos)
    name = raw[pos + 4: pos + 4 + n].decode()
    pos += 4 + n
    (nargs,) = struct.unpack_from("<I", raw, pos)
    pos += 4 + 20 * nargs
    ops[name] += 1
print("   launches: " + ", ".join(f"{k.replace('mi355x_sd_', '')} x{v}" for k, v in ops.most_common()))
