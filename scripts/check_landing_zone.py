#!/usr/bin/env python3
"""Build-time check of the early-residual GEMM kernels (csrc/gemm_pipe.hip gemm_pipe_pre_kernel): the residual rows land in
v216 .. v255, registers the compiler is told not to allocate (amdgpu_num_vgpr(216)). That attribute is a budget, not a
reservation -- round 4's first interleaved loop made the allocator use v216+ as MFMA temporaries in an unrolled tail and the
landed rows were overwritten (NaN on the first hardware run). This script disassembles the built libraries and fails if any
instruction of such a kernel other than the hand-written ones (the asm buffer loads that fill the zone, the v_mov reads that empty
it) names a register >= v216. WHICH kernels: every kernel whose code contains the marker instruction gemm_epilogue_pre emits
(`s_mov_b32 m0, m0`), not a list of names.

    python scripts/check_landing_zone.py [lib.so ...]          (default: the three built libraries; runs without a GPU)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BASE = 216
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def code_objects(lib):
    """the gfx950 code objects inside a fat library -> paths of extracted ELF files"""
    tmp = os.path.join("/tmp", "landing_zone_" + os.path.basename(lib))
    os.makedirs(tmp, exist_ok=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={lib}"], capture_output=True)
    out = []
    # the device code sits in the .hip_fatbin section as an offload bundle; roc-obj-ls / clang-offload-bundler unpack it
    sec = os.path.join(tmp, "fatbin")
    subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, sec], check=True)
    data = open(sec, "rb").read()
    if b"CCOB" in data[:64] and b"__CLANG_OFFLOAD_BUNDLE__" not in data:
        sys.exit(f"{lib}: the device code is a COMPRESSED offload bundle (CCOB): build without --offload-compress, or unbundle with "
                 "clang-offload-bundler first -- this script reads uncompressed bundles")
    # uncompressed bundles: "__CLANG_OFFLOAD_BUNDLE__" header, entries (offset, size, triple)
    pos = 0
    idx = 0
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    while True:
        pos = data.find(magic, pos)
        if pos < 0:
            break
        import struct
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        p = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                path = os.path.join(tmp, f"co{idx}.elf")
                open(path, "wb").write(data[pos + off:pos + off + size])
                out.append(path)
                idx += 1
        pos += len(magic)
    return out


MARKER = re.compile(r"^s_mov_b32\s+m0,\s*m0$")


def check_disassembly(text, label):
    """Every kernel that carries the landing-zone MARKER (gemm_epilogue.h: `s_mov_b32 m0, m0`, emitted by gemm_epilogue_pre and by
    nothing else -- hipcc has no reason to move m0 onto itself) is held to the rule, whatever it is called; a kernel NAMED
    gemm_pipe_pre_kernel without the marker is an error too (the marker was lost)."""
    kernels, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is not None and "\t" in ln:
            cur.append(ln.split("//")[0].strip())
    bad, n_k = [], 0
    for name, body in kernels.items():
        marked = any(MARKER.match(i) for i in body)
        if not marked:
            if "gemm_pipe_pre_kernel" in name:
                bad.append((name, "<no landing-zone marker in an early-residual kernel>"))
            continue
        n_k += 1
        for ins in body:
            hi = max([int(g[1] or g[2]) for g in REG.findall(ins)] or [0])
            if hi < BASE:
                continue
            op = ins.split()[0]
            mv = re.match(r"v_mov_b32(?:_e32)?\s+v(\d+),\s*v(2\d\d)$", ins)
            ok = (op in ("buffer_load_dwordx4", "buffer_load_dwordx2") and re.match(r"\S+\s+v\[2\d\d:2\d\d\]", ins)) or \
                 (mv is not None and int(mv.group(1)) < BASE)
            if not ok:
                bad.append((name, ins))
    print(f"{label}: {n_k} kernels with the landing-zone marker, {len(bad)} compiler-generated uses of v{BASE}+")
    for k, ins in bad[:20]:
        print("   ", k[:70], "|", ins)
    return n_k, bad


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "paddlemix_amd", n) for n in ("libmi355x_sd.so", "libmi355x_sd_f16.so", "libmi355x_sd_dbg.so")]
    fail = False
    for lib in libs:
        if lib.endswith(".s"):   # an assembly listing (hipcc -S): same rule, labels instead of symbols
            text = re.sub(r"^(_Z\S+):.*$", lambda m: "0 <" + m.group(1) + ">:", open(lib).read(), flags=re.M)
            text = "\n".join(("\t" + ln.strip() if ln.startswith("\t") and not ln.strip().startswith((".", ";")) else ln) for ln in text.splitlines())
            n_k, bad = check_disassembly(text, lib)
        else:
            text = ""
            for co in code_objects(lib):
                text += subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
            n_k, bad = check_disassembly(text, os.path.basename(lib))
        fail |= bool(bad) or n_k == 0
    sys.exit(1 if fail else 0)


if __name__ == "__main__":
    main()
