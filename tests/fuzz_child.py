"""Child process of tests/test_boundary_fuzz.py: feeds damaged inputs to the two parsers of the C library that read what a host hands
them -- the program file (mi355x_sd_program_load) and the UNet config text (mi355x_sd_unet_create) -- through ctypes, host side
only. A parser fault would kill this process (that is why it is a child); every call must come back with a status code, and a
non-zero one with a message. Prints one summary line.

    python tests/fuzz_child.py program <file.mi3prg> <n> <seed>
    python tests/fuzz_child.py config <n> <seed>"""
import ctypes
import json
import os
import random
import struct
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paddlemix_amd import _lib  # noqa: E402


def _status(lib, rc):
    if rc != 0:
        msg = lib.mi355x_sd_last_error()
        assert msg, "a refusal carries a message"
    return rc


def fuzz_program(path, n, seed):
    lib = _lib.load()
    raw = open(path, "rb").read()
    rng = random.Random(seed)
    # header, region table, I/O table and launch records live in front of the region contents: the smallest content offset of the
    # region table (paddlemix_amd/export.py) is where the structured part ends
    nreg = struct.unpack_from("<I", raw, 20)[0]
    at, offs = 40, []
    for _ in range(nreg):
        _, _, off = struct.unpack_from("<IQQ", raw, at)
        at += 20
        at += 4 + struct.unpack_from("<I", raw, at)[0]
        if off:
            offs.append(off)
    head = min(offs) if offs else len(raw)
    tmp = tempfile.NamedTemporaryFile(suffix=".mi3prg", delete=False)
    tmp.close()
    accepted = refused = 0
    try:
        for it in range(n):
            b = bytearray(raw)
            kind = rng.randrange(6)
            if kind == 0:      # truncate anywhere (mostly inside the structured part)
                b = b[:rng.randrange(0, head) if rng.random() < 0.8 else rng.randrange(0, len(raw))]
            elif kind == 1:    # flip a few bytes of the structured part
                for _ in range(rng.randrange(1, 6)):
                    b[rng.randrange(0, head)] ^= 1 << rng.randrange(8)
            elif kind == 2:    # a 32-bit field becomes an extreme value
                at = rng.randrange(0, head - 4) & ~3
                struct.pack_into("<I", b, at, rng.choice([0, 1, 0x7fffffff, 0x80000000, 0xffffffff, 0xfffffff0, 1 << 20]))
            elif kind == 3:    # a 64-bit field becomes an extreme value
                at = rng.randrange(0, head - 8) & ~3
                struct.pack_into("<Q", b, at, rng.choice([0, 1 << 31, 1 << 32, (1 << 63) - 1, 1 << 63, (1 << 64) - 1, (1 << 64) - 256]))
            elif kind == 4:    # a slice of the structured part repeated or zeroed
                a0 = rng.randrange(0, head - 64)
                ln = rng.randrange(4, 64)
                b[a0:a0 + ln] = bytes(ln) if rng.random() < 0.5 else b[a0 + ln:a0 + 2 * ln]
            else:              # garbage appended / the file cut to its first bytes
                b = b + bytes(rng.randrange(256) for _ in range(rng.randrange(1, 64))) if rng.random() < 0.5 else b[:rng.randrange(0, 40)]
            open(tmp.name, "wb").write(bytes(b))
            h = ctypes.c_void_p()
            rc = _status(lib, lib.mi355x_sd_program_load(tmp.name.encode(), ctypes.byref(h)))
            if rc == 0:
                accepted += 1
                # what an accepting loader hands back must be usable without a device: sizes, io table
                nb = ctypes.c_size_t()
                _status(lib, lib.mi355x_sd_program_device_bytes(h, ctypes.byref(nb)))
                assert lib.mi355x_sd_program_num_launches(h) >= 0
                nio = lib.mi355x_sd_program_num_io(h)
                assert 0 <= nio < 4096
                for i in range(nio):
                    name = ctypes.c_char_p()
                    is_out, dt, nd = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                    shp = (ctypes.c_int64 * 4)()
                    ptr = ctypes.c_void_p()
                    nbytes = ctypes.c_size_t()
                    _status(lib, lib.mi355x_sd_program_io_info(h, i, ctypes.byref(name), ctypes.byref(is_out), ctypes.byref(dt), shp,
                                                                ctypes.byref(nd), ctypes.byref(nbytes), ctypes.byref(ptr)))
                _status(lib, lib.mi355x_sd_program_destroy(h))
            else:
                refused += 1
    finally:
        os.unlink(tmp.name)
    print(f"program fuzz: {n} inputs, {accepted} accepted, {refused} refused, no fault")


def fuzz_config(n, seed):
    lib = _lib.load()
    rng = random.Random(seed)
    bases = [open(os.path.join(ROOT, "scripts", "c", f)).read() for f in ("sdxl_unet_config.json", "sd15_unet_config.json")]
    accepted = refused = 0
    for it in range(n):
        s = rng.choice(bases)
        kind = rng.randrange(7)
        if kind == 0:
            s = s[:rng.randrange(0, len(s))]
        elif kind == 1:
            cs = list(s)
            for _ in range(rng.randrange(1, 5)):
                cs[rng.randrange(len(cs))] = rng.choice('{}[]",:0-9.eE tnf\\\x00\xff')
            s = "".join(cs)
        elif kind == 2:    # a number becomes extreme / negative / fractional
            d = json.loads(s)
            k = rng.choice([k for k, v in d.items() if isinstance(v, (int, list))])
            v = rng.choice([0, -1, -320, 1, 3, 7, 2 ** 31 - 1, 2 ** 31, 2 ** 40, 1e30, 0.5, 1e-9])
            d[k] = [v if rng.random() < 0.5 else x for x in d[k]] if isinstance(d[k], list) and rng.random() < 0.7 else v
            s = json.dumps(d)
        elif kind == 3:    # wrong types
            d = json.loads(s)
            k = rng.choice(list(d))
            d[k] = rng.choice([None, True, "x", [], {}, [[1, 2], [3]], [None], ["a", 1], 1.5, {"a": [1]}])
            s = json.dumps(d)
        elif kind == 4:    # list lengths that disagree
            d = json.loads(s)
            k = rng.choice([k for k, v in d.items() if isinstance(v, list)])
            d[k] = d[k][:rng.randrange(0, len(d[k]))] if rng.random() < 0.5 else d[k] * rng.randrange(2, 40)
            s = json.dumps(d)
        elif kind == 5:    # deep nesting / very long tokens
            s = rng.choice(["[" * 5000, "[" * 1000000, "{\"a\":[" * 400000, "{\"a\":" * 3000, "\"" + "x" * 100000, "{\"block_out_channels\":[" + "1," * 50000 + "1]}",
                            "{\"" + "k" * 70000 + "\":1}", "-" * 1000, "1e" + "9" * 400])
        else:              # unknown and duplicated keys
            d = json.loads(s)
            d["not_a_unet_key_%d" % it] = rng.choice([1, "s", [1, 2]])
            s = json.dumps(d)
            s = s[:-1] + "," + s[1:] if rng.random() < 0.5 else s
        h = ctypes.c_void_p()
        try:
            text = s.encode("utf-8", "surrogatepass")
        except UnicodeEncodeError:
            text = s.encode("latin-1", "replace")
        rc = _status(lib, lib.mi355x_sd_unet_create(text.split(b"\x00")[0] if rng.random() < 0.5 else text, ctypes.byref(h)))
        if rc == 0:
            accepted += 1
            np_ = lib.mi355x_sd_unet_num_params(h)
            assert 0 <= np_ < 100000
            for i in range(0, np_, max(1, np_ // 50)):
                name = ctypes.c_char_p()
                shp = (ctypes.c_int64 * 4)()
                nd = ctypes.c_int()
                _status(lib, lib.mi355x_sd_unet_param_info(h, i, ctypes.byref(name), shp, ctypes.byref(nd)))
                assert 1 <= nd.value <= 4 and all(shp[d] > 0 for d in range(nd.value)), (name.value, list(shp), nd.value)
            _status(lib, lib.mi355x_sd_unet_destroy(h))
        else:
            refused += 1
    print(f"config fuzz: {n} inputs, {accepted} accepted, {refused} refused, no fault")


if __name__ == "__main__":
    if sys.argv[1] == "program":
        fuzz_program(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        fuzz_config(int(sys.argv[2]), int(sys.argv[3]))
