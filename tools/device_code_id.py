#!/usr/bin/env python3
"""Identity of the DEVICE code inside a built libempc.so: sha256 over its `.hip_fatbin` section (the bundled gfx950 code objects),
read with a few lines of ELF parsing -- no LLVM tools, so the same id is computed on the GPU box.

    python3 tools/device_code_id.py [library]           prints the 16-hex-digit id

bench.py stamps its JSON line with this id; tools/profile_summarize.py stamps the PMC summaries it writes.  A bench line takes
`roofline.traffic` / `roofline.compute` from a committed PMC summary only when the ids agree (counters measured on other kernels
say nothing about the ones that ran)."""
import hashlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def section(path, name):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise ValueError("not a 64-bit ELF file: " + path)
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    hdr = lambda i: struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
    stro = hdr(shstrndx)[4]
    for i in range(shnum):
        h = hdr(i)
        end = data.index(b"\0", stro + h[0])
        if data[stro + h[0]:end].decode() == name:
            return data[h[4]:h[4] + h[5]]
    return None


def device_code_id(path=None):
    """16 hex digits, or None when the library is missing / carries no device code"""
    path = path or os.environ.get("EMPC_LIB_PATH") or os.path.join(ROOT, "eagle-mpc_amd", "libempc.so")
    try:
        fat = section(path, ".hip_fatbin")
    except (OSError, ValueError):
        return None
    return hashlib.sha256(fat).hexdigest()[:16] if fat else None


if __name__ == "__main__":
    print(device_code_id(sys.argv[1] if len(sys.argv) > 1 else None))
