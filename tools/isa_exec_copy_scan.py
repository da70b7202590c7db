#!/usr/bin/env python3
"""Static scan of the built kernels for one code-generation hazard seen in round 4 (LABNOTES.md, "GPU memory fault"):

the register allocator splits the live range of a long-lived vector register and leaves the split's copy
(`v_mov_b64 vA, vB`) as the LAST instruction of a block that runs under a narrowed EXEC mask (an `if (lane == 0)` body), right
in front of the `s_or_b64 exec, exec, ...` that re-enables the other lanes.  Lanes that sat the block out never get the copy;
a later use of vA under the full mask reads whatever those lanes held before.  (k_linearize<Dims<6,6>, 6, 64, 256, true>: the
record pointer, valid in lane 0 only -> wild global stores, HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.)

The scan lists every  v_mov vA, vB  that is directly followed by  s_or_b64 exec, exec, <saved>  and whose destination is read
again later without an intervening full redefinition -- per kernel, with the line of the disassembly.  A hit is a candidate,
not a proof (the value may be dead for the other lanes); zero hits in the shipped objects is what `make -C eagle-mpc_amd` is
expected to give, and tests/test_build_artifacts.py asserts it for the instantiations that faulted.

    python3 tools/isa_exec_copy_scan.py [object files ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOV = re.compile(r"^\s*v_mov_b(32|64)(?:_e32)?\s+(v\[\d+:\d+\]|v\d+),\s*(v\[\d+:\d+\]|v\d+)\s*$")
FAR = int(os.environ.get("FAR", "300"))  # instructions between the copy and the read that make a candidate
EXEC_OR = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(tok[1:])}


def tokens(line):
    out = set()
    for t in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        out |= regs(t)
    return out


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        txt = subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None and line.strip():
            kernels[cur].append(line.split("//")[0].rstrip())
    return kernels


def scan(lines):
    hits = []
    for i, line in enumerate(lines[:-1]):
        m = MOV.match(line)
        if not m or not EXEC_OR.match(lines[i + 1]):
            continue
        dst = regs(m.group(2))
        # is the destination read later before it is fully rewritten?  (linear scan: a conservative approximation)
        for j in range(i + 2, min(len(lines), i + 20000)):
            l = lines[j]
            parts = l.strip().split(None, 1)
            if len(parts) < 2:
                continue
            ops = parts[1].split(",")
            first = tokens(ops[0])
            rest = tokens(",".join(ops[1:]))
            store = parts[0].startswith(("global_store", "ds_write", "scratch_store", "buffer_store"))
            if (rest & dst) or (store and (first & dst)):
                hits.append((i, line.strip(), j, l.strip()))
                break
            if not store and dst <= first and not parts[0].startswith(("v_fmac", "v_mac", "v_accvgpr_write")):
                break  # redefined
    return hits


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "eagle-mpc_amd", "build", "csrc", "empc_inst_*.o")))
    total = 0
    dm = None
    for obj in objs:
        for name, lines in disassemble(obj).items():
            h = scan(lines)
            # a split of a long live range is read far from the copy; the copies of an `x = c ? a : b` join are read at once
            h = [x for x in h if x[2] - x[0] >= FAR]
            if h:
                d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                print("%s: %s: %d candidate(s)" % (os.path.basename(obj), d[:110], len(h)))
                for i, a, j, b in h[:6]:
                    print("    +%d  %s   ... read at +%d  %s" % (i, a, j, b[:80]))
                total += len(h)
    print("candidates:", total)
    return 0


if __name__ == "__main__":
    sys.exit(main())
