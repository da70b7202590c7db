#!/usr/bin/env python3
"""Static instruction mix of the loops of a kernel (from `llvm-objdump -d --symbolize-operands` of the built object).

For the chain kernels the time of a launch is (knots) x (cycles per knot), and a wave64 FP64 instruction costs ~5 issue cycles
whether or not it depends on its predecessor (LABNOTES.md section 3.1): the instruction count of the knot loop is the first-order
model of the kernel.  This prints, for every backward branch (a loop) of the selected kernels, the number of instructions
between the loop head and the branch by class: FP64 vector, MFMA, other vector, LDS, global / scratch memory, scalar loads,
other scalar, waits, barriers.  Nested loops are listed separately (the outer one includes the inner one's body once).

    python3 tools/isa_loop_profile.py <object> <kernel substring> [min instructions]
    e.g. python3 tools/isa_loop_profile.py eagle-mpc_amd/build/csrc/empc_inst_4_6.o "k_backward4<empc::Dims<4, 6, empc::RuntimeModel>, false>"
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")], check=True)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--symbolize-operands", co], capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(_Z\S+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None and line.strip():
            out[cur].append(line.split("//")[0].rstrip())
    return out


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_") and "f64" in op:
        return "fp64"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load")):
        return "s_load"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    obj, want = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    ks = kernels(obj)
    names = list(ks)
    dm = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for name, d in zip(names, dm):
        if want not in d:
            continue
        lines = ks[name]
        label_at, instr = {}, []
        for l in lines:
            m = re.match(r"^[0-9a-f]+ <(L\d+)>:", l)
            if m:
                label_at[m.group(1)] = len(instr)
                continue
            parts = l.strip().split(None, 1)
            if parts:
                instr.append((parts[0], parts[1] if len(parts) > 1 else ""))
        total = {}
        for op, _ in instr:
            total[classify(op)] = total.get(classify(op), 0) + 1
        print(d[:140])
        print("  whole kernel: %d instructions  %s" % (len(instr), " ".join("%s %d" % kv for kv in sorted(total.items()))))
        loops = []
        for i, (op, args) in enumerate(instr):
            if op.startswith(("s_cbranch", "s_branch")):
                m = re.search(r"(L\d+)", args)
                if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
                    loops.append((label_at[m.group(1)], i))
        for a, b in sorted(loops, key=lambda ab: ab[0] - ab[1]):
            if b - a < floor:
                continue
            mix = {}
            for op, _ in instr[a:b + 1]:
                mix[classify(op)] = mix.get(classify(op), 0) + 1
            print("  loop +%d..+%d: %d instructions  %s" % (a, b, b - a + 1, " ".join("%s %d" % kv for kv in sorted(mix.items()))))
            if os.environ.get("MNEMONICS") and (a, b) == sorted(loops, key=lambda ab: ab[0] - ab[1])[0]:
                # MNEMONICS=1: the largest loop broken down by mnemonic (what the "other vector" and scalar instructions are)
                mn = {}
                for op, _ in instr[a:b + 1]:
                    mn[op] = mn.get(op, 0) + 1
                print("    " + "  ".join("%s %d" % kv for kv in sorted(mn.items(), key=lambda kv: -kv[1]) if kv[1] >= 3))


if __name__ == "__main__":
    main()
