#!/usr/bin/env python3
"""CPU-side verdicts of the build-time variants of the backward pass (csrc/empc_variants.hpp), no GPU needed:

  static   the backward kernels compiled alone (tools/variant_harness/bwd_only.hip) with the variant's macros: VGPRs, AGPRs, spilled
           registers, scratch, and the instruction mix of the knot loop (tools/isa_loop_profile.py) next to the default build's;
  bitwise  tools/emulator_variant_equal.py: the kernel bodies on the lane emulator with and without the macros -- tape, gains, sums,
           trial rollouts and whole box-solver solves bit for bit (variants that leave the arithmetic alone must pass);
  parity   the emulator parity tests (phase parity against the oracle, step-wise parity in both directions, the north-star contract)
           on the variant (EMU_MACROS): the verdict of a variant that MOVES a rounding.

    python3 tools/variant_verdicts.py [--static-only] [--out profiles/r06_variant_resources.md] [tag ...]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_loop_profile as ilp
import kernel_resources as kr

# tag -> (macros, what, arithmetic)
BWD = ["EMPC_BWD_R4B=1", "EMPC_BWD_SYMTILES=1", "EMPC_BWD_GLDS=1", "EMPC_BOX_LDS=1", "EMPC_ANY_BALLOT=1", "EMPC_BWD_OVERLAP=1", "EMPC_BWD_FUSE=1",
       "EMPC_BWD_VPTR=1"]
VARIANTS = {
    "default": ([], "the shipped build (every switch 0)", "-"),
    "r4b": (["EMPC_BWD_R4B=1"], "round-4 late changes: triangle symmetrisation, 16-byte record moves", "same"),
    "sym": (["EMPC_BWD_SYMTILES=1"], "mirror-image tiles of Q / Vxx not computed", "moves"),
    "glds": (["EMPC_BWD_GLDS=1"], "record by LDS-DMA into a second LDS buffer", "same"),
    "boxlds": (["EMPC_BOX_LDS=1"], "box QP out of line, its vectors in LDS", "same"),
    "mfma4": (["EMPC_BWD_MFMA4=1"], "products on v_mfma_f64_4x4x4_4b", "same"),
    "overlap": (["EMPC_BWD_OVERLAP=1"], "Qxx tiles issued between the pieces of the LLT", "same"),
    "fuse": (["EMPC_BWD_FUSE=1"], "k, Quu k and the LLT's verdict as wave broadcasts instead of LDS hand-overs", "same"),
    "vptr": (["EMPC_BWD_VPTR=1"], "output pointers per lane in vector registers (scalar-register relief)", "same"),
    "tri": (["EMPC_REC_TRI=1"], "Lxx / Luu as upper triangles in the record (1104 -> 912 doubles)", "moves"),
    "bwd": (BWD, "every backward variant on the 16 x 16 x 4 form", "moves"),
    "bwdm4": (BWD + ["EMPC_BWD_MFMA4=1"], "every backward variant on the 4 x 4 x 4 form", "moves"),
    "alltri": (BWD + ["EMPC_ROLL_CAP_LDS=1", "EMPC_ROLL_GAP_EARLY=1", "EMPC_REC_TRI=1"], "everything", "moves"),
}
KERNELS = [("9-DoF", "Dims<4, 6, RuntimeModel>, false>", "4, 6, empc::RuntimeModel>, false"),
           ("9-DoF box", "Dims<4, 6, RuntimeModel>, true>", "4, 6, empc::RuntimeModel>, true"),
           ("11-DoF", "Dims<6, 6, RuntimeModel>, false>", "6, 6, empc::RuntimeModel>, false"),
           ("11-DoF box", "Dims<6, 6, RuntimeModel>, true>", "6, 6, empc::RuntimeModel>, true")]


def compile_harness(macros, out):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "eagle-mpc_amd", "csrc"),
           "--offload-arch=gfx950"] + ["-D" + m for m in macros] + ["-c", os.path.join(ROOT, "tools", "variant_harness", "bwd_only.hip"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)


def resources(obj):
    rows = {}
    names = []
    blks = kr.notes(obj).split("- .agpr_count")[1:]
    for blk in blks:
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        names.append(g("name"))
    dm = kr.demangle(names)
    for blk, name in zip(blks, names):
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        rows[dm[name]] = dict(agpr=int(re.match(r":\s*(\d+)", blk).group(1)), vgpr=int(g("vgpr_count")), sgpr_spill=int(g("sgpr_spill_count")),
                              vgpr_spill=int(g("vgpr_spill_count")), scratch=int(g("private_segment_fixed_size")))
    return rows


def knot_loop(obj, want):
    """instruction mix of the largest loop of the kernel; a body that exists twice (EMPC_BWD_GLDS) is reported per knot"""
    ks = ilp.kernels(obj)
    names = list(ks)
    dm = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for name, d in zip(names, dm):
        if want not in d or "k_backward4<" not in d:
            continue
        label_at, instr = {}, []
        for l in ks[name]:
            m = re.match(r"^[0-9a-f]+ <(L\d+)>:", l)
            if m:
                label_at[m.group(1)] = len(instr)
                continue
            parts = l.strip().split(None, 1)
            if parts:
                instr.append((parts[0], parts[1] if len(parts) > 1 else ""))
        loops = []
        for i, (op, args) in enumerate(instr):
            if op.startswith(("s_cbranch", "s_branch")):
                m = re.search(r"(L\d+)", args)
                if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
                    loops.append((label_at[m.group(1)], i))
        a, b = max(loops, key=lambda ab: ab[1] - ab[0])
        mix = {}
        for op, _ in instr[a:b + 1]:
            c = ilp.classify(op)
            mix[c] = mix.get(c, 0) + 1
            if op.startswith("v_accvgpr"):
                mix["accvgpr"] = mix.get("accvgpr", 0) + 1
            if op.startswith("scratch_"):
                mix["scratch"] = mix.get("scratch", 0) + 1
            if op.startswith("flat_"):
                mix["flat"] = mix.get("flat", 0) + 1
        mix["total"] = b - a + 1
        return mix
    return None


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    static_only = "--static-only" in sys.argv
    out = os.path.join(ROOT, "profiles", "r06_variant_resources.md")
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
        args.remove(out)
    tags = args or list(VARIANTS)
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for tag in tags:
            macros, what, arith = VARIANTS[tag]
            obj = os.path.join(d, tag + ".o")
            compile_harness(macros, obj)
            r = resources(obj)
            entry = {"macros": macros, "what": what, "arithmetic": arith, "kernels": {}}
            for label, key, want in KERNELS:
                k = next((v for n, v in r.items() if key in n.replace("empc::", "")), None)
                mix = knot_loop(obj, want)
                glds = any(m.startswith("EMPC_BWD_GLDS") for m in macros) and label.startswith("9-DoF")
                if mix and glds:  # two copies of the knot body in the loop
                    mix = {kk: vv / 2.0 for kk, vv in mix.items()}
                entry["kernels"][label] = {"resources": k, "knot_loop": mix}
            res[tag] = entry
            print(tag, json.dumps(entry["kernels"]["9-DoF"]), flush=True)
    verdicts = {}
    if not static_only:
        for tag in tags:
            macros, what, arith = VARIANTS[tag]
            if not macros:
                continue
            t0 = time.time()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emulator_variant_equal.py")] + macros, capture_output=True, text=True, cwd=ROOT)
            lines = [l for l in r.stdout.splitlines() if "bitwise equal" in l]
            v = {"bitwise": "%d of %d comparisons bit-identical" % (sum("bitwise equal: True" in l for l in lines), len(lines)), "bitwise_all": r.returncode == 0}
            if arith == "moves" or r.returncode != 0:
                env = dict(os.environ, EMU_MACROS=" ".join(macros))
                p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "tests/test_emulator_parity.py", "tests/test_teacher_forced_emulator.py",
                                    "tests/test_parity_contract_emulator.py", "tests/test_two_contacts_emulator.py"], capture_output=True, text=True, cwd=ROOT, env=env)
                v["parity"] = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "no output"
                v["parity_ok"] = p.returncode == 0
            v["seconds"] = round(time.time() - t0)
            verdicts[tag] = v
            print(tag, v, flush=True)
    write_md(out, res, verdicts, static_only)


def write_md(out, res, verdicts, static_only):
    L = []
    L.append("# Build-time variants of the backward pass: static resources and CPU verdicts (round 6)\n")
    L.append("Written by `python3 tools/variant_verdicts.py` (hipcc of this image, gfx950; `tools/variant_harness/bwd_only.hip`).  No GPU ran any of this:")
    L.append("the pool was closed to this repository (gpurun_out/r06a_call.log).  Registers / spills / scratch from the code-object metadata;")
    L.append("knot loop = the largest loop of the kernel (instructions by class per knot; `accvgpr` = v_accvgpr_read/write, the cost of living")
    L.append("above 256 VGPRs; `scratch` = scratch_load/store inside the loop).  First-order model: a wave64 instruction ~ 4-5 issue cycles, a")
    L.append("v_mfma_f64_16x16x4 64 cycles, a v_mfma_f64_4x4x4_4b 16 (to be measured: tools/probes/mfma_f64_4x4_probe.hip).\n")
    for label, _, _ in KERNELS:
        L.append("## %s: `k_backward4`\n" % label)
        L.append("| variant | macros | VGPR | AGPR | spilled VGPR | spilled SGPR | scratch B | knot loop | fp64 | mfma | lds | valu other (accvgpr) | salu | vmem (scratch, flat) | wait |")
        L.append("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
        for tag, e in res.items():
            k = e["kernels"][label]
            r, m = k["resources"], k["knot_loop"]
            if r is None or m is None:
                continue
            f = lambda x: ("%g" % x)
            L.append("| %s | %s | %d | %d | %d | %d | %d | %s | %s | %s | %s | %s (%s) | %s | %s (%s, %s) | %s |" % (
                tag, " ".join(x.replace("EMPC_", "").replace("=1", "") for x in e["macros"]) or "-", r["vgpr"], r["agpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch"],
                f(m["total"]), f(m.get("fp64", 0)), f(m.get("mfma", 0)), f(m.get("lds", 0)), f(m.get("valu_other", 0)), f(m.get("accvgpr", 0)),
                f(m.get("salu", 0)), f(m.get("vmem", 0)), f(m.get("scratch", 0)), f(m.get("flat", 0)), f(m.get("wait", 0))))
        L.append("")
    if verdicts:
        L.append("## CPU verdicts (lane emulator)\n")
        L.append("| variant | what | arithmetic | `tools/emulator_variant_equal.py` | emulator parity tests on the variant (`EMU_MACROS`) |")
        L.append("|---|---|---|---|---|")
        for tag, v in verdicts.items():
            e = res[tag]
            L.append("| %s | %s | %s | %s | %s |" % (tag, e["what"], {"same": "unchanged", "moves": "a rounding moves"}.get(e["arithmetic"], "-"),
                                                 v["bitwise"] + (" -- **bit-identical**" if v["bitwise_all"] else ""),
                                                 v.get("parity", "not needed (bit-identical)")))
        L.append("")
    open(out, "w").write("\n".join(L))
    print("wrote", out)


if __name__ == "__main__":
    main()
