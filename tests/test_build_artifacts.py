"""What the compiler made of the hot kernels, read from the built objects (no GPU needed): registers, spills, scratch and the
one code-generation hazard of round 4.  These are regression guards for properties DESIGN.md quotes -- a change that pushes the
shipped backward pass or the baked rollout into scratch memory, or brings back a vector-register split copy in front of an EXEC
restore inside a linearize kernel, fails here instead of showing up as a slower (or faulting) kernel on the GPU box.
Skipped when the objects are not there (`python -c "import __graft_entry__ as g; g.build()"` makes them)."""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "eagle-mpc_amd", "build", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def need(*names):
    paths = [os.path.join(OBJ, n) for n in names]
    if not all(os.path.isfile(p) for p in paths) or not os.path.isfile("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("built objects or the LLVM tools are missing")
    return paths


def resources(obj):
    """{demangled kernel name: {vgpr, agpr, sgpr_spill, vgpr_spill, scratch}} of one object"""
    import kernel_resources as kr
    rows = {}
    blocks = kr.notes(obj).split("- .agpr_count")[1:]
    names = [re.search(r"\.name:\s*(\S+)", b).group(1) for b in blocks]
    dm = kr.demangle(names)
    for b, n in zip(blocks, names):
        g = lambda k: int(re.search(r"\." + k + r":\s*(\d+)", b).group(1))
        rows[dm[n].replace("empc::", "")] = {"agpr": int(re.match(r":\s*(\d+)", b).group(1)), "vgpr": g("vgpr_count"), "sgpr_spill": g("sgpr_spill_count"),
                                              "vgpr_spill": g("vgpr_spill_count"), "scratch": g("private_segment_fixed_size")}
    return rows


def pick(rows, *parts):
    hit = [v for k, v in rows.items() if all(p in k for p in parts)]
    assert len(hit) == 1, (parts, [k for k in rows if parts[0] in k])
    return hit[0]


def test_chain_kernels_stay_out_of_scratch():
    """the shipped backward pass (9-dof class) and the baked rollout / linearize of the north-star robot: no spilled vector
    registers, no scratch; the baked linearize keeps its scalar spills at the level DESIGN.md quotes (<= 32)"""
    generic, baked, contact = need("empc_inst_4_6.o", "empc_inst_baked_arm3.o", "empc_inst_baked_arm3_contact.o")
    bwd = pick(resources(generic), "k_backward4<Dims<4, 6, RuntimeModel>, false>")
    assert bwd["vgpr_spill"] == 0 and bwd["scratch"] == 0, bwd
    for obj, ct in ((baked, "0"), (contact, "3")):
        r = resources(obj)
        roll = pick(r, "k_rollout6<Dims<4, 6, BakedHex370Arm3>, %s, false>" % ct)
        assert roll["vgpr_spill"] == 0 and roll["scratch"] == 0, roll
        lean = pick(r, "k_linearize<Dims<4, 6, BakedHex370Arm3>, %s, 32, 256, false>" % ct)
        assert lean["sgpr_spill"] <= 32 and lean["vgpr"] <= 256, lean  # two wavefronts per SIMD


def test_no_split_copy_in_front_of_an_exec_restore_in_linearize():
    """the hazard behind the GPU memory fault of round 4 (LABNOTES.md section 3.0b): none of its shape in any linearize or calc
    kernel of the library (the per-knot kernels with the largest register footprints)"""
    import isa_exec_copy_scan as scan
    need("empc_inst_6_6_contact6.o", "empc_inst_4_6_contact6.o")  # (the two instantiations that showed it)
    for obj in sorted(glob.glob(os.path.join(OBJ, "empc_inst_*.o"))):
        for name, lines in scan.disassemble(obj).items():
            if "k_linearize" not in name and "k_calc" not in name:
                continue
            far = [h for h in scan.scan(lines) if h[2] - h[0] >= scan.FAR]
            assert not far, (os.path.basename(obj), name, far[:3])


def test_nan_guards_survive_the_nan_free_build_of_the_baked_units():
    """The baked translation units are compiled -fno-honor-nans -fno-signed-zeros (Makefile BAKEDFLAGS) so that products with a
    robot's structural zeros fold; the solver's NaN / overflow guards are tests on the bit pattern (is_nan, bad_number:
    empc_dev_math.hpp) precisely so that this flag cannot fold them.  Checked on the machine code: every
    model-touching kernel of a baked unit carries as many 64-bit integer compares (and v_cmp_class) as its runtime-model sibling,
    which is built without the flag -- and more than none where the source has guards (the rollouts)."""
    import isa_exec_copy_scan as scan
    guard = re.compile(r"\bv_cmpx?_[a-z]+_u64|\bv_cmp_class_f64")
    pairs = [("empc_inst_baked_arm3.o", "empc_inst_4_6.o", "BakedHex370Arm3"), ("empc_inst_baked_arm3_contact.o", "empc_inst_4_6_contact.o", "BakedHex370Arm3"),
             ("empc_inst_baked_arm5.o", "empc_inst_6_6.o", "BakedHextiltArm5")]
    checked = with_guards = 0
    for baked_o, runtime_o, model in pairs:
        b, r = need(baked_o, runtime_o)
        kb, kr = scan.disassemble(b), scan.disassemble(r)
        count = lambda lines: sum(1 for l in lines if guard.search(l))
        # mangled names: the model type is the only difference (N4empc15BakedHex370Arm3E <-> NS0_12RuntimeModelE / N4empc12RuntimeModelE)
        for name, lines in kb.items():
            if model not in name or not re.search(r"k_rollout|k_linearize|k_calc|k_plant", name):
                continue
            sib = [n for n in kr if re.sub(r"\d+%s" % model, "12RuntimeModel", name) == n]
            assert len(sib) == 1, (name, baked_o)
            nb, nr = count(lines), count(kr[sib[0]])
            assert nb == nr, (name, nb, nr)
            checked += 1
            with_guards += nb > 0
    assert checked >= 12 and with_guards >= 6, (checked, with_guards)


def test_no_flat_memory_instructions_in_any_kernel():
    """Every global access of the kernels is a global_* (or scalar) instruction.  A FLAT load / store -- what the compiler emits when
    a pointer lost its address space, e.g. after passing through an inline-asm operand -- also counts on the LDS counter, so every
    later wait for an LDS read waits for it: found in round 6 in a variant of the backward pass (EMPC_BWD_VPTR), where 64 flat
    stores per knot pair would have sat in front of every `s_waitcnt lgkmcnt`.  Checked on all shipped objects and, when the variant
    harness compiles (hipcc present), on the backward kernels with every variant switch on."""
    import tempfile
    LLVM = "/opt/rocm/lib/llvm/bin"
    objs = sorted(glob.glob(os.path.join(OBJ, "empc_*.o")))
    if not objs or not os.path.isfile(LLVM + "/llvm-objdump"):
        pytest.skip("built objects or the LLVM tools are missing")

    def flat_by_function(obj):
        """{function symbol: number of flat_* instructions} of the gfx950 code object inside `obj` (empty: no device code)"""
        with tempfile.TemporaryDirectory() as d:
            fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
            if subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")], capture_output=True).returncode != 0:
                return {}
            subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--output=" + co], check=True)
            txt = subprocess.run("%s/llvm-objdump -d %s | grep -E '^[0-9a-f]+ <|^\\s+flat_'" % (LLVM, co), shell=True, capture_output=True, text=True).stdout
        out, cur = {}, None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
            elif cur is not None:
                out[cur] = out.get(cur, 0) + 1
        return out

    bad = {}
    for o in objs:
        for name, n in flat_by_function(o).items():
            bad[os.path.basename(o) + ":" + name[:60]] = n
    # Known exception, found by this very test and left alone (the default device code is locked to the hardware-verified one): in
    # the RUNTIME-MODEL family of the 11-DoF class a lambda of linearize_unit2 is not inlined (a called function: generic pointers,
    # 42 flat accesses).  The shipped 11-DoF robot runs the baked family (empc_inst_baked_arm5.o: clean); to be fixed with the next
    # hardware run (force-inline + a new manifest).
    known = {k: v for k, v in bad.items() if k.startswith("empc_inst_6_6")
             and "linearize_unit2" in k and v == 42}
    bad = {k: v for k, v in bad.items() if k not in known}
    assert not bad, bad
    if not os.path.isfile("/opt/rocm/bin/hipcc"):
        return
    import variant_verdicts as vv
    with tempfile.TemporaryDirectory() as d:
        for tag in ("bwd", "bwdm4"):
            obj = os.path.join(d, tag + ".o")
            vv.compile_harness(vv.VARIANTS[tag][0], obj)
            assert not flat_by_function(obj), (tag, flat_by_function(obj))
