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
