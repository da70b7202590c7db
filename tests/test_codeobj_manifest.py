"""The standing rule of round 5, enforced on the built objects (no GPU needed): the device code of the shipped library is the
device code that last passed the GPU suite on hardware.

`profiles/verified_codeobj_manifest.json` holds, per kernel, hashes of the machine code, the kernel descriptor and the metadata
block of the tree named in its `commit` / `evidence` fields (tools/codeobj_compare.py --write-manifest).  A change to a default
kernel body changes a hash and fails here; the ways to green are (a) put the change behind a switch of
`eagle-mpc_amd/csrc/empc_variants.hpp`, off, or (b) run the whole GPU suite on the new tree on an MI355X and write a new manifest
that cites the log.  Objects the manifest does not know must be opt-in at run time (listed below with their switch).
Skipped when the objects are not built (`python -c "import __graft_entry__ as g; g.build()"` builds them)."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "eagle-mpc_amd", "build")
MANIFEST = os.path.join(ROOT, "profiles", "verified_codeobj_manifest.json")
sys.path.insert(0, os.path.join(ROOT, "tools"))

# objects whose kernels have not run on hardware: reachable only with the named environment switch (empc_solver.hip find_table)
OPT_IN = {"empc_inst_1_4_contact.o": "EMPC_EXPERIMENTAL_CONTACT", "empc_inst_1_6_contact.o": "EMPC_EXPERIMENTAL_CONTACT",
          "empc_inst_3_6_contact.o": "EMPC_EXPERIMENTAL_CONTACT", "empc_inst_6_6_contact_mixed.o": "EMPC_EXPERIMENTAL_CONTACT",
          # two ContactModel3D per stage (CT_PAIR3, round 6)
          "empc_inst_4_6_contact_pair.o": "EMPC_EXPERIMENTAL_CONTACT", "empc_inst_6_6_contact_pair.o": "EMPC_EXPERIMENTAL_CONTACT"}


def test_shipped_device_code_is_the_hardware_verified_one():
    if not glob.glob(os.path.join(BUILD, "csrc", "empc_inst_*.o")) or not os.path.isfile("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("built objects or the LLVM tools are missing")
    import codeobj_compare as cc
    objs = glob.glob(os.path.join(BUILD, "**", "*.o"), recursive=True)
    before = {o: os.stat(o).st_mtime_ns for o in objs}
    n, diff, unknown, missing = cc.check_manifest(MANIFEST, BUILD)
    import kernel_resources as kr
    kr.notes(sorted(glob.glob(os.path.join(BUILD, "csrc", "empc_inst_*.o")))[0])
    # the inspectors work on copies: certifying the shipped objects must not touch them (make would relink libempc.so)
    assert before == {o: os.stat(o).st_mtime_ns for o in objs}
    man = json.load(open(MANIFEST))
    assert n == man["kernels"] and not missing, (n, man["kernels"], missing)
    assert not diff, "device code differs from the tree verified on hardware (%s): %s" % (man["commit"], list(cc.demangle(diff).values())[:6])
    assert sorted(unknown) == sorted(OPT_IN), unknown


def test_opt_in_objects_are_refused_without_their_switch():
    src = open(os.path.join(ROOT, "eagle-mpc_amd", "csrc", "empc_solver.hip")).read()
    for switch in set(OPT_IN.values()):
        assert 'getenv("%s")' % switch in src


def test_variant_switches_default_off():
    """every switch of empc_variants.hpp is 0 unless the build line says otherwise, and the product Makefile sets none"""
    import re
    txt = open(os.path.join(ROOT, "eagle-mpc_amd", "csrc", "empc_variants.hpp")).read()
    sw = re.findall(r"#ifndef (EMPC_\w+)\n#define \1 (\d+)", txt)
    assert len(sw) >= 3 and all(v == "0" for _, v in sw), sw
    mk = open(os.path.join(ROOT, "eagle-mpc_amd", "Makefile")).read()
    assert not re.search(r"^(CXXFLAGS|DEVFLAGS|BAKEDFLAGS)\s*:?=.*-DEMPC_", mk, re.M)
    out = subprocess.run(["strings", "-a", os.path.join(ROOT, "eagle-mpc_amd", "libempc.so")], capture_output=True, text=True).stdout
    assert "empc_solver_create" in out
