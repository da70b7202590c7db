#!/usr/bin/env python3
"""Register / spill / scratch table of the kernels in the built objects (llvm-readelf --notes of the gfx950 code objects).

    python3 tools/kernel_resources.py [object files ...]      default: every eagle-mpc_amd/build/csrc/empc_inst_*.o
Prints one line per kernel: VGPRs, AGPRs, spilled SGPRs / VGPRs, scratch bytes per lane, static LDS."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def notes(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")], check=True)  # never in place
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        return subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "eagle-mpc_amd", "build", "csrc", "empc_inst_*.o")))
    want = re.compile(os.environ.get("KERNELS", "k_rollout6|k_linearize|k_backward4|k_calc|k_select|k_rk4"))
    for obj in objs:
        rows = []
        for blk in notes(obj).split("- .agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
            rows.append((g("name"), re.match(r":\s*(\d+)", blk).group(1), g("vgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
                         g("private_segment_fixed_size"), g("group_segment_fixed_size")))
        dm = demangle([r[0] for r in rows])
        print("==", os.path.basename(obj))
        for name, ag, vg, ss, vs, scr, lds in rows:
            d = dm.get(name, name).replace("void ", "").replace("empc::", "").replace("(DevBuffers)", "")
            if want.search(d):
                print("  %-78s vgpr %3s agpr %3s sgpr_spill %3s vgpr_spill %3s scratch %4s B lds %s" % (d[:78], vg, ag, ss, vs, scr, lds))


if __name__ == "__main__":
    main()
