#!/usr/bin/env python3
"""Per-kernel comparison of the gfx950 code objects of two builds: is the machine code of a kernel the same, byte for byte?

    python3 tools/codeobj_compare.py <build dir A> <build dir B> [--json out.json]
    python3 tools/codeobj_compare.py --write-manifest <file> <build dir> <commit> <evidence>     hashes of one build
    python3 tools/codeobj_compare.py --check-manifest <file> <build dir>                         (tests/test_codeobj_manifest.py)

For every `csrc/empc_*.o` present in both directories the gfx950 code object is taken out of the fat binary
(llvm-objcopy + clang-offload-bundler) and every kernel (each `<name>.kd` descriptor and the function it describes) is hashed:
the bytes of the function in .text, the 64-byte kernel descriptor, and the kernel's metadata block (registers, scratch, LDS,
argument layout).  Used to show that the shipped library, with every switch of empc_variants.hpp off, runs the same device code
as the last tree whose GPU suite passed on hardware (build that commit from `git archive` into a scratch directory first).
Exit code 0: every kernel present in both builds is identical; 1: at least one differs."""
import glob
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj, d):
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    # an explicit output file: with none, llvm-objcopy rewrites its input in place (same content, new mtime -> make relinks)
    r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")],
                       capture_output=True, text=True)
    if r.returncode != 0:
        return None  # an object without device code (the host driver)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    return co


def sections(co):
    """name -> (address, file offset, size)"""
    out = {}
    txt = subprocess.run([LLVM + "/llvm-readelf", "-S", "-W", co], capture_output=True, text=True, check=True).stdout
    for m in re.finditer(r"\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", txt):
        out[m.group(1)] = (int(m.group(2), 16), int(m.group(3), 16), int(m.group(4), 16))
    return out


def symbols(co):
    """name -> (value, size, type)"""
    out = {}
    txt = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", co], capture_output=True, text=True, check=True).stdout
    for line in txt.splitlines():
        f = line.split()
        if len(f) >= 8 and f[0].rstrip(":").isdigit():
            out[f[7]] = (int(f[1], 16), int(f[2]), f[3])
    return out


def metadata_blocks(co):
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = {}
    for blk in txt.split("  - .agpr_count")[1:]:
        m = re.search(r"\.name:\s*(\S+)", blk)
        if m:
            end = blk.find("\namdhsa.target")
            out[m.group(1)] = blk if end < 0 else blk[:end]
    return out


def kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        co = code_object(obj, d)
        if co is None:
            return {}
        sec, sym, meta = sections(co), symbols(co), metadata_blocks(co)
        data = open(co, "rb").read()

        def read(addr, size):
            for name, (a, off, sz) in sec.items():
                if a <= addr and addr + size <= a + sz and name not in (".bss", ".note"):
                    return data[off + addr - a: off + addr - a + size]
            raise KeyError(hex(addr))

        out = {}
        for name, (val, size, typ) in sym.items():
            if not name.endswith(".kd"):
                continue
            fn = name[:-3]
            if fn not in sym:
                continue
            fv, fs, _ = sym[fn]
            out[fn] = {"text": hashlib.sha256(read(fv, fs)).hexdigest(), "text_bytes": fs,
                       "descriptor": hashlib.sha256(read(val, size)).hexdigest(),
                       "metadata": hashlib.sha256(meta.get(fn, "").encode()).hexdigest()}
        return out


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def build_hashes(build):
    out = {}
    for p in sorted(glob.glob(os.path.join(build, "csrc", "empc_*.o"))):
        k = kernels(p)
        if k:
            out[os.path.basename(p)] = {name: h["text"][:16] + h["descriptor"][:16] + h["metadata"][:16] for name, h in k.items()}
    return out


def check_manifest(manifest, build):
    """-> (kernels checked, [different], [kernels of objects the manifest does not know])"""
    want = json.load(open(manifest))
    have = build_hashes(build)
    diff, unknown, n = [], [], 0
    for obj, ks in have.items():
        if obj not in want["objects"]:
            unknown.append(obj)
            continue
        for name, h in ks.items():
            n += 1
            if want["objects"][obj].get(name) != h:
                diff.append(obj + ": " + name)
    missing = [o for o in want["objects"] if o not in have]
    return n, diff, unknown, missing


def main():
    if "--write-manifest" in sys.argv:
        i = sys.argv.index("--write-manifest")
        out, build, commit, evidence = sys.argv[i + 1:i + 5]
        h = build_hashes(build)
        json.dump({"what": "per-kernel hashes (sha256 prefixes of .text bytes, kernel descriptor, metadata block) of the gfx950 code "
                           "objects of the tree whose GPU suite last passed on hardware; the shipped build must reproduce them "
                           "(tests/test_codeobj_manifest.py) or come with a new hardware run and a new manifest",
                   "commit": commit, "evidence": evidence, "compiler": subprocess.run(["/opt/rocm/bin/hipcc", "--version"],
                                                                                  capture_output=True, text=True).stdout.splitlines()[0:2],
                   "kernels": sum(len(v) for v in h.values()), "objects": h}, open(out, "w"), indent=0, sort_keys=True)
        print("wrote", out, sum(len(v) for v in h.values()), "kernels")
        return
    if "--check-manifest" in sys.argv:
        i = sys.argv.index("--check-manifest")
        n, diff, unknown, missing = check_manifest(sys.argv[i + 1], sys.argv[i + 2])
        print("kernels checked %d, different %d, objects not in the manifest %s, objects missing from the build %s" % (n, len(diff), unknown, missing))
        for d in demangle(diff).values():
            print("  ", d[:160])
        sys.exit(1 if diff or missing else 0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if jout:
        args.remove(jout)
    a, b = args
    names = sorted(set(os.path.basename(p) for p in glob.glob(os.path.join(a, "csrc", "empc_*.o"))) &
                   set(os.path.basename(p) for p in glob.glob(os.path.join(b, "csrc", "empc_*.o"))))
    report = {"a": a, "b": b, "objects": {}, "kernels_compared": 0, "kernels_identical": 0, "different": [], "only_in_a": [], "only_in_b": []}
    for n in names:
        if n == "empc_capi.o":
            continue
        ka, kb = kernels(os.path.join(a, "csrc", n)), kernels(os.path.join(b, "csrc", n))
        both = sorted(set(ka) & set(kb))
        same = [k for k in both if ka[k] == kb[k]]
        diff = [k for k in both if ka[k] != kb[k]]
        dm = demangle(diff + sorted(set(ka) ^ set(kb)))
        report["objects"][n] = {"kernels": len(both), "identical": len(same), "text_bytes": sum(ka[k]["text_bytes"] for k in both)}
        report["kernels_compared"] += len(both)
        report["kernels_identical"] += len(same)
        report["different"] += [n + ": " + dm[k] for k in diff]
        report["only_in_a"] += [n + ": " + dm[k] for k in sorted(set(ka) - set(kb))]
        report["only_in_b"] += [n + ": " + dm[k] for k in sorted(set(kb) - set(ka))]
        print("%-40s kernels %3d identical %3d" % (n, len(both), len(same)) + ("" if not diff else "   DIFFERENT: %d" % len(diff)))
        for k in diff:
            what = [f for f in ("text", "descriptor", "metadata") if ka[k][f] != kb[k][f]]
            print("      %s  (%s)" % (dm[k][:150], ", ".join(what)))
    print("kernels compared %d, identical %d; only in A %d, only in B %d" %
          (report["kernels_compared"], report["kernels_identical"], len(report["only_in_a"]), len(report["only_in_b"])))
    if jout:
        json.dump(report, open(jout, "w"), indent=1)
    sys.exit(0 if report["kernels_compared"] == report["kernels_identical"] else 1)


if __name__ == "__main__":
    main()
