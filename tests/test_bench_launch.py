"""bench.py starts its own workers for --gpus N (VERDICT r01 missing #3): the parent spawns N processes through
torch.distributed.run before it touches torch / HIP, and exits with their code.  No GPU here, so the workers stop at
"bench.py needs a GPU" -- which is exactly what this test looks for: the message must come from two ranks that got
RANK / WORLD_SIZE from the launcher, and the parent must report the failure."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_spawns_ranks_and_relays_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the 2-rank dry run of tools/gpu_multi_dryrun.sh")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    text = r.stdout + r.stderr
    assert text.count("bench.py needs a GPU") >= 2, text[-3000:]


def test_single_rank_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stdout + r.stderr)


def test_counter_summaries_are_used_only_for_the_device_code_they_were_taken_on(tmp_path, monkeypatch):
    """roofline.traffic / roofline.compute come from a committed PMC summary only when its `device_code_id` is the id of the
    library that is running; the committed round-4 summaries carry no id and are refused"""
    import json
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import device_code_id as dci
    lib = os.path.join(ROOT, "eagle-mpc_amd", "libempc.so")
    if not os.path.exists(lib):
        pytest.skip("libempc.so is not built")
    cid = dci.device_code_id(lib)
    assert cid and len(cid) == 16 and cid == dci.device_code_id(lib)
    k, f, note = bench.committed_counters("eagle_catch", 1024, False, cid)
    assert k == {} and f is None and "no committed counter pass matches" in note  # until a round-5 pass of THIS build is committed
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "r05_pmc_eagle_catch.json").write_text(json.dumps({"device_code_id": cid, "commit": "abc", "kernels": {"backward": {"hbm_bytes_per_launch": 1.0}}}))
    (prof / "r04_pmc_eagle_catch.json").write_text(json.dumps({"kernels": {"backward": {"hbm_bytes_per_launch": 2.0}}}))
    k, f, note = bench.committed_counters("eagle_catch", 1024, False, cid)
    assert k["backward"]["hbm_bytes_per_launch"] == 1.0 and f.endswith("r05_pmc_eagle_catch.json") and "abc" in note
    k, f, note = bench.committed_counters("eagle_catch", 1024, False, "0" * 16)
    assert k == {} and f is None
    assert bench.committed_counters("eagle_catch", 512, False, cid)[0] == {}


def test_gpu_launcher_builds_its_command_for_any_number_of_arguments():
    """`shift 3` with fewer than three arguments shifted nothing and the mode / tag landed in front of `bash` as a
    command ("check: command not found" on a scarce GPU slot).  The command line for 0, 2, 3 and 5 arguments:"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def cmd(*a):
        r = subprocess.run(["bash", os.path.join(root, "tools", "gpurun_r5.sh")] + list(a), capture_output=True, text=True,
                           env=dict(os.environ, DRY_RUN="1"))
        assert r.returncode == 0, r.stderr
        return r.stdout.strip()

    import re
    for args, mode, tag, tmo, extra in (((), "check", "r05", "2700", ""), (("tests", "r06x"), "tests", "r06x", "2700", ""),
                                        (("tests", "r06x", "900"), "tests", "r06x", "900", ""),
                                        (("variants", "r06v", "5000", "VARIANTS=r6o", "CONFIGS=eagle_catch"), "variants", "r06v", "5000",
                                         "VARIANTS=r6o CONFIGS=eagle_catch")):
        out = cmd(*args)
        m = re.fullmatch(r"gpurun --timeout (\d+) -- EMPC_COMMIT=(\S+) (.*?) ?bash tools/gpu_r5.sh (\S+) (\S+)", out)
        assert m, out
        assert m.group(1) == tmo and m.group(4) == mode and m.group(5) == tag and m.group(3).strip() == extra, out
