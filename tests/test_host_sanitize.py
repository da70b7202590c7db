"""AddressSanitizer / UBSan / LeakSanitizer over the host-side C++ (parsers, factories, MPC controller) on the CPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_sanitizer_clean(tmp_path):
    host = os.path.join(ROOT, "eagle-mpc_amd", "host")
    exe = str(tmp_path / "host_sanitize")
    srcs = [os.path.join(ROOT, "tests", "csrc", "host_sanitize.cpp")] + [os.path.join(host, f) for f in
            ("yaml_lite.cpp", "params.cpp", "robot_model.cpp", "trajectory.cpp", "sbfddp.cpp", "mpc.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert "mpc knots 30 T 29" in r.stdout and r.stdout.count("rejected") == 2
