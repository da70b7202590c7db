"""bench.py's report assembly executed on the CPU: main() runs end to end with the solver replaced by a fake that returns
synthetic statistics (the host factory, the sharding, the argument handling, the roofline / counters / self-check code are the
real ones).  The GPU box is the only place the real path runs and the pool is not always open: a typo in the report block must
not be found there.  Nothing here measures anything."""
import json
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeSolver:
    def __init__(self, problem, batch=1, device=0, params=None):
        d = problem.desc
        self.batch, self.T, self.nx, self.nu = batch, d.T, d.nx, d.nu
        self.jobs = 0

    def stats_na(self):
        return 10

    def stream_begin(self, x0s):
        self.jobs = len(x0s)

    def stream_run(self, maxiter):
        pass

    def solve(self, xs, us, maxiter, x0s=None):
        self.jobs = self.batch

    def stats(self):
        B, T = self.batch, self.T
        sweeps = 60 * max(self.jobs // B, 1)
        timed = sweeps // 4
        return dict(total_iters=57 * self.jobs, sweeps=sweeps, max_iters=200, timing_every=4, ms_total=sweeps * 1.7,
                    ms_linearize=0.50 * timed, ms_backward=0.55 * timed, ms_rollout=0.52 * timed, ms_select=0.06 * timed, ms_calc=0.02 * timed,
                    n_linearize=timed, n_backward=timed, n_rollout=timed, n_select=timed, n_calc=timed,
                    linearize_units=timed * B * (T + 1), backward_units=timed * B * T, rollout_units=timed * B * 10 * (T + 1),
                    linearize_units_all=sweeps * B * (T + 1), backward_units_all=sweeps * B * T, rollout_units_all=sweeps * B * 10 * (T + 1))

    def stream_results(self):
        n, T, nx, nu = self.jobs, self.T, self.nx, self.nu
        return dict(xs=np.zeros((n, T + 1, nx)), us=np.zeros((n, T, nu)), us_squash=np.zeros((n, T, nu)), cost=np.ones(n),
                    iter=np.full(n, 56, dtype=np.int32), status=np.ones(n, dtype=np.int32))


@pytest.mark.parametrize("mode", ["stream", "batch"])
def test_report_block_runs_and_is_consistent(monkeypatch, capsys, tmp_path, mode):
    import torch
    import empc_loader
    import bench
    real = empc_loader.load()
    if not os.path.exists(real.LIB_PATH):
        pytest.skip("libempc.so is not built")
    fake = types.SimpleNamespace(**{k: getattr(real, k) for k in dir(real) if not k.startswith("__")})
    fake.device_count = lambda: 1
    fake.SolverSbFDDP = FakeSolver
    monkeypatch.setattr(empc_loader, "load", lambda: fake)
    for name in ("set_device", "synchronize"):
        monkeypatch.setattr(torch.cuda, name, lambda *a, **k: None)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    # a committed counter summary taken on THIS device code is used; one taken on another is not
    import device_code_id as dci
    cid = dci.device_code_id(real.LIB_PATH)
    prof = tmp_path / "profiles"
    prof.mkdir()
    kern = {"hbm_bytes_per_launch": 1.1e9, "fp64_flop_per_launch": 15.6e9, "units_per_launch": 1024 * 100, "valu_active_frac": 0.28, "wait_frac": 0.26}
    (prof / "r05_pmc_eagle_catch.json").write_text(json.dumps({"device_code_id": cid, "commit": "abc1234", "kernels": {"backward": kern}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--mode", mode, "--no-secondary",
                                      "--no-single-batch", "--no-slots-sweep", "--no-cpu-baseline"])
    bench.main()
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["dtype"] == "f64" and out["scaling"] == "weak"
    assert out["device_code_id"] == cid and out["vs_baseline"] is None and out["higher_is_better"] is True
    r = out["roofline"]
    T = int(r["units_per_launch"] / 1024)  # knots of the workload (eagle_catch at 32 ms: 99 running nodes)
    assert T in (99, 100)
    assert r["kernel"] == "backward" and r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    # achieved = algorithmic bytes per launch / average launch time: 1024 x T units x 8 784 B / 0.55 ms
    assert abs(r["achieved"] - 1024 * T * 8784 / 0.55e-3 / 1e9) < 1e-6 * r["achieved"] and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["traffic"] == 1.1e9 and r["counters_commit"] == "abc1234" and "r05_pmc_eagle_catch.json" in r["traffic_source"]
    assert r["compute"]["source"].endswith("r05_pmc_eagle_catch.json")
    # rollout: units are (trajectory, knot) -- the ten step lengths are not charged their inputs ten times
    k = out["kernels"]["rollout"]
    assert abs(k["algorithmic_GBs"] - 1024 * (T + 1) * 2104 / 0.52e-3 / 1e9) < 1e-6 * k["algorithmic_GBs"]
    assert out["config"]["workload"] and "model" not in out["config"]
    if mode == "stream":
        assert out["stream_rows_checked"]["rows"] == 3 * 1024 == out["stream_rows_checked"]["rows_with_a_final_status"]
    # counters of another build are refused
    (prof / "r05_pmc_eagle_catch.json").write_text(json.dumps({"device_code_id": "0" * 16, "commit": "zzz", "kernels": {"backward": kern}}))
    bench.main()
    out = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert out["roofline"]["traffic"] is None and "no committed counter pass matches" in out["roofline"]["traffic_source"]
    assert "compute" not in out["roofline"] and out["roofline"]["counters_commit"] is None
