"""bench.py --gpus N means N (CPU): no launcher -> N ranks are started as a child torch.distributed.run; a launcher
that started another number of ranks -> a loud non-zero exit; never a silent single-GPU line."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    sys.path.insert(0, ROOT)
    import bench as b
    return b


def test_launch_plan(bench, capsys):
    argv = ["--gpus", "4", "--steps", "2"]
    assert bench.launch_plan(1, {}, ["--gpus", "1"]) is None                       # plain single-GPU run
    assert bench.launch_plan(4, {"WORLD_SIZE": "4"}, argv) is None                 # the driver's launcher line
    assert bench.launch_plan(1, {"WORLD_SIZE": "1"}, ["--gpus", "1"]) is None
    plan = bench.launch_plan(4, {}, argv, device_count=8)                          # no launcher: start one, as a child
    assert plan[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert plan[plan.index("--nproc-per-node") + 1] == "4" and plan[plan.index("--master-addr") + 1] == "127.0.0.1"
    assert plan[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and plan[-len(argv):] == argv
    for gpus, env, dc in ((4, {"WORLD_SIZE": "1"}, None), (1, {"WORLD_SIZE": "8"}, None), (8, {"WORLD_SIZE": "4"}, None),
                          (4, {}, 1), (0, {}, None)):
        with pytest.raises(SystemExit) as e:
            bench.launch_plan(gpus, env, argv, device_count=dc)
        assert e.value.code not in (0, None)
    # several ranks sharing one GPU over gloo is a development configuration the caller has to ask for
    assert bench.launch_plan(2, {"MUNDY_BENCH_BACKEND": "gloo"}, argv, device_count=1) is not None


def test_gpus_2_without_a_launcher_starts_two_ranks_or_fails_loudly():
    # in this container there is no GPU: the ranks that get started each say so and the run ends non-zero; on a box
    # with fewer GPUs than asked for the parent refuses before starting anything.  Either way: no JSON line.
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--bodies", "1000"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(env, MUNDY_BENCH_BACKEND="gloo"))
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "starting" in p.stderr and "--nproc-per-node 2" in p.stderr
    # (the elastic agent ends the other rank as soon as one exits non-zero, so only ONE of them is sure to get its
    # message out)
    assert p.stderr.count("bench.py needs a GPU") >= 1, p.stderr[-3000:]
    # a launcher that started ONE rank for --gpus 2: refused by name
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode == 2 and "--gpus 2 but the launcher started WORLD_SIZE=1" in p.stderr
    assert not p.stdout.strip()


def test_traffic_figure_is_refused_when_the_sweep_kernels_changed(bench, monkeypatch):
    # roofline.traffic comes from committed PMC passes (profiles/traffic.json); the file carries a stamp of the sweep
    # kernels it was measured on, and a running library with another stamp gets `traffic: null` and the reason instead of
    # a figure measured on other kernels (round-3 review: "silently goes stale with the next kernel edit")
    import json
    from mundy_amd import build as hip_build
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert "_kernel_stamp" in tj and tj["k_body"]["hbm_bytes_per_launch"] > 0

    def entries():
        return {"traffic": None, "traffic_source": None}, {"k_constraint": {"traffic": None, "traffic_source": None}}

    roof, extra = entries()
    monkeypatch.setattr(hip_build, "sweep_kernels_stamp", lambda: tj["_kernel_stamp"])
    bench.attach_traffic(roof, extra, "k_body", "k_constraint", 1_000_000, 0.1)
    assert roof["traffic"] == tj["k_body"]["hbm_bytes_per_launch"]
    assert extra["k_constraint"]["traffic"] == tj["k_constraint"]["hbm_bytes_per_launch"]
    roof, extra = entries()
    monkeypatch.setattr(hip_build, "sweep_kernels_stamp", lambda: "0000000000000000")
    bench.attach_traffic(roof, extra, "k_body", "k_constraint", 1_000_000, 0.1)
    assert roof["traffic"] is None and "dropped" in roof["traffic_source"]
    assert extra["k_constraint"]["traffic"] is None
    # another workload than the one the passes were taken on: no figure either way
    roof, extra = entries()
    bench.attach_traffic(roof, extra, "k_body", "k_constraint", 500_000, 0.1)
    assert roof["traffic"] is None and roof["traffic_source"] is None


def test_committed_traffic_matches_the_committed_kernels():
    # the stamp in profiles/traffic*.json must be the stamp of the sources in this tree: a kernel edit without a new
    # profile run shows up here, on the CPU, before the GPU line prints `traffic: null`
    # (a warning, not a failure: between a kernel edit and the next profile run the tree is legitimately in that state,
    # and bench.py handles it by printing `traffic: null` with the reason)
    import json
    import warnings
    from mundy_amd import build as hip_build
    for name in ("traffic.json", "traffic_mixed.json"):
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert "_kernel_stamp" in tj, name
        if tj["_kernel_stamp"] != hip_build.sweep_kernels_stamp():
            warnings.warn("profiles/%s was measured on other sweep kernels (stamp %s, tree %s): re-run "
                          "scripts/profile_bench.sh" % (name, tj["_kernel_stamp"], hip_build.sweep_kernels_stamp()))
