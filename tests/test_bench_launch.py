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
