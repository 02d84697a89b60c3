"""bench.py's output contract on a small workload: one JSON line on stdout with the driver's keys, the roofline and
cpu_baseline objects, and sane values."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bodies", "30000", "--steps", "2", "--warmup",
                        "1", "--cpu-iters", "2"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("tier", ["default", "any-size"])
def test_default_line_has_the_contract_keys(tier):
    # (30000 rods are below the size from which the cold tier pays: the default run sweeps every contact; the second
    # run switches the tier on regardless, for the byte accounting of the tiered iterations)
    d = _run([] if tier == "default" else ["--cold-tier-any-size"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "timesteps/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["scaling"] == "strong" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-3)
    assert all(d["config"]["converged"]) and d["config"]["contacts_per_gpu"] > 100_000
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert key in r, key
    # the body sweep's bytes are those its activity masks require, with the active fraction measured in the run
    kb = r if r["kernel"].startswith("k_body") else d["k_body"]
    assert 0.0 < kb["active_contact_fraction"] < 1.0
    # (+ 72 B per body over the tiered share of the iterations: its drift, read and written, its firing threshold, and --
    # round 4, systems up to 1.75e6 bodies -- the row of the previous iterate the drift is the difference to)
    ct = d["cold_tier"]
    share = min(1.0, ct["tiered_iterations"] / d["config"]["bbpgd_iters_per_step"][-1])
    assert kb["bytes_per_launch"] == pytest.approx(
        88.0 * kb["active_contact_fraction"] * d["config"]["contacts_per_gpu"] +
        (112.0 + 72.0 * share) * d["config"]["bodies_per_gpu"], rel=1e-3)
    # the constraint sweep's: 88 B per contact swept (hot range + awake part of the tail) over the tiered iterations
    kc = r if r["kernel"].startswith("k_constraint") else d["k_constraint"]
    C, N, h = d["config"]["contacts_per_gpu"], d["config"]["bodies_per_gpu"], ct["mean_hot_fraction"]
    if tier == "default":
        assert ct["tiered_iterations"] == 0 and ct["renumberings"] == 0
    else:
        assert ct["tiered_iterations"] > 0 and 0.0 < h < 1.0 and ct["renumberings"] >= 1
    assert kc["bytes_per_launch"] == pytest.approx(
        (1 - share) * (88.0 * C + 48.0 * N) + share * (88.0 * h * C + 48.0 * N), rel=1e-3)
    # the second, labelled figure: the same step from the relaxed packing
    rp = d["relaxed_packing"]
    assert "NOT the headline" in rp["what"] and all(rp["converged"]) and rp["timesteps_per_sec"] > 0
    assert max(rp["bbpgd_iters_per_step"]) < min(d["config"]["bbpgd_iters_per_step"])
    tr = rp["trajectory"]     # consecutive steps with the rebuild rule deciding about the neighbour list
    assert len(tr["bbpgd_iters_per_step"]) == 8 and all(tr["converged"]) and 0 <= tr["rebuilds"] <= 8
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] < 1.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=1e-3)
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and 0.0 < c["value"] < d["value"]


def test_friction_extension_line_is_labelled():
    d = _run(["--friction", "0.3", "--no-cpu-baseline"])
    assert "EXTENSION" in d["config"]["workload"] and "build extension" in d["metric"]
    assert d["cpu_baseline"] is None
    # the usable configuration of the extension: the packing relaxed by two steps of the reference's frictionless path
    rp = d["relaxed_packing"]
    assert all(rp["converged"]) and max(rp["bbpgd_iters_per_step"]) < 2000


def test_mixed_line_names_configs4():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mixed", "--bodies", "9000", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert "configs[4]" in d["config"]["workload"] and "mixed" in d["metric"]
    assert d["config"]["converged"] == [True] and d["stage_ms"]["narrowphase"] > 0 and d["roofline"]["bound"] == "hbm"
    # the fp64-vector roofline of the ellipsoid classes and the CPU oracle timed on a bounded sample of the same input
    assert d["narrow_phase_roofline"]["bound"] == "fp64-vector" and 0.0 < d["narrow_phase_roofline"]["frac"] < 1.0
    assert 0.0 < d["roofline"]["frac"] < 1.0 and 0.0 < d["k_constraint" if "k_constraint" in d else "k_body"]["frac"] < 1.0
    # both arithmetics of the minimisation classes side by side; the line itself ran the default (bit-exact) one
    ar = d["narrow_phase_roofline"]["arithmetics"]
    assert d["narrow_phase_roofline"]["arithmetic"].startswith("no contraction")
    assert ar["separations_within_1e-4"] >= 0.995 and 0.0 < ar["contracted"]["frac_of_fp64_vector_peak"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and 0.0 < c["value"] < d["value"]
    assert "pair list equal to the GPU's: True" in c["sample"]


def _torchrun(extra, fail_rccl=False):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # fail_rccl: bench.py run through a wrapper that puts a refusing function in the place of the library's
    # communicator constructor (the product reads no environment variable for this)
    script = os.path.join(ROOT, "tests", "bench_rccl_refused.py") if fail_rccl else os.path.join(ROOT, "bench.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), script, "--gpus", "1", "--distributed",
           "--bodies", "30000", "--steps", "1", "--warmup", "1"] + extra
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                          env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


def test_partitioned_line_is_configs3_and_refuses_a_host_staged_halo():
    # one rank over RCCL: the partitioned path prints configs[3]'s line ("strong": the one fixed-size system) ...
    p = _torchrun([])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["scaling"] == "strong" and "configs[3]" in d["config"]["workload"] and d["config"]["transport"] == "rccl"
    assert d["config"]["bodies_total"] == 30000 and "constraint_updates_per_sec" in d
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-3)
    assert d["config"]["contact_imbalance_max_over_mean"] == 1.0
    # ... a run whose RCCL communicator could not be made falls back to the host-staged transport and prints NO line
    p = _torchrun([], fail_rccl=True)
    assert p.returncode != 0 and "refusing to print a scaling line" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # ... unless told to (development boxes), and then the line says which wire ran
    p = _torchrun(["--allow-host-transport"], fail_rccl=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["transport"] == "host"
    # the weak line keeps its own metric name
    p = _torchrun(["--weak"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["scaling"] == "weak" and "PER GPU" in d["metric"]


def test_gpus_2_without_a_launcher_prints_a_two_rank_line():
    # `python bench.py --gpus 2` with no launcher around it starts its own (a child torch.distributed.run) and relays
    # rank 0's line: n_gpus == --gpus.  Two ranks share the test box's one GPU here (gloo; the halo of the partitioned
    # path goes through the inboxes, the records through the mailbox), which is why the host transport is allowed.
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "40000", "--steps", "1",
                        "--warmup", "1", "--allow-host-transport"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(env, MUNDY_BENCH_BACKEND="gloo"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["bodies_total"] == 40000
    assert d["config"]["velocity_halo"] in ("IPC-mapped inboxes", "grouped send / recv")
    assert all(d["config"]["converged"])
    # what every rank's solves actually used in the timed steps (the library reports it) ...
    assert len(d["config"]["paths_active_per_rank"]) == 2
    assert all(p_["halo"] in ("inboxes", "send/recv") for p_ in d["config"]["paths_active_per_rank"])
    # ... and the untimed A/B of the transport mechanisms that makes an N > 1 line explain itself: four variants, each
    # with the mechanism ACTUALLY active on every rank and where an iteration's microseconds go
    ab = d["transport_ab"]
    assert "error" not in ab, ab
    assert [v["requested"] for v in ab["variants"]] == ["inboxes + mailbox", "send/recv + mailbox",
                                                        "inboxes + all-gather", "send/recv + all-gather"]
    for v in ab["variants"]:
        assert len(v["per_rank"]) == 2 and v["us_per_iteration"] > 0 and v["iterations_per_step"] > 0
        for e in v["per_rank"]:
            assert e["body_us"] > 0 and e["constraint_us"] > 0 and e["record_us"] > 0
    assert ab["variants"][1]["halo_active"] == ["send/recv"] and ab["variants"][3]["halo_active"] == ["send/recv"]
    assert ab["variants"][2]["records_active"] == ["all-gather"] and ab["variants"][3]["records_active"] == ["all-gather"]
    # (the test box grants IPC between processes: the first variant really runs inboxes + mailbox)
    assert ab["variants"][0]["halo_active"] == ["inboxes"] and ab["variants"][0]["records_active"] == ["mailbox (fused)"]
    its = {v["iterations_per_step"] for v in ab["variants"]}
    assert len(its) == 1                      # the wire does not reach the iterates


def test_mixed_system_over_two_ranks_prints_the_configs4_line():
    # BASELINE configs[4] is the mixed system ON SEVERAL GPUs: `bench.py --mixed --gpus N` partitions it like the rods.
    # Two ranks share the test box's GPU (gloo); same contacts as the single-GPU line of the same system
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    common = ["--mixed", "--bodies", "30000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--allow-host-transport"] + common,
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env, MUNDY_BENCH_BACKEND="gloo"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and "mixed" in d["metric"] and d["config"]["workload"].startswith("configs[4]")
    assert d["config"]["bodies_total"] == 30000 and all(d["config"]["converged"])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-3000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert d1["config"]["contacts_per_gpu"] == d["config"]["contacts_total"]
    # (the iteration counts are close, not equal: the single-GPU line Z-orders the bodies, the partitioned one orders
    #  them along the Hilbert curve, and an ellipsoid pair is evaluated in list orientation -- the reference's E-E
    #  minimisation parametrises the FIRST body's normal, so (i, j) and (j, i) agree to its 1e-4 only.  Same order,
    #  same bits: tests/test_gpu_distributed.py)
    it1, it2 = d1["config"]["bbpgd_iters_per_step"][-1], d["config"]["bbpgd_iters_per_step"][-1]
    assert abs(it1 - it2) <= 0.1 * it1, (it1, it2)
