"""Domain-decomposed path on real kernels: W ranks share the one GPU of the test box (gloo staging), each owning a
Hilbert range of one rod system; the distributed solve must reproduce the single-rank solve (same neighbour list,
same LCP gradient to 20 tol, bit-identical duplicated contacts).  The nccl transport differs only inside
mundy_amd.distributed.Comm (exercised with world_size 1 here)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, port=None, extra_env=None):
    import socket
    with socket.socket() as sk:   # a fixed port can still sit in TIME_WAIT from an earlier run
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    print(p.stdout[-5000:], p.stderr[-3000:])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = "_mixed" if (extra_env or {}).get("DIST_MIXED") == "1" else ""
    with open(os.path.join(ROOT, "gpurun_out", "dist_world%d%s.log" % (world, tag)), "w") as f:
        f.write(p.stdout[-20000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_RESULT PASS" in p.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_equals_single_rank(world):
    _run(world, 29610 + world)


def test_distributed_mixed_shapes_equals_single_rank():
    # BASELINE configs[4] as a parity case: spheres + spherocylinders + ellipsoids, Hilbert-partitioned over 2 ranks
    _run(2, 29620, {"DIST_MIXED": "1", "DIST_BODIES": "9000"})


def test_nccl_transport_single_rank():
    # world_size 1 over the nccl (RCCL) backend: the Comm fast path and the staged solver on the production transport
    import torch
    import torch.distributed as dist
    from gpu_util import dev
    import numpy as np
    from mundy_amd import distributed as D, ops, pipeline, synth
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        comm = D.Comm()
        assert comm.direct and comm.world == 1
        t = torch.arange(3, dtype=torch.float64, device="cuda")
        assert torch.equal(comm.all_gather(t), t.reshape(1, 3))
        h = torch.tensor([4, 5], dtype=torch.int64)          # host scalars ride through the GPU with nccl
        assert comm.all_gather(h).tolist() == [[4, 5]] and not comm.all_gather(h).is_cuda
        b = synth.spherocylinders(8000, seed=3)
        cfg = ops.PGDConfig(max_iters=20000, tol=1e-6)
        st = D.DistributedContactStepper(dev(b["center"]), dev(b["quat"]), dev(b["radius"]), dev(b["length"]), 0,
                                         comm=comm, cfg=cfg)
        s = st.step(integrate=False)
        ref = pipeline.ContactStepper("spherocylinder", dev(b["center"]), dev(b["radius"]), dev(b["quat"]),
                                      dev(b["length"]), search_buffer=0.1, cfg=cfg)
        r = ref.step(integrate=False)
        assert s["converged"] and r.converged and s["local_contacts"] == r.num_contacts
        assert torch.equal(st.pairs, ref.links.pairs)
        assert s["num_iters"] == r.num_iters            # one rank: identical arithmetic to the fused driver
        assert torch.equal(st.lam, ref.lam)
    finally:
        dist.destroy_process_group()
