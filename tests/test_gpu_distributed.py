"""Domain-decomposed path on real kernels: W ranks share the one GPU of the test box (gloo staging), each owning a
Hilbert range of one rod system; the distributed solve must reproduce the single-rank solve (same neighbour list,
same LCP gradient to 20 tol, bit-identical duplicated contacts).  The whole iteration loop runs in the library
(mhip_bbpgd_solve_contact_distributed); gloo is only the message layer behind its host-callback transport.  The RCCL
transport is exercised with world_size 1 (self-addressed messages, all-gather, the full step)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, port=None, extra_env=None):
    import socket
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    for attempt in range(3):
        with socket.socket() as sk:   # a fixed port can still sit in TIME_WAIT from an earlier run
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        # (the probed port can be taken by somebody else before the launcher binds it: nothing has run yet, ask again)
        if p.returncode == 0 or "EADDRINUSE" not in p.stderr:
            break
    print(p.stdout[-5000:], p.stderr[-3000:])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = "_mixed" if (extra_env or {}).get("DIST_MIXED") == "1" else ""
    tag += "_migrate" if (extra_env or {}).get("DIST_MIGRATE") == "1" else ""
    with open(os.path.join(ROOT, "gpurun_out", "dist_world%d%s.log" % (world, tag)), "w") as f:
        f.write(p.stdout[-20000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_RESULT PASS" in p.stdout
    return p.stdout


def test_reduction_records_through_the_transport_instead_of_the_mailbox():
    # by default the ranks of a node exchange the per-iteration record through the mailbox (slots in each other's device
    # memory); this one takes the transport's all-gather, as ranks on several nodes would
    out = _run(3, extra_env={"DIST_NO_MAILBOX": "1"})
    assert out.count("MAILBOX rank") == 3 and "opened 1" not in out


def test_mailbox_opens_between_the_ranks_of_the_test_box():
    # three processes sharing the GPU: every rank maps the others' boxes (hipIpc), the trial exchanges go through, and
    # the solve that used it equals the single-rank solve like any other
    out = _run(3)
    assert out.count("MAILBOX rank") == 3 and "opened 0" not in out


def test_velocity_halo_through_the_inboxes_of_the_test_box():
    # three processes sharing the GPU: owned boundary rows written straight into the peers' IPC-mapped inboxes after
    # the body sweep, ghost rows collected from the own inbox before the boundary sweep -- no send / recv launch inside
    # the iteration.  Bit for bit the single-rank solve (the worker's checks), over a trajectory with rebuilds (every
    # ghost plan re-plans the inboxes) and list reuse (the old plan's halo again)
    out = _run(3, None, {"DIST_STEPS": "6", "DIST_BODIES": "9000"})
    assert out.count("HALO_IPC rank") == 3 and "active 0" not in out


def test_velocity_halo_through_the_transport_instead_of_the_inboxes():
    out = _run(3, None, {"DIST_NO_HALO_IPC": "1", "DIST_STEPS": "3", "DIST_BODIES": "9000"})
    assert out.count("HALO_IPC rank") == 3 and "active 1" not in out


@pytest.mark.parametrize("fault", ["sym", "asym"])
def test_a_failed_solve_leaves_the_communicator_usable(fault):
    # round-3 review: the exchange numbers of the inbox halo were advanced only by a solve that succeeded -- a rank that
    # returned early would have met its own stale inbox words in the next solve.  They are retired at ONE exit now
    # whatever the outcome, and every ghost plan re-agrees them (and the mailbox's count) across the ranks.  Two
    # back-to-back solves, the first ending in an injected error: on every rank at the same poll (the next solve reuses
    # the plan), and on one rank only (the others time out; a rebuild re-agrees).  The second solve is the single-rank
    # solve bit for bit, through inboxes + mailbox.
    out = _run(3, None, {"DIST_FAULT": fault, "DIST_BODIES": "9000"})
    assert out.count("FAULT rank") == 3
    assert out.count("HALO_IPC rank") == 3 and "active 0" not in out
    assert out.count("MAILBOX rank") == 3 and "opened 0" not in out


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_equals_single_rank(world):
    _run(world, 29610 + world)


@pytest.mark.parametrize("bodies", [2, 5])
def test_distributed_fewer_bodies_than_ranks_or_barely_more(bodies):
    # 3 ranks, 2 bodies: one rank owns nothing, holds no ghosts and still takes part in every exchange and reduction
    _run(3, None, {"DIST_BODIES": str(bodies), "DIST_PHI": "0.5"})


def test_distributed_without_any_contact():
    # a dilute system: empty neighbour lists and empty LCPs on every rank, zero iterations, collectives still matched
    _run(3, None, {"DIST_BODIES": "60", "DIST_PHI": "0.0005", "DIST_STEPS": "2"})


def test_distributed_trajectory_tracks_single_rank():
    # three more full steps (ghost plan, neighbour list, solve, Euler update on every rank) after the checked one
    _run(2, None, {"DIST_STEPS": "3", "DIST_BODIES": "8000"})


def test_distributed_trajectory_with_list_reuse():
    # the rebuild rule across ranks: ghosts refreshed through the old plan, one all-gathered decision, list / partition /
    # incidence index reused when nobody moved more than half the buffer; decisions equal the single-rank ones
    _run(3, None, {"DIST_STEPS": "8", "DIST_REUSE": "1", "DIST_BODIES": "6000", "DIST_PHI": "0.2", "DIST_BUFFER": "0.4"})


def test_three_ranks_40_steps_bodies_migrate_and_the_curve_is_recut_by_work():
    # ownership follows the bodies (a body whose lattice cell now lies in another rank's range of
    # the Hilbert curve is handed over at the rebuild), the curve is re-cut every third rebalance at equal WORK (weight
    # 1 + contacts per body), and the 40-step trajectory still is the single-rank one
    _run(3, None, {"DIST_STEPS": "40", "DIST_MIGRATE": "1", "DIST_BODIES": "6000", "DIST_PHI": "0.3",
                   "DIST_BUFFER": "0.3"})


@pytest.mark.parametrize("world,mode", [(2, 3), (3, 3), (3, 2)])
def test_cold_tier_of_the_staged_solver(world, mode):
    # the staged (multi-rank) driver keeps interior contacts -- both bodies owned by the rank -- in the cold tier: each
    # rank renumbers its own contacts, wakes them through its own bodies' drifts, and on return rebuilds the previous
    # iterate's rows for the sleepers' x_tmp / g_tmp.  Mode 3 tiers whatever the size; mode 2 also makes every rank leave
    # the tiers mid-solve.  Bit for bit the single-rank solve, and a 6-step trajectory on top
    _run(world, None, {"DIST_TIER": str(mode), "DIST_BODIES": "9000", "DIST_STEPS": "6"})


@pytest.mark.parametrize("bodies", [2, 7])
def test_migration_when_ranks_own_nothing(bodies):
    # the curve cut, the migration plan and the exchange with ranks that own no body at all (3 ranks, 2 or 7 bodies):
    # every rank still takes part in every collective, nobody is lost, the trajectory is the single-rank one
    _run(3, None, {"DIST_STEPS": "6", "DIST_MIGRATE": "1", "DIST_MIGRATE_ANY": "1", "DIST_BODIES": str(bodies),
                   "DIST_PHI": "0.5"})


def test_distributed_mixed_shapes_equals_single_rank():
    # BASELINE configs[4] as a parity case: spheres + spherocylinders + ellipsoids, Hilbert-partitioned over 2 ranks
    _run(2, 29620, {"DIST_MIXED": "1", "DIST_BODIES": "9000"})


def test_configs3_at_full_size_two_ranks_sharing_the_gpu():
    # BASELINE configs[3] at its stated size through mhip_bbpgd_solve_contact_distributed: the bench system (10^6
    # spherocylinders, seed 1234, 40 %) cut into two Hilbert ranges, two processes sharing the test GPU, the velocity halo
    # through the inboxes and the reduction records through the mailbox (both insisted on), every rank's interior
    # contacts in the cold tier (3.8 M per rank: above the size from which it is on), polled every 64 iterations as in
    # the bench.  The worker's rank 0 also runs the fused single-GPU solve: same neighbour list, 770 iterations on both,
    # multipliers bit-identical, LCP conditions at 10 tol, duplicated cross-rank contacts bit-identical.
    out = _run(2, None, {"DIST_BODIES": "1000000", "DIST_SEED": "1234", "DIST_EXACT": "1", "DIST_EXPECT_ITERS": "770",
                         "DIST_POLL": "64"})
    assert out.count("HALO_IPC rank") == 2 and "active 0" not in out
    assert out.count("MAILBOX rank") == 2 and "opened 0" not in out
    assert "multipliers bit-identical to the fused solve" in out and "world 2 contacts 7621833" in out


def _nccl_world_of_one():
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    return dist


def test_configs3_at_full_size_world_one_over_rccl():
    # the same system through the staged / distributed driver on the production transport (RCCL; one rank -- the test
    # box has one GPU and RCCL refuses two ranks on a device): ghost plan, partitioned operator, the distributed loop
    # with its reduction exchange, cold tier, polls every 64 iterations.  770 iterations, multipliers and gradient
    # bit-identical to the fused solve, complementarity at 10 tol.
    import numpy as np
    import torch
    from gpu_util import dev
    from mundy_amd import distributed as D, ops, pipeline, synth
    dist = _nccl_world_of_one()
    try:
        tol = 1e-5
        b = synth.spherocylinders(1_000_000, seed=1234)
        order = D.hilbert_order(b["center"], 0.0, b["box"], level=7)
        c, q, r, ln = (dev(b[k][order]) for k in ("center", "quat", "radius", "length"))
        cfg = ops.PGDConfig(max_iters=10000, tol=tol)
        comm = D.Comm()
        assert comm.direct and comm.transport == "rccl" and comm.world == 1
        st = D.DistributedContactStepper(c.clone(), q.clone(), r, ln, 0, comm=comm, search_buffer=0.1, cfg=cfg,
                                         poll_every=64)
        st.profile = True
        s = st.step(integrate=False)
        ref = pipeline.ContactStepper("spherocylinder", c.clone(), r, q.clone(), ln, search_buffer=0.1, cfg=cfg)
        rs = ref.step(integrate=False)
        assert s["converged"] and rs.converged and s["local_contacts"] == rs.num_contacts == 7_621_833
        assert s["num_iters"] == rs.num_iters == 770, (s["num_iters"], rs.num_iters)
        assert torch.equal(st.pairs, ref.links.pairs)
        assert torch.equal(st.lam, ref.lam) and torch.equal(st.grad, ref.grad)
        assert st.op.tier_stats()["renumberings"] >= 1           # the staged driver's cold tier was on
        x, g = st.lam, st.grad
        assert float(x.min()) >= 0 and float(g.min()) >= -10 * tol
        assert float(torch.minimum(x, g).abs().max()) <= 10 * tol
        assert st.prof["record_path"].startswith("mailbox") or st.prof["record_path"] == "all-gather"
        print("configs[3] at world 1 over RCCL: %d iterations, records through %s, %.1f us per sampled iteration "
              "(body %.1f + constraint %.1f + record %.1f)" % (
                  s["num_iters"], st.prof["record_path"],
                  1e3 * (st.prof["body_ms"] + st.prof["con_ms"] + st.prof["record_ms"]) / st.prof["iters"],
                  1e3 * st.prof["body_ms"] / st.prof["iters"], 1e3 * st.prof["con_ms"] / st.prof["iters"],
                  1e3 * st.prof["record_ms"] / st.prof["iters"]))
        st.op.close()
        comm.close()
    finally:
        dist.destroy_process_group()


def test_nccl_transport_single_rank():
    # world_size 1 over the nccl (RCCL) backend: the Comm fast path and the staged solver on the production transport
    import torch
    import torch.distributed as dist
    from gpu_util import dev
    import numpy as np
    import ctypes as C
    from mundy_amd import capi, distributed as D, ops, pipeline, synth
    if not dist.is_initialized():
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        comm = D.Comm()
        assert comm.direct and comm.world == 1
        t = torch.arange(3, dtype=torch.float64, device="cuda")
        assert torch.equal(comm.all_gather(t), t.reshape(1, 3))
        comm.self_check()
        h = torch.tensor([4.0, 5.0], dtype=torch.float64)    # host scalars ride through the GPU
        assert comm.all_gather(h).tolist() == [[4.0, 5.0]] and not comm.all_gather(h).is_cuda
        # grouped ncclSend / ncclRecv with the stream hand-over, on the one rank there is: RCCL delivers a message to
        # oneself as a local copy.  Producer and consumer are ordinary stream work right before / after the exchange.
        for n in (1, 6 * 1000, 6 * 250_000):
            src = torch.zeros(n, dtype=torch.float64, device="cuda")
            dst = torch.full((n,), -1.0, dtype=torch.float64, device="cuda")
            for rep in range(3):
                src.copy_(torch.arange(n, dtype=torch.float64, device="cuda") + rep)   # producer on the stream
                comm.exchange({0: src}, {0: dst})
                got = dst.clone()                                                       # consumer on the stream
                src.zero_()                                                             # must not overtake the send
                assert torch.equal(got, torch.arange(n, dtype=torch.float64, device="cuda") + rep)
        a, b = torch.ones(12, dtype=torch.float64, device="cuda"), torch.full((18,), 2.0, dtype=torch.float64, device="cuda")
        ra, rb = torch.empty_like(a), torch.empty_like(b)
        lib, i2, p2, z2 = capi.load(), C.c_int * 2, C.c_void_p * 2, C.c_size_t * 2
        capi.check(lib.mhip_comm_exchange_start(comm._h, 2, i2(0, 0), p2(a.data_ptr(), b.data_ptr()), z2(12, 18),
                                                2, i2(0, 0), p2(ra.data_ptr(), rb.data_ptr()), z2(12, 18), None))
        with pytest.raises(RuntimeError, match="in flight"):
            comm.all_gather(t)
        capi.check(lib.mhip_comm_exchange_finish(comm._h, None))
        torch.cuda.synchronize()
        assert torch.equal(ra, a) and torch.equal(rb, b)      # two messages between the same pair keep their order
        with pytest.raises(RuntimeError, match="no exchange in flight"):
            capi.check(lib.mhip_comm_exchange_finish(comm._h, None))
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
        # how often the host polls for convergence changes nothing but the number of idle launches after convergence;
        # sampled timing on: the profile reports only iterations that did work
        for poll in (1, 5, 64):
            st2 = D.DistributedContactStepper(dev(b["center"]), dev(b["quat"]), dev(b["radius"]), dev(b["length"]), 0,
                                              comm=comm, cfg=cfg, poll_every=poll)
            st2.profile = True
            s2 = st2.step(integrate=False)
            assert s2["num_iters"] == r.num_iters and torch.equal(st2.lam, ref.lam), poll
            # every 8th iteration of a chunk is bracketed by events (all of them when a chunk is a single iteration)
            most = r.num_iters + 1 if poll < 8 else r.num_iters // 8 + r.num_iters // poll + 2
            assert 0 < st2.prof["iters"] <= most and st2.prof["body_ms"] > 0 and st2.prof["con_ms"] > 0, st2.prof
            st2.op.close()
        # an iteration cap below convergence: not an error, converged = False, num_iters = the cap (convex.hpp:642-675)
        st3 = D.DistributedContactStepper(dev(b["center"]), dev(b["quat"]), dev(b["radius"]), dev(b["length"]), 0,
                                          comm=comm, cfg=ops.PGDConfig(max_iters=7, tol=1e-6), poll_every=3)
        s3 = st3.step(integrate=False)
        assert not s3["converged"] and s3["num_iters"] == 7
        st3.op.close()
        # the cold tier of the staged solver leaves all four solver vectors where the untiered staged solve leaves them:
        # converged, stopped at an even and at an odd iteration cap, tiers kept to the end (3) or left mid-solve (2)
        for cap in (20000, 61, 62):
            got, tiered = {}, {}
            for mode in (0, 3, 2):
                st5 = D.DistributedContactStepper(dev(b["center"]), dev(b["quat"]), dev(b["radius"]), dev(b["length"]),
                                                  0, comm=comm, cfg=ops.PGDConfig(max_iters=cap, tol=1e-6),
                                                  poll_every=8)
                st5.tiering = mode
                s5 = st5.step(integrate=False)
                got[mode] = (s5["num_iters"], st5.lam.clone(), st5.grad.clone(), st5.lam_prev.clone(),
                             st5.grad_prev.clone(), st5.vel.clone())
                tiered[mode] = st5.op.tier_stats()
                if mode and cap > 100:
                    assert tiered[mode]["renumberings"] > 0
                st5.op.close()
            if cap > 100:   # mode 2 did leave the tiers mid-solve
                assert 0 < tiered[2]["tiered_iterations"] < tiered[3]["tiered_iterations"], tiered
            for mode in (3, 2):
                assert got[mode][0] == got[0][0], (cap, mode)
                for u, v in zip(got[mode][1:], got[0][1:]):
                    assert torch.equal(u, v), (cap, mode)
        comm.close()
        # if the library's RCCL communicator cannot be made, every rank falls back -- together -- to the host-callback
        # transport over a gloo group: same step, same result
        def refuse(*a, **k):
            raise RuntimeError("RCCL refused (injected by the test)")

        real = D._rccl_create
        D._rccl_create = refuse
        try:
            comm2 = D.Comm()
        finally:
            D._rccl_create = real
        assert comm2.transport == "host" and not comm2.direct
        st4 = D.DistributedContactStepper(dev(b["center"]), dev(b["quat"]), dev(b["radius"]), dev(b["length"]), 0,
                                          comm=comm2, cfg=cfg)
        s4 = st4.step(integrate=False)
        assert s4["num_iters"] == r.num_iters and torch.equal(st4.lam, ref.lam)
        st4.op.close()
        comm2.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 63, 64, 1023, 1024, 1025, 5000, 300_001])
def test_ballot_compactions_match_numpy(n):
    # mhip_filter_pairs_owned / mhip_select_aabb_overlap: order-preserving compaction by wavefront ballots, against the
    # same selections written in numpy (the filter_view step of GenNeighborLinkers.hpp:141-183)
    import ctypes as C
    import numpy as np
    import torch
    from gpu_util import dev, host
    from mundy_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(n)
    nb = max(4, n // 3)
    pairs = np.sort(rng.integers(0, nb, (n, 2)).astype(np.int32), axis=1)
    for first, count in ((0, nb), (nb // 4, nb // 2), (nb, 0), (nb // 2, 1)):
        out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
        counted = torch.empty(n, dtype=torch.uint8, device="cuda")
        cnt = C.c_size_t(0)
        capi.check(lib.mhip_filter_pairs_owned(n, C.c_void_p(dev(pairs).data_ptr()), first, count,
                                               C.c_void_p(out.data_ptr()), C.c_void_p(counted.data_ptr()),
                                               C.byref(cnt), None))
        own = (pairs >= first) & (pairs < first + count)
        keep = own.any(axis=1)
        assert cnt.value == int(keep.sum())
        np.testing.assert_array_equal(host(out)[: cnt.value], pairs[keep])
        np.testing.assert_array_equal(host(counted)[: cnt.value].astype(bool), own[keep][:, 0])  # lower body = column 0
        # the same selection split for overlap: interior pairs (both owned) first, boundary pairs (one owned) after
        out2 = torch.empty((n, 2), dtype=torch.int32, device="cuda")
        counted2 = torch.empty(n, dtype=torch.uint8, device="cuda")
        n_int, cnt2 = C.c_size_t(0), C.c_size_t(0)
        capi.check(lib.mhip_partition_pairs_owned(n, C.c_void_p(dev(pairs).data_ptr()), first, count,
                                                  C.c_void_p(out2.data_ptr()), C.c_void_p(counted2.data_ptr()),
                                                  C.byref(n_int), C.byref(cnt2), None))
        both = own.all(axis=1)
        assert cnt2.value == cnt.value and n_int.value == int(both.sum())
        np.testing.assert_array_equal(host(out2)[: n_int.value], pairs[both])
        np.testing.assert_array_equal(host(out2)[n_int.value: cnt2.value], pairs[keep & ~both])
        np.testing.assert_array_equal(host(counted2)[: cnt2.value].astype(bool),
                                      np.concatenate([own[both][:, 0], own[keep & ~both][:, 0]]))
    lo = rng.uniform(0, 10, (n, 3))
    aabb = np.concatenate([lo, lo + rng.uniform(0.1, 1.0, (n, 3))], axis=1)
    for box, buf in ((np.array([2.0, 2, 2, 6, 6, 6]), 0.25), (np.array([-5.0, -5, -5, -4, -4, -4]), 0.0),
                     (np.array([-1.0, -1, -1, 12, 12, 12]), 0.0)):
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        cnt = C.c_size_t(0)
        capi.check(lib.mhip_select_aabb_overlap(n, C.c_void_p(dev(aabb).data_ptr()), buf, (C.c_double * 6)(*box),
                                                C.c_void_p(idx.data_ptr()), C.byref(cnt), None))
        disjoint = ((aabb[:, 3:] + buf) < box[:3]).any(axis=1) | (box[3:] < (aabb[:, :3] - buf)).any(axis=1)
        np.testing.assert_array_equal(host(idx)[: cnt.value], np.nonzero(~disjoint)[0])


@pytest.mark.parametrize("n,nchunks", [(1, 4), (50, 64), (1000, 7), (70_000, 64)])
def test_chunk_boxes_and_any_box_selection(n, nchunks):
    # mhip_aabb_chunk_bounds / mhip_select_aabb_overlap_any: a rank's region as several boxes and the ghost candidates
    # against it, versus the same written in numpy (closed interval tests, ascending order, empty chunks inverted)
    import ctypes as C
    import numpy as np
    import torch
    from gpu_util import dev, host
    from mundy_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(n + nchunks)
    # a "rank" whose bodies are in a spatially coherent order (sorted along x) and a set of query boxes around it
    lo = rng.uniform(0, 20, (n, 3))
    lo = lo[np.argsort(lo[:, 0])]
    aabb = np.concatenate([lo, lo + rng.uniform(0.1, 1.0, (n, 3))], axis=1)
    buf = 0.15
    boxes = torch.empty((nchunks + 1, 6), dtype=torch.float64, device="cuda")
    capi.check(lib.mhip_aabb_chunk_bounds(n, C.c_void_p(dev(aabb).data_ptr()), buf, nchunks,
                                          C.c_void_p(boxes.data_ptr()), None))
    per = -(-n // nchunks)
    want = np.empty((nchunks + 1, 6))
    for k in range(nchunks):
        ch = aabb[k * per:(k + 1) * per]
        want[k] = ([np.finfo(float).max] * 3 + [-np.finfo(float).max] * 3) if len(ch) == 0 else \
            np.concatenate([ch[:, :3].min(0) - buf, ch[:, 3:].max(0) + buf])
    want[nchunks] = np.concatenate([want[:nchunks, :3].min(0), want[:nchunks, 3:].max(0)])
    np.testing.assert_array_equal(host(boxes), want)
    m = 5000
    qlo = rng.uniform(-2, 22, (m, 3))
    q = np.concatenate([qlo, qlo + rng.uniform(0.05, 0.6, (m, 3))], axis=1)
    idx = torch.empty(m, dtype=torch.int32, device="cuda")
    cnt = C.c_size_t(0)
    capi.check(lib.mhip_select_aabb_overlap_any(m, C.c_void_p(dev(q).data_ptr()), buf, nchunks,
                                                C.c_void_p(boxes.data_ptr()), C.c_void_p(idx.data_ptr()), C.byref(cnt),
                                                None))
    hit = np.zeros(m, bool)
    for k in range(nchunks):
        b = want[k]
        hit |= ~(((q[:, 3:] + buf) < b[:3]).any(axis=1) | (b[3:] < (q[:, :3] - buf)).any(axis=1))
    np.testing.assert_array_equal(host(idx)[: cnt.value], np.nonzero(hit)[0])
    # conservative: whatever meets a member's grown box meets its chunk's box
    member = np.zeros(m, bool)
    for j in rng.integers(0, n, 50):
        b = np.concatenate([aabb[j, :3] - buf, aabb[j, 3:] + buf])
        member |= ~(((q[:, 3:] + buf) < b[:3]).any(axis=1) | (b[3:] < (q[:, :3] - buf)).any(axis=1))
    assert not np.any(member & ~hit)
