"""Host logic of the domain-decomposed path, on CPU: the Hilbert ordering against the reference's KAT lists, the
partition / halo bookkeeping, and the Comm wrapper in a 2-rank gloo run (no GPU, no compute calls)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_hilbert_positions_match_reference_kats(oracle):
    # mundy/math/tests/unit_tests/UnitTestHilbert.cpp:48-295 (s = 2, 4, 8) and the oracle's recursive generator
    from mundy_amd import distributed as D
    kat = json.load(open(os.path.join(GOLD, "hilbert_kat.json")))
    for level, name in ((1, "Cube2"), (2, "Cube4"), (3, "Cube8")):
        np.testing.assert_array_equal(D.hilbert_positions(level), np.array(kat[name]["positions"], dtype=np.int64))
    for level in (4, 5):
        np.testing.assert_array_equal(D.hilbert_positions(level), oracle.hilbert_3d(1 << level).astype(np.int64))
    t = D.hilbert_key_table(3)
    pos = D.hilbert_positions(3)
    assert sorted(t.ravel().tolist()) == list(range(512))
    np.testing.assert_array_equal(t[pos[:, 0], pos[:, 1], pos[:, 2]], np.arange(512))
    # consecutive curve points are lattice neighbours (what makes contiguous ranges compact domains)
    assert np.all(np.abs(np.diff(D.hilbert_positions(5), axis=0)).sum(axis=1) == 1)


def test_library_hilbert_table_is_the_reference_curve(oracle):
    # mhip_hilbert_key_table (host code of the library, what the C++ stepper feeds to mhip_curve_keys): the same table
    # as the numpy generator pinned above, and the inverse of the oracle's restatement of Hilbert.hpp:48-83
    import pytest
    from mundy_amd import distributed as D, ops
    for level in range(0, 6):
        table = ops.hilbert_key_table(level)
        np.testing.assert_array_equal(table, D.hilbert_key_table(level).astype(np.int32))
        if level >= 1:
            pos = oracle.hilbert_3d(1 << level).astype(np.int64)
            np.testing.assert_array_equal(table[pos[:, 0], pos[:, 1], pos[:, 2]], np.arange(len(pos)))
    with pytest.raises(ValueError, match="level"):   # MHIP_ERR_INVALID_ARGUMENT -> std::invalid_argument / ValueError
        ops.hilbert_key_table(9)


def test_hilbert_order_and_partition():
    from mundy_amd import distributed as D
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 10, (5000, 3))
    order = D.hilbert_order(c, 0.0, 10.0, level=4)
    assert sorted(order.tolist()) == list(range(5000))
    cell = np.clip(np.floor(c / 10.0 * 16).astype(int), 0, 15)
    key = D.hilbert_key_table(4)[cell[:, 0], cell[:, 1], cell[:, 2]]
    assert np.all(np.diff(key[order]) >= 0)
    same = np.diff(key[order]) == 0
    assert np.all(np.diff(order)[same] > 0)          # ties keep index order (stable)
    starts = D.partition_ranges(5003, 4)
    assert starts.tolist() == [0, 1251, 2502, 3753, 5003]
    # compactness: a quarter of the curve spans far less than the whole box in at least one direction
    q = c[order[:1250]]
    assert (q.max(axis=0) - q.min(axis=0)).min() < 7.0


def _halo_layout(count_matrix, rank):
    """independent restatement of the ghost bookkeeping: ghosts of lower ranks, owned, ghosts of higher ranks"""
    world = len(count_matrix)
    recv = [int(count_matrix[p][rank]) if p != rank else 0 for p in range(world)]
    return recv, sum(recv[:rank]), sum(recv[rank + 1:])


def _c_layout(counts, rank, n_owned):
    import ctypes as C
    from mundy_amd import capi
    lib = capi.load()
    w = len(counts)
    flat = (C.c_size_t * (w * w))(*[counts[s][d] for s in range(w) for d in range(w)])
    lo, hi, ns, nr = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int()
    sp, sr = (C.c_int * w)(), (C.c_size_t * w)()
    rp, rf, rr = (C.c_int * w)(), (C.c_size_t * w)(), (C.c_size_t * w)()
    capi.check(lib.mhip_ghost_layout_from_counts(w, rank, n_owned, flat, C.byref(lo), C.byref(hi), C.byref(ns), sp, sr,
                                                 C.byref(nr), rp, rf, rr))
    return dict(n_lo=lo.value, n_hi=hi.value, send=[(sp[k], sr[k]) for k in range(ns.value)],
                recv=[(rp[k], rf[k], rr[k]) for k in range(nr.value)])


def test_ghost_layout_bookkeeping():
    # the host arithmetic of mhip_ghost_plan (C++, no device needed) against the restatement above
    counts = [[0, 5, 0, 2], [3, 0, 7, 0], [0, 4, 0, 1], [6, 0, 8, 0]]
    lay = _c_layout(counts, 2, 100)
    assert lay["n_lo"] == 7 and lay["n_hi"] == 8
    assert lay["send"] == [(1, 4), (3, 1)]
    assert lay["recv"] == [(1, 0, 7), (3, 107, 8)]          # lower ghosts from row 0, higher ones after lo + owned
    lay = _c_layout(counts, 0, 10)
    assert (lay["n_lo"], lay["n_hi"]) == (0, 9) and lay["recv"] == [(1, 10, 3), (3, 13, 6)] and lay["send"] == [(1, 5), (3, 2)]
    rng = np.random.default_rng(5)
    for w in (1, 2, 3, 5, 8):
        cm = rng.integers(0, 4, (w, w)) * rng.integers(0, 50, (w, w))
        np.fill_diagonal(cm, 0)
        cm = cm.tolist()
        lays = [_c_layout(cm, r, 1000 + r) for r in range(w)]
        for r in range(w):
            recv, n_lo, n_hi = _halo_layout(cm, r)
            assert (lays[r]["n_lo"], lays[r]["n_hi"]) == (n_lo, n_hi)
            # rows are contiguous, in peer order, with the owned block between the lower and the higher ghosts
            row = 0
            want = []
            for p in range(w):
                if p == r:
                    row = n_lo + 1000 + r
                    continue
                if recv[p]:
                    want.append((p, row, recv[p]))
                row += recv[p]
            assert lays[r]["recv"] == want
            # what r sends to p is what p expects from r
            for p, rows in lays[r]["send"]:
                assert (r, rows) in [(q, k) for q, _, k in lays[p]["recv"]]
            assert sum(k for _, k in lays[r]["send"]) == sum(cm[r])


WORKER = r'''
import os, sys, ctypes as C, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group(backend="gloo")
from mundy_amd import capi, distributed as D
comm = D.Comm()            # host-callback transport of the C library; its message layer is what runs here (no GPU)
r, w = comm.rank, comm.world
assert w == 2 and not comm.direct
rank, world, is_rccl = C.c_int(-1), C.c_int(-1), C.c_int(-1)
capi.check(capi.load().mhip_comm_info(comm._h, C.byref(rank), C.byref(world), C.byref(is_rccl)))
assert (rank.value, world.value, is_rccl.value) == (r, 2, 0)
g = comm.host_all_gather(torch.tensor([float(r), 10.0 + r, 20.0 + r], dtype=torch.float64))
assert g.tolist() == [[0.0, 10.0, 20.0], [1.0, 11.0, 21.0]], g
send = {1 - r: torch.full((3 + r, 6), float(r), dtype=torch.float64)}
recv = {1 - r: torch.empty((4 - r, 6), dtype=torch.float64)}
comm.host_exchange(send, recv)
assert torch.all(recv[1 - r] == float(1 - r))
comm.host_exchange({}, {})
# the ghost bookkeeping both sides derive from the all-gathered count matrix agrees: what r sends is what 1-r expects
counts = comm.host_all_gather(torch.tensor([0.0, 5.0] if r == 0 else [3.0, 0.0], dtype=torch.float64)).to(torch.int64).tolist()
assert counts == [[0, 5], [3, 0]]
import ctypes as C2
flat = (C2.c_size_t * 4)(*[counts[s][d] for s in range(2) for d in range(2)])
lo, hi, ns, nr = C2.c_size_t(), C2.c_size_t(), C2.c_int(), C2.c_int()
sp, sr, rp, rf, rr = (C2.c_int * 2)(), (C2.c_size_t * 2)(), (C2.c_int * 2)(), (C2.c_size_t * 2)(), (C2.c_size_t * 2)()
capi.check(capi.load().mhip_ghost_layout_from_counts(2, r, 100, flat, C2.byref(lo), C2.byref(hi), C2.byref(ns), sp, sr,
                                                     C2.byref(nr), rp, rf, rr))
assert lo.value + hi.value == counts[1 - r][r] and (sp[0], sr[0]) == (1 - r, counts[r][1 - r])
assert (rp[0], rr[0]) == (1 - r, counts[1 - r][r]) and rf[0] == (0 if r == 1 else 100)
# argument validation of the exchange that needs no device: a rank cannot message itself
one_i, one_p, one_z = (C.c_int * 1)(r), (C.c_void_p * 1)(8), (C.c_size_t * 1)(1)
st = capi.load().mhip_comm_exchange_start(comm._h, 1, one_i, one_p, one_z, 0, None, None, None, None)
assert st == capi.ERR_INVALID_ARGUMENT and b"not another rank" in capi.load().mhip_last_error()
comm.close()
dist.destroy_process_group()
print("COMM_OK", r)
'''


def _free_port():
    import socket
    with socket.socket() as sk:   # a fixed port can still sit in TIME_WAIT from the previous run
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_comm_two_ranks_gloo(tmp_path):
    script = tmp_path / "comm_worker.py"
    script.write_text(WORKER % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert p.stdout.count("COMM_OK") == 2


NEGOTIATE_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group(backend="gloo")
from mundy_amd import capi, distributed as D
r = dist.get_rank()
cpu = torch.device("cpu")
made = []
def run(fail_id_on_rank0, fail_create_on):
    def make_id():
        if fail_id_on_rank0:
            raise RuntimeError("no unique id (librccl not loadable)")
        return bytes(range(capi.COMM_ID_BYTES %% 256)) + bytes(capi.COMM_ID_BYTES - capi.COMM_ID_BYTES %% 256)
    def create(raw):
        assert raw == bytes(range(capi.COMM_ID_BYTES %% 256)) + bytes(capi.COMM_ID_BYTES - capi.COMM_ID_BYTES %% 256)
        if r in fail_create_on:
            raise RuntimeError("create refused on rank %%d" %% r)
        made.append(1)
    return D.negotiate_direct_transport(None, r, make_id, create, cpu)
# every combination ends in the same decision on both ranks and no rank is left inside a collective (the advisor's
# case: rank 0 alone fails to make the id; before, it skipped the broadcast the other ranks were waiting in)
ok, err = run(True, ())
assert not ok and err is not None
ok, err = run(False, (1,))
assert not ok and ((err is not None) == (r == 1))
ok, err = run(False, (0,))
assert not ok and ((err is not None) == (r == 0))
ok, err = run(False, ())
assert ok and err is None and len(made) >= 1
# a sub-group keeps its rank numbering in the fallback group
g = dist.new_group(ranks=[1, 0])
ok, err = D.negotiate_direct_transport(g, dist.get_rank(g), lambda: bytes(capi.COMM_ID_BYTES), lambda raw: None, cpu)
assert ok
dist.barrier()
dist.destroy_process_group()
print("NEGOTIATE_OK", r)
'''


def test_direct_transport_negotiation_two_ranks_gloo(tmp_path):
    # the RCCL -> host fallback must keep the control flow identical on all ranks whichever rank
    # fails at whichever point
    script = tmp_path / "negotiate_worker.py"
    script.write_text(NEGOTIATE_WORKER % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert p.stdout.count("NEGOTIATE_OK") == 2
