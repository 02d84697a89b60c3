"""GPU parity, neighbour lists: pair sets from libmundy_hip.so are BIT-EXACT (same pairs, same (i,j)-sorted order)
against the CPU oracle's brute-force / cell-list search with the same predicate.  The reference's own search is third
party (stk::search, GenNeighborLinkers.hpp:658): parity unpinned at that boundary (see oracle header)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import torch
    assert torch.cuda.is_available()
    from mundy_amd import ops as o
    return o


def _links(ops, kind, buffer, symmetric=False, box=None):
    return (ops.GenNeighborLinks().set_search_kind(kind).set_search_buffer(buffer)
            .set_enforce_source_target_symmetry(symmetric).set_periodic_box(box).concretize())


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("symmetric", [False, True])
def test_pairs_bit_exact_vs_bruteforce(ops, oracle, kind, periodic, symmetric):
    from gpu_util import dev, host, random_rods
    rng = np.random.default_rng(100 + 10 * kind + 2 * periodic + symmetric)
    box = np.array([9.0, 11.0, 13.0])
    c, q, r, L = random_rods(rng, 3000, box)
    if periodic:
        c += rng.integers(-2, 3, c.shape) * box  # unwrapped images must not matter
    aabb = oracle.compute_aabb_spherocylinders(c, q, r, L)
    brad = oracle.bounding_radius_spherocylinders(r, L)
    lo, hi, R = oracle.grow(aabb, brad, 0.15)
    b = box if periodic else None
    exp = oracle.search(kind, lo, hi, c, R, box=b, symmetric=symmetric, method="brute")
    g = _links(ops, kind, 0.15, symmetric, b)
    assert g.generate(dev(aabb), dev(c), dev(brad)) is True
    got = host(g.pairs)
    assert len(exp) > 5000
    np.testing.assert_array_equal(got, exp)
    # CSR view is the same list
    rp, col = host(g.row_ptr), host(g.col)
    assert rp[0] == 0 and rp[-1] == len(exp) and np.all(np.diff(rp) >= 0)
    np.testing.assert_array_equal(col, exp[:, 1])
    np.testing.assert_array_equal(np.repeat(np.arange(len(c)), np.diff(rp)), exp[:, 0])
    g.close()


def test_config1_10k_spheres_periodic(ops, oracle):
    # BASELINE.json configs[0]: 10k spheres, periodic box, pairwise distance + neighbour list only
    from gpu_util import assert_bits_equal, dev, host
    from mundy_amd import synth
    s = synth.spheres(10_000)
    box = np.full(3, s["box"])
    c, r = s["center"], s["radius"]
    aabb = oracle.compute_aabb_spheres(c, r)
    for buffer in (0.0, 1.0):
        lo, hi, R = oracle.grow(aabb, r, buffer)
        for kind in (0, 1):
            exp = oracle.search(kind, lo, hi, c, R, box=box, method="cell")
            g = _links(ops, kind, buffer, False, box)
            g.generate(ops.compute_aabb_spheres(dev(c), dev(r)), dev(c), dev(r))
            np.testing.assert_array_equal(host(g.pairs), exp)
            sep, nrm = ops.contact_spheres(g.pairs, dev(c), dev(r), box=box)
            osep, onrm = oracle.contact_spheres(exp, c, r, box=box)
            assert_bits_equal(host(sep), osep, "sep")
            assert_bits_equal(host(nrm), onrm, "normal")
            if kind == 0 and buffer == 0.0:  # bounding spheres with no buffer: every pair overlaps or touches
                assert osep.max() <= 1e-12
            g.close()


def test_edge_cases(ops, oracle):
    from gpu_util import dev, host
    import torch
    z3 = torch.zeros((0, 3), dtype=torch.float64, device="cuda")
    z6 = torch.zeros((0, 6), dtype=torch.float64, device="cuda")
    z1 = torch.zeros(0, dtype=torch.float64, device="cuda")
    for kind in (0, 1):
        g = _links(ops, kind, 0.1)
        g.generate(z6, z3, z1)
        assert g.num_pairs == 0 and host(g.row_ptr).tolist() == [0]
        c = np.zeros((1, 3))
        g.generate(dev(np.array([[-1.0, -1, -1, 1, 1, 1]])), dev(c), dev(np.ones(1)), force=True)
        assert g.num_pairs == 0 and host(g.row_ptr).tolist() == [0, 0]
        g.close()
        # UnitTestGenNeighborLinks.cpp:73-152: two coincident spheres -> one link (two when symmetric)
        c2, r2 = np.zeros((2, 3)), np.ones(2)
        a2 = oracle.compute_aabb_spheres(c2, r2)
        for sym, exp in ((False, [[0, 1]]), (True, [[0, 1], [1, 0]])):
            g = _links(ops, kind, 0.0, sym)
            g.generate(dev(a2), dev(c2), dev(r2))
            assert host(g.pairs).tolist() == exp
            g.close()
        # touching volumes are a hit (closed predicate), separated ones are not
        c3 = np.array([[0.0, 0, 0], [2.0, 0, 0], [4.5, 0, 0]])
        g = _links(ops, kind, 0.0)
        g.generate(dev(oracle.compute_aabb_spheres(c3, np.ones(3))), dev(c3), dev(np.ones(3)))
        assert host(g.pairs).tolist() == [[0, 1]]
        g.close()


def test_all_bodies_in_one_cell_and_clusters(ops, oracle):
    # degenerate grids: everything coincident-ish (one cell) and two distant clusters (sparse grid)
    from gpu_util import dev, host
    rng = np.random.default_rng(3)
    for c in (rng.uniform(0, 0.5, (300, 3)),
              np.concatenate([rng.uniform(0, 3, (200, 3)), 1e4 + rng.uniform(0, 3, (200, 3))])):
        r = np.full(len(c), 0.4)
        aabb = oracle.compute_aabb_spheres(c, r)
        lo, hi, R = oracle.grow(aabb, r, 0.05)
        for kind in (0, 1):
            exp = oracle.search(kind, lo, hi, c, R, method="brute")
            g = _links(ops, kind, 0.05)
            g.generate(dev(aabb), dev(c), dev(r))
            np.testing.assert_array_equal(host(g.pairs), exp)
            g.close()


def test_crowded_cells_need_several_lds_tiles(ops, oracle):
    # one large body sets the cell edge, thousands of small ones crowd a few cells: the LDS-staged search (k_pairs_lds)
    # must walk a run of records in several 512-record tiles, and rows get long (the big body touches hundreds)
    from gpu_util import dev, host
    rng = np.random.default_rng(8)
    c = np.concatenate([rng.uniform(0, 4, (3500, 3)), [[2.0, 2.0, 2.0]], rng.uniform(30, 34, (1200, 3))])
    r = np.concatenate([np.full(3500, 0.06), [3.0], np.full(1200, 0.08)])
    aabb = oracle.compute_aabb_spheres(c, r)
    lo, hi, R = oracle.grow(aabb, r, 0.02)
    for kind in (0, 1):
        for symmetric in (False, True):
            exp = oracle.search(kind, lo, hi, c, R, symmetric=symmetric, method="brute")
            g = _links(ops, kind, 0.02, symmetric)
            g.generate(dev(aabb), dev(c), dev(r))
            assert len(exp) > 3000
            np.testing.assert_array_equal(host(g.pairs), exp)
            g.close()


@pytest.mark.parametrize("periodic", [False, True])
def test_grid_rows_beyond_the_slab_are_searched_again(ops, oracle, periodic):
    # round 4: the cell grid's counting pass parks up to 32 partners per body in a slab and the filling pass copies the
    # parked rows; a body with more partners walks the stencil again (its workgroup-mates only help staging the tiles).
    # A dense cluster (rows of 40-200 partners) beside a dilute cloud (rows of 0-5) in ONE system, the grid forced, free
    # and periodic cell (the periodic search is the per-lane kernel): the brute-force oracle's lists, pairs and order
    from gpu_util import dev, host
    rng = np.random.default_rng(19 + periodic)
    box = np.array([24.0, 24.0, 24.0])
    c = np.concatenate([rng.uniform(0, 3.0, (2500, 3)), rng.uniform(6, 24, (6000, 3))])
    r = np.full(len(c), 0.3)
    aabb = oracle.compute_aabb_spheres(c, r)
    lo, hi, R = oracle.grow(aabb, r, 0.2)
    for kind in (0, 1):
        for symmetric in (False, True):
            exp = oracle.search(kind, lo, hi, c, R, symmetric=symmetric, method="brute", box=box if periodic else None)
            rows = np.bincount(exp[:, 0], minlength=len(c))
            assert rows.max() > 64 and (rows > 32).sum() > 500 and (rows <= 32).sum() > 3000
            g = (ops.GenNeighborLinks().set_search_kind(kind).set_search_buffer(0.2)
                 .set_search_method(ops.SEARCH_METHOD_GRID).set_enforce_source_target_symmetry(symmetric))
            if periodic:
                g = g.set_periodic_box(box)
            g = g.concretize()
            g.generate(dev(aabb), dev(c), dev(r))
            assert g.method_used() == ops.SEARCH_METHOD_GRID
            np.testing.assert_array_equal(host(g.pairs), exp)
            g.close()


def test_rebuild_rule(ops):
    # GenNeighborLinkers.hpp:510-543, :603-615: generate() returns False until a centre moves > buffer/2
    from gpu_util import dev
    rng = np.random.default_rng(1)
    c = rng.uniform(0, 10, (500, 3))
    r = np.full(500, 0.5)
    g = _links(ops, 0, 1.0)
    dc, dr = dev(c), dev(r)
    aabb = ops.compute_aabb_spheres(dc, dr)
    assert g.generate(aabb, dc, dr) is True
    assert g.generate(aabb, dc, dr) is False
    dc[7, 1] += 0.5
    assert g.generate(aabb, dc, dr) is False       # exactly half the buffer: not "more than"
    dc[7, 1] += 1e-6
    assert g.generate(aabb, dc, dr) is True
    assert g.generate(aabb, dc, dr) is False       # snapshot was refreshed by the rebuild
    g.close()


def test_full_size_properties_1M_rods(ops):
    # BASELINE.json configs[2] size: properties that do not need the oracle -- sorted unique rows, i<j, symmetric
    # list = both orientations of the unique list, every listed pair passes the predicate, determinism
    import torch
    from gpu_util import dev
    from mundy_amd import synth
    b = synth.spherocylinders(1_000_000)
    c, q, r, L = dev(b["center"]), dev(b["quat"]), dev(b["radius"]), dev(b["length"])
    aabb = ops.compute_aabb_spherocylinders(c, q, r, L)
    brad = ops.bounding_radius_spherocylinders(r, L)
    g = _links(ops, 1, 0.25)
    g.generate(aabb, c, brad)
    p = g.pairs.to(torch.int64)
    assert g.num_pairs > 5_000_000
    assert bool((p[:, 0] < p[:, 1]).all())
    key = p[:, 0] * 1_000_000 + p[:, 1]
    assert bool((key[1:] > key[:-1]).all())          # strictly increasing: sorted and duplicate free
    lo, hi = aabb[:, :3] - 0.25, aabb[:, 3:] + 0.25
    ok = ((hi[p[:, 0]] >= lo[p[:, 1]]) & (hi[p[:, 1]] >= lo[p[:, 0]])).all()
    assert bool(ok)
    g2 = _links(ops, 1, 0.25, symmetric=True)
    g2.generate(aabb, c, brad)
    assert g2.num_pairs == 2 * g.num_pairs
    p2 = g2.pairs.to(torch.int64)
    fwd = p2[p2[:, 0] < p2[:, 1]]
    assert torch.equal(fwd, p)
    g3 = _links(ops, 1, 0.25)
    g3.generate(aabb, c, brad)
    assert torch.equal(g3.pairs, g.pairs)           # atomics inside, deterministic outside
    for x in (g, g2, g3):
        x.close()


def _polydisperse(rng, n, sigma=0.8):
    """spheres with log-normal radii (a few bodies tens of times larger than the median) at ~30 % volume fraction"""
    r = np.exp(rng.normal(0.0, sigma, n)) * 0.3
    L = (4.0 / 3.0 * np.pi * (r ** 3).sum() / 0.30) ** (1.0 / 3.0)
    return rng.uniform(0, L, (n, 3)), r


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("symmetric", [False, True])
def test_lbvh_lists_equal_grid_and_bruteforce(ops, oracle, kind, symmetric):
    # the Morton LBVH (the reference's search method) against the cell grid and the brute-force oracle: same pairs, same
    # order -- the structure only prunes, the predicate decides
    from gpu_util import dev, host, random_rods
    rng = np.random.default_rng(7 + kind)
    c, q, r, L = random_rods(rng, 4000, np.array([10.0, 12.0, 9.0]))
    aabb = oracle.compute_aabb_spherocylinders(c, q, r, L)
    brad = oracle.bounding_radius_spherocylinders(r, L)
    lo, hi, R = oracle.grow(aabb, brad, 0.1)
    exp = oracle.search(kind, lo, hi, c, R, symmetric=symmetric, method="brute")
    for method in (ops.SEARCH_METHOD_GRID, ops.SEARCH_METHOD_MORTON_LBVH):
        g = (ops.GenNeighborLinks().set_search_kind(kind).set_search_buffer(0.1).set_search_method(method)
             .set_enforce_source_target_symmetry(symmetric).concretize())
        g.generate(dev(aabb), dev(c), dev(brad))
        assert g.method_used() == method
        np.testing.assert_array_equal(host(g.pairs), exp)
        np.testing.assert_array_equal(host(g.col), exp[:, 1])
        g.close()


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("symmetric", [False, True])
def test_periodic_lbvh_lists_equal_grid_and_bruteforce(ops, oracle, kind, symmetric):
    # periodic cell: the tree holds the volumes translated into the primary cell and is walked once per image of the
    # query that can meet it.  Bodies given as arbitrary unwrapped images, some sitting exactly on the faces and corners
    # of the cell; same pairs, same order as the grid and the brute-force oracle (minimum-image predicate)
    from gpu_util import dev, host, random_rods
    rng = np.random.default_rng(70 + 2 * kind + symmetric)
    box = np.array([9.0, 11.0, 13.0])
    c, q, r, L = random_rods(rng, 3000, box)
    c[:64] = rng.integers(0, 2, (64, 3)) * box                 # corners of the cell
    c[64:256, 0] = 0.0                                          # a face
    c[256:300, 1] = box[1]
    c += rng.integers(-2, 3, c.shape) * box
    aabb = oracle.compute_aabb_spherocylinders(c, q, r, L)
    brad = oracle.bounding_radius_spherocylinders(r, L)
    lo, hi, R = oracle.grow(aabb, brad, 0.15)
    exp = oracle.search(kind, lo, hi, c, R, box=box, symmetric=symmetric, method="brute")
    assert len(exp) > 5000
    for method in (ops.SEARCH_METHOD_GRID, ops.SEARCH_METHOD_MORTON_LBVH):
        g = (ops.GenNeighborLinks().set_search_kind(kind).set_search_buffer(0.15).set_search_method(method)
             .set_enforce_source_target_symmetry(symmetric).set_periodic_box(box).concretize())
        g.generate(dev(aabb), dev(c), dev(brad))
        assert g.method_used() == method and g.minimum_image_complete()
        np.testing.assert_array_equal(host(g.pairs), exp)
        g.close()
    # a cell with an edge below four times the largest reach is the grid's case, whatever was asked for
    small = np.array([3.0, 11.0, 13.0])
    g = (ops.GenNeighborLinks().set_search_kind(kind).set_search_buffer(0.15)
         .set_search_method(ops.SEARCH_METHOD_MORTON_LBVH).set_periodic_box(small).concretize())
    g.generate(dev(aabb[:500]), dev(c[:500]), dev(brad[:500]))
    assert g.method_used() == ops.SEARCH_METHOD_GRID
    # ... and the handle says that the minimum-image list is not the multi-image list there (mundy_hip.h)
    assert not g.minimum_image_complete()
    np.testing.assert_array_equal(host(g.pairs), oracle.search(kind, lo[:500], hi[:500], c[:500], R[:500], box=small,
                                                                method="brute"))
    g.close()


def test_periodic_size_disperse_spheres_pick_the_lbvh(ops, oracle):
    # log-normal radii in a periodic cell: AUTO goes to the tree (largest reach > 2 x mean reach) and returns the list of
    # the grid
    from gpu_util import dev, host
    rng = np.random.default_rng(5)
    c, r = _polydisperse(rng, 20_000, sigma=0.6)
    box = np.full(3, (4.0 / 3.0 * np.pi * (r ** 3).sum() / 0.30) ** (1.0 / 3.0))
    assert 4.0 * (r.max() + 0.05) < box[0] and r.max() > 4.0 * r.mean()
    aabb = oracle.compute_aabb_spheres(c, r)
    lists = {}
    for name, method in (("grid", ops.SEARCH_METHOD_GRID), ("auto", ops.SEARCH_METHOD_AUTO)):
        g = (ops.GenNeighborLinks().set_search_kind(ops.SEARCH_SPHERES).set_search_buffer(0.05).set_search_method(method)
             .set_periodic_box(box).concretize())
        g.generate(dev(aabb), dev(c), dev(r))
        lists[name] = host(g.pairs).copy()
        if name == "auto":
            assert g.method_used() == ops.SEARCH_METHOD_MORTON_LBVH
        g.close()
    assert len(lists["grid"]) > 10_000
    np.testing.assert_array_equal(lists["auto"], lists["grid"])


@pytest.mark.parametrize("n", [1, 2, 3, 65, 1000])
def test_lbvh_small_and_degenerate_inputs(ops, oracle, n):
    # coincident centres (equal Morton keys: the position breaks the tie), a single body, touching boxes
    from gpu_util import dev, host
    rng = np.random.default_rng(n)
    c = np.repeat(rng.uniform(0, 1, ((n + 2) // 3, 3)), 3, axis=0)[:n] if n > 2 else rng.uniform(0, 1, (n, 3))
    assert c.shape == (n, 3)
    r = np.full(n, 0.25)
    aabb = oracle.compute_aabb_spheres(c, r)
    lo, hi, R = oracle.grow(aabb, r, 0.0)
    for kind in (0, 1):
        exp = oracle.search(kind, lo, hi, c, R, method="brute").reshape(-1, 2)
        g = (ops.GenNeighborLinks().set_search_kind(kind).set_search_method(ops.SEARCH_METHOD_MORTON_LBVH).concretize())
        g.generate(dev(aabb), dev(c), dev(r))
        np.testing.assert_array_equal(host(g.pairs).reshape(-1, 2), exp)
        g.close()


def test_size_disperse_system_picks_the_lbvh_and_sorts_long_rows(ops, oracle):
    # one large body must not set the cell edge for everybody.  10^5 spheres with log-normal radii:
    # AUTO picks the LBVH (largest reach > 2 x mean reach); rows of hundreds of partners go through the workgroup radix
    # sort; lists equal the grid's and the CPU oracle's; build times are printed
    import time
    import torch
    from gpu_util import dev, host
    rng = np.random.default_rng(3)
    c, r = _polydisperse(rng, 100_000)
    assert r.max() > 10 * np.median(r)
    aabb = oracle.compute_aabb_spheres(c, r)
    lo, hi, R = oracle.grow(aabb, r, 0.2)
    exp = oracle.search(oracle.SEARCH_SPHERES, lo, hi, c, R, fast=True)
    deg = np.bincount(exp[:, 0], minlength=len(r))
    assert deg.max() > 200 and (deg > 32).sum() > 50    # long rows exist: the workgroup sort is exercised
    daabb, dc, dr = dev(aabb), dev(c), dev(r)
    times = {}
    for name, method in (("auto", ops.SEARCH_METHOD_AUTO), ("grid", ops.SEARCH_METHOD_GRID),
                         ("lbvh", ops.SEARCH_METHOD_MORTON_LBVH)):
        g = (ops.GenNeighborLinks().set_search_kind(ops.SEARCH_SPHERES).set_search_buffer(0.2)
             .set_search_method(method).concretize())
        g.generate(daabb, dc, dr)                # warm-up: workspaces
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.generate(daabb, dc, dr, force=True)
        torch.cuda.synchronize()
        times[name] = 1e3 * (time.perf_counter() - t0)
        if name == "auto":
            assert g.method_used() == ops.SEARCH_METHOD_MORTON_LBVH
        np.testing.assert_array_equal(host(g.pairs), exp)
        g.close()
    print("size-disperse 1e5 spheres, %d pairs, longest row %d: build ms %s" % (len(exp), deg.max(), times))
    assert times["lbvh"] < times["grid"]
    # a monodisperse system stays on the grid
    g = ops.GenNeighborLinks().set_search_kind(ops.SEARCH_SPHERES).set_search_buffer(0.2).concretize()
    g.generate(daabb, dc, dev(np.full(len(r), 0.3)))
    assert g.method_used() == ops.SEARCH_METHOD_GRID
    g.close()


@pytest.mark.parametrize("method", [1, 2])
@pytest.mark.parametrize("symmetric", [False, True])
def test_source_target_sets_exclusions_and_identities(ops, oracle, method, symmetric):
    # seam S3: acts_on(source, target), ExcludeConnectedEntities / existing links, (entity id, owner) identities
    import torch
    from gpu_util import dev, host, random_rods
    rng = np.random.default_rng(11)
    n = 3000
    c, q, r, L = random_rods(rng, n, np.array([9.0, 9.0, 9.0]))
    aabb = oracle.compute_aabb_spherocylinders(c, q, r, L)
    brad = oracle.bounding_radius_spherocylinders(r, L)
    lo, hi, R = oracle.grow(aabb, brad, 0.1)
    full = oracle.search(1, lo, hi, c, R, symmetric=True, method="brute")          # every ordered pair, i != j
    src = (rng.random(n) < 0.6).astype(np.uint8)
    tgt = (rng.random(n) < 0.5).astype(np.uint8)
    # a chain: every body is bonded to its two index neighbours; plus 500 random already-linked pairs
    ex = [set() for _ in range(n)]
    for i in range(n):
        for j in (i - 1, i + 1):
            if 0 <= j < n:
                ex[i].add(j)
    for k in rng.choice(len(full), 500, replace=False):
        ex[int(full[k, 0])].add(int(full[k, 1]))
    ex_ptr = np.zeros(n + 1, dtype=np.int32)
    ex_ptr[1:] = np.cumsum([len(e) for e in ex])
    ex_idx = np.array([j for e in ex for j in sorted(e)], dtype=np.int32)
    keep = (src[full[:, 0]] == 1) & (tgt[full[:, 1]] == 1)
    keep &= np.array([int(j) not in ex[int(i)] for i, j in full])
    if not symmetric:
        keep &= full[:, 0] < full[:, 1]
    exp = full[keep]
    assert 1000 < len(exp) < len(full)
    ids = (rng.permutation(n).astype(np.int64) + (1 << 40))
    owner = rng.integers(0, 8, n).astype(np.int32)
    g = (ops.GenNeighborLinks().set_search_kind(1).set_search_buffer(0.1).set_search_method(method)
         .set_enforce_source_target_symmetry(symmetric).acts_on(dev(src), dev(tgt)).concretize())
    g.set_excluded_partners(dev(ex_ptr), dev(ex_idx)).set_identities(dev(ids), dev(owner))
    g.generate(dev(aabb), dev(c), dev(brad))
    np.testing.assert_array_equal(host(g.pairs), exp)
    sid, sp, tid, tp = (host(t) for t in g.ident_pairs())
    np.testing.assert_array_equal(sid, ids[exp[:, 0]])
    np.testing.assert_array_equal(tid, ids[exp[:, 1]])
    np.testing.assert_array_equal(sp, owner[exp[:, 0]])
    np.testing.assert_array_equal(tp, owner[exp[:, 1]])
    # changing the filter invalidates the list: the next generate searches again without a forced rebuild
    g.set_excluded_partners(None, None)
    assert g.generate(dev(aabb), dev(c), dev(brad)) is True
    keep2 = (src[full[:, 0]] == 1) & (tgt[full[:, 1]] == 1)
    if not symmetric:
        keep2 &= full[:, 0] < full[:, 1]
    np.testing.assert_array_equal(host(g.pairs), full[keep2])
    # ---- the list in MuNDy's link layout ------------------------------------------------------------
    P = g.num_pairs
    pairs = host(g.pairs)
    lid, linked, ranks = (host(t) for t in g.export_coo(first_link_id=1000, source_rank=3, target_rank=3))
    np.testing.assert_array_equal(lid, 1000 + np.arange(P))
    np.testing.assert_array_equal(linked, ids[pairs])
    assert np.all(ranks == 3)
    cap = 512
    num, offs, conn, begin = (host(t) for t in g.export_crs(first_link_id=1000, bucket_capacity=cap))
    want = [[] for _ in range(n)]
    for k, (i, j) in enumerate(pairs):
        want[i].append(1000 + k)
        want[j].append(1000 + k)
    np.testing.assert_array_equal(num, [len(w) for w in want])
    nb = (n + cap - 1) // cap
    assert offs.shape == (nb, cap + 1) and begin.shape == (nb + 1,) and begin[-1] == 2 * P
    for b in range(nb):
        for k in range(min(cap, n - b * cap)):
            e = b * cap + k
            seg = conn[begin[b] + offs[b, k]:begin[b] + offs[b, k + 1]]   # LinkCRSBucketConn::get_connected_links
            assert seg.tolist() == want[e], e
    g.close()


def test_self_interactions_are_a_filter(ops, oracle):
    # the reference's only test of the seam (UnitTestGenNeighborLinks.cpp:73-152, disabled there): two coincident spheres,
    # source == target == all, symmetry enforced, ExcludeSelfInteractions -> the two are linked (either orientation);
    # without that filter every sphere also meets itself
    from gpu_util import dev, host
    c = np.zeros((2, 3))
    r = np.ones(2)
    aabb = oracle.compute_aabb_spheres(c, r)
    for method in (1, 2):
        g = (ops.GenNeighborLinks().set_search_kind(0).set_search_buffer(0.0).set_search_method(method)
             .set_enforce_source_target_symmetry(True).set_exclude_self_interactions(True).concretize())
        g.generate(dev(aabb), dev(c), dev(r))
        assert host(g.pairs).tolist() == [[0, 1], [1, 0]]
        g.close()
        g = (ops.GenNeighborLinks().set_search_kind(0).set_search_method(method)
             .set_enforce_source_target_symmetry(True).set_exclude_self_interactions(False).concretize())
        g.generate(dev(aabb), dev(c), dev(r))
        assert host(g.pairs).tolist() == [[0, 0], [0, 1], [1, 0], [1, 1]]
        g.close()


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("symmetric", [False, True])
def test_triclinic_cell_lists_equal_brute_force(ops, oracle, kind, symmetric):
    # the neighbour search in a TRICLINIC periodic cell (PeriodicMetric, periodicity.hpp:233-332).  Grid and
    # Morton tree work in scaled fractional coordinates, the predicate tests the Cartesian volumes at the image
    # PeriodicMetric::sep picks; both must give the brute-force oracle's list, row for row -- bodies on faces and
    # corners of the cell and unwrapped images included.
    from gpu_util import dev, host
    rng = np.random.default_rng(21 + kind)
    n = 10_000
    # a strongly sheared cell: lattice vectors (columns) a = (30,0,0), b = (9,28,0), c = (-7,11,26)
    cell = np.array([[30.0, 9.0, -7.0], [0.0, 28.0, 11.0], [0.0, 0.0, 26.0]])
    f = rng.uniform(0, 1, (n, 3))
    f[:64] = rng.integers(0, 2, (64, 3))            # corners of the cell
    f[64:192, 0] = 0.0                              # a face
    f[192:256, 2] = 1.0
    c = f @ cell.T
    c += (rng.integers(-2, 3, (n, 3)) @ cell.T)     # unwrapped images must not matter
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r, L = rng.uniform(0.3, 0.6, n), rng.uniform(0.5, 2.5, n)
    aabb = oracle.compute_aabb_spherocylinders(c, q, r, L)
    brad = oracle.bounding_radius_spherocylinders(r, L)
    buffer = 0.15
    lo, hi, R = oracle.grow(aabb, brad, buffer)
    exp = oracle.search(kind, lo, hi, c, R, box=cell, symmetric=symmetric, method="brute")
    assert len(exp) > 3 * n
    # the orthorhombic statement on the same bodies is ANOTHER list (the shear matters)
    ortho = oracle.search(kind, lo, hi, c, R, box=np.array([30.0, 28.0, 26.0]), symmetric=symmetric, method="brute")
    assert len(ortho) != len(exp) or not np.array_equal(ortho, exp)
    for method in (ops.SEARCH_METHOD_GRID, ops.SEARCH_METHOD_MORTON_LBVH):
        g = (ops.GenNeighborLinks().set_search_kind(kind).set_search_buffer(buffer).set_search_method(method)
             .set_enforce_source_target_symmetry(symmetric).set_periodic_box(cell).concretize())
        g.generate(dev(aabb), dev(c), dev(brad))
        assert g.method_used() == method and g.minimum_image_complete()
        np.testing.assert_array_equal(host(g.pairs), exp)
        g.close()
    # a diagonal unit cell is the orthorhombic box
    diag = np.diag([30.0, 28.0, 26.0])
    g = (ops.GenNeighborLinks().set_search_kind(kind).set_search_buffer(buffer)
         .set_enforce_source_target_symmetry(symmetric).set_periodic_box(diag).concretize())
    g.generate(dev(aabb), dev(c), dev(brad))
    np.testing.assert_array_equal(host(g.pairs), ortho)
    g.close()
    # a singular cell is refused
    bad = ops.GenNeighborLinks().set_search_kind(kind).set_periodic_box(np.array([[1.0, 2, 3], [2, 4, 6], [0, 0, 1]])).concretize()
    with pytest.raises(ValueError):
        bad.generate(dev(aabb), dev(c), dev(brad))
    bad.close()
