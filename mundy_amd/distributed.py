"""Domain-decomposed contact step over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI), Hilbert-curve partition of the bodies, ghost-body halo at every neighbour-list
rebuild and a ghost-velocity halo + one 3-double all-gather per BBPGD iteration.

What the reference does at the same places (all through STK/MPI): RCB partition `stk::balance::balanceStkMesh`
(scrap/lcp_spheres/NGPSpheresLCP.cpp:956), `stk::search::coarse_search(..., comm, ...)` + `change_ghosting`
(mundy_mesh/GenNeighborLinkers.hpp:658, :687-711), per-iteration `stk::all_reduce_max` + 3 x `stk::all_reduce_sum`
(NGPSpheresLCP.cpp:371, :450-452) and a ghost field refresh it leaves as a TODO (:1057).

Scheme (the reference's symmetric-pair trick, NGPSpheresLCP.cpp:455-499, made exact):
  * bodies are sorted along a Hilbert curve (same visiting order as mundy_math/Hilbert.hpp:48-83) and cut into equal
    contiguous ranges; global id = position in that order, so every rank's local index order (ghosts from lower ranks,
    owned, ghosts from higher ranks) is also global-id order and local pairs (i < j) keep their global orientation;
  * a contact between bodies of two ranks is kept on BOTH ranks; each rank sums forces only for the bodies it owns
    (`mhip_contact_op_set_partition`), so no force reduction is needed; the duplicated contact carries bit-identical
    (x, g) on both ranks because its inputs (geometry, both body velocities, global step size) are identical;
  * per iteration: body sweep -> send owned boundary velocities to the ranks that hold them as ghosts -> constraint
    sweep -> all-gather of (max, num, den) -> every rank reduces the triples in rank order (same step everywhere).
    A duplicated contact is counted once in the reductions, by the owner of its lower body.
torch.distributed carries the messages; with the gloo backend (tests, several ranks sharing one GPU) buffers are
staged through the host.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import capi, ops


# ---- Hilbert ordering -------------------------------------------------------------------------------------------------
def hilbert_positions(level):
    """Lattice points of a (2^level)^3 cube in the visiting order of mundy::math::hilbert_3d
    (mundy_math/Hilbert.hpp:48-83) started from the origin with axes (x, y, z) -- a level-by-level, vectorised form of
    that recursion (each state expands into its 8 children in curve order)."""
    cur = np.zeros((1, 3), dtype=np.int64)
    d = np.eye(3, dtype=np.int64)[None, :, :]  # [M, 3 axes (dr1, dr2, dr3), xyz]
    s = 1 << level
    while s > 1:
        h = s // 2
        neg = (d < 0).astype(np.int64)                       # stencil: 1 where an axis component is negative
        cn = cur - h * (neg * d).sum(axis=1)                 # current_position_new
        d1, d2, d3 = d[:, 0], d[:, 1], d[:, 2]
        kids = [
            (cn, d2, d3, d1),
            (cn + h * d1, d3, d1, d2),
            (cn + h * (d1 + d2), d3, d1, d2),
            (cn + h * d2, -d1, -d2, d3),
            (cn + h * (d2 + d3), -d1, -d2, d3),
            (cn + h * (d1 + d2 + d3), -d3, d1, -d2),
            (cn + h * (d1 + d3), -d3, d1, -d2),
            (cn + h * d3, d2, -d3, -d1),
        ]
        cur = np.stack([k[0] for k in kids], axis=1).reshape(-1, 3)
        d = np.stack([np.stack(k[1:], axis=1) for k in kids], axis=1).reshape(-1, 3, 3)
        s = h
    return cur


_KEY_TABLES = {}


def hilbert_key_table(level):
    """key[ix, iy, iz] = index of the lattice point along the curve"""
    if level not in _KEY_TABLES:
        pos = hilbert_positions(level)
        n = 1 << level
        table = np.empty((n, n, n), dtype=np.int64)
        table[pos[:, 0], pos[:, 1], pos[:, 2]] = np.arange(len(pos))
        _KEY_TABLES[level] = table
    return _KEY_TABLES[level]


def hilbert_order(center, lo, hi, level=6):
    """permutation that sorts points along the Hilbert curve of a (2^level)^3 lattice over [lo, hi]; ties (same
    cell) keep index order.  Host-side set-up step (the reference repartitions on the host too)."""
    center = np.asarray(center)
    n = 1 << level
    span = np.maximum(np.asarray(hi, dtype=np.float64) - np.asarray(lo, dtype=np.float64), 1e-300)
    cell = np.clip(np.floor((center - lo) / span * n).astype(np.int64), 0, n - 1)
    key = hilbert_key_table(level)[cell[:, 0], cell[:, 1], cell[:, 2]]
    return np.argsort(key, kind="stable")


def partition_ranges(n_total, world):
    """equal contiguous ranges of the curve order: rank r owns global ids [start[r], start[r+1])"""
    base, rem = divmod(n_total, world)
    counts = np.array([base + (1 if r < rem else 0) for r in range(world)], dtype=np.int64)
    return np.concatenate([[0], np.cumsum(counts)])


# ---- communication ------------------------------------------------------------------------------------------------------
def negotiate_direct_transport(group, rank, make_id, create, device):
    """Decides, identically on every rank of `group`, whether the direct (RCCL) communicator exists everywhere.

    rank 0 calls make_id() -> COMM_ID_BYTES bytes; every rank then calls create(id_bytes).  Either may raise on any
    subset of the ranks.  The sequence of collectives is the same on every rank whatever fails where: ONE broadcast of
    (id, status byte) from rank 0 -- a failed make_id travels as status 0 instead of skipping the broadcast, which
    would leave the other ranks inside it -- then ONE all-reduce of the per-rank outcome.  Returns (ok, error seen on
    this rank or None).  `device`: where the two small tensors live (cuda for nccl groups, cpu for gloo)."""
    n = capi.COMM_ID_BYTES
    buf = torch.zeros(n + 1, dtype=torch.uint8)
    err = None
    if rank == 0:
        try:
            raw = bytes(make_id())
            if len(raw) != n:
                raise RuntimeError("unique id has %d bytes, expected %d" % (len(raw), n))
            buf[:n] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            buf[n] = 1
        except Exception as e:  # noqa: BLE001
            err = e
    t = buf.to(device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    t = t.cpu()
    bad = 0.0
    if int(t[n]) == 1:
        try:
            create(t[:n].numpy().tobytes())
        except Exception as e:  # noqa: BLE001
            err, bad = e, 1.0
    else:
        bad = 1.0
        err = err or RuntimeError("rank 0 could not make the unique id")
    flag = torch.tensor([bad], dtype=torch.float32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return flag.item() == 0.0, err


# the two calls that make the library's RCCL communicator; module-level so that the tests can put a failing one in
# their place (monkeypatch) -- the product reads no environment variable to decide anything
def _rccl_unique_id(lib):
    ident = C.create_string_buffer(capi.COMM_ID_BYTES)
    capi.check(lib.mhip_comm_unique_id(ident))
    return ident.raw


def _rccl_create(lib, handle, raw, rank, world):
    ident = C.create_string_buffer(raw, capi.COMM_ID_BYTES)
    capi.check(lib.mhip_comm_create_rccl(C.byref(handle), ident, rank, world))


class Comm:
    """One rank's communicator of the C library (mhip_comm_*, csrc/dist.hip).  torch.distributed is only the launcher:
    it carries the RCCL unique id from rank 0 to the others, and, when the process group is not nccl (gloo in the
    tests, where several ranks share one GPU and RCCL refuses duplicate devices), it is the message layer behind the
    library's host-callback transport.  All halo traffic and reductions of the step go through the library."""

    def __init__(self, group=None, mailbox=True, halo_ipc=True):
        lib = capi.load()
        self.group = group
        self.mailbox = False
        self.enabled = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.direct = self.enabled and dist.get_backend(group) == "nccl"
        self._h = C.c_void_p()
        self._callbacks = None
        self.transport = "host"
        if self.direct:
            # the library's own RCCL communicator.  Should creating it fail on ANY rank (decided together, so that all
            # ranks take the same road) the step still runs, staged through host memory over a gloo group -- slow, and
            # said so loudly; it is not a second compute path, only a second wire.  bench.py refuses to time it.
            def make_id():
                return _rccl_unique_id(lib)

            def create(raw):
                _rccl_create(lib, self._h, raw, self.rank, self.world)

            ok, err = negotiate_direct_transport(group, self.rank, make_id, create, torch.device("cuda"))
            if not ok:
                import sys
                print("mundy_amd: the RCCL communicator could not be created on every rank (%s); falling back to the "
                      "host-callback transport over gloo -- expect a host-bound iteration" % (err,), file=sys.stderr)
                if self._h:
                    lib.mhip_comm_destroy(self._h)
                    self._h = C.c_void_p()
                self.direct = False
                # the same ranks, in the same order, as the group the caller gave (its rank numbering is kept)
                if group is None:
                    self.group = dist.new_group(backend="gloo")
                else:
                    self.group = dist.new_group(ranks=dist.get_process_group_ranks(group), backend="gloo",
                                                use_local_synchronization=True)
            else:
                self.transport = "rccl"
        if not self.direct:
            self._callbacks = (capi.EXCHANGE_FN(self._exchange_cb), capi.ALL_GATHER_FN(self._all_gather_cb))
            capi.check(lib.mhip_comm_create_host(C.byref(self._h), self.rank, self.world, self._callbacks[0],
                                                 self._callbacks[1], None))
        self.halo_ipc = False
        one_node = (mailbox or halo_ipc) and torch.cuda.is_available() and self._ranks_share_a_node()
        if mailbox and one_node:
            self.mailbox = self._open_mailbox()
        if halo_ipc and one_node and self.world > 1:
            # the velocity halo of the iteration through IPC-mapped inboxes instead of send / recv; the inboxes are
            # opened by the next ghost plan (collective), and anything short of success on every rank keeps send / recv
            capi.check(lib.mhip_comm_halo_ipc_enable(self._h, 1))
            self.halo_ipc = True

    def set_exchange_timeout(self, seconds):
        """bound of every wait on a peer's words (mailbox records, inbox rows); default 20 s"""
        capi.check(capi.load().mhip_comm_set_exchange_timeout(self._h, float(seconds)))

    def inject_fault(self, at_poll):
        """TEST HOOK: this rank's next distributed solve fails at its at_poll-th convergence poll (one shot)"""
        capi.check(capi.load().mhip_comm_inject_fault(self._h, int(at_poll)))

    def set_mailbox(self, on):
        """Opens / closes the mailbox of the reduction records (collective: every rank makes the same call).  Returns
        whether it is open afterwards."""
        if on and not self.mailbox:
            self.mailbox = self._open_mailbox()
        elif not on and self.mailbox:
            capi.check(capi.load().mhip_comm_mailbox_close(self._h))
            self.mailbox = False
        return self.mailbox

    def set_halo_ipc(self, on):
        """The velocity halo through the inboxes (on) or through the transport's send / recv (off), from the next
        ghost plan on (collective: every rank makes the same call)."""
        capi.check(capi.load().mhip_comm_halo_ipc_enable(self._h, 1 if on else 0))
        self.halo_ipc = bool(on)

    def halo_ipc_active(self):
        """True when the current ghost plan's velocity halo travels through the inboxes (known after a ghost plan)"""
        a = C.c_int(0)
        capi.check(capi.load().mhip_comm_halo_ipc_active(self._h, C.byref(a)))
        return bool(a.value)

    def _ranks_share_a_node(self):
        import socket
        import zlib
        if self.enabled and self.world > 1:
            dev = torch.device("cuda") if self.direct else torch.device("cpu")
            host = torch.tensor([zlib.crc32(socket.gethostname().encode())], dtype=torch.int64, device=dev)
            hosts = [torch.zeros_like(host) for _ in range(self.world)]
            dist.all_gather(hosts, host, group=self.group)
            return all(int(h) == int(host) for h in hosts)
        return True

    def _open_mailbox(self):
        """The per-iteration reduction record through slots in the ranks' device memory instead of a collective launch
        (mhip_comm_mailbox_open): ranks of one node only."""
        opened = C.c_int(0)
        capi.check(capi.load().mhip_comm_mailbox_open(self._h, C.byref(opened), _stream()))
        return bool(opened.value)

    def close(self):
        if self._h:
            capi.check(capi.load().mhip_comm_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- message layer of the host-callback transport: host tensors through the process group -----------------------
    def host_exchange(self, send, recv):
        """send / recv: {peer: contiguous float64 HOST tensor}; one message per (peer, direction); recv filled in place"""
        works = [dist.irecv(t, src=self._global(p), group=self.group) for p, t in sorted(recv.items()) if t.numel()]
        works += [dist.isend(t, dst=self._global(p), group=self.group) for p, t in sorted(send.items()) if t.numel()]
        for w in works:
            w.wait()

    def host_all_gather(self, t):
        """t: 1-D float64 HOST tensor -> [world, len] host tensor"""
        if self.world == 1:
            return t.reshape(1, -1).clone()
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t.contiguous(), group=self.group)
        return torch.stack(parts)

    # -- the callbacks the library calls (after synchronising the stream): device pointers in, staged through the host
    def _exchange_cb(self, _user, nsend, send_peer, send_buf, send_count, nrecv, recv_peer, recv_buf, recv_count):
        try:
            lib = capi.load()
            send, recv, where = {}, {}, {}
            for k in range(nrecv):
                if recv_count[k]:
                    recv[recv_peer[k]] = torch.empty(recv_count[k], dtype=torch.float64)
                    where[recv_peer[k]] = recv_buf[k]
            for k in range(nsend):
                if send_count[k]:
                    b = send[send_peer[k]] = torch.empty(send_count[k], dtype=torch.float64)
                    capi.check(lib.mhip_memcpy_d2h(b.data_ptr(), send_buf[k], 8 * b.numel(), None))
            self.host_exchange(send, recv)
            for p, b in recv.items():
                capi.check(lib.mhip_memcpy_h2d(where[p], b.data_ptr(), 8 * b.numel(), None))
            return 0
        except Exception:  # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1

    def _all_gather_cb(self, _user, send, count, recv):
        try:
            lib = capi.load()
            src = torch.empty(count, dtype=torch.float64)
            capi.check(lib.mhip_memcpy_d2h(src.data_ptr(), send, 8 * count, None))
            out = self.host_all_gather(src).contiguous()
            capi.check(lib.mhip_memcpy_h2d(recv, out.data_ptr(), 8 * out.numel(), None))
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _global(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    # -- what the stepper calls ----------------------------------------------------------------------------------------
    def all_gather(self, t):
        """t: 1-D float64 tensor (device or host) -> [world, len] tensor on the same device"""
        src = (t if t.is_cuda else t.cuda()).to(torch.float64).contiguous()
        out = torch.empty((self.world, src.numel()), dtype=torch.float64, device=src.device)
        capi.check(capi.load().mhip_comm_all_gather(self._h, _p(src), src.numel(), _p(out), _stream()))
        return out if t.is_cuda else out.cpu()

    def self_check(self):
        """Every rank messages every other rank a pattern only that pair knows and all-gathers a rank-stamped triple;
        raises on any rank that sees something else.  bench.py runs it once before the timed region, so a transport
        that misroutes fails loudly instead of timing wrong physics."""
        dev, r, w = torch.device("cuda", torch.cuda.current_device()), self.rank, self.world
        pattern = lambda src, dst, n: torch.arange(n, dtype=torch.float64, device=dev) + 1000.0 * src + 7.0 * dst  # noqa: E731
        size = lambda src, dst: 6 * (1 + (3 * src + 5 * dst) % 11)  # noqa: E731
        send = {p: pattern(r, p, size(r, p)) for p in range(w) if p != r}
        recv = {p: torch.full((size(p, r),), -1.0, dtype=torch.float64, device=dev) for p in range(w) if p != r}
        self.exchange(send, recv)
        for p, t in recv.items():
            if not torch.equal(t, pattern(p, r, size(p, r))):
                raise RuntimeError("rank %d: the message from rank %d arrived damaged" % (r, p))
        g = self.all_gather(torch.tensor([r, 2.0 * r, -1.0 * r], dtype=torch.float64, device=dev))
        want = torch.tensor([[q, 2.0 * q, -1.0 * q] for q in range(w)], dtype=torch.float64, device=dev)
        if not torch.equal(g, want):
            raise RuntimeError("rank %d: all_gather returned %s" % (r, g.tolist()))

    def exchange(self, send, recv):
        """send / recv: {peer: contiguous float64 device tensor}; recv tensors are filled in place."""
        lib = capi.load()
        sp = [p for p, t in sorted(send.items()) if t.numel()]
        rp = [p for p, t in sorted(recv.items()) if t.numel()]
        if not sp and not rp:
            return
        for p in sp:
            assert send[p].is_contiguous() and send[p].dtype == torch.float64
        for p in rp:
            assert recv[p].is_contiguous() and recv[p].dtype == torch.float64
        arr_i, arr_p, arr_z = C.c_int * len(sp), C.c_void_p * len(sp), C.c_size_t * len(sp)
        brr_i, brr_p, brr_z = C.c_int * len(rp), C.c_void_p * len(rp), C.c_size_t * len(rp)
        capi.check(lib.mhip_comm_exchange_start(
            self._h, len(sp), arr_i(*sp), arr_p(*[send[p].data_ptr() for p in sp]), arr_z(*[send[p].numel() for p in sp]),
            len(rp), brr_i(*rp), brr_p(*[recv[p].data_ptr() for p in rp]), brr_z(*[recv[p].numel() for p in rp]),
            _stream()))
        capi.check(lib.mhip_comm_exchange_finish(self._h, _stream()))


# ---- the distributed stepper ----------------------------------------------------------------------------------------------
def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DistributedContactStepper:
    """This rank's slice of a system of spherocylinders -- or of mixed spheres / spherocylinders / ellipsoids
    (BASELINE configs[4]) -- owned bodies in global-id order (a contiguous range of the Hilbert order), plus per-step
    ghosts.  step() = AABB -> ghost halo -> local neighbour list (ghost-ghost pairs dropped) -> contacts -> staged
    BBPGD with a velocity halo per iteration -> Euler update of the owned bodies.

    Rod systems use the rod-compressed operator; a mixed system (kind / shape given) bins its contacts by shape class
    (mhip_contact_mixed) and uses the vector-arm operator."""

    # gid, centre 3, quat 4, shape 3 (r,L,- for rods), kind, translational and rotational mobility, persistent entity id
    RECORD = 15

    def __init__(self, center, quat, radius, length, gid_first, *, comm=None, dt=5e-3, viscosity=1e-3,
                 search_buffer=0.1, cfg=None, poll_every=64, kind=None, shape=None, entity_id=None, domain=None,
                 curve_level=5, recut_every=4):
        """entity_id [n] (float64 or int64): ids that stay with a body when it changes owner (default: the initial
        global ids).  domain = (lo, hi): the fixed box whose (2^curve_level)^3 Hilbert lattice decides ownership when
        bodies migrate (rebalance / step(migrate=True)); recut_every: re-cut the curve by work every that many
        rebalances, 0 = keep the cuts."""
        from . import synth
        self.comm = comm or Comm()
        self.mixed = kind is not None
        if self.mixed:
            if shape is None:
                raise ValueError("a mixed system needs shape [n, 3] next to kind [n]")
            self.kind, self.shape = kind.to(torch.int32).contiguous(), shape.contiguous()
        else:
            self.kind = None
            self.shape = torch.stack([radius, length, torch.zeros_like(radius)], dim=1).contiguous()
        self.center, self.quat = center, quat
        self.n = center.shape[0]
        # dry local-drag mobilities of the owned bodies, once (they travel with the ghost records)
        brad0 = self._aabb(center, quat, self.shape, self.kind)[1] if self.n else torch.zeros(0, dtype=torch.float64)
        mt0, mr0 = synth.dry_mobility(brad0.cpu().numpy(), viscosity=float(viscosity))
        self.mob_t = torch.from_numpy(mt0).to(center.device)
        self.mob_r = torch.from_numpy(mr0).to(center.device)
        self.gid_first = int(gid_first)
        self.entity_id = (torch.arange(self.gid_first, self.gid_first + self.n, dtype=torch.float64, device=center.device)
                          if entity_id is None else entity_id.to(torch.float64).contiguous())
        self.domain = None if domain is None else ([float(v) for v in np.broadcast_to(domain[0], 3)],
                                                   [float(v) for v in np.broadcast_to(domain[1], 3)])
        if not 1 <= int(curve_level) <= 7:
            raise ValueError("curve_level must be in [1, 7]")
        self.curve_level, self.recut_every = int(curve_level), int(recut_every)
        self.splitters = None            # [world - 1] last lattice cell (Hilbert position) of every rank but the last
        self._rebalances = 0
        self._body_weight = None
        self.tiering = None              # ContactOperator.set_tiering mode (None: the library default): time only
        self.dt, self.viscosity, self.buffer = float(dt), float(viscosity), float(search_buffer)
        self.cfg = cfg or ops.PGDConfig(max_iters=10000, tol=1e-5)
        self.poll_every = int(poll_every)
        self.links = ops.GenNeighborLinks().set_search_buffer(search_buffer).set_search_kind(ops.SEARCH_AABB).concretize()
        self._synth = synth
        self.op = None
        self.stats = {}
        self.profile = False          # per-stage torch.cuda.Event timing (same stream as the kernels)
        self.prof = dict(body_ms=0.0, con_ms=0.0, iters=0)

    def _aabb(self, center, quat, shape, kind):
        """(aabb, bounding radius) of rods or of a mixed set"""
        if self.mixed:
            return ops.compute_aabb_mixed(kind, center, quat, shape)
        r, ln = shape[:, 0].contiguous(), shape[:, 1].contiguous()
        return ops.compute_aabb_spherocylinders(center, quat, r, ln), ops.bounding_radius_spherocylinders(r, ln)

    def _records(self):
        """one RECORD-wide row per owned body: what travels as a ghost and what travels when a body changes owner"""
        n, dev = self.n, self.center.device
        gid = torch.arange(self.gid_first, self.gid_first + n, dtype=torch.float64, device=dev)
        kcol = self.kind.to(torch.float64)[:, None] if self.mixed else torch.ones((n, 1), dtype=torch.float64, device=dev)
        return torch.cat([gid[:, None], self.center, self.quat, self.shape, kcol, self.mob_t[:, None],
                          self.mob_r[:, None], self.entity_id[:, None]], dim=1).contiguous()

    # -- ownership: work-weighted cuts of the Hilbert curve, bodies migrate to the rank whose range their cell is in ----
    def _cell_keys(self, center):
        lo, hi = self.domain
        if getattr(self, "_key_table", None) is None:
            self._key_table = torch.from_numpy(hilbert_key_table(self.curve_level).astype(np.int32)).to(center.device)
        return ops.curve_keys(center, lo, hi, self.curve_level, self._key_table).long()

    def _cell_keys32(self, center):
        """the same keys as the library takes them (uint32 bit patterns in an int32 tensor)"""
        return self._cell_keys(center).to(torch.int32).contiguous()

    def rebalance(self, recut=True, weights=None):
        """Moves every owned body to the rank that owns its lattice cell (replaces the RCB repartition of
        stk::balance::balanceStkMesh, scrap/lcp_spheres/NGPSpheresLCP.cpp:956, called every load_balance_frequency steps
        in Bacteria.cpp:1076-1078).  recut=True first re-cuts the curve so that every rank gets the same WORK: weight per
        body = 1 + its contacts in the last step (or `weights`), histogrammed over the lattice cells, all-gathered, cut
        at equal cumulative weight -- every rank derives the same cuts.  One grouped send / recv of body records per
        peer, through the communicator the ghost halo uses.  Returns a dict of counts."""
        if self.domain is None:
            raise RuntimeError("rebalance() needs the domain=(lo, hi) the stepper was given at construction")
        lib, comm, dev, world, rank = capi.load(), self.comm, self.center.device, self.comm.world, self.comm.rank
        keys = self._cell_keys32(self.center) if self.n else torch.zeros(0, dtype=torch.int32, device=dev)
        if recut or self.splitters is None:
            w = weights if weights is not None else self._body_weight
            if w is not None and w.shape[0] != self.n:
                w = None
            if w is not None:
                w = w.to(torch.float64).contiguous()
            splitters = np.zeros(max(world - 1, 1), dtype=np.int64)
            capi.check(lib.mhip_curve_cut(comm._h, self.n, _p(keys), _p(w) if w is not None else None,
                                          8 ** self.curve_level, splitters.ctypes.data, _stream()))
            self.splitters = splitters[:world - 1].copy()
        # rank r owns the cells  splitters[r - 1] < key <= splitters[r]
        rec = self._records()
        n_new, sent, received = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        spl = np.ascontiguousarray(self.splitters if world > 1 else np.zeros(1, dtype=np.int64), dtype=np.int64)
        capi.check(lib.mhip_migrate_plan(comm._h, self.n, _p(keys), spl.ctypes.data, C.byref(n_new), C.byref(sent),
                                         C.byref(received), _stream()))
        new = torch.empty((n_new.value, self.RECORD), dtype=torch.float64, device=dev)
        capi.check(lib.mhip_migrate_exchange(comm._h, self.RECORD, _p(rec), _p(new), _stream()))
        n_new = new.shape[0]
        if n_new:
            # owned bodies in curve order, ties by entity id: the local order every rank and the single-rank run agree on
            key2 = (self._cell_keys(new[:, 1:4].contiguous()) << 40) | new[:, 14].to(torch.int64)
            new = ops.gather_rows(ops.sort_by_key(key2.contiguous()), new)
        self.center, self.quat = new[:, 1:4].contiguous(), new[:, 4:8].contiguous()
        self.shape = new[:, 8:11].contiguous()
        if self.mixed:
            self.kind = new[:, 11].to(torch.int32).contiguous()
        self.mob_t, self.mob_r = new[:, 12].contiguous(), new[:, 13].contiguous()
        self.entity_id = new[:, 14].contiguous()
        self.n = n_new
        sizes = comm.all_gather(torch.tensor([float(n_new)], dtype=torch.float64, device=dev)).cpu().numpy().ravel()
        self.gid_first = int(sizes[:rank].sum())
        # everything indexed by the old numbering goes
        self.links.invalidate()
        if self.op is not None:
            self.op.close()
            self.op = None
        self._layout = None
        self._body_weight = None
        self._rebalances += 1
        out = dict(sent=int(sent.value), received=int(received.value), owned=n_new, recut=bool(recut))
        self.stats.update(migrated_out=out["sent"], migrated_in=out["received"])
        return out

    # -- ghost halo -----------------------------------------------------------------------------------------------------
    def _exchange_ghosts(self, replan=True):
        """ghost plan + body-record exchange, both inside the library (mhip_ghost_plan / mhip_ghost_exchange).
        replan=False moves the current records of the same ghosts through the plan of the last rebuild."""
        lib, comm = capi.load(), self.comm
        n, dev = self.n, self.center.device
        if replan:
            aabb, _ = self._aabb(self.center, self.quat, self.shape, self.kind)
            lay = self._layout = capi.GhostLayout()
            capi.check(lib.mhip_ghost_plan(comm._h, n, _p(aabb), self.buffer, C.byref(lay), _stream()))
        lay = self._layout
        n_lo, n_hi = int(lay.num_ghost_lo), int(lay.num_ghost_hi)
        rec = self._records()
        local = torch.empty((n_lo + n + n_hi, self.RECORD), dtype=torch.float64, device=dev)
        capi.check(lib.mhip_ghost_exchange(comm._h, self.RECORD, _p(rec), _p(local), _stream()))
        self.n_lo, self.n_hi, self.n_local = n_lo, n_hi, local.shape[0]
        self.local = dict(gid=local[:, 0].contiguous(), center=local[:, 1:4].contiguous(),
                          quat=local[:, 4:8].contiguous(), shape=local[:, 8:11].contiguous(),
                          kind=local[:, 11].to(torch.int32).contiguous(), mob_t=local[:, 12].contiguous(),
                          mob_r=local[:, 13].contiguous(), entity=local[:, 14].contiguous())
        self.stats.update(ghosts=n_lo + n_hi, halo_send_bodies=int(lay.num_sent))

    # -- one step -------------------------------------------------------------------------------------------------------------
    def step(self, integrate=True, force_rebuild=True, migrate=False):
        """migrate=True: before a rebuild, bodies whose lattice cell now belongs to another rank change owner
        (rebalance); every recut_every-th time the curve is re-cut by work first.
        force_rebuild=False applies the reference's rebuild rule across the ranks (GenNeighborLinkers.hpp:603-615 and
        the all-reduce of its parallel build): the ghosts' current state travels through the plan of the last rebuild,
        every rank tests its local bodies (owned + ghosts) against half the search buffer, one all-gather of the flags
        decides for everybody; without a rebuild the ghost layout, the pair list, its interior / boundary split and the
        operator's incidence index are kept and only the contact geometry is refreshed."""
        lib, comm = capi.load(), self.comm
        self.phase_ms = {}
        t_last = [time.perf_counter()]

        def tick(name):   # host wall time per phase, with a device sync, only when profiling
            if self.profile:
                torch.cuda.synchronize()
                now = time.perf_counter()
                self.phase_ms[name] = self.phase_ms.get(name, 0.0) + 1e3 * (now - t_last[0])
                t_last[0] = now

        tick("start")
        reuse = False
        if not force_rebuild and self.op is not None and getattr(self, "_layout", None) is not None:
            self._exchange_ghosts(replan=False)
            moved = self.links.needs_rebuild(self.local["center"])
            flags = comm.all_gather(torch.tensor([1.0 if moved else 0.0], dtype=torch.float64))
            reuse = not bool(flags.max().item() > 0.0)
        if not reuse:
            if migrate:
                self.rebalance(recut=self.splitters is None or
                               (self.recut_every > 0 and self._rebalances % self.recut_every == 0))
                tick("migrate")
            self._exchange_ghosts()
        tick("ghost_exchange")
        L, dev = self.local, self.center.device
        nl = self.n_local
        self.stats["rebuilt"] = not reuse
        if reuse:
            pairs, counted, nci = self.pairs, self.counted, self._nci
            nc = pairs.shape[0]
        else:
            aabb, brad = self._aabb(L["center"], L["quat"], L["shape"], L["kind"])
            self.links.generate(aabb, L["center"], brad, force=True)
            tick("aabb_neighbour_list")
            c_all = self.links.num_pairs
            pairs = torch.empty((c_all, 2), dtype=torch.int32, device=dev)
            counted = torch.empty(c_all, dtype=torch.uint8, device=dev)
            cnt = C.c_size_t(0)
            n_int = C.c_size_t(0)  # interior contacts (both bodies owned) first, boundary contacts (one ghost) after
            capi.check(lib.mhip_partition_pairs_owned(c_all, _p(self.links.pairs), self.n_lo, self.n, _p(pairs),
                                                      _p(counted), C.byref(n_int), C.byref(cnt), _stream()))
            nc, nci = int(cnt.value), int(n_int.value)
            pairs, counted = pairs[:nc].contiguous(), counted[:nc].contiguous()
            self._nci = nci
            tick("partition_pairs")
        mob_t, mob_r = L["mob_t"], L["mob_r"]
        if self.op is not None and not reuse:
            self.op.close()
        if self.mixed:
            seg = None
            con = ops.contact_mixed(pairs, L["kind"], L["center"], L["quat"], L["shape"])
            if reuse:
                self.op.refresh(con["normal"], ra=con["ra"], rb=con["rb"])
            else:
                self.op = ops.ContactOperator(pairs, con["normal"], mob_t, self.dt, ra=con["ra"], rb=con["rb"],
                                              mob_rot=mob_r, priority=con["sep"])
        else:
            seg = ops.spherocylinder_segments(L["center"], L["quat"], L["shape"][:, 0].contiguous(),
                                              L["shape"][:, 1].contiguous())
            con = ops.contact_spherocylinders(pairs, seg, L["center"], want_points=False, arms="arclength")
            # rod-compressed kinematics: velocity rows (and the halo) carry (U, W x u); (U, W) = body_velocity()
            if reuse:
                self.op.refresh(con["normal"], rod=(con["s"], con["t"], seg))
            else:
                self.op = ops.ContactOperator(pairs, con["normal"], mob_t, self.dt, mob_rot=mob_r,
                                              rod=(con["s"], con["t"], seg), priority=con["sep"])
        op = self.op
        if self.tiering is not None:
            op.set_tiering(self.tiering)
        tick("narrow_phase_operator")
        self.vel = torch.zeros((nl, 6), dtype=torch.float64, device=dev)
        self._keep = (pairs, counted, con, mob_t, mob_r, seg)
        capi.check(lib.mhip_contact_op_set_partition(op._h, self.n_lo, self.n, _p(counted), _p(self.vel)))
        # staged BBPGD
        x = torch.zeros(nc, dtype=torch.float64, device=dev)
        g, x_tmp, g_tmp = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        sp = capi.Space(ops.SPACE_LOWER_BOUND, 0.0, 0.0)
        pc = capi.PgdConfig(int(self.cfg.max_iters), float(self.cfg.tol), int(self.cfg.residual_kind))
        # the whole staged loop runs inside the library (csrc/dist.hip): body sweep -> velocity halo in flight during
        # the interior sweep -> boundary sweep -> 3-double all-gather -> finalize, polled every poll_every iterations
        halo = self._layout.halo   # the velocity halo of the ghost layout (lists owned by the communicator)
        halo.velocity = self.vel.data_ptr()
        res = capi.SolveResult()
        dprof = capi.DistProfile()
        # (what a later step without a rebuild continues from is in place BEFORE the solve: a solve that ends in an
        # error -- a peer's words that never came -- leaves a stepper that can step again)
        self.contacts, self.pairs, self.counted = con, pairs, counted
        capi.check(lib.mhip_bbpgd_solve_contact_distributed(
            op._h, comm._h, C.byref(halo), nci, _p(con["sep"]), C.byref(sp), C.byref(pc), _p(x), _p(g), _p(x_tmp),
            _p(g_tmp), self.poll_every, C.byref(res), C.byref(dprof) if self.profile else None, _stream()))
        if self.profile:
            self.prof["body_ms"] += dprof.body_ms
            self.prof["con_ms"] += dprof.constraint_ms
            self.prof["halo_wait_ms"] = self.prof.get("halo_wait_ms", 0.0) + dprof.halo_wait_ms
            self.prof["halo_post_ms"] = self.prof.get("halo_post_ms", 0.0) + dprof.halo_post_ms
            self.prof["record_ms"] = self.prof.get("record_ms", 0.0) + dprof.record_ms
            self.prof["iters"] += int(dprof.timed_iterations)
            # what the solve ACTUALLY used (the library reports it, not the flags it was asked with)
            self.prof["halo_path"] = ("none", "inboxes", "send/recv")[int(dprof.halo_path)]
            self.prof["record_path"] = ("?", "mailbox (fused)", "mailbox", "all-gather")[int(dprof.record_path)]
        tick("solve")
        self.lam, self.grad, self.contacts, self.pairs, self.counted = x, g, con, pairs, counted
        self.lam_prev, self.grad_prev = x_tmp, g_tmp     # the iterate before (what a warm restart continues from)
        if integrate:
            a, b = self.n_lo, self.n_lo + self.n
            own_c, own_q = L["center"][a:b], L["quat"][a:b]
            ops.integrate_euler(self.dt, op.body_velocity()[a:b], own_c, own_q)
            self.center.copy_(own_c)
            self.quat.copy_(own_q)
        tick("integrate")
        owned_contacts = int(counted.sum().item()) if nc else 0
        # work per owned body for the next re-cut: 1 + the contacts it takes part in (every contact of an owned body is
        # present locally)
        deg = torch.bincount(pairs.reshape(-1).long(), minlength=nl) if nc else torch.zeros(nl, dtype=torch.int64, device=dev)
        self._body_weight = 1.0 + deg[self.n_lo:self.n_lo + self.n].to(torch.float64)
        self.stats.update(local_bodies=nl, local_contacts=nc, owned_contacts=owned_contacts,
                          num_iters=int(res.num_iters), residual=float(res.residual), converged=bool(res.converged))
        return dict(self.stats)
