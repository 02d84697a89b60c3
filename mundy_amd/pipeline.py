"""The timestep composition of the hot path -- what the reference's step loops call, in their order
(scrap/lcp_spheres/NgpLcp.cpp:835-920; operator pipeline of
scrap/parameter_interface/alens/tests/performance_tests/Bacteria.cpp:755-805):

    compute_aabb -> GenNeighborLinks.generate (rebuild only when the search buffer is violated)
                 -> signed separation + contact normal (+ lever arms) per link
                 -> resolve_collisions: BBPGD on the LCP  0 <= dt D^T M D lambda + sep  _|_  lambda >= 0
                 -> Euler update x += dt U (and q <- rotate(q, W dt) for rods)

Everything is device resident; host logic here only sequences library calls.
"""
from dataclasses import dataclass, field

import torch

from . import ops, synth


@dataclass
class StepStats:
    num_bodies: int = 0
    num_contacts: int = 0
    rebuilt: bool = False
    num_iters: int = 0
    residual: float = 0.0
    converged: bool = False
    timings_ms: dict = field(default_factory=dict)


class ContactStepper:
    """One rank's bodies (spheres, spherocylinders, or a mix with ellipsoids) and the per-timestep contact resolution."""

    def __init__(self, kind, center, radius, quat=None, length=None, *, dt=5e-3, viscosity=1e-3, search_buffer=0.25,
                 search_kind=ops.SEARCH_AABB, periodic_box=None, cfg=None, warm_start=False, mob_trans=None,
                 mob_rot=None, rod_kinematics=True, kinds=None, shape=None, friction=None, contact_cutoff=None,
                 conservative_ellipsoid_box=False, friction_method="apgd"):
        """kind = "sphere" | "spherocylinder" | "mixed".  Mixed systems (BASELINE configs[4]) pass kinds [n] int32
        (0 sphere, 1 spherocylinder, 2 ellipsoid) and shape [n, 3] = (r,-,-) / (r,L,-) / (r1,r2,r3) instead of
        radius / length."""
        if kind not in ("sphere", "spherocylinder", "mixed"):
            raise ValueError("kind must be 'sphere', 'spherocylinder' or 'mixed'")
        if kind == "spherocylinder" and (quat is None or length is None):
            raise ValueError("spherocylinders need quat and length")
        if kind == "mixed" and (quat is None or kinds is None or shape is None):
            raise ValueError("mixed systems need quat, kinds and shape")
        if friction is not None and kind != "spherocylinder":
            raise ValueError("the friction extension is wired for spherocylinders")
        self.kind = kind
        # BUILD EXTENSION, mixed systems: the tight conservative ellipsoid box in the neighbour search instead of the
        # reference's (compute_aabb.hpp:82-103), which can miss overlapping ellipsoids of general orientation
        self.conservative_ellipsoid_box = bool(conservative_ellipsoid_box)
        # BUILD EXTENSION (parity unpinned: the reference has no frictional solver): Coulomb coefficient, None = the
        # reference's frictionless LCP
        self.friction = None if friction is None else float(friction)
        # "apgd" (Mazhar et al. 2015; 10^6 rods at mu = 0.3: 2 189 sweeps from the raw packing, 318 from the relaxed one)
        # or "bbpgd" (the reference's iteration with a cone projection: 23 227 / 754)
        self.friction_method = friction_method
        self.center, self.radius, self.quat, self.length = center, radius, quat, length
        self.kinds, self.shape = kinds, shape
        self.dt, self.viscosity = float(dt), float(viscosity)
        self.box = periodic_box
        self.cfg = cfg or ops.PGDConfig(max_iters=10000, tol=1e-5)  # NgpLcp.cpp:851-852
        self.warm_start = warm_start
        # spherocylinders: lever arms as one arclength per contact (mhip_contact_op_create_rods) -- same operator up to
        # rounding, 18 % fewer bytes per solver iteration; False selects the (ra, rb) vector form
        self.rod_kinematics = bool(rod_kinematics)
        self.links = (ops.GenNeighborLinks().set_search_buffer(search_buffer).set_search_kind(search_kind)
                      .set_periodic_box(periodic_box).concretize())
        n = center.shape[0]
        # dry local drag U = F/(6 pi mu r), W = T/(8 pi mu r^3) (NgpLcp.cpp:484-486, Bacteria.cpp:810-848); the
        # per-body coefficients are set-up data computed once on the host (same numbers feed the CPU oracle)
        if kind == "sphere":
            self.bounding_radius = radius.clone()
            eff = radius
        elif kind == "mixed":
            self.bounding_radius = ops.compute_aabb_mixed(kinds, center, quat, shape)[1]
            eff = self.bounding_radius
        else:
            self.bounding_radius = ops.bounding_radius_spherocylinders(radius, length)
            eff = self.bounding_radius
            self.seg = torch.empty((n, 8), dtype=torch.float64, device=center.device)
        if mob_trans is None:
            mt, mr = synth.dry_mobility(eff.cpu().numpy(), viscosity=self.viscosity)
            mob_trans = torch.from_numpy(mt).to(center.device)
            mob_rot = torch.from_numpy(mr).to(center.device) if kind != "sphere" else None
        self.mob_trans, self.mob_rot = mob_trans, (mob_rot if kind != "sphere" else None)
        self.op = None
        self.lam = None
        self.contacts = None
        self.work_mapping = None  # (xcd_tile, lanes_per_body) for ContactOperator.set_work_mapping: time only
        self.tiering = None       # ContactOperator.set_tiering mode (None: the library default): time only
        # BUILD OPTION (None = off, the reference's behaviour: every neighbour pair is a constraint, NgpLcp.cpp:346-373):
        # only pairs within contact_cutoff of touching become constraints of this step (ballot compaction of the
        # candidate list); the dropped pairs are checked afterwards (they must satisfy g >= 0, i.e. stay inactive) and
        # the step is redone on the full list if one does not.
        self.contact_cutoff = None if contact_cutoff is None else float(contact_cutoff)
        self.cutoff_fallbacks = 0

    # -- stages -----------------------------------------------------------------------------------------------------
    _BODY_ARRAYS = ("center", "radius", "quat", "length", "bounding_radius", "mob_trans", "mob_rot", "shape", "kinds")

    def snapshot(self):
        """device copies of every per-body array (to restart a step from the same input)"""
        return {k: getattr(self, k).clone() for k in self._BODY_ARRAYS if getattr(self, k, None) is not None}

    def restore(self, snap):
        for k, v in snap.items():
            getattr(self, k).copy_(v)

    def reorder_bodies(self, cell_size=None, lo=None, curve="morton", hi=None, level=7):
        """Space-filling-curve permutation of all per-body arrays by centre (what the reference's zmorton /
        Hilbert helpers are advertised for): neighbours in space become neighbours in memory, so every gather of the
        contact sweeps hits nearby lines.  curve = "morton" (lattice of edge cell_size anchored at lo) or "hilbert"
        (the hilbert_3d order of a (2^level)^3 lattice over [lo, hi]; both give the same sweep times).  Returns the
        permutation (new position k holds old body perm[k])."""
        if lo is None:
            lo = self.center.min(dim=0).values.tolist() if self.box is None else [0.0, 0.0, 0.0]
        if curve == "hilbert":
            from . import distributed
            if hi is None:
                hi = self.center.max(dim=0).values.tolist() if self.box is None else list(self.box)
            table = torch.from_numpy(distributed.hilbert_key_table(level).astype("int32")).to(self.center.device)
            perm = ops.curve_order(self.center, lo, hi, level, table)
        elif curve == "morton":
            if cell_size is None:
                cell_size = 2.0 * float(self.bounding_radius.max())
            perm = ops.morton_order(self.center, lo, cell_size)
        else:
            raise ValueError("curve must be 'morton' or 'hilbert'")
        for name in self._BODY_ARRAYS:
            t = getattr(self, name, None)
            if t is not None:
                t.copy_(ops.gather_rows(perm, t) if t.dtype == torch.float64 else t[perm.long()])
        # the neighbour list, the operator's incidence index and the multipliers are in the old numbering: a reused list
        # would pair the wrong bodies unless the displacement test happened to fire, so force the rebuild
        self.links.invalidate()
        if self.op is not None:
            self.op.close()
            self.op = None
        self.lam = None
        return perm

    def compute_aabb(self):
        if self.kind == "sphere":
            self.aabb = ops.compute_aabb_spheres(self.center, self.radius)
        elif self.kind == "mixed":
            self.aabb = ops.compute_aabb_mixed(self.kinds, self.center, self.quat, self.shape,
                                               conservative_ellipsoids=self.conservative_ellipsoid_box)[0]
        else:
            self.aabb = ops.compute_aabb_spherocylinders(self.center, self.quat, self.radius, self.length)
        return self.aabb

    def generate_neighbor_links(self, force=False):
        return self.links.generate(self.aabb, self.center, self.bounding_radius, force=force)

    def compute_contacts(self):
        pairs = self.links.pairs
        if self.kind == "sphere":
            sep, normal = ops.contact_spheres(pairs, self.center, self.radius, box=self.box)
            self.contacts = dict(sep=sep, normal=normal, ra=None, rb=None)
        elif self.kind == "mixed":  # pairs binned by shape class, one distance routine per class
            self.contacts = ops.contact_mixed(pairs, self.kinds, self.center, self.quat, self.shape, box=self.box)
        else:
            ops.spherocylinder_segments(self.center, self.quat, self.radius, self.length, out=self.seg)
            rodk = self.rod_kinematics and self.friction is None
            self.contacts = ops.contact_spherocylinders(pairs, self.seg, self.center, want_points=False,
                                                        arms="arclength" if rodk else "vector", box=self.box)
        return self.contacts

    def _compact_contacts(self):
        """contacts within the cutoff -> (pairs, contact arrays) of this step's constraints, the rest kept aside"""
        c = self.contacts
        kept = ops.select_contacts(c["sep"], self.contact_cutoff)
        keep_mask = torch.zeros(c["sep"].shape[0], dtype=torch.bool, device=kept.device)
        keep_mask[kept.long()] = True
        pairs64 = self.links.pairs.view(torch.float64).reshape(-1)       # one (i, j) row = 8 bytes
        out = {k: (ops.gather_rows(kept, v) if isinstance(v, torch.Tensor) and v.shape[0] == keep_mask.shape[0] else v)
               for k, v in c.items()}
        pairs = ops.gather_rows(kept, pairs64).view(torch.int32).reshape(-1, 2)
        return pairs, out, ~keep_mask

    def _dropped_pairs_stay_inactive(self, dropped):
        """g = sep + dt * sdot >= 0 on the pairs left out of the solve, from the body velocities it produced"""
        if not bool(dropped.any()):
            return True
        c, p = self.full_contacts, self.links.pairs[dropped].long()
        vel = self.op.body_velocity()
        n = c["normal"][dropped]
        vi, vj = vel[p[:, 0], :3].clone(), vel[p[:, 1], :3].clone()
        if self.kind != "sphere":
            if c.get("ra") is not None:
                ra, rb = c["ra"][dropped], c["rb"][dropped]
            else:  # arclength form: arm = (s - 1/2) (p1 - p0)
                u = self.seg[:, 3:6] - self.seg[:, 0:3]
                ra = (c["s"][dropped] - 0.5).unsqueeze(1) * u[p[:, 0]]
                rb = (c["t"][dropped] - 0.5).unsqueeze(1) * u[p[:, 1]]
            vi += torch.cross(vel[p[:, 0], 3:], ra, dim=1)
            vj += torch.cross(vel[p[:, 1], 3:], rb, dim=1)
        g = c["sep"][dropped] - self.dt * ((vi - vj) * n).sum(dim=1)
        return bool((g >= -self.cfg.tol).all())

    def resolve_collisions(self, rebuilt):
        if self.contact_cutoff is not None and self.friction is None:
            self.full_contacts = self.contacts
            pairs, self.contacts, dropped = self._compact_contacts()
            res = self._resolve(True, pairs)       # the constraint set changes from step to step: new operator
            if self._dropped_pairs_stay_inactive(dropped):
                self.contact_pairs = pairs
                return res
            self.cutoff_fallbacks += 1             # a dropped pair would have carried an impulse: the full list decides
            self.contacts = self.full_contacts
            self.contact_pairs = self.links.pairs
            return self._resolve(True, self.links.pairs)
        self.contact_pairs = self.links.pairs
        return self._resolve(rebuilt, self.links.pairs)

    def _resolve(self, rebuilt, pairs):
        c = self.contacts
        # a step that reuses the neighbour list keeps the operator's incidence index and only refreshes its geometry
        reuse = (not rebuilt and self.op is not None and self.friction is None and
                 self.op.num_constraints == pairs.shape[0] and getattr(self.op, "_h", None))
        if self.op is not None and not reuse:
            self.op.close()
        if self.friction is not None:
            ra, rb = ops.surface_lever_arms(pairs, c["normal"], c["ra"], c["rb"], self.radius)
            self.op = ops.ContactOperator(pairs, c["normal"], self.mob_trans, self.dt, ra=ra, rb=rb,
                                          mob_rot=self.mob_rot)
            p, g, res = ops.solve_friction_contact(self.op, c["sep"], self.friction, cfg=self.cfg,
                                                   method=self.friction_method)
            self.impulse, self.lam = p, (p * c["normal"]).sum(dim=1)
            return res
        if self.kind == "spherocylinder" and self.rod_kinematics:
            if reuse:
                self.op.refresh(c["normal"], rod=(c["s"], c["t"], self.seg))
            else:
                self.op = ops.ContactOperator(pairs, c["normal"], self.mob_trans, self.dt,
                                              mob_rot=self.mob_rot, rod=(c["s"], c["t"], self.seg), priority=c["sep"])
        elif reuse:
            self.op.refresh(c["normal"], ra=c.get("ra"), rb=c.get("rb"))
        else:
            self.op = ops.ContactOperator(pairs, c["normal"], self.mob_trans, self.dt, ra=c.get("ra"),
                                          rb=c.get("rb"), mob_rot=self.mob_rot, priority=c["sep"])
        if self.work_mapping is not None:
            self.op.set_work_mapping(*self.work_mapping)
        if self.tiering is not None:
            self.op.set_tiering(self.tiering)
        if getattr(self, "profile_next", False):
            self.op.set_profiling(True)  # per-kernel HIP-event timing of the fused iteration (bench.py roofline)
        nc = pairs.shape[0]
        if rebuilt or self.lam is None or not self.warm_start or self.lam.shape[0] != nc:
            self.lam = torch.zeros(nc, dtype=torch.float64, device=self.center.device)  # NgpLcp.cpp:890-891
        x, g, res = ops.solve_lcp(self.op, c["sep"], self.lam, self.cfg)
        self.lam, self.grad = x, g
        return res

    def integrate(self):
        vel = self.op.body_velocity()
        ops.integrate_euler(self.dt, vel, self.center, self.quat)
        if self.box is not None:
            # wrap_rigid_inplace of a Sphere / Spherocylinder / Ellipsoid: the centre goes back into the box,
            # orientation and size are untouched (periodicity.hpp:1088-1113, :1156-1160)
            ops.wrap_rigid(self.box, self.center)

    # -- one timestep -------------------------------------------------------------------------------------------------
    def step(self, integrate=True, force_rebuild=False, timed=False):
        st = StepStats(num_bodies=self.center.shape[0])
        ev = []

        def mark(name):
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((name, e))

        mark("start")
        self.compute_aabb()
        mark("aabb")
        st.rebuilt = self.generate_neighbor_links(force=force_rebuild)
        mark("broadphase")
        self.compute_contacts()
        mark("narrowphase")
        res = self.resolve_collisions(st.rebuilt)
        mark("solve")
        if integrate:
            self.integrate()
        mark("integrate")
        st.num_contacts = self.contact_pairs.shape[0]
        st.num_iters, st.residual, st.converged = res.num_iters, res.residual, res.converged
        if timed:
            torch.cuda.synchronize()
            for (_, a), (name, b) in zip(ev[:-1], ev[1:]):
                st.timings_ms[name] = a.elapsed_time(b)
        return st
