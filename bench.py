#!/usr/bin/env python
"""bench.py -- timesteps/s and contact-pairs/s of the contact hot path on 10^6 spherocylinders.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run, one rank per GPU.  W untimed warm-up steps, then exactly K timed steps bracketed by a barrier +
torch.cuda.synchronize() on both sides, MAX over ranks; rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input, i.e. the whole body of the reference's
single-device app (scrap/lcp_spheres/NgpLcp.cpp:835-920) on BASELINE.json configs[2]: 10^6 spherocylinders (r = 0.5,
L = 2) at 40 % volume fraction, random positions/orientations (overlaps allowed, the LCP resolves them):
    compute_aabb -> neighbour list (AABB + buffer, rebuilt every step) -> signed separations / normals / lever arms ->
    BBPGD solve of the frictionless LCP (tol 1e-5, dt 5e-3, mu 1e-3, dry mobility) -> Euler update.
Every step starts from the same pristine input (restored by a device copy inside the timed region), so all steps do
identical work.  Inputs are resident in HBM before the timed region starts.

N > 1 runs BASELINE configs[3] as written: the SAME 10^6-spherocylinder system, cut along a Hilbert curve into N
contiguous ranges, one per GPU ("scaling": "strong"; `value` = timesteps/s of that one system): ghost-body halo at the
neighbour-list build, and per BBPGD iteration a ghost-velocity halo (RCCL send/recv) + one 5-double all-gather.  The
iteration loop and the transport are C++ inside libmundy_hip (csrc/dist.hip); torch.distributed only launches (RCCL id
broadcast, barrier, max over ranks).  Every sum that feeds the BB step is a double-double pair rounded once, so the
partitioned run takes the single-GPU run's iterates and iteration count.  The line is refused (exit code 3) unless the
halo really travels over RCCL: a run that fell back to the host-staged transport is not a measurement of xGMI
(`--allow-host-transport` for development boxes where several ranks share one GPU).
`--weak` instead grows the system with N (--bodies per GPU, "scaling": "weak", `value` = N x global timesteps/s); the
BBPGD iteration count grows with the system, which that line folds into what reads as a scaling curve -- so both lines
also carry `constraint_updates_per_sec` (contacts x iterations per second), which does not.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); must be in the
# environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak, FMA = 2 flops
# fp64 operations (+ - x / sqrt, each counted once) of ONE objective evaluation of the ellipsoid shared-normal distance:
# two sin/cos pairs (2 x 45), the normal (3), two foot-point maps (each: two quaternion rotations of 56 + the body-frame
# map ~40 + 3), the distance (8) ~ 410, plus ~50 of minimiser logic per evaluation
FLOPS_PER_EVALUATION = 460.0
# one metric for every N: the N = 1 line is BASELINE configs[2], the N > 1 lines are configs[3] -- the same system
METRIC = "timesteps/sec, 10^6 spherocylinders, frictionless LCP contact (BBPGD)"
MIXED_PHI_DEFAULT = 0.40   # configs[4]'s box is derived "from phi" = 0.40, like configs[2]


def kernel_bytes(contacts, bodies, active_contacts=None, tier=None, iterations=None, kin="rod", drift_rows=True):
    """Algorithmic bytes per launch of the two sweeps of one fused BBPGD iteration (rod-compressed kinematics): every
    array the launch REQUIRES, counted once (a gathered table once per row, not once per reader) -- what HBM must move
    even with perfect caches.  This is what roofline.achieved divides.
      k_constraint  every constraint is evaluated every iteration: pair 8 + normal 24 + arclengths 16 + packed (x, g) 16
                    + q 8 read, packed (x, g) 16 written = 88 B per constraint; the 48-byte (U, W x u) body rows once
                    each = 48 B per body
      k_body        walks only the half edges its activity masks flag (a contact with x = 0, g >= 0 stays at
                    Proj(0 - step g) = 0 and adds nothing), streamed from the compact active lists: per ACTIVE half
                    edge incidence entry 4 + (n, s - 1/2) record 32 = 36 B; per ACTIVE contact its packed iterate 16 B
                    (gathered by both of its half edges, counted once); per body row pointer 4 + active-list pointer 4
                    + mask 8 + snapshot mask 8 + mobilities 16 + axis 24 + velocity row 48 = 112 B (the angular
                    velocity, 24 B, is written once per solve, by a sweep of the final iterate); tiered iterations
                    add 72 B per body: drift read + written 16, firing threshold 8, the row of the previous iterate
                    48.  `active_contacts` is measured in the run (state at the end of the solve); None = every
                    contact (the pre-mask count).
    SURVEY 8(d)'s own figure, 368 C + 96 N per iteration, charges a gathered row to every contact that reads it and
    assumes vector lever arms; this implementation streams scalar arclengths and serves the 48 MB row table from L2 /
    Infinity Cache, so that figure divided by the measured time exceeds the HBM peak (1.39 x at 10^6 rods) -- it is
    unusable as a denominator here and is not printed."""
    act = contacts if active_contacts is None else active_contacts
    # kin = "rigid" (mixed shapes: explicit lever arms): the constraint streams 48 B of arms instead of 16 B of
    # arclengths (120 B per constraint), a half-edge record is (n, r) 48 B instead of (n, s - 1/2) 32 B (entry + record
    # 52 B), a body has no axis (88 B) -- and in tiered iterations reads its longest arm (8 B) on top of the drift words
    # tier_body (round 4): + 48 B -- the drift of a tiered sweep is the difference of the body's new row and its row of
    # the previous iterate, which the sweep therefore reads (rounds 2-3 kept the force change in registers instead: 18
    # VGPRs and a fourth LDS plane, which capped the vector-arm sweep at three workgroups per CU)
    # (systems beyond 1.75e6 bodies keep the force change in registers instead -- drift_rows False: no such read)
    per_con, per_edge, per_body, tier_body = (88.0, 36.0, 112.0, 72.0) if kin == "rod" else (120.0, 52.0, 88.0, 80.0)
    if not drift_rows:
        tier_body -= 48.0
    con = per_con * contacts + 48.0 * bodies
    body = 2 * per_edge * act + 16.0 * act + per_body * bodies
    if tier and tier.get("tiered_iterations", 0) > 0 and iterations:
        # cold tier: over the tiered iterations only the hot share h of the contacts is swept (plus the few
        # thousand awake contacts of the cold tail, not counted here); nothing scans the sleepers.  The body sweep additionally
        # updates each body's drift (8 + 8) and reads its firing threshold (8)
        w = min(1.0, tier["tiered_iterations"] / float(iterations))
        h = tier["mean_hot_fraction"]
        con = (1.0 - w) * con + w * (per_con * h * contacts + 48.0 * bodies)
        body += w * tier_body * bodies
    return {"k_constraint": con, "k_body": body}


def roofline_entries(contacts, bodies, con_ms, body_ms, launches, label="", active_contacts=None, tier=None,
                     iterations=None, kin="rod", drift_rows=True):
    kb = kernel_bytes(contacts, bodies, active_contacts, tier, iterations, kin, drift_rows)
    ms = {"k_constraint": con_ms, "k_body": body_ms}
    ent = {}
    for k in ms:
        a = kb[k] / (ms[k] * 1e-3) / 1e9
        ent[k] = {"bound": "hbm", "kernel": k + ("<X_SOLVE,KIN_ROD>" if kin == "rod" else "<X_SOLVE,KIN_RIGID>") + label, "achieved": round(a, 1),
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4), "traffic": None,
                  "traffic_source": None, "avg_launch_ms": round(ms[k], 4), "launches": launches,
                  "bytes_per_launch": kb[k]}
    if active_contacts is not None:
        ent["k_body"]["active_contact_fraction"] = round(active_contacts / max(1, contacts), 4)
    dominant = max(ms, key=ms.get)  # the sweep with the longer launch is the one the step waits on
    other = "k_body" if dominant == "k_constraint" else "k_constraint"
    return ent[dominant], {other: ent[other]}, dominant, other


def attach_traffic(roof, extra, dom, oth, n, buffer, name="traffic.json"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (scripts/profile_bench.sh:
    separate FETCH_SIZE / WRITE_SIZE runs, 2 x FETCH_SIZE + WRITE_SIZE as the gfx950 guide prescribes).  They are NOT
    measured in this run -- the source is named next to the number."""
    tpath = os.path.join(ROOT, "profiles", name)
    if os.path.exists(tpath) and n == 1_000_000 and buffer == 0.1:
        from mundy_amd import build as hip_build
        tj = json.load(open(tpath))
        stamp = hip_build.sweep_kernels_stamp()
        if tj.get("_kernel_stamp") != stamp:
            # measured on other kernels (or flags) than the ones running: not a measurement of these
            for ent in (roof, extra[oth]):
                ent["traffic"] = None
                ent["traffic_source"] = ("profiles/%s was measured on sweep kernels with stamp %s, the running library's is "
                                         "%s: dropped (scripts/profile_bench.sh regenerates it)"
                                         % (name, tj.get("_kernel_stamp"), stamp))
            return
        src = "profiles/%s (%s)" % (name, tj.get("_source", "rocprofv3 --pmc passes of `python bench.py`, committed"))
        for ent, k in ((roof, dom), (extra[oth], oth)):
            ent["traffic"] = tj.get(k, {}).get("hbm_bytes_per_launch")
            ent["traffic_source"] = src if ent["traffic"] is not None else None


def measured_ceiling(roof, extra):
    """SURVEY 8d asks for a measured STREAM-like ceiling ALONGSIDE the spec peak: a 1-GiB mhip_deep_copy (the library's
    own streaming kernel: 8 B read + 8 B written per element, 2 GiB of traffic, eight times the Infinity Cache) timed
    in this run with HIP events, best of four batches.  `peak` stays the 8 TB/s spec; `peak_measured` and
    `frac_of_measured` say what this box's memory system delivered to a pure stream on the same day."""
    from mundy_amd import capi
    lib = capi.load()
    n = 1 << 27
    x = torch.empty(n, dtype=torch.float64, device="cuda").fill_(1.0)
    z = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    px, pz = x.data_ptr(), z.data_ptr()
    for _ in range(3):
        capi.check(lib.mhip_deep_copy(n, pz, px, stream))
    best = None
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            capi.check(lib.mhip_deep_copy(n, pz, px, stream))
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10
        best = t if best is None else min(best, t)
    gbs = 16.0 * n / (best * 1e-3) / 1e9
    del x, z
    for ent in [roof] + list(extra.values()):
        if ent is not None and ent.get("unit") == "GB/s":
            ent["peak_measured"] = round(gbs, 1)
            ent["peak_measured_how"] = "mhip_deep_copy of 2^27 doubles (1 GiB read + 1 GiB written) in this run, HIP events, best of 4 x 10"
            ent["frac_of_measured"] = round(ent["achieved"] / gbs, 4)
    return gbs


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--bodies", type=int, default=1_000_000, help="spherocylinders per GPU (default: configs[2])")
    p.add_argument("--buffer", type=float, default=0.1, help="search buffer added to every AABB")
    p.add_argument("--tol", type=float, default=1e-5)
    p.add_argument("--max-iters", type=int, default=10000)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-reorder", dest="reorder", action="store_false", help="skip the per-step Z-order reordering")
    p.add_argument("--reorder-cell", type=float, default=3.0, help="lattice edge of the Morton keys")
    p.add_argument("--cpu-iters", type=int, default=12, help="BBPGD iterations timed on the host cores")
    p.add_argument("--weak", action="store_true",
                   help="N > 1: --bodies rods PER GPU (one system of N x --bodies, weak scaling) instead of the default, "
                        "BASELINE configs[3] as written: ONE system of --bodies rods partitioned over the N GPUs")
    p.add_argument("--strong", action="store_true", help="(the default for N > 1; kept for older command lines)")
    p.add_argument("--distributed", action="store_true",
                   help="take the partitioned path even with one rank (under torch.distributed.run): one rank over RCCL")
    p.add_argument("--no-mailbox", action="store_true",
                   help="N > 1: the per-iteration reduction records through ncclAllGather instead of the node's mailbox")
    p.add_argument("--no-halo-ipc", action="store_true",
                   help="N > 1: the per-iteration velocity halo through grouped ncclSend / ncclRecv instead of the "
                        "node's IPC-mapped inboxes")
    p.add_argument("--no-transport-ab", action="store_true",
                   help="N > 1: skip the untimed A/B of the transport mechanisms after the timed steps (`transport_ab`)")
    p.add_argument("--allow-host-transport", action="store_true",
                   help="print a line even when the halo is staged through host memory (gloo) instead of RCCL; "
                        "without it such a run exits with code 3")
    p.add_argument("--xcd-tile", type=int, default=-1, help="tile -> XCD mapping of the sweeps (time only)")
    p.add_argument("--lanes-per-body", type=int, default=-1, help="lanes per body of the body sweep (time only)")
    p.add_argument("--no-cold-tier", action="store_true",
                   help="sweep every contact every iteration (time only: the iterates are the same bits)")
    p.add_argument("--cold-tier-any-size", action="store_true",
                   help="cold tier even below the size (1.5M contacts) from which it pays -- for tests of the accounting")
    p.add_argument("--relaxed-steps", type=int, default=2,
                   help="N = 1: after the headline steps, advance the packing by this many full steps and time --steps "
                        "more from THAT state (labelled `relaxed_packing`; SURVEY 8d.3 allows one relaxation pre-pass)")
    p.add_argument("--mixed", action="store_true",
                   help="BASELINE configs[4] instead of the headline workload: --bodies bodies, one third each spheres "
                        "(r 0.5), spherocylinders (r 0.5, L 2) and ellipsoids (0.8, 0.5, 0.4), random orientations, "
                        "--mixed-phi volume fraction; shape classes binned, L-BFGS ellipsoid distances, vector-arm "
                        "operator.  With --gpus N > 1 the one system is Hilbert-partitioned over the ranks like the rods "
                        "(configs[4] as BASELINE states it).  Never the default line.")
    p.add_argument("--mixed-phi", type=float, default=MIXED_PHI_DEFAULT)
    p.add_argument("--ellipsoid-fma", action="store_true",
                   help="--mixed: LABELLED build option -- the ellipsoid minimisation classes from the build with "
                        "floating-point contraction on (results at the reference's 1e-4 tolerance instead of bit parity "
                        "with the oracle).  Never the default; the line says which arithmetic ran.")
    p.add_argument("--friction-method", choices=("bbpgd", "apgd"), default="apgd",
                   help="--friction: the reference's BBPGD iteration with a cone projection, or APGD (Mazhar et al. 2015)")
    p.add_argument("--friction", type=float, default=None,
                   help="BUILD EXTENSION, parity unpinned: Coulomb coefficient of the cone-complementarity solver "
                        "(the reference has no frictional solver; default = its frictionless LCP).  N = 1 only.")
    return p.parse_args()


def launch_plan(gpus, env, argv, device_count=None):
    """--gpus N must MEAN N.  Called before anything touches the GPU.  Returns None when this process is one of the N
    ranks it should be (N == WORLD_SIZE, or N == 1 without a launcher), the command line of the launcher to start as a
    CHILD when N > 1 and no launcher started us (never an exec: a process that has initialised the GPU must not be
    replaced on this pool, and this way the rule holds whatever gets added above), and raises SystemExit(2) on any
    mismatch a launcher left us with -- a scaling run must never be recorded from fewer ranks than it names."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; launch as\n  python -m "
                             "torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 "
                             "--master-port P bench.py --gpus %d ...\n" % (gpus, world, gpus, gpus))
            raise SystemExit(2)
        return None
    if gpus == 1:
        return None
    if device_count is not None and device_count < gpus and env.get("MUNDY_BENCH_BACKEND", "nccl") == "nccl":
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s)\n" % (gpus, device_count))
        raise SystemExit(2)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    # (torch.cuda.device_count() does not initialise the GPU on this image; nothing before this line has either)
    plan = launch_plan(args.gpus, os.environ, sys.argv[1:], device_count=torch.cuda.device_count())
    if plan is not None:
        import subprocess
        sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(plan)))
        sys.stderr.flush()
        raise SystemExit(subprocess.call(plan))   # the ranks inherit stdout: rank 0's JSON line is the output
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: mundy_amd has no CPU path")
    # MUNDY_BENCH_BACKEND=gloo lets several ranks share one GPU (development boxes have one): same code path, the halo
    # is staged through the host.  The driver's multi-GPU runs use the default, nccl (= RCCL).
    backend = os.environ.get("MUNDY_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1 or args.distributed:
        # a transport that stalls must end the run, not sit on the node: dump the stacks and exit after the limit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("MUNDY_BENCH_WATCHDOG_S", "900")), exit=True)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)

    from mundy_amd import build as hip_build
    if rank == 0:
        hip_build.build()
    if dist is not None:
        dist.barrier()
    from mundy_amd import ops, pipeline, synth

    n = args.bodies
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    if world > 1 or args.distributed:
        return main_distributed(args, rank, world, dist, ops, synth, dev)
    if args.mixed:
        return main_mixed(args, ops, pipeline, synth, dev)
    b = synth.spherocylinders(n, seed=1234)
    center, quat = dev(b["center"]), dev(b["quat"])
    radius, length = dev(b["radius"]), dev(b["length"])
    cfg = ops.PGDConfig(max_iters=args.max_iters, tol=args.tol)
    stepper = pipeline.ContactStepper("spherocylinder", center, radius, quat, length, dt=5e-3, viscosity=1e-3,
                                      search_buffer=args.buffer, search_kind=ops.SEARCH_AABB, cfg=cfg,
                                      friction=args.friction, friction_method=args.friction_method)
    if args.xcd_tile >= 0 or args.lanes_per_body > 0:
        stepper.work_mapping = (args.xcd_tile, args.lanes_per_body)
    if args.no_cold_tier:
        stepper.tiering = 0
    elif args.cold_tier_any_size:
        stepper.tiering = 3
    pristine = stepper.snapshot()
    prof = dict(body_ms=0.0, con_ms=0.0, iters=0)

    def one_step(timed_kernels, timed_stages=False):
        stepper.restore(pristine)
        if args.reorder:
            # the synthetic input is in random order; the Z-order body reordering operator is part of the step
            stepper.reorder_bodies(cell_size=args.reorder_cell, lo=[0.0, 0.0, 0.0])
        stepper.profile_next = timed_kernels
        st = stepper.step(integrate=True, force_rebuild=True, timed=timed_stages)
        if timed_kernels and args.friction is None:
            a, c, k = stepper.op.get_profile()
            prof["body_ms"] += a
            prof["con_ms"] += c
            prof["iters"] += k
            prof["tier"] = stepper.op.tier_stats()
            prof["solve_iters"] = st.num_iters
            prof["drift_rows"] = stepper.op.drift_source() == 1
        return st

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(False)
    sync()
    t0 = time.perf_counter()
    stats = [one_step(True) for _ in range(args.steps)]
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # contacts the body sweep's masks flag at the end of the solve: not (x == 0 and 0 <= g < inf)
    active = None
    if args.friction is None and stepper.lam is not None and getattr(stepper, "grad", None) is not None:
        lam_t, g_t = stepper.lam, stepper.grad
        active = int((~((lam_t == 0) & (g_t >= 0) & torch.isfinite(g_t))).sum().item())
    stage_ms = one_step(False, timed_stages=True).timings_ms  # one extra, untimed step for the stage breakdown
    contacts = stats[-1].num_contacts
    iters = [s.num_iters for s in stats]
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed

    # ---- roofline of the dominant sweep of the fused BBPGD iteration (see kernel_bytes) -----------------------------
    roof, extra = None, {}
    if prof["iters"] > 0:
        roof, extra, dom, oth = roofline_entries(contacts, n, prof["con_ms"] / prof["iters"],
                                                 prof["body_ms"] / prof["iters"], prof["iters"], active_contacts=active,
                                                 tier=prof.get("tier"), iterations=prof.get("solve_iters"),
                                                 drift_rows=prof.get("drift_rows", True))
        attach_traffic(roof, extra, dom, oth, n, args.buffer)
        measured_ceiling(roof, extra)

    # ---- a second, labelled figure: the same step from a RELAXED packing (what a running simulation sees) -----------
    relaxed = None
    if args.relaxed_steps > 0 and dist is None:
        stepper.restore(pristine)
        if args.reorder:
            stepper.reorder_bodies(cell_size=args.reorder_cell, lo=[0.0, 0.0, 0.0])
        # the relaxation pre-pass: full steps of the reference's (frictionless) path, untimed -- also for the friction
        # extension, whose timed steps then start from the same relaxed packing as the headline's
        mu_kept, stepper.friction = stepper.friction, None
        for _ in range(args.relaxed_steps):
            stepper.step(integrate=True, force_rebuild=True)
        stepper.friction = mu_kept
        pristine_relaxed = stepper.snapshot()

        def relaxed_step():
            stepper.restore(pristine_relaxed)
            stepper.links.invalidate()
            stepper.profile_next = False
            return stepper.step(integrate=True, force_rebuild=True)

        relaxed_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rstats = [relaxed_step() for _ in range(args.steps)]
        torch.cuda.synchronize()
        r_el = time.perf_counter() - t1
        relaxed = {"what": "the same step (reorder excluded: bodies stay Z-ordered) from the packing reached after %d "
                           "full steps of this path from the raw input; NOT the headline value" % args.relaxed_steps,
                   "timesteps_per_sec": round(args.steps / r_el, 4), "ms_per_step": round(1e3 * r_el / args.steps, 3),
                   "contacts": rstats[-1].num_contacts, "bbpgd_iters_per_step": [s.num_iters for s in rstats],
                   "converged": [bool(s.converged) for s in rstats]}
        # ... and a short trajectory from there as a simulation runs it: consecutive steps, the neighbour list rebuilt
        # only when the rebuild rule asks for it (GenNeighborLinkers.hpp:603-615: a body moved more than half the buffer)
        stepper.restore(pristine_relaxed)
        stepper.links.invalidate()
        stepper.step(integrate=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tstats = [stepper.step(integrate=True) for _ in range(8)]
        torch.cuda.synchronize()
        t_el = time.perf_counter() - t2
        relaxed["trajectory"] = {"what": "8 consecutive steps from the relaxed packing, neighbour list reused until the "
                                         "rebuild rule fires", "ms_per_step": round(1e3 * t_el / 8, 3),
                                 "timesteps_per_sec": round(8 / t_el, 4),
                                 "rebuilds": int(sum(1 for s in tstats if s.rebuilt)),
                                 "bbpgd_iters_per_step": [s.num_iters for s in tstats],
                                 "converged": [bool(s.converged) for s in tstats]}

    # ---- CPU baseline: the oracle (CPU restatement of the reference path) on this box's host cores, rank 0 ---------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and args.friction is None:
        cpu = cpu_baseline(b, stepper, args, int(np.mean(iters)))

    if rank == 0:
        out = {
            "metric": METRIC,
            "value": round(value, 4), "unit": "timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2]: %.3gM spherocylinders r=0.5 L=2 at 40%% volume fraction, random packing, "
                                   "AABB+%.2g neighbour list, frictionless LCP tol %.0e" % (n / 1e6, args.buffer, args.tol),
                       "bodies_per_gpu": n, "bodies_total": n, "contacts_per_gpu": contacts, "contacts_total": contacts,
                       "bbpgd_iters_per_step": iters, "converged": [bool(s.converged) for s in stats],
                       "parallelism": "single GPU (N > 1: the same system, hilbert domain decomposition with RCCL halo)"},
            "contact_pairs_per_sec": round(world * contacts * args.steps / elapsed, 1),
            "bbpgd_iterations_per_sec": round(world * sum(iters) / elapsed, 1),
            # contacts x iterations per second: the rate that stays comparable when the iteration count changes
            "constraint_updates_per_sec": round(world * contacts * sum(iters) / elapsed, 1),
            "roofline": roof, "cpu_baseline": cpu,
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "relaxed_packing": relaxed,
            # cold tier of the solve (time only, same iterates): share of contacts swept in full over the tiered iterations
            "cold_tier": prof.get("tier"),
        }
        if args.friction is not None:  # never the default line: an extension without a reference to pin it on
            out["metric"] = "timesteps/sec, 10^6 spherocylinders per GPU, FRICTIONAL cone-complementarity contact (build extension)"
            out["config"]["workload"] = out["config"]["workload"].replace(
                "frictionless LCP", "EXTENSION (parity unpinned): Coulomb friction mu = %g as a cone complementarity "
                "problem (%s)," % (args.friction, args.friction_method.upper()))
        out.update(extra)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main_mixed(args, ops, pipeline, synth, dev):
    """BASELINE configs[4] on one GPU: mixed spheres / spherocylinders / ellipsoids ("divergent distance kernels").  The
    ellipsoid classes run the reference's 9-start L-BFGS shared-normal distance (compute bound, no HBM roofline); the
    roofline object is that of the BBPGD sweeps with explicit lever arms (KIN_RIGID)."""
    n = args.bodies
    b = synth.mixed_bodies(n, volume_fraction=args.mixed_phi, seed=1234)
    ops.contact_mixed_set_contraction(args.ellipsoid_fma)
    cfg = ops.PGDConfig(max_iters=args.max_iters, tol=args.tol)
    st = pipeline.ContactStepper("mixed", dev(b["center"]), None, dev(b["quat"]), search_buffer=args.buffer, cfg=cfg,
                                 kinds=dev(b["kind"]), shape=dev(b["shape"]))
    if args.xcd_tile >= 0 or args.lanes_per_body > 0:
        st.work_mapping = (args.xcd_tile, args.lanes_per_body)
    pristine = st.snapshot()
    prof = dict(body_ms=0.0, con_ms=0.0, iters=0)

    def one_step(timed_kernels, timed_stages=False):
        st.restore(pristine)
        if args.reorder:
            st.reorder_bodies(cell_size=args.reorder_cell, lo=[0.0, 0.0, 0.0])
        st.profile_next = timed_kernels
        s = st.step(integrate=True, force_rebuild=True, timed=timed_stages)
        if timed_kernels:
            a, c, k = st.op.get_profile()
            prof["body_ms"] += a
            prof["con_ms"] += c
            prof["iters"] += k
            prof["tier"] = st.op.tier_stats()
            prof["solve_iters"] = s.num_iters
        return s

    for _ in range(args.warmup):
        one_step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = [one_step(True) for _ in range(args.steps)]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    active = None   # contacts the body sweep's masks flag at the end of the solve (as in the headline line)
    if st.lam is not None and getattr(st, "grad", None) is not None:
        active = int((~((st.lam == 0) & (st.grad >= 0) & torch.isfinite(st.grad))).sum().item())
    stage_ms = one_step(False, timed_stages=True).timings_ms
    contacts, iters = stats[-1].num_contacts, [s.num_iters for s in stats]
    # The narrow phase of this config is NOT bandwidth bound: the E-E class runs the reference's 9-start L-BFGS
    # shared-normal minimisation, ~10^3 objective evaluations per pair (S-E and R-E, build extensions without a reference
    # routine, are closed-form since round 3 and report no evaluations).  Its roofline is the fp64 vector rate:
    # algorithmic flops = evaluations (counted by the kernels) x FLOPS_PER_EVALUATION, over the stage time -- the whole
    # stage, class binning and the closed-form classes included.
    evals = ops.contact_mixed_last_evaluations()
    narrow_s = 1e-3 * stage_ms["narrowphase"]
    flops = FLOPS_PER_EVALUATION * float(sum(evals.values()))
    ell_roof = {"bound": "fp64-vector", "kernel": "k_contact_class_lockstep<E-E> (the minimisation class; S-E / R-E closed-form)",
                "achieved": round(flops / narrow_s / 1e12, 3), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(flops / narrow_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 4), "traffic": None,
                "objective_evaluations": evals, "flops_per_evaluation": FLOPS_PER_EVALUATION,
                "stage_ms": round(stage_ms["narrowphase"], 3),
                "arithmetic": "contracted (fused multiply-adds): LABELLED build option, results at the reference's "
                              "1e-4" if args.ellipsoid_fma else "no contraction: bit-identical to the CPU oracle",
                "note": "peak counts an FMA as two flops; the default build never fuses a*b+c (bit parity with the "
                        "scalar reference order), so 0.5 x peak is what it could reach"}
    # both arithmetics side by side on this step's neighbour list: the narrow phase alone, HIP-event timed, and how far
    # the contracted build's separations are from the default's (the reference's bar: 1e-4, UnitTestEllipsoidEllipsoid.cpp:53)
    both = {}
    seps = {}
    for name, on in (("default", False), ("contracted", True)):
        ops.contact_mixed_set_contraction(on)
        ops.contact_mixed(st.links.pairs, st.kinds, st.center, st.quat, st.shape)   # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = ops.contact_mixed(st.links.pairs, st.kinds, st.center, st.quat, st.shape)
        e1.record()
        torch.cuda.synchronize()
        ev = ops.contact_mixed_last_evaluations()
        ms = e0.elapsed_time(e1)
        fl = FLOPS_PER_EVALUATION * float(sum(ev.values()))
        both[name] = {"narrowphase_ms": round(ms, 3), "objective_evaluations": sum(ev.values()),
                      "achieved_tflops": round(fl / (1e-3 * ms) / 1e12, 3),
                      "frac_of_fp64_vector_peak": round(fl / (1e-3 * ms) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 4)}
        seps[name] = o["sep"]
    ops.contact_mixed_set_contraction(args.ellipsoid_fma)
    dsep = (seps["default"] - seps["contracted"]).abs()
    both["separations_within_1e-4"] = round(float((dsep <= 1e-4).double().mean().item()), 6)
    both["separations_bit_identical"] = round(float((dsep == 0).double().mean().item()), 6)
    ell_roof["arithmetics"] = both
    roof, extra = None, {}
    if prof["iters"] > 0:
        # the same accounting as the headline line (kernel_bytes: what the launch REQUIRES, with the activity masks and
        # the cold tier as measured in this run), for explicit lever arms
        roof, extra, dom, oth = roofline_entries(contacts, n, prof["con_ms"] / prof["iters"], prof["body_ms"] / prof["iters"],
                                                 prof["iters"], active_contacts=active, tier=prof.get("tier"),
                                                 iterations=prof.get("solve_iters"), kin="rigid")
        if args.mixed_phi == MIXED_PHI_DEFAULT:
            attach_traffic(roof, extra, dom, oth, n, args.buffer, name="traffic_mixed.json")
        measured_ceiling(roof, extra)
    out = {
        "metric": "timesteps/sec, 10^6 mixed sphere / spherocylinder / ellipsoid bodies, frictionless LCP contact (BBPGD)",
        "value": round(args.steps / elapsed, 4), "unit": "timesteps/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[4] on one GPU: %.3gM bodies, one third each spheres r=0.5, spherocylinders r=0.5 L=2, "
                               "ellipsoids (0.8, 0.5, 0.4), random orientations, %.0f%% volume fraction, AABB+%.2g neighbour "
                               "list, class-binned narrow phase (L-BFGS ellipsoid distances), frictionless LCP tol %.0e"
                               % (n / 1e6, 100 * args.mixed_phi, args.buffer, args.tol),
                   "bodies_per_gpu": n, "contacts_per_gpu": contacts, "bbpgd_iters_per_step": iters,
                   "converged": [bool(s.converged) for s in stats], "parallelism": "single GPU"},
        "contact_pairs_per_sec": round(contacts * args.steps / elapsed, 1),
        "bbpgd_iterations_per_sec": round(sum(iters) / elapsed, 1),
        "roofline": roof,
        "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline_mixed(b, st, pristine, args, int(np.mean(iters))),
        "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        "narrow_phase_roofline": ell_roof,
        "ellipsoid_aabb": "reference (centre -/+ q*radii, compute_aabb.hpp:82-103: not conservative for general orientations)",
    }
    out.update(extra)
    print(json.dumps(out))


def transport_ab(st, comm, dist, one_step, rank, world, steps=2):
    """UNTIMED, after the timed steps of a partitioned run: the same step with each combination of the two per-iteration
    transport mechanisms -- velocity halo through the IPC-mapped inboxes or through grouped ncclSend / ncclRecv, reduction
    records through the mailbox or through ncclAllGather -- so that the first run on a real multi-GPU node explains
    itself: microseconds per BBPGD iteration, where they go (HIP events of sampled iterations on every rank), and which
    mechanism was ACTUALLY active on every rank (reported by the library, not by the flags)."""
    import torch
    out = {"what": "untimed A/B after the timed steps: %d steps per variant (after one warm step that re-plans the "
                   "ghosts); us_per_iteration = max over ranks of the solve phase's wall time / iterations; per_rank = "
                   "HIP-event means over the sampled iterations of that rank (body sweep, halo post, interior + boundary "
                   "constraint sweeps, wait for the halo, record = reduction + exchange + finalize incl. the wait for "
                   "the slowest rank)" % steps, "variants": []}
    want_ipc, want_mbox = comm.halo_ipc, comm.mailbox
    for name, ipc, mbox in (("inboxes + mailbox", True, True), ("send/recv + mailbox", False, True),
                            ("inboxes + all-gather", True, False), ("send/recv + all-gather", False, False)):
        comm.set_halo_ipc(ipc)
        comm.set_mailbox(mbox)
        one_step(False)                    # (every step of this bench rebuilds: the ghost plan makes the switch effective)
        st.prof = dict(body_ms=0.0, con_ms=0.0, iters=0)
        torch.cuda.synchronize()
        dist.barrier()
        solve_ms, iters = 0.0, 0
        t0 = time.perf_counter()
        for _ in range(steps):
            s = one_step(True)
            solve_ms += st.phase_ms.get("solve", 0.0)
            iters += s["num_iters"]
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        k = max(1, st.prof["iters"])
        mine = {"rank": rank, "halo": st.prof.get("halo_path", "?"), "records": st.prof.get("record_path", "?"),
                "solve_us_per_iteration": round(1e3 * solve_ms / max(1, iters), 2),
                "body_us": round(1e3 * st.prof["body_ms"] / k, 2), "constraint_us": round(1e3 * st.prof["con_ms"] / k, 2),
                "halo_post_us": round(1e3 * st.prof.get("halo_post_ms", 0.0) / k, 2),
                "halo_wait_us": round(1e3 * st.prof.get("halo_wait_ms", 0.0) / k, 2),
                "record_us": round(1e3 * st.prof.get("record_ms", 0.0) / k, 2), "step_ms": round(1e3 * el / steps, 3)}
        every = [None] * world
        dist.all_gather_object(every, mine)
        out["variants"].append({"requested": name, "iterations_per_step": iters // steps,
                                "ms_per_step": max(e["step_ms"] for e in every),
                                "us_per_iteration": max(e["solve_us_per_iteration"] for e in every),
                                "halo_active": sorted({e["halo"] for e in every}),
                                "records_active": sorted({e["records"] for e in every}), "per_rank": every})
    comm.set_halo_ipc(want_ipc)
    comm.set_mailbox(want_mbox)
    return out


def main_distributed(args, rank, world, dist, ops, synth, dev):
    """N > 1: one Hilbert-partitioned system of world x bodies rods, RCCL halo (see module docstring)."""
    from mundy_amd import distributed as D
    strong = not args.weak
    n = args.bodies // world if strong else args.bodies
    n_total = n * world
    # every rank derives the same global order from the counter-based generator (no set-up communication)
    if args.mixed:   # BASELINE configs[4]: the mixed system over the ranks (every rank generates it, keeps its slice)
        whole = synth.mixed_bodies(n_total, volume_fraction=args.mixed_phi, seed=1234)
        centers, box = whole["center"], whole["box"]
    else:
        centers, box = synth.spherocylinder_centers(np.arange(n_total), n_total, seed=1234)
    order = D.hilbert_order(centers, 0.0, box, level=7)
    starts = D.partition_ranges(n_total, world)
    a, e = int(starts[rank]), int(starts[rank + 1])
    mine = order[a:e]
    del centers
    if args.mixed:
        b = {k: np.ascontiguousarray(whole[k][mine]) for k in ("center", "quat", "kind", "shape")}
        del whole
    else:
        b = synth.spherocylinders(len(mine), seed=1234, n_total=n_total, indices=mine)
    cfg = ops.PGDConfig(max_iters=args.max_iters, tol=args.tol)
    comm = D.Comm(mailbox=not args.no_mailbox, halo_ipc=not args.no_halo_ipc)
    if comm.transport != "rccl" and not args.allow_host_transport:
        # every rank takes this branch together (the transport is negotiated): a halo staged through host memory is not
        # a measurement of the xGMI path, so no line is printed
        if rank == 0:
            print("bench.py: the halo transport is %r, not RCCL -- refusing to print a scaling line "
                  "(--allow-host-transport overrides, for development only)" % comm.transport, file=sys.stderr, flush=True)
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(3)
    comm.self_check()  # pairwise messages + all-gather with known contents, before anything is timed
    if args.mixed:
        st = D.DistributedContactStepper(dev(b["center"]), dev(b["quat"]), None, None, a, comm=comm,
                                         search_buffer=args.buffer, cfg=cfg, poll_every=64, kind=dev(b["kind"]),
                                         shape=dev(b["shape"]), domain=(0.0, float(box)), curve_level=6)
    else:
        st = D.DistributedContactStepper(dev(b["center"]), dev(b["quat"]), dev(b["radius"]), dev(b["length"]), a,
                                         comm=comm, search_buffer=args.buffer, cfg=cfg, poll_every=64,
                                         domain=(0.0, float(box)), curve_level=6)
    # set-up, untimed: one solve on the equal-COUNT cut to measure the work per body, then the curve is re-cut at equal
    # WORK (1 + contacts per body) and the bodies move to their new owners (the reference rebalances with
    # stk::balance, NGPSpheresLCP.cpp:956)
    s0 = st.step(integrate=False)
    before = torch.zeros(world, dtype=torch.float64, device="cuda")
    before[rank] = s0["owned_contacts"]
    dist.all_reduce(before, op=dist.ReduceOp.SUM)
    imbalance_before = float(before.max().item() * world / max(1.0, before.sum().item()))
    moved = st.rebalance(recut=True)
    n = st.n
    pristine_c, pristine_q = st.center.clone(), st.quat.clone()

    def one_step(profile):
        st.center.copy_(pristine_c)
        st.quat.copy_(pristine_q)
        st.profile = profile
        return st.step(integrate=True)

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(False)
    sync()
    t0 = time.perf_counter()
    stats = [one_step(True) for _ in range(args.steps)]
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    tot = torch.tensor([stats[-1]["owned_contacts"], stats[-1]["ghosts"], stats[-1]["local_contacts"]],
                       dtype=torch.float64, device="cuda")
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    contacts_global, ghosts_global = int(tot[0].item()), int(tot[1].item())
    per_rank = torch.zeros(world, dtype=torch.float64, device="cuda")
    per_rank[rank] = stats[-1]["owned_contacts"]
    dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    per_rank_contacts = [int(v) for v in per_rank.tolist()]
    iters = [s["num_iters"] for s in stats]
    roof, extra = None, {}
    if st.prof["iters"] > 0:
        roof, extra, _, _ = roofline_entries(stats[-1]["local_contacts"], n, st.prof["con_ms"] / st.prof["iters"],
                                             st.prof["body_ms"] / st.prof["iters"], st.prof["iters"], ", rank 0",
                                             kin="rigid" if args.mixed else "rod")
    timed_prof = dict(st.prof)
    timed_phase_ms = dict(st.phase_ms)
    paths = [None] * world   # what every rank's solve actually used during the timed steps
    dist.all_gather_object(paths, (timed_prof.get("halo_path", "?"), timed_prof.get("record_path", "?")))

    def finish(ab):
        if rank == 0:
            out["transport_ab"] = ab
            print(json.dumps(out), flush=True)

    out = None
    if rank == 0:
        out = {
            "metric": ("timesteps/sec, 10^6 mixed sphere / spherocylinder / ellipsoid bodies, frictionless LCP contact (BBPGD)"
                       if args.mixed else METRIC) if strong else
            "timesteps/sec, 10^6 bodies PER GPU (one system of N x 10^6), frictionless LCP contact (BBPGD)",
            # strong: timesteps per second of the one fixed-size system; weak: world x (10^6-rod workloads per second)
            "value": round((1 if strong else world) * args.steps / elapsed, 4), "unit": "timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("configs[4]: %.3gM mixed bodies (one third each spheres r=0.5, spherocylinders r=0.5 "
                                    "L=2, ellipsoids (0.8, 0.5, 0.4)) at %.0f%% volume fraction, one system "
                                    % (n_total / 1e6, 100 * args.mixed_phi) if args.mixed else
                                    "configs[3]: %.3gM spherocylinders r=0.5 L=2 at 40%% volume fraction, one system "
                                    % (n_total / 1e6)) +
                                   "Hilbert-partitioned over %d GPUs, AABB+%.2g neighbour list, frictionless LCP tol %.0e"
                                   % (world, args.buffer, args.tol),
                       "bodies_per_gpu": n, "bodies_total": n_total, "contacts_total": contacts_global,
                       "ghost_bodies_total": ghosts_global, "bbpgd_iters_per_step": iters,
                       "owned_contacts_per_rank": per_rank_contacts,
                       "contact_imbalance_max_over_mean": round(max(per_rank_contacts) * world / max(1, sum(per_rank_contacts)), 4),
                       "contact_imbalance_equal_count_cut": round(imbalance_before, 4),
                       "bodies_moved_by_work_recut_rank0": moved["sent"],
                       "converged": [bool(s["converged"]) for s in stats],
                       "parallelism": "hilbert domain decomposition dd%d: ghost-body halo per rebuild; per BBPGD "
                                      "iteration ghost-velocity send/recv + 5-double record per rank (%s)"
                                      % (world, "mailbox: IPC-mapped device slots" if comm.mailbox else "ncclAllGather"),
                       "transport": comm.transport,
                       # the per-iteration 5-double record: slots in the ranks' device memory, or the transport's all-gather
                       "reduction_records": "mailbox" if comm.mailbox else "all-gather",
                       "velocity_halo": "IPC-mapped inboxes" if comm.halo_ipc_active() else "grouped send / recv",
                       # ... and what the library reports every rank's solves USED in the timed steps
                       "paths_active_per_rank": [{"halo": h, "records": r} for h, r in paths]},
            "contact_pairs_per_sec": round(contacts_global * args.steps / elapsed, 1),
            "bbpgd_iterations_per_sec": round(sum(iters) / elapsed, 1),
            # contacts x iterations per second over all ranks: separates the growth of the BBPGD iteration count with
            # the system size (algorithmic) from what the halo and the all-gather cost per iteration
            "constraint_updates_per_sec": round(contacts_global * sum(iters) / elapsed, 1),
            "halo_wait_ms_per_iteration": round(timed_prof.get("halo_wait_ms", 0.0) / max(1, timed_prof["iters"]), 4),
            "record_ms_per_iteration": round(timed_prof.get("record_ms", 0.0) / max(1, timed_prof["iters"]), 4),
            # host wall time per phase of the last step on rank 0 (each phase ends with a device sync)
            "stage_ms": {k: round(v, 3) for k, v in timed_phase_ms.items() if k != "start"},
            "roofline": roof, "cpu_baseline": None,
        }
        out.update(extra)
    # ---- untimed: the transport mechanisms side by side.  The line must come out whatever happens in here (this is the
    # first code that ever runs these mechanisms between real GPUs): an exception is recorded in its place, and a
    # watchdog prints the line without it if a rank gets stuck in a collective
    if world > 1 and not args.no_transport_ab:
        import threading

        def give_up():
            finish({"error": "the A/B did not finish within 300 s on rank %d: skipped" % rank})
            sys.stdout.flush()
            os._exit(0)

        dog = threading.Timer(300.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            ab = transport_ab(st, comm, dist, one_step, rank, world)
        except Exception as e:  # noqa: BLE001
            ab = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
        dog.cancel()
        finish(ab)
    else:
        finish(None)
    st.op.close()
    comm.close()   # the library's RCCL communicator goes before the launcher's process group
    dist.barrier()
    dist.destroy_process_group()


def host_core_share(omp_max):
    """Cores this job may actually use: the cgroup CPU quota when there is one (the GPU boxes give a 16-core share of a
    256-thread host; 128 OpenMP threads on that quota ran the baseline at half the speed of 16), else the affinity
    mask -- never more than OpenMP's own maximum."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(round(float(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, omp_max))


def cpu_baseline(b, stepper, args, gpu_iters):
    """Times the CPU oracle (kind "port") on a bounded sample of the same workload: the full AABB / neighbour search /
    narrow phase once at full size, and `--cpu-iters` BBPGD iterations of the full-size LCP on this job's host cores; the
    solve is extrapolated to the iteration count the GPU needed.  Reported, never the target."""
    import oracle
    oracle.build()
    threads = host_core_share(oracle.num_threads())
    oracle.set_num_threads(threads)
    t = {}
    t0 = time.perf_counter()
    aabb = oracle.compute_aabb_spherocylinders(b["center"], b["quat"], b["radius"], b["length"], fast=True)
    brad = oracle.bounding_radius_spherocylinders(b["radius"], b["length"])
    t["aabb"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    lo, hi, R = oracle.grow(aabb, brad, args.buffer)
    pairs = oracle.search(oracle.SEARCH_AABB, lo, hi, b["center"], R, fast=True)
    t["search"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    seg = oracle.spherocylinder_segments(b["center"], b["quat"], b["radius"], b["length"], fast=True)
    out = oracle.contact_spherocylinders(pairs, seg, b["center"], fast=True)
    t["narrow"] = time.perf_counter() - t0
    mt = stepper.mob_trans.cpu().numpy()
    mr = stepper.mob_rot.cpu().numpy()
    k = max(2, args.cpu_iters)
    t0 = time.perf_counter()
    oracle.solve_cqpp_contact(pairs, out["normal"], out["ra"], out["rb"], mt, mr, 5e-3, out["sep"],
                              np.zeros(len(pairs)), max_iters=k, tol=args.tol, threads=True, fast=True)
    t["solve_sample"] = time.perf_counter() - t0
    per_iter = t["solve_sample"] / (k + 1)  # k iterations + the initial operator apply
    step_s = t["aabb"] + t["search"] + t["narrow"] + per_iter * (gpu_iters + 1)
    return {"value": round(1.0 / step_s, 6), "unit": "timesteps/s", "cores": threads, "kind": "port",
            "sample": "full-size AABB (%.2fs, OpenMP), cell-list search (%.2fs, OpenMP), narrow phase (%.2fs, "
                      "OpenMP) once + %d BBPGD iterations on %d OpenMP threads (%.3fs/iter), solve extrapolated to the "
                      "GPU's %d iterations; contacts %d" % (t["aabb"], t["search"], t["narrow"], k, threads, per_iter,
                                                           gpu_iters, len(pairs))}


def cpu_baseline_mixed(b, st, pristine, args, gpu_iters):
    """The CPU oracle on a bounded sample of the mixed workload: full-size AABBs and cell-list search once; the narrow
    phase (the reference's 9-start L-BFGS for the ellipsoid classes, ~10^3 objective evaluations per pair) on every
    `stride`-th pair of the list, scaled by the stride; `--cpu-iters` BBPGD iterations of the full-size LCP (contact
    geometry taken from the GPU step of the same input -- the parity tests hold the two bit-identical), extrapolated to
    the GPU's iteration count."""
    import oracle
    oracle.build()
    threads = host_core_share(oracle.num_threads())
    oracle.set_num_threads(threads)
    t = {}
    t0 = time.perf_counter()
    aabb, brad = oracle.aabb_mixed(b["kind"], b["center"], b["quat"], b["shape"], fast=True)
    t["aabb"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    lo, hi, R = oracle.grow(aabb, brad, args.buffer)
    pairs = oracle.search(oracle.SEARCH_AABB, lo, hi, b["center"], R, fast=True)
    t["search"] = time.perf_counter() - t0
    stride = 16
    sample = np.ascontiguousarray(pairs[::stride])
    t0 = time.perf_counter()
    oracle.contact_mixed(sample, b["kind"], b["center"], b["quat"], b["shape"], fast=True)
    t["narrow_sample"] = time.perf_counter() - t0
    narrow = t["narrow_sample"] * len(pairs) / max(1, len(sample))
    st.restore(pristine)      # original body order: the GPU's pair list is the oracle's
    st.step(integrate=False, force_rebuild=True)
    gp = st.links.pairs.cpu().numpy()
    same = len(gp) == len(pairs) and bool(np.array_equal(gp, pairs))
    c = {k: v.cpu().numpy() for k, v in st.contacts.items() if v is not None}
    k = max(2, args.cpu_iters)
    t0 = time.perf_counter()
    oracle.solve_cqpp_contact(gp, c["normal"], c["ra"], c["rb"], st.mob_trans.cpu().numpy(), st.mob_rot.cpu().numpy(),
                              st.dt, c["sep"], np.zeros(len(gp)), max_iters=k, tol=args.tol, threads=True, fast=True)
    t["solve_sample"] = time.perf_counter() - t0
    per_iter = t["solve_sample"] / (k + 1)
    step_s = t["aabb"] + t["search"] + narrow + per_iter * (gpu_iters + 1)
    return {"value": round(1.0 / step_s, 6), "unit": "timesteps/s", "cores": threads, "kind": "port",
            "sample": "full-size AABB (%.2fs, OpenMP) and cell-list search (%.2fs, OpenMP) once; narrow phase on every "
                      "%dth pair (%d pairs, %.2fs on %d OpenMP threads; x%d = %.1fs); %d BBPGD iterations of the full LCP "
                      "(%.3fs/iter) extrapolated to the GPU's %d; contacts %d (pair list equal to the GPU's: %s)"
                      % (t["aabb"], t["search"], stride, len(sample), t["narrow_sample"], threads, stride, narrow, k,
                         per_iter, gpu_iters, len(pairs), same)}


if __name__ == "__main__":
    main()
