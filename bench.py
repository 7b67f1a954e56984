#!/usr/bin/env python3
"""Headline benchmark: latent-frames/sec of the latent-NeRF render path, forward + backward,
at 64x64x4 latents with a 128^3 occupancy grid (BASELINE.json metric / configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one optimisation step on one synthetic view per GPU, through the drop-in surface
(`NeRFNetwork.render()` -> `image.backward(gradient=g)`): HIP ray generation, AABB test,
occupancy-pruned march, hash-grid gather, sigma/latent MLP, compositing, their backward kernels,
the gradient all-reduce (N > 1, RCCL) and the fused Adam update of every parameter.  Inputs are
synthetic and resident in HBM before the timed region: table ~ N(0, 0.1),
nn.Linear-default MLP, analytic sphere occupancy (|x| < 0.5), camera r=1.25, theta=60 deg,
phi = 45 deg * rank, fovy 55 deg, upstream gradient g = randn * sqrt(a)(1-a), a = 0.5
(the SDS weighting form of the reference's src/stable_diffusion.py:320-321).

Prints ONE JSON line (rank 0).  `roofline` is the hash-grid gather (the kernel the metric names):
algorithmic bytes (1164 B/sample f32 table, 588 B/sample bf16) x samples / the
kernel's duration measured with HIP events inside the timed steps; `roofline.traffic` and `mfma.busy_frac` come
from the committed rocprofv3 --pmc passes (profiles/pmc_latest.json) and are null unless that file was collected on
THIS build of the library (build tags compared).  `mfma` prices the MLP kernels (the only MFMA users) against the
dense bf16 peak from their live HIP-event times.  The occupancy-grid refresh (H10, `update_extra_state()`) runs INSIDE
the timed region at the trainer's cadence (every `update_extra_interval` = 16 steps, eager launches between replays)
and is part of `value`; it writes a shadow copy of the occupancy state so that the analytic scene of the workload
stays pinned (`refresh` on the line says so and carries the refresh-free rate of the same run).  A run shorter than
100 steps times 5 regions of --steps steps each and reports the median region (`repeats`).  `trainer` is the product
entry point -- `Trainer.train()` with the seeded synthetic guidance, captured step, refreshes included -- on the same
configuration.  `f32` is the same step with the exact-f32 parity configuration.  `cpu_baseline` is the oracle
(oracle/nerf_oracle.py, pure PyTorch fp32) doing the same step on the host cores: the only use of `oracle/` here.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "latent-nerf-test_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense peaks of the same guide (bf16 MFMA; f32-input MFMA)
MLP_FLOP_PER_SAMPLE = (12928, 25856)  # forward, backward (32 -> 64 -> 64 -> 5)
# sample-buffer capacity of the bench: 1.5 x the ~430 k samples of its view, rounded up to 64 Ki -- what the renderer's
# own budget (NeRFRenderer.update_sample_budget) settles on; the worst case 4096 rays x 256 would be 1 Mi
BENCH_CAPACITY = 10 * 65536
H = W = 64
GRID = 128
FOVY = 55.0
# Learning rate of the timed steps.  The work of a step does not depend on it, but the STATE of the field does, and
# the kernels skip samples whose upstream gradient is exactly zero (rays past their termination point): with the
# trainer's 1e-3 and this bench's fixed synthetic upstream gradient the field degenerates within ~100 steps (98 % of
# the samples dead at step 230, tools/dead_fraction.py), which makes the steps cheaper and the number depend on
# --warmup/--steps.  1e-7 keeps the field at its random-init state (8 % dead samples) for any run length: every
# parameter still takes a real Adam step, the workload stays the one the config names.
LR = 1e-7


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--precision", default=os.environ.get("LNERF_BENCH_PRECISION", "bf16"), choices=["f32", "bf16"],
                    help="bf16 (BASELINE configs[1]): bf16 shadow table + bf16 features + bf16 MFMA MLP, f32 "
                         "master weights/accumulation/compositing; f32: f32 table + exact-f32 MFMA MLP")
    ap.add_argument("--table", default="auto", choices=["auto", "f32", "bf16"],
                    help="dtype of the hash table the gather reads (auto: follows --precision, the SURVEY §8(b) dtype policy; "
                         "bf16 = half-size shadow refreshed by the fused Adam pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=5, help="timed CPU frames (~2 s each on 16 cores; + 1 warm-up)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed companions (f32 step, occupancy refresh)")
    ap.add_argument("--gather-variant", type=int, default=0)
    ap.add_argument("--gridtype", default="auto", choices=["auto", "hash", "tiled", "blocked"],
                    help="layout of the levels larger than their table: auto (default) = what TrainConfig picks -- `blocked` "
                         "(4 x 2 x 2 vertex blocks hashed together, one 64-byte line of the bf16 table each) with the bf16 "
                         "table, `hash` (Instant-NGP's vertex hash) with the f32 table; `tiled` = the upstream encoder's other "
                         "layout.  The default run reports the OTHER of blocked / hash as the `layout_companion`")
    ap.add_argument("--fuse-table-update", default="auto", choices=["auto", "0", "1"],
                    help="hash-table Adam step applied inside the scatter's reduce pass (single GPU only; auto = on at N=1)")
    ap.add_argument("--jitter-rng", default="kernel", choices=["kernel", "torch"],
                    help="source of the march jitter: the library's counter-based generator or torch.rand(N) per step")
    ap.add_argument("--lr", type=float, default=None,
                    help="learning rate (default 1e-7: holds the field at its random-init state, see LR above)")
    ap.add_argument("--perturb", type=int, default=1, help="1 (training default): per-ray jitter of the march start")
    ap.add_argument("--breakdown", action="store_true", help="extra untimed pass: per-kernel event timings")
    ap.add_argument("--graph", type=int, default=1,
                    help="1: replay the captured hipGraph of the whole step (every --probe-every-th timed step still runs "
                         "eagerly so that the gather can be bracketed by HIP events); a capture failure is FATAL; "
                         "0: eager launches only")
    ap.add_argument("--probe-every", type=int, default=8)
    ap.add_argument("--grad-transport", default="auto", choices=["auto", "f32", "bf16"],
                    help="dtype of the table-gradient all-reduce at N > 1 (auto: follows --precision)")
    ap.add_argument("--prefetch-rays", type=int, default=0,
                    help="1: rays + march of step k+1 on a side stream while step k runs (two steps per captured graph). "
                         "Measured on MI355X / ROCm 7.2: 0.4725 vs 0.4721 ms per step -- the branch does overlap the gather, "
                         "but the cross-queue hand-offs at the fork and the join cost what it hides (DESIGN.md section 5) -- "
                         "hence off by default (profiles/r02_exp_ray_prefetch.jsonl)")
    ap.add_argument("--fragment-shadow", type=int, default=1,
                    help="1: the optimiser mirrors the MLP weights into their bf16 fragments (no per-step fragment build)")
    ap.add_argument("--tail", type=int, default=1,
                    help="1: the step's tail (MLP slab sum + Adam of the MLP's tensors + step-counter tick + clearing of the "
                         "scatter's level maxima) runs inside the scatter's pass 2 -- no launch behind it; 0: separate launches")
    ap.add_argument("--tune", default="", help="lnerf_set_tuning overrides for an experiment: key=value,key=value "
                    "(recorded in the output line; the default run sets none)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 step at N = 1: process group (RCCL, communicator of one rank) initialised before any "
                         "GPU call, bf16 gradient sink, pipelined per-group all-reduce, row-group Adam, graph A + eager "
                         "exchange -- the un-fused path the driver's 8-GPU run takes, measured on one card")
    ap.add_argument("--graph-collectives", default="auto", choices=["auto", "0", "1"],
                    help="N > 1 (or --force-dist) on RCCL: 1 = the exchange and the optimiser are captured into the step "
                         "graph (one graph launch per step and rank); 0 = graph A + eager exchange + eager optimiser; "
                         "auto (default) = captured IF a pre-flight passes on this very job: every rank runs a few steps "
                         "both ways in a supervised child process (time limit: a hang is a verdict, not a stall) and the "
                         "ranks agree on the outcome -- captured == eager bit for bit and replicas identical, on all of "
                         "them -- before the timed run picks its form (`preflight` on the output line)")
    ap.add_argument("--preflight", action="store_true", help=argparse.SUPPRESS)   # (the pre-flight child itself)
    ap.add_argument("--preflight-timeout", type=float, default=120.0,
                    help="seconds until a pre-flight child is killed and its verdict is 'no' (it takes 3-7 s on one rank)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 started without a launcher: seconds until the parent kills every rank")
    ap.add_argument("--refresh", type=int, default=1,
                    help="1: the occupancy refresh runs inside the timed region every update_extra_interval steps (shadow "
                         "state: the analytic scene stays pinned); 0: the refresh-free step only")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of --steps steps each (0 = auto: 5 below 100 steps, else 1); the median region is reported")
    ap.add_argument("--trainer-steps", type=int, default=200, help="steps of the `trainer` companion (0 = skip)")
    ap.add_argument("--shard-optimizer", type=int, default=0,
                    help="N > 1, bf16 on the wire, pipelined groups: row-sharded table optimiser -- reduce-scatter of the "
                         "gradient, the owning rank steps its 1/N of the rows, all-gather of the bf16 shadow")
    ap.add_argument("--views-per-rank", type=int, default=1,
                    help="views every rank renders per optimisation step, as ONE batch through the fused captured step "
                         "(render.batch_size of the reference's fork: src/latent_paint_mesh/configs/train_config.py:32); "
                         "`value` counts every view")
    ap.add_argument("--exchange-groups", default="auto",
                    help="N > 1, bf16 on the wire: level groups the table gradient is exchanged in, each group's all-reduce "
                         "launched behind its own sums (0 = one collective after the whole scatter).  auto (default): the "
                         "ranks TIME the step with 1, 2, 4 and 8 groups on the job's own links before the timed region "
                         "(a dozen replays each, maximum over ranks) and take the fastest -- more groups start the wire "
                         "earlier and cost ~20 us of launch tails each, so the best depth depends on which all-reduce RCCL "
                         "picks over xGMI (on one rank, where nothing travels: 1); `exchange_tuning` on the output line")
    return ap.parse_args()


def build(dev, precision, variant, rank, table="f32", jitter_rng="kernel", gridtype="hash", views=1):
    from src.latent_nerf.configs.render_config import RenderConfig
    from src.latent_nerf.models.network_grid import NeRFNetwork
    from src.latent_nerf.models.nerf_utils import intrinsics_from_fov, pose_from_angles

    torch.manual_seed(0)
    cfg = RenderConfig(grid_size=GRID, train_h=H, train_w=W, mlp_precision=precision, table_dtype=table,
                       gather_variant=variant, noise_seed=(0x5EED + rank) if jitter_rng == "kernel" else None,
                       max_samples=BENCH_CAPACITY * views, gridtype=gridtype)
    net = NeRFNetwork(cfg)
    net.encoder.embeddings.data.normal_(0, 0.1)
    net = net.to(dev).train()
    sphere_scene(net)
    # `views` views per rank and step, rendered as ONE batch (one march / gather / MLP / composite / scatter over all of
    # their rays and samples, one optimiser step): view v of rank r looks from phi = 45 deg * (r * views + v)
    pose = torch.stack([pose_from_angles(math.radians(60.0), math.radians(45.0 * (rank * views + v)), 1.25)
                        for v in range(views)]).to(dev)
    intr = intrinsics_from_fov(FOVY, H, W)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    bg = torch.rand(views * H * W, 4, generator=g).to(dev)
    grad = (torch.randn(views, H * W, 4, generator=g) * math.sqrt(0.5) * 0.5).to(dev)
    return net, pose, intr, bg, grad


def sphere_scene(net):
    """SURVEY.md section 8(d) synthetic occupancy: density 10 inside |x| < 0.5, 0 outside, packed at threshold 0.01
    (product helper: NeRFRenderer.seed_density_grid evaluates the function at the cell centres on the device)."""
    net.seed_density_grid(lambda x: (x.norm(dim=-1) < 0.5).float() * 10.0, thresh=0.01)


def make_step(net, pose, intr, bg, grad, opt, world, transport=torch.float32, perturb=True, exchange_groups=0,
              prefetch=False, shard=False):
    """Returns (eager_step, fwd_bwd, opt_step, sync).
    prefetch: the rays + occupancy march of step k+1 (NeRFRenderer.prepare_rays: they read neither the hash table
    nor the MLP) run on a side stream while step k is shaded and back-propagated; the two sample-buffer sets of the
    renderer alternate.  Every step still does one ray generation, one march, one shade, one backward, one update."""
    from src.latent_nerf.raymarching import raymarching as rm
    from src.latent_nerf.training.distributed import GradSync
    small = [p for p in net.parameters() if p is not net.encoder.embeddings]
    sync = GradSync([net.encoder.embeddings], small, transport=transport, shard_optimizer=shard)
    # N > 1 with bf16 on the wire: backward writes the wire buffer directly; with exchange_groups >= 1 it only bins the
    # scatter records and the exchange sums + sends one level group at a time (GradSync.allreduce_pipelined)
    sink = sync.attach_sink(net.encoder, pipeline_groups=exchange_groups)
    pipelined = sink is not None and sink.groups is not None
    state = {}

    side = torch.cuda.Stream() if prefetch else None

    def prepare(slot):
        rays_o, rays_d = rm.get_rays(pose, intr, H, W)
        return net.prepare_rays(rays_o, rays_d, bg_color=bg, perturb=perturb, slot=slot)

    def fwd_bwd(flip=True):
        if not prefetch:
            # (ray generation runs inside the march's count pass: NeRFRenderer.render(camera=...))
            out = net.render(None, None, camera=(pose, intr, H, W), bg_color=bg, perturb=perturb)
            opt.arm()                        # N = 1: the scatter applies the table's Adam step (no-op otherwise)
            out["image"].backward(gradient=grad)
            return out
        main = torch.cuda.current_stream()
        if "prep" not in state:
            state["prep"], state["slot"] = prepare(0), 0
        cur, slot = state["prep"], state["slot"]
        if flip:   # fork: the next step's march goes to the other buffer set on the side stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                nxt = prepare(1 - slot)
        out = net.render(None, None, prepared=cur)
        opt.arm()
        out["image"].backward(gradient=grad)
        if flip:
            main.wait_stream(side)   # join
            state["prep"], state["slot"] = nxt, 1 - slot
        else:      # eager probe step between graph replays: same buffer set again, after the backward that reads it
            state["prep"] = prepare(slot)
        return out

    def allreduce():
        if pipelined:
            state["ex"] = sync.allreduce_pipelined()
            state["ex"].finish_small()
        else:
            sync.allreduce(copy_back=False)  # no-op without an exchange; bf16 sums stay in the wire buffer

    def opt_step():
        if pipelined:   # the optimiser waits for a level group's all-reduce right before it steps those rows
            ex = state.pop("ex")
            opt.step(grad_scale=1.0 / world, grads=sync.reduced(), row_groups={net.encoder.embeddings: ex.table_groups})
            ex.finish_gathers()   # (--shard-optimizer: the owners' new shadow rows; no-op otherwise)
        else:
            opt.step(grad_scale=1.0 / world, grads=sync.reduced() if sync.active else None)

    def step():
        out = fwd_bwd(False) if prefetch else fwd_bwd()
        allreduce()
        opt_step()
        return out

    allreduce.gradsync = sync     # (the GradSync behind the exchange: gather_rows() of the row-sharded optimiser)
    return step, fwd_bwd, opt_step, allreduce


class KernelTimer:
    """HIP events around selected C-ABI calls (same stream the kernels are launched on)."""

    def __init__(self, names):
        self.names = set(names)
        self.pairs = {n: [] for n in names}
        self._open = {}

    def hook(self, name, when):
        if name not in self.names:
            return
        if when == "pre":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._open[name] = e
        else:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.pairs[name].append((self._open.pop(name), e))

    def mean_ms(self, name):
        p = self.pairs[name]
        return sum(a.elapsed_time(b) for a, b in p) / max(len(p), 1)


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block.  RCCL prints a version banner on STDOUT when its first
    communicator is created; rank 0's stdout must carry the ONE JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def cpu_baseline(frames):
    """The oracle's pure-PyTorch fp32 step (render fwd+bwd + Adam) on the host cores."""
    from oracle import nerf_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    log("cpu baseline on %d threads" % cores)
    torch.manual_seed(0)
    lv = O.make_grid_levels()
    table = (torch.randn(lv.n_rows, 2) * 0.1).requires_grad_()
    mp = {k: v.requires_grad_() for k, v in O.init_mlp_params().items()}
    grid = O.sphere_density_grid(G=GRID, radius=0.5)
    bits = O.packbits(grid.reshape(-1), 0.01)
    f = H / (2 * math.tan(math.radians(FOVY) / 2))
    c2w = O.pose_from_angles(math.radians(60.0), 0.0, 1.25)
    bg = torch.rand(H * W, 4)
    g = torch.randn(H * W, 4) * math.sqrt(0.5) * 0.5
    opt = torch.optim.Adam([{"params": [table], "lr": LR * 10}, {"params": list(mp.values()), "lr": LR}],
                           betas=(0.9, 0.99), eps=1e-15)
    times = []
    M = 0
    for i in range(frames + 1):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        ro, rd = O.get_rays(c2w, f, f, W / 2, H / 2, H, W)
        out = O.render_frame(ro[0], rd[0], table, mp, lv, bits, G=GRID, noises=torch.rand(H * W), bg_color=bg)
        out["image"].backward(g)
        opt.step()
        dt = time.perf_counter() - t0
        M = out["M"]
        log("cpu frame %d: %.2f s" % (i, dt))
        if i > 0:
            times.append(dt)
        if i == 0 and dt > 15.0:  # keep the default run bounded (~30 s of CPU work)
            frames = min(frames, 1)
        if i >= frames:
            break
    med = sorted(times)[len(times) // 2]
    return {"value": 1.0 / med, "unit": "latent-frames/sec", "cores": cores, "kind": "port",
            "sample": "%d timed frames (+1 warm-up) of the same 64x64x4 / 128^3 step, M=%d samples/frame, "
                      "oracle/nerf_oracle.py fp32 PyTorch, %d threads" % (len(times), M, cores)}


def load_pmc(build_tag):
    """profiles/pmc_latest.json (tools/run_pmc_all.sh) if it was collected on this very build of the library."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    if d.get("build") != build_tag:
        log("profiles/pmc_latest.json was collected on build %r, this is %r: counter-derived fields are null"
            % (d.get("build"), build_tag))
        return None
    return d


def companion_layout(dev, rank, precision, gridtype, steps=60, warmup=10):
    """The OTHER table layout through the same captured step, beside the headline: frames/s and the gather's kernel time.
    `blocked` (render.gridtype = "blocked": 4 x 2 x 2 vertex blocks in one 64-byte line of the bf16 table, 2.8 instead of
    4.25 lines per sample and level) is what the bf16 configuration trains with since round 4 (equal or better training
    error in the A/B of profiles/r04_ab_layout.jsonl); `hash` is Instant-NGP's vertex hash."""
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.training.graph_step import GraphedTrainStep
    from src.latent_nerf.training.optimizer import FusedAdam
    net, pose, intr, bg, grad = build(dev, precision, 0, rank, precision, gridtype=gridtype)
    opt = FusedAdam(net.get_params(LR), betas=(0.9, 0.99), eps=1e-15, encoder=net.encoder, capturable=True,
                    fuse_table_update=True, mlp=net)
    step, fwd_bwd, opt_step, sync = make_step(net, pose, intr, bg, grad, opt, 1)
    gstep = GraphedTrainStep(fwd_bwd, opt_step, list(net.parameters()), sync=sync, world=1, warmup=3,
                             stream=torch.cuda.current_stream())
    for _ in range(warmup):
        gstep()
    timer = KernelTimer(["lnerf_grid_encode_forward"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if i % 8 == 7:
            B.set_profile_hook(timer.hook)
            out = step()
            B.set_profile_hook(None)
        else:
            out = gstep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    M = int(out["counter"][0].item())
    g_ms = timer.mean_ms("lnerf_grid_encode_forward")
    bps = 16 * 8 * 2 * 2 + 12 + 32 * 2 if precision == "bf16" else 1164
    return {"value": steps / dt, "unit": "latent-frames/sec", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "samples_per_view": M, "gather_kernel_ms": g_ms, "gather_GBps": M * bps / (g_ms * 1e-3) / 1e9,
            "gather_frac_of_hbm_peak": M * bps / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "gridtype": gridtype,
            "what": "render.gridtype = %s: the same step on the other table layout (hash = Instant-NGP's vertex hash; blocked "
                    "= 4 x 2 x 2 vertex blocks hashed together, one 64-byte line of the bf16 table each)" % gridtype}


def companion_views(args, dev, rank, k=8, steps=30, warmup=6):
    """k views per rank and step as ONE batch through the same fused, captured step (`--views-per-rank k`): one march /
    gather / MLP / composite / scatter over the k views' rays and samples, the 318 MB table update, the step's tail and
    the tick paid once per STEP (BASELINE configs[3]'s 8 views per step on one GPU; render.batch_size of the reference's
    fork, src/latent_paint_mesh/configs/train_config.py:32)."""
    a = argparse.Namespace(**vars(args))
    a.views_per_rank, a.prefetch_rays, a.fuse_table_update = k, 0, "auto"
    S = build_step(a, dev, rank, 1, False)
    gstep = make_gstep(S, False, False, torch.cuda.current_stream())
    for _ in range(warmup):
        gstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = gstep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if int(out["counter"][2].item()) != 0:
        raise SystemExit("bench (views companion): rays did not fit the sample capacity")
    return {"value": k * steps / dt, "unit": "latent-frames/sec", "views_per_step": k, "ms_per_step": 1e3 * dt / steps,
            "ms_per_view": 1e3 * dt / steps / k, "steps": steps, "samples_per_step": int(out["counter"][0].item()),
            "what": "--views-per-rank %d: the views of a step rendered and back-propagated as one batch inside the fused, "
                    "captured step; one optimiser step per batch" % k}


def companion_f32(dev, rank, steps=40, warmup=6):
    """The exact-f32 parity configuration (f32 table, f32 features, exact-f32 MFMA MLP, 12-byte scatter records) through
    the same captured step: frames/s beside the headline (bf16) number."""
    from src.latent_nerf.training.graph_step import GraphedTrainStep
    from src.latent_nerf.training.optimizer import FusedAdam
    net, pose, intr, bg, grad = build(dev, "f32", 0, rank, "f32")
    opt = FusedAdam(net.get_params(LR), betas=(0.9, 0.99), eps=1e-15, encoder=net.encoder, capturable=True,
                    fuse_table_update=True)
    step, fwd_bwd, opt_step, sync = make_step(net, pose, intr, bg, grad, opt, 1)
    gstep = GraphedTrainStep(fwd_bwd, opt_step, list(net.parameters()), sync=sync, world=1, warmup=3,
                             stream=torch.cuda.current_stream())
    for _ in range(warmup):
        gstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = gstep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "latent-frames/sec", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "samples_per_view": int(out["counter"][0].item()),
            "what": "--precision f32 (f32 table + features, exact-f32 MFMA MLP, exact 12-byte scatter records): the "
                    "configuration the fp32-tolerance parity tests run"}


def trainer_companion(dev, steps, precision, fixed_pose=False):
    """The product entry point on the bench configuration: `Trainer(cfg).train()` (scripts/train_latent_nerf.py) with
    the seeded synthetic guidance, 64x64x4 / 128^3, bf16, one view per step -- captured step (graph F: render / eager
    guidance / graph B: backward + optimiser), a new random pose and field of view every step, an occupancy refresh
    every 16 steps, sparsity term on.  Same scene and learning rate as the headline (analytic sphere, lr 1e-7: the field
    stays at its random-init state).  Whole-loop wall clock, evaluation excluded."""
    import shutil
    import tempfile
    from src.latent_nerf.configs.train_config import TrainConfig, apply_overrides
    from src.latent_nerf.training.trainer import Trainer
    root = tempfile.mkdtemp(prefix="lnerf_bench_trainer_")
    try:
        warm = 40   # eager steps, captures and the first budget-driven re-capture happen here
        cfg = apply_overrides(TrainConfig(), {
            "log.exp_name": "bench", "log.exp_root": root, "render.train_h": H, "render.train_w": W,
            "render.grid_size": GRID, "render.eval_h": 8, "render.eval_w": 8, "log.eval_size": 1, "log.full_eval_size": 1,
            "log.save_interval": 10 ** 9, "log.quiet": True, "optim.lr": LR, "optim.fp16": precision == "bf16", "guide.text": "bench",
            "optim.iters": warm})
        if fixed_pose:   # the bench's own view (theta 60, phi 0, r 1.25, fovy 55) and sample capacity: same GPU work per
            cfg.render.train_pose = (60.0, 0.0, 1.25, FOVY)   # step as the headline + guidance + sparsity term + real refreshes
            cfg.render.max_samples = BENCH_CAPACITY
        tr = Trainer(cfg, device=dev)
        sphere_scene(tr.nerf)
        tr.nerf.iter_density = 16          # steady-state refreshes (G^3/4 random + G^3/4 occupied cells)
        tr.full_eval = lambda: None        # the loop only
        tr.train()
        torch.cuda.synchronize()
        c0 = dict(tr.graph_stats)
        t0 = time.perf_counter()
        tr.train(iters=warm + steps)
        host = time.perf_counter() - t0    # (train() returns with its stream's work enqueued, not finished)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        M = int(tr.nerf._march.counter[0].item())
        iv = cfg.render.update_extra_interval
        return {"value": steps / dt, "unit": "steps/sec (= latent-frames/sec at 1 view per step)", "steps": steps,
                "ms_per_step": 1e3 * dt / steps, "host_ms_per_step": 1e3 * host / steps,
                # (host_ms_per_step includes the wait at every refresh, where the sample budget is read back; this one is
                # the host time of the replayed steps alone: pose, upload, one graph launch)
                "host_ms_per_replayed_step": 1e3 * (tr.graph_stats["host_s"] - c0["host_s"])
                / max(tr.graph_stats["replayed_steps"] - c0["replayed_steps"], 1),
                "refreshes_in_region": len([k for k in range(warm + 1, warm + steps + 1) if (k - 1) % iv == 0]),
                "replayed_steps": tr.graph_stats["replayed_steps"] - c0["replayed_steps"],
                "eager_steps": tr.graph_stats["eager_steps"] - c0["eager_steps"],
                "captures_total": tr.graph_stats["captures"], "samples_per_view_last": M,
                "sample_capacity": tr.nerf._march.capacity, "whole_step_graph": bool(tr._whole),
                "what": ("Trainer.train() (src/latent_nerf/training/trainer.py), SyntheticGuidance, %s, sparsity term, "
                         "occupancy refresh every %d steps feeding the march%s"
                         % ("the bench's fixed view and sample capacity" if fixed_pose else "random poses / fov", iv,
                            "" if fixed_pose else ", sample budget from observed marches"))}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process -- which has not
    touched the GPU and never will -- starts the N ranks as children (one per GPU, a free rendezvous port), relays rank 0's
    JSON line, kills every rank when one fails or the time limit passes, and exits with the worst exit code.  Under
    `python -m torch.distributed.run ... bench.py --gpus N` the ranks arrive with WORLD_SIZE set and none of this runs."""
    from src.latent_nerf.training.launch import spawn_ranks
    n = args.gpus
    backend = os.environ.get("LNERF_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    if ndev == 0:
        raise SystemExit("bench: no GPU visible")
    if backend == "nccl" and ndev < n:
        raise SystemExit("bench: --gpus %d but %d GPU(s) visible (RCCL needs one GPU per rank; LNERF_DIST_BACKEND=gloo lets "
                         "ranks share a card for a functional rehearsal)" % (n, ndev))
    log("starting %d ranks (backend %s)" % (n, backend))
    rc, out = spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], n, timeout_s=args.launch_timeout,
                          log=log)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if rc == 0 and not lines:
        log("rank 0 printed no result line")
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


def dist_setup(args):
    """Rank / device / process group of this process, before anything touches the GPU."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one process per GPU; LNERF_DIST_BACKEND=gloo lets several ranks share one card for a functional rehearsal
    backend = os.environ.get("LNERF_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local_dev = local if backend == "nccl" else local % max(ndev, 1)
    dev = torch.device("cuda", local_dev)
    if args.force_dist:
        os.environ["LNERF_FORCE_DIST"] = "1"   # (GradSync / Trainer read it: exchange at any world size)
    dist_on = world > 1 or args.force_dist     # gradients are exchanged (collectives run; the table update is not fused)
    store = None
    if dist_on:
        from src.latent_nerf.training.launch import free_port, open_store
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:    # (only a one-rank --force-dist run gets here without a launcher)
            if world > 1:
                raise SystemExit("bench: WORLD_SIZE > 1 needs MASTER_PORT (a launcher sets it)")
            os.environ["MASTER_PORT"] = str(free_port())
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        store = open_store(rank, world)
    return rank, world, local_dev, dev, backend, dist_on, store


def join_group(store, rank, world, dev, backend):
    import torch.distributed as dist
    torch.cuda.set_device(dev)
    with stdout_to_stderr():   # (the communicator is created by the first collective: do one here)
        if backend == "nccl":
            dist.init_process_group("nccl", store=store, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, store=store, rank=rank, world_size=world)
        dist.barrier()
        torch.cuda.synchronize()


def decide_graph_collectives(args, store, rank, world, backend):
    """-> (capture the exchange?, verdict dict for the output line).  Runs BEFORE this process touches the GPU."""
    from src.latent_nerf.training.launch import agree, free_port, run_child
    env_sw = os.environ.get("LNERF_GRAPH_COLLECTIVES", "")
    if backend != "nccl" or not args.graph:
        return False, {"ran": False, "why": "collectives of backend %r are not captured" % backend}
    if args.graph_collectives != "auto" or env_sw in ("0", "1"):
        on = (args.graph_collectives == "1") if env_sw not in ("0", "1") else env_sw == "1"
        return on, {"ran": False, "why": "forced %s by %s" % ("on" if on else "off", "LNERF_GRAPH_COLLECTIVES"
                                                              if env_sw in ("0", "1") else "--graph-collectives")}
    # every rank starts ONE child that joins a process group of its own (a fresh port rank 0 publishes), runs the step
    # with the eager and with the captured exchange from the same seeded state and compares; a child that hangs is
    # killed at the time limit.  The ranks then read each other's exit codes: all take the same turn.
    if rank == 0:
        store.set("lnerf_preflight_port", str(free_port()))
    port = store.get("lnerf_preflight_port").decode()
    env = dict(os.environ, MASTER_PORT=port, LNERF_BENCH_PREFLIGHT="1")
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    argv = [sys.executable, os.path.abspath(__file__), "--preflight", "--gpus", str(world), "--precision", args.precision,
            "--table", args.table, "--grad-transport", args.grad_transport,
            "--exchange-groups", "4" if str(args.exchange_groups) == "auto" else str(args.exchange_groups),
            "--gridtype", args.gridtype, "--views-per-rank", str(args.views_per_rank), "--perturb", str(args.perturb),
            "--shard-optimizer", str(args.shard_optimizer)]
    if args.force_dist:
        argv.append("--force-dist")
    t0 = time.perf_counter()
    gave_up = lambda: store.check(["lnerf_preflight_abort"])
    rc = run_child(argv, env, args.preflight_timeout, poll=gave_up)
    if rc != 0:
        store.set("lnerf_preflight_abort", "1")      # the other ranks stop waiting for their children
    codes = agree(store, "lnerf_preflight_rc", rank, world, rc, timeout_s=args.preflight_timeout + 60)
    ok = all(c == "0" for c in codes)
    return ok, {"ran": True, "passed": ok, "exit_codes": [int(c) for c in codes], "seconds": time.perf_counter() - t0,
                "what": "per rank: one supervised child, 3 + 4 steps with the eager exchange and 3 eager + 4 REPLAYED steps "
                        "with exchange and optimiser captured, from the same seeded state: tables / MLP bit-identical "
                        "between the two forms and across ranks; 124 = killed at the time limit"}


def build_step(args, dev, rank, world, dist_on):
    """Model, optimiser and the step closures of this rank."""
    from src.latent_nerf.training.optimizer import FusedAdam
    table = args.precision if args.table == "auto" else args.table
    if args.gridtype == "auto":   # (TrainConfig's rule: the blocked layout goes with the bf16 table)
        args.gridtype = "blocked" if table == "bf16" else "hash"
    net, pose, intr, bg, grad = build(dev, args.precision, args.gather_variant, rank, table, args.jitter_rng, args.gridtype,
                                      views=args.views_per_rank)
    fuse = (not dist_on) if args.fuse_table_update == "auto" else (args.fuse_table_update == "1")
    if fuse and dist_on:
        raise SystemExit("--fuse-table-update 1 needs one rank without --force-dist (the gradient all-reduce sits between "
                         "backward and Adam)")
    opt = FusedAdam(net.get_params(LR if args.lr is None else args.lr), betas=(0.9, 0.99), eps=1e-15, encoder=net.encoder, capturable=True,
                    fuse_table_update=fuse, mlp=net if args.fragment_shadow else None,
                    tail=bool(args.tail) and fuse and bool(args.fragment_shadow))
    opt.grad_scale = 1.0 / (world * args.views_per_rank)
    tr = args.precision if args.grad_transport == "auto" else args.grad_transport
    groups = int(args.exchange_groups) if (dist_on and tr == "bf16") else 0
    prefetch = bool(args.prefetch_rays)
    if prefetch and (dist_on or args.views_per_rank != 1):
        raise SystemExit("--prefetch-rays 1 needs one rank without --force-dist and one view per rank")
    step, fwd_bwd, opt_step, sync = make_step(net, pose, intr, bg, grad, opt, world * args.views_per_rank,
                                              torch.bfloat16 if tr == "bf16" else torch.float32, bool(args.perturb), groups,
                                              prefetch, shard=bool(args.shard_optimizer) and bool(groups))
    return dict(net=net, opt=opt, step=step, fwd_bwd=fwd_bwd, opt_step=opt_step, sync=sync, table=table, fuse=fuse, tr=tr,
                groups=groups, prefetch=prefetch)


def make_gstep(S, dist_on, in_graph, stream):
    from src.latent_nerf.training.graph_step import GraphedTrainStep
    kw = dict(sync=S["sync"], world=2 if dist_on else 1, warmup=3, stream=stream, opt_in_graph=not S["groups"],
              steps_per_graph=2 if S["prefetch"] else 1)
    return GraphedTrainStep(S["fwd_bwd"], S["opt_step"], list(S["net"].parameters()), sync_in_graph=bool(in_graph), **kw)


def tune_exchange_groups(args, dev, rank, world, in_graph, stream, candidates=(1, 2, 4, 8), steps=12):
    """`--exchange-groups auto`: the step with each candidate depth of the pipelined exchange, on the job's own ranks and
    links: build, capture, `steps` replays between barriers, maximum over ranks; every rank gets the same timings (one
    all-reduce each) and so the same choice.  Outside the timed region.  -> (best, {depth: ms per step})."""
    import torch.distributed as dist
    timings = {}
    for G in candidates:
        a = argparse.Namespace(**vars(args))
        a.exchange_groups = G
        S = build_step(a, dev, rank, world, True)
        g = make_gstep(S, True, in_graph, stream)
        for _ in range(3):
            g()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        timings[G] = 1e3 * float(t.item()) / steps
        del S, g
        torch.cuda.empty_cache()
    best = min(timings, key=lambda k: (timings[k], k))
    return best, timings


def preflight_main(args):
    """The pre-flight child of one rank (decide_graph_collectives): exit code 0 = the captured exchange reproduces the
    eager one bit for bit on this rank and the replicas agree; 3 = it does not; anything else = it failed to run."""
    import hashlib
    import torch.distributed as dist
    rank, world, _local, dev, backend, dist_on, store = dist_setup(args)
    join_group(store, rank, world, dev, backend)
    from src.latent_nerf.raymarching import backend as B
    B.get_lib()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    def digest(S):
        # (row-sharded optimiser: every rank steps only the rows it owns -- the f32 master is whole again after the gather,
        # a collective every rank reaches here; the bf16 shadow, what renders, is hashed as well)
        if args.shard_optimizer and S["groups"]:
            S["sync"].gradsync.gather_rows([S["net"].encoder.embeddings.data])
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for p in S["net"].parameters():
            h.update(p.detach().contiguous().cpu().numpy().tobytes())
        sh = S["net"].encoder.shadow()
        if sh is not None:
            h.update(sh.detach().contiguous().view(torch.int16).cpu().numpy().tobytes())
        return h.hexdigest()

    K = 4
    A = build_step(args, dev, rank, world, True)
    for _ in range(3 + K):            # (GraphedTrainStep below runs 3 eager steps before it captures)
        A["step"]()
    da = digest(A)
    del A
    Bs = build_step(args, dev, rank, world, True)
    g = make_gstep(Bs, True, True, stream)
    for _ in range(K):
        g()
    db = digest(Bs)
    same = torch.tensor([1 if da == db else 0], device=dev, dtype=torch.int32)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    chk = torch.frombuffer(bytearray(bytes.fromhex(db)), dtype=torch.uint8).to(dev)
    allc = [torch.empty_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    replicas = all(torch.equal(allc[0], c) for c in allc)
    ok = bool(int(same.item())) and replicas
    log("pre-flight rank %d: captured %s eager, replicas %s" % (rank, "==" if da == db else "!=",
                                                                  "identical" if replicas else "DIFFER"))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 3


def main():
    args = parse()
    if args.preflight:
        raise SystemExit(preflight_main(args))
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1") or "1") == 1:
        # no launcher started N ranks (WORLD_SIZE unset -- or a stray WORLD_SIZE=1 in the environment): start them here
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            os.environ.pop(k, None)
        raise SystemExit(self_launch(args))
    import torch.distributed as dist
    rank, world, _local, dev, backend, dist_on, store = dist_setup(args)
    in_graph, preflight = False, None
    if dist_on:
        in_graph, preflight = decide_graph_collectives(args, store, rank, world, backend)   # (no GPU call before this)
        join_group(store, rank, world, dev, backend)
    torch.cuda.set_device(dev)

    from src.latent_nerf.raymarching import backend as B
    B.get_lib()  # no fallback: raise here if the HIP library is missing
    tuned = {}
    for kv in [t for t in args.tune.split(",") if t]:
        k, v = kv.split("=")
        B.call("lnerf_set_tuning", k.encode(), int(v))
        tuned[k] = int(v)
    # everything (eager steps, graph capture, replays, collectives) runs on one non-default stream
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    exchange_tuning = None
    if str(args.exchange_groups) == "auto":
        tr0 = args.precision if args.grad_transport == "auto" else args.grad_transport
        if dist_on and tr0 == "bf16" and args.graph:
            best, timings = tune_exchange_groups(args, dev, rank, world, in_graph, main_stream)
            exchange_tuning = {"chosen": best, "ms_per_step": {str(k): round(v, 4) for k, v in timings.items()},
                               "what": "refresh-free step with 1 / 2 / 4 / 8 level groups, 12 replays each on this job's "
                                       "ranks and links, maximum over ranks; the fastest is what the timed region runs"}
            log("exchange groups: %s -> %d" % (exchange_tuning["ms_per_step"], best))
            args.exchange_groups = best
        else:
            args.exchange_groups = 4
    S = build_step(args, dev, rank, world, dist_on)
    net, opt, step, sync = S["net"], S["opt"], S["step"], S["sync"]
    table, fuse, tr, groups, prefetch = S["table"], S["fuse"], S["tr"], S["groups"], S["prefetch"]
    inline_tail = bool(opt.fused is not None and opt.fused.inline_tail)   # pass 2 of the scatter closes the step itself
    scatter_call = ("lnerf_grid_encode_backward_adam_tail" if inline_tail else "lnerf_grid_encode_backward_adam" if fuse else
                    "lnerf_grid_scatter_bin" if groups else
                    "lnerf_grid_encode_backward_bf16" if (dist_on and tr == "bf16") else "lnerf_grid_encode_backward")

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    log("model built on %s (rank %d/%d%s)" % (dev, rank, world, ", forced exchange" if args.force_dist else ""))
    launch = "eager"
    gstep = None
    if args.graph:
        # (a capture failure raises: the line must not silently describe eager launches; use --graph 0 for those)
        launch = "hipgraph"
        if in_graph:
            # the pre-flight passed on every rank (or the operator forced the form): a failure here is fatal on every
            # rank alike -- no per-rank fallback that could leave ranks on different paths
            gstep = make_gstep(S, dist_on, True, main_stream)
            launch = "hipgraph (exchange + optimiser captured)"
        else:
            gstep = make_gstep(S, dist_on, False, main_stream)
            if dist_on:
                launch = "hipgraph (render + backward; exchange and optimiser eager)"
    emb0 = net.encoder.embeddings.detach().clone()
    spg = gstep.steps_per_call if gstep is not None else 1
    # occupancy refresh (H10) at the trainer's cadence, inside the timed region: update_extra_state() in its
    # steady-state form, eager launches between replays, on a SHADOW copy of the occupancy state (the analytic scene
    # the workload is defined on stays what the march reads; the refresh does all of its work)
    iv = int(net.cfg.update_extra_interval)
    do_refresh = bool(args.refresh) and net.cuda_ray
    net.iter_density = max(net.iter_density, 16)
    refresh_events = []

    def refresh():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        with net.shadow_extra_state():
            net.update_extra_state()
        b.record()
        refresh_events.append((a, b))

    i = 0
    while i < args.warmup:   # eager and replayed steps alternate; never more than --warmup steps
        if gstep is not None and (i // spg) % 2 and i + spg <= args.warmup:
            gstep()
            i += spg
        else:
            step()
            i += 1
    if do_refresh:
        refresh()
        refresh_events.clear()
    torch.cuda.synchronize()
    log("warm-up done (%s)" % launch)
    timer = KernelTimer(["lnerf_grid_encode_forward", scatter_call, "lnerf_mlp_forward", "lnerf_mlp_backward",
                         "lnerf_grid_scatter_reduce_bf16"])
    pe = max(1, args.probe_every)
    state = {"done": 0, "n_probe": 0}   # steps done over all regions (the refresh cadence runs across regions)

    def region(steps):
        """EXACTLY `steps` steps between barrier + synchronize on both sides.  Returns (elapsed, host enqueue time,
        last output, refreshes)."""
        barrier()
        t0 = time.perf_counter()
        n_probe0, n_ref = state["n_probe"], 0
        # eager probe steps: every --probe-every-th step; a run shorter than that still gets one (its last step), so that
        # the roofline object can always be measured live, whatever --steps the caller picks
        i, since, out = 0, 0, None   # steps done in this region; steps since the last probe
        while i < steps:
            if do_refresh and state["done"] % iv == 0:
                refresh()
                n_ref += 1
            left = steps - i
            probed = state["n_probe"] > n_probe0
            # (a replay of `spg` steps must not jump over a refresh point)
            room = iv - state["done"] % iv if do_refresh else left
            if gstep is None or since >= pe - 1 or (not probed and left == 1):
                B.set_profile_hook(timer.hook)   # eager step: the gather is bracketed by HIP events on its stream
                out = step()
                B.set_profile_hook(None)
                state["n_probe"] += 1
                n, since = 1, 0
            elif left < spg or room < spg or (not probed and left <= spg):
                out = step()                     # a remainder shorter than one replay
                n, since = 1, since + 1
            else:
                out = gstep()
                n, since = spg, since + spg
            i += n
            state["done"] += n
        host_enqueue = time.perf_counter() - t0   # host time to enqueue the steps (GPU runs behind)
        barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        if dist_on:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), host_enqueue, out, n_ref

    repeats = args.repeats if args.repeats > 0 else (5 if args.steps < 100 else 1)
    regions = [region(args.steps) for _ in range(repeats)]
    order = sorted(range(repeats), key=lambda k: regions[k][0])
    elapsed, host_enqueue, out, n_ref = regions[order[repeats // 2]]     # the median region
    n_probe = state["n_probe"]
    M = int(out["counter"][0].item())
    if int(out["counter"][2].item()) != 0:
        raise SystemExit("bench: %d rays did not fit the sample capacity %d" % (int(out["counter"][2].item()), net.cfg.max_samples))
    if rank == 0:
        log("timed region: %.4f s for %d steps (median of %d: %s)" % (elapsed, args.steps, repeats,
                                                                   ", ".join("%.4f" % r[0] for r in regions)))
    # outside the timed region: the run must have trained, not diverged (fail loudly rather than report a number)
    emb = net.encoder.embeddings.detach()
    if not bool(torch.isfinite(emb).all()) or not all(bool(torch.isfinite(p.detach()).all()) for p in net.parameters()):
        raise SystemExit("bench: non-finite parameters after the timed steps")
    if dist_on and args.shard_optimizer and groups:   # the owners' rows of the f32 master, whole again for the checks
        sync.gradsync.gather_rows([net.encoder.embeddings.data])
    if world > 1:  # data parallel: every rank applied the same update, the replicas must still be bit-identical
        sh = net.encoder.shadow()
        chk = torch.stack([emb.double().sum(), emb.double().abs().sum(), net.w2.detach().double().sum(),
                           (sh.double().sum() if sh is not None else emb.double().sum())])
        allc = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        if not all(torch.equal(allc[0], c) for c in allc):
            raise SystemExit("bench: the replicas diverged (rank checksums differ)")
    if float((emb - emb0).abs().max()) == 0.0 and (args.lr is None or args.lr != 0.0):
        raise SystemExit("bench: the hash table did not change during the timed steps (optimiser not applied?)")

    breakdown = None
    if args.breakdown and rank == 0:
        names = ["lnerf_get_rays", "lnerf_near_far_from_aabb", "lnerf_march_rays_train", "lnerf_march_rays_train_aabb",
                 "lnerf_march_rays_train_pose", "lnerf_grid_encode_forward",
                 "lnerf_mlp_forward", "lnerf_composite_rays_train_forward", "lnerf_composite_rays_train_backward",
                 "lnerf_mlp_backward", scatter_call, "lnerf_adam_step", "lnerf_adam_step_multi"]
        bt = KernelTimer(names)
        B.set_profile_hook(bt.hook)
        for _ in range(20):
            if gstep is not None:  # keep the GPU backlogged: an eager step alone is host-bound
                for _r in range(4):
                    gstep()
            step()
        torch.cuda.synchronize()
        B.set_profile_hook(None)
        breakdown = {n.replace("lnerf_", ""): round(bt.mean_ms(n) * (len(bt.pairs[n]) / 20.0), 4) for n in names}

    kv = int(args.views_per_rank)
    if rank == 0:
        # algorithmic bytes per sample of the gather: 16 levels x 8 vertices x 2 features x
        # sizeof(table entry) gathered + 12 B position + 16 x 2 x sizeof(feature) written
        bytes_per_sample = 16 * 8 * 2 * (2 if table == "bf16" else 4) + 12 + 32 * (2 if args.precision == "bf16" else 4)
        if n_probe == 0:
            raise SystemExit("no eager probe step ran inside the timed region (lower --probe-every)")
        g_ms = timer.mean_ms("lnerf_grid_encode_forward")
        s_ms = timer.mean_ms(scatter_call)
        if groups:   # pipelined exchange: pass 1 in the backward pass + pass 2 once per level group (collectives in between)
            s_ms += timer.mean_ms("lnerf_grid_scatter_reduce_bf16") * len(timer.pairs["lnerf_grid_scatter_reduce_bf16"]) \
                / max(n_probe, 1)
        achieved = M * bytes_per_sample / (g_ms * 1e-3) / 1e9
        scatter = M * 1164 / (s_ms * 1e-3) / 1e9 if s_ms > 0 else None
        build_tag = B.get_lib().lnerf_build_info().decode()
        pmc = load_pmc(build_tag)
        traffic = None   # HBM-side bytes per launch of the gather from the committed PMC passes (profiles/)
        if pmc:
            traffic = pmc.get("gather", {}).get("%s_table_%s_out" % (table, args.precision), {}).get("hbm_bytes_per_launch")
        # the MLP kernels (H7) are the only MFMA users: FLOP / live kernel time against the dense peak of the MFMA type
        f_ms, b_ms = timer.mean_ms("lnerf_mlp_forward"), timer.mean_ms("lnerf_mlp_backward")
        peak = MFMA_PEAK_TFLOPS[args.precision]
        mfma = {"kernels": "k_mlp_forward_%s / k_mlp_backward_%s (H7, 32 -> 64 -> 64 -> 5)" % ((args.precision,) * 2),
                "fwd_ms": f_ms, "bwd_ms": b_ms,
                "fwd_tflops": M * MLP_FLOP_PER_SAMPLE[0] / (f_ms * 1e-3) / 1e12,
                "bwd_tflops": M * MLP_FLOP_PER_SAMPLE[1] / (b_ms * 1e-3) / 1e12,
                "peak_tflops": peak, "busy_frac": None}
        mfma["frac_of_peak"] = {"fwd": mfma["fwd_tflops"] / peak, "bwd": mfma["bwd_tflops"] / peak}
        if pmc and args.precision == "bf16":   # SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), per kernel
            mfma["busy_frac"] = {k: pmc.get("mfma", {}).get(k, {}).get("busy_frac") for k in ("fwd", "bwd")}
        # refresh share of the median region (HIP events around each refresh; the GPU is backlogged, so the events see
        # GPU time): the refresh-free rate of the SAME run
        ref_ms = sorted(a.elapsed_time(b) for a, b in refresh_events)
        ref_med = ref_ms[len(ref_ms) // 2] if ref_ms else None
        res = {
            "metric": "latent-frames/sec (64x64x4, 128^3 grid), render forward+backward",
            "value": world * kv * args.steps / elapsed,
            "unit": "latent-frames/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "repeats": repeats,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "region_ms_per_step": [round(1e3 * r[0] / args.steps, 5) for r in regions],
            "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
            "launch": launch, "eager_probe_steps": n_probe,
            "refresh": ({"in_timed_region": True, "every_steps": iv, "refreshes_in_region": n_ref,
                         "ms_per_refresh": ref_med, "ms_per_step_amortised": (ref_med / iv) if ref_med else None,
                         "value_without_refresh": (world * kv * args.steps / (elapsed - 1e-3 * ref_med * n_ref))
                         if ref_med else None,
                         "what": "NeRFRenderer.update_extra_state() (H10), steady-state form, every %d steps between "
                                 "replays, INCLUDED in `value`; it writes a shadow copy of the occupancy state, so the "
                                 "analytic scene of the workload stays pinned" % iv}
                        if do_refresh else {"in_timed_region": False}),
            "ray_prefetch": ("rays + occupancy march of step k+1 on a side stream during step k "
                             "(NeRFRenderer.prepare_rays, two buffer sets, two steps per captured graph)") if prefetch else None,
            "gridtype": args.gridtype,
            "mlp_fragment_shadow": bool(opt.mlp is not None),
            "step_tail": ("inside the scatter's pass 2" if inline_tail else "one launch") if opt._tail is not None else False,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": "configs[1]: unconstrained latent-NeRF 64x64x4, 128^3 occupancy grid, hash grid "
                                   "L=16 F=2 T=2^19 (%s layout), %d view%s/GPU/step, fwd+bwd+grad all-reduce+Adam, occupancy refresh every "
                                   "%d steps" % (args.gridtype, kv, "" if kv == 1 else "s (one batch)", iv),
                       "rays_per_view": H * W, "samples_per_view": M // kv, "sample_capacity": int(net.cfg.max_samples),
                       "views_per_step": world * kv, "views_per_rank": kv,
                       "parallelism": "dp%d (%d view%s per GPU, RCCL all-reduce of gradients, %s on the wire%s)%s"
                                      % (world, kv, "" if kv == 1 else "s", tr,
                                         (", table in %d pipelined level groups" % groups if groups else "")
                                         + ("; row-sharded table optimiser (reduce-scatter / owner steps / all-gather of "
                                            "the shadow)" if (args.shard_optimizer and groups) else ""),
                                         "; --force-dist: the exchange path on ONE rank (communicator of size 1)"
                                         if args.force_dist else "")},
            "roofline": {"kernel": "k_grid_forward (hash-grid gather, H5)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "bytes_per_sample": bytes_per_sample, "samples_per_launch": M, "kernel_ms": g_ms},
            "mfma": mfma,
            "scatter": {"kernel": scatter_call.replace("lnerf_", "") + " (H6: two-pass bucketed scatter"
                                  + (" + fused Adam step of the table + the step's tail in the same launch)" if inline_tail else
                                     " + fused Adam step of the table)" if fuse else
                                     ", gradient written in the bf16 wire format)" if scatter_call.endswith("bf16") else
                                     " + grid_scatter_reduce_bf16 per level group: pipelined exchange)" if groups else ")"),
                        "algorithmic_GBps": scatter, "kernel_ms": s_ms},
        }
        if breakdown:
            res["kernel_ms_per_step"] = breakdown
        res["build"] = build_tag
        if preflight is not None:
            res["preflight"] = preflight
        if exchange_tuning is not None:
            res["exchange_tuning"] = exchange_tuning
        if tuned:
            res["tuning_overrides"] = tuned
        if not args.no_extras and not dist_on and args.gridtype in ("hash", "blocked") and kv == 1:
            if args.trainer_steps > 0:
                # the product loop twice: on the bench's own view (same GPU work per step as the headline: what the loop
                # itself costs) and on the training pose distribution (random radius / angles / field of view per step)
                res["trainer"] = trainer_companion(dev, args.trainer_steps, args.precision, fixed_pose=True)
                res["trainer"]["frac_of_value"] = res["trainer"]["value"] / res["value"]
                res["trainer_random_views"] = trainer_companion(dev, args.trainer_steps, args.precision)
            res["views8"] = companion_views(args, dev, rank, 8)
            res["layout_companion"] = companion_layout(dev, rank, args.precision,
                                                       "hash" if args.gridtype == "blocked" else "blocked")
            res["f32"] = companion_f32(dev, rank)
        if not args.no_cpu_baseline and not dist_on:
            res["cpu_baseline"] = cpu_baseline(args.cpu_frames)
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
