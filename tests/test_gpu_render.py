"""End-to-end parity of NeRFRenderer.render()/run_cuda() (HIP path through the C ABI) against the
oracle's render_frame on identical seeds: rendered latents and back-propagated gradients,
at a small configuration and at the BASELINE configuration (64x64x4, 128^3 grid, L=16, T=2^19)."""
import math

import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built_lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    return torch.device("cuda:0")


def _make(dev, G, HW, log2_T, base_res, seed=0, table_std=0.1, **cfg_kw):
    from src.latent_nerf.configs.render_config import RenderConfig
    from src.latent_nerf.models.network_grid import NeRFNetwork
    torch.manual_seed(seed)
    cfg = RenderConfig(grid_size=G, train_h=HW, train_w=HW, **cfg_kw)
    net = NeRFNetwork(cfg, base_resolution=base_res, log2_hashmap_size=log2_T)
    net.encoder.embeddings.data.normal_(0, table_std)
    net = net.to(dev)
    grid = O.sphere_density_grid(G=G, radius=0.5)
    net.density_grid.copy_(grid.to(dev))
    net.density_bitfield.copy_(O.packbits(grid.reshape(-1), 0.01).to(dev))
    lv = O.make_grid_levels(16, 2, base_res, 2048, log2_T)
    assert lv.offsets == net.encoder.levels.offsets
    params = {k: getattr(net, k).detach().cpu().clone().requires_grad_() for k in ("w1", "b1", "w2", "b2", "w3", "b3")}
    table = net.encoder.embeddings.detach().cpu().clone().requires_grad_()
    return net, cfg, lv, table, params, grid


def _rays(HW, theta=60.0, phi=0.0, radius=1.25):
    f = HW / (2 * math.tan(math.radians(55) / 2))
    c2w = O.pose_from_angles(math.radians(theta), math.radians(phi), radius)
    ro, rd = O.get_rays(c2w, f, f, HW / 2, HW / 2, HW, HW)
    return ro, rd  # [1, HW*HW, 3]


def _err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()), float(b.abs().max())


@pytest.mark.parametrize("size", ["small", "baseline"])
def test_render_train_matches_oracle(dev, size):
    if size == "small":
        G, HW, log2_T, base = 32, 16, 12, 16
    else:
        G, HW, log2_T, base = 128, 64, 19, 16
    net, cfg, lv, table, params, grid = _make(dev, G, HW, log2_T, base)
    net.train()
    ro, rd = _rays(HW)
    N = HW * HW
    torch.manual_seed(7)
    bg = torch.rand(N, 4)
    noises = torch.rand(N)
    g = torch.randn(1, N, 4) * math.sqrt(0.5) * 0.5  # SDS-like upstream gradient w = sqrt(a)(1-a), a = .5

    # HIP path through the renderer surface; same jitter noise as the oracle
    from src.latent_nerf.raymarching import raymarching as rm
    orig = rm.march_rays_train

    def patched(*a, **k):
        k["noises"] = noises.to(dev)
        return orig(*a, **k)

    rm.march_rays_train = patched
    try:
        out = net.render(ro.to(dev), rd.to(dev), bg_color=bg.to(dev), perturb=True)
    finally:
        rm.march_rays_train = orig
    assert out["image"].shape == (1, N, 4) and out["depth"].shape == (1, N)
    out["image"].backward(g.to(dev))

    ref = O.render_frame(ro[0], rd[0], table, params, lv, O.packbits(grid.reshape(-1), 0.01), G=G, noises=noises,
                         bg_color=bg)
    ref["image"].backward(g[0])
    M = ref["M"]
    assert int(out["counter"][0]) == M
    assert torch.equal(out["rays"].cpu(), ref["rays"])
    e, s = _err(out["image"][0], ref["image"])
    assert e <= 1e-4 * max(s, 1.0), ("image", e, s)
    e, s = _err(out["weights_sum"][0], ref["weights_sum"])
    assert e <= 1e-4, ("weights_sum", e)
    e, s = _err(out["depth"][0], ref["depth"])
    assert e <= 2e-4 * max(s, 1.0), ("depth", e, s)
    e, s = _err(out["sigmas"][:M], ref["sigmas"])
    assert e <= 1e-4 * s + 1e-5, ("sigmas", e, s)
    # gradients: hash table (float atomics) and MLP weights
    e, s = _err(net.encoder.embeddings.grad, table.grad)
    assert e <= 2e-3 * s + 1e-7, ("dtable", e, s)
    for k in ("w1", "b1", "w2", "b2", "w3", "b3"):
        e, s = _err(getattr(net, k).grad, params[k].grad)
        assert e <= 2e-3 * s + 1e-7, ("d" + k, e, s)


def test_render_train_bf16_matches_bf16_oracle(dev):
    """BASELINE config 2 precision (bf16 shadow table, bf16 features, bf16 MFMA MLP, 8-byte scatter records, f32
    compositing) against the oracle with the SAME roundings in both directions: bf16 operands of every forward product,
    and in the backward pass the pre-activation gradients rounded once to bf16 before they feed dX / dW / db
    (oracle _Bf16Linear), the table gradient's addends rounded to the 26-bit record format (oracle _CornerGather).  With
    the rounding points restated, what is left is f32 summation order: discrete outputs bit-exact; forward outputs within
    1e-4 of the value range for the composited image and opacity and 5e-4 of the largest density for the per-sample
    sigmas (measured on three seeds: 2e-6, 8e-7 and 3e-5 -- round 3 still allowed 2e-2 here); every gradient within
    1e-2 of its maximum."""
    G, HW, log2_T, base = 64, 32, 14, 16
    net, cfg, lv, table, params, grid = _make(dev, G, HW, log2_T, base, seed=2, mlp_precision="bf16", table_dtype="bf16")
    net.train()
    ro, rd = _rays(HW, 70.0, 30.0, 1.3)
    N = HW * HW
    torch.manual_seed(3)
    bg = torch.rand(N, 4)
    g = torch.randn(1, N, 4) * 0.35
    out = net.render(ro.to(dev), rd.to(dev), bg_color=bg.to(dev), perturb=False)
    out["image"].backward(g.to(dev))
    ref = O.render_frame(ro[0], rd[0], table, params, lv, O.packbits(grid.reshape(-1), 0.01), G=G, bg_color=bg,
                         bf16_mlp=True, bf16_table=True)
    ref["image"].backward(g[0])
    assert int(out["counter"][0]) == ref["M"] and torch.equal(out["rays"].cpu(), ref["rays"])
    e, s = _err(out["image"][0], ref["image"])
    assert e <= 1e-4 * max(s, 1.0), ("image", e, s)
    e, s = _err(out["weights_sum"][0], ref["weights_sum"])
    assert e <= 1e-4, ("weights_sum", e)
    e, s = _err(out["sigmas"][:ref["M"]], ref["sigmas"])
    assert e <= 5e-4 * s, ("sigmas", e, s)
    e, s = _err(net.encoder.embeddings.grad, table.grad)
    assert e <= 1e-2 * s, ("dtable", e, s)
    for k in ("w1", "b1", "w2", "b2", "w3", "b3"):
        e, s = _err(getattr(net, k).grad, params[k].grad)
        assert e <= 1e-2 * s + 1e-7, ("d" + k, e, s)


def test_render_contract_and_properties(dev):
    """The renderer->trainer contract of src/latent_paint/models/textured_mesh.py:181-220 /
    src/stable_diffusion.py:259: 'image' reshapes to [B,4,H,W] latents, and
    `pred.backward(gradient=grad)` (src/latent_paint_mesh/training/trainer.py:657-658) works."""
    G, HW = 64, 32
    net, cfg, lv, table, params, grid = _make(dev, G, HW, 14, 16, seed=3)
    net.train()
    ro, rd = _rays(HW, 75.0, 140.0, 1.4)
    out = net.render(ro.to(dev), rd.to(dev), bg_color=None, perturb=True)
    pred = out["image"].reshape(1, HW, HW, 4).permute(0, 3, 1, 2).contiguous()
    assert pred.shape == (1, 4, HW, HW)
    pred.backward(gradient=torch.randn_like(pred))
    assert net.encoder.embeddings.grad is not None and float(net.encoder.embeddings.grad.abs().sum()) > 0
    ws = out["weights_sum"]
    assert float(ws.min()) >= 0 and float(ws.max()) <= 1 + 1e-5
    # rays with no samples are exactly background (default bg = 1)
    empty = (out["rays"][:, 2] == 0)
    assert bool(empty.any())
    assert float((out["image"][0][empty] - 1.0).abs().max()) == 0.0
    # zero density (empty bitfield) -> pure background everywhere, no samples
    net.density_bitfield.zero_()
    out2 = net.render(ro.to(dev), rd.to(dev), bg_color=torch.full((4,), 0.25, device=dev))
    assert int(out2["counter"][0]) == 0
    assert float((out2["image"] - 0.25).abs().max()) == 0.0


def test_inference_path_matches_training_path(dev):
    """eval-mode run_cuda (march_rays / composite_rays / compact_rays loop) renders the same
    picture as the training kernels (no jitter); the lattices differ only by float rounding."""
    G, HW = 64, 32
    net, cfg, lv, table, params, grid = _make(dev, G, HW, 14, 16, seed=5)
    ro, rd = _rays(HW, 50.0, 200.0, 1.3)
    bg = torch.rand(HW * HW, 4, device=dev)
    net.train()
    with torch.no_grad():
        a = net.render(ro.to(dev), rd.to(dev), bg_color=bg, perturb=False)
    net.eval()
    with torch.no_grad():
        b = net.render(ro.to(dev), rd.to(dev), bg_color=bg, perturb=False)
    e, s = _err(b["image"], a["image"])
    assert e <= 5e-3 * max(s, 1.0), ("image", e, s)
    e, _ = _err(b["weights_sum"], a["weights_sum"])
    assert e <= 5e-3
    e, s = _err(b["depth"], a["depth"])
    assert e <= 5e-3 * max(s, 1.0)


def test_uniform_sampler_run_and_occupancy_refresh(dev):
    G, HW = 32, 16
    net, cfg, lv, table, params, grid = _make(dev, G, HW, 12, 16, seed=9, cuda_ray=False)
    net.train()
    ro, rd = _rays(HW)
    out = net.render(ro.to(dev), rd.to(dev), bg_color=None, num_steps=64)
    assert out["image"].shape == (1, HW * HW, 4) and bool(torch.isfinite(out["image"]).all())
    out["image"].sum().backward()
    assert float(net.encoder.embeddings.grad.abs().sum()) > 0
    # occupancy refresh: density blob (5 exp(-r^2/0.08)) dominates a near-zero table -> centre cells on
    net2, *_ = _make(dev, G, HW, 12, 16, seed=9, table_std=1e-4)
    net2.reset_extra_state()
    net2.density_thresh = 10.0
    for _ in range(3):
        net2.update_extra_state()
    assert net2.iter_density == 3
    dg = net2.density_grid[0].cpu()
    coords = O.morton3d_invert(torch.arange(G ** 3)).float()
    r = (((coords + 0.5) / G) * 2 - 1).norm(dim=-1)
    assert float(dg[r < 0.15].min()) > float(dg[r > 0.9].max())
    mean = float(net2.mean_density_dev)
    assert abs(mean - float(dg.clamp(min=0).mean())) < 1e-3 * max(mean, 1.0)
    bits = net2.density_bitfield.cpu()
    assert torch.equal(bits, O.packbits(dg, min(mean, 10.0)))


def test_graphed_step_matches_eager(dev):
    """hipGraph capture of render -> backward (training/graph_step.py): a replay produces the same image
    and the same gradients as the eager launches, and a captured optimiser step advances the device-side
    step counter.  Everything runs on one non-default stream (see GraphedTrainStep)."""
    from src.latent_nerf.training.graph_step import GraphedTrainStep
    from src.latent_nerf.training.optimizer import FusedAdam
    G, HW = 64, 32
    net, cfg, lv, table, params, grid = _make(dev, G, HW, 14, 16, seed=4)
    net.train()
    ro, rd = _rays(HW, 65.0, 10.0, 1.3)
    ro, rd = ro.to(dev), rd.to(dev)
    bg = torch.rand(HW * HW, 4, device=dev)
    g = torch.randn(1, HW * HW, 4, device=dev) * 0.3
    plist = list(net.parameters())

    def fwd_bwd():
        out = net.render(ro, rd, bg_color=bg, perturb=False)
        out["image"].backward(gradient=g)
        return out

    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ref = fwd_bwd()
        ref_img = ref["image"].detach().clone()
        ref_grads = [p.grad.detach().clone() for p in plist]
        for p in plist:
            p.grad = None
        gs = GraphedTrainStep(fwd_bwd, lambda: None, plist, world=1, warmup=2, stream=stream)
        for _ in range(2):
            out = gs()
        stream.synchronize()
        e, s = _err(out["image"], ref_img)
        assert e <= 1e-5 * max(s, 1.0), ("graph image", e, s)
        for got, want in zip(gs.static_grads, ref_grads):
            e, s = _err(got, want)
            assert e <= 1e-4 * s + 1e-8, ("graph grad", e, s)
        # with the optimiser inside the graph: parameters move on every replay, the step counter advances
        opt = FusedAdam(net.get_params(1e-3), encoder=net.encoder, capturable=True)
        gs2 = GraphedTrainStep(fwd_bwd, lambda: opt.step(), plist, world=1, warmup=1, stream=stream)
        before = net.w2.detach().clone()
        c0 = int(opt.step_dev[0].item())
        for _ in range(3):
            gs2()
        stream.synchronize()
        assert int(opt.step_dev[0].item()) == c0 + 3
        assert float((net.w2.detach() - before).abs().max()) > 0
        assert bool(torch.isfinite(net.encoder.embeddings).all())
    torch.cuda.synchronize()


def test_camera_form_generates_the_same_rays_and_render(dev):
    """render(camera=(pose, intrinsics, H, W)) -- rays generated inside the march's count pass
    (lnerf_march_rays_train_pose) -- is the render of get_rays' rays, bit for bit: rays, spans, samples, image, gradients."""
    from src.latent_nerf.raymarching import raymarching as rm
    G, HW = 64, 32
    net, cfg, lv, table, params, grid = _make(dev, G, HW, 14, 16, seed=12)
    net.train()
    f = HW / (2 * math.tan(math.radians(55) / 2))
    intr = (f, f, HW / 2, HW / 2)
    pose = O.pose_from_angles(math.radians(58.0), math.radians(33.0), 1.3).to(dev)
    bg = torch.rand(HW * HW, 4, device=dev)
    g = torch.randn(1, HW * HW, 4, device=dev) * 0.3
    plist = list(net.parameters())
    ro, rd = rm.get_rays(pose, intr, HW, HW)
    ref = net.render(ro, rd, bg_color=bg, perturb=False)
    ref["image"].backward(gradient=g)
    want = (ref["image"].detach().clone(), ref["rays"].clone(), int(ref["counter"][0]), ref["xyzs"][:int(ref["counter"][0])].clone(),
            [p.grad.detach().clone() for p in plist])
    for p in plist:
        p.grad = None
    out = net.render(None, None, camera=(pose, intr, HW, HW), bg_color=bg, perturb=False)
    out["image"].backward(gradient=g)
    assert torch.equal(net._ray_slots[0][1].view_as(ro), ro) and torch.equal(net._ray_slots[0][2].view_as(rd), rd)
    assert int(out["counter"][0]) == want[2] and torch.equal(out["rays"], want[1])
    assert torch.equal(out["xyzs"][:want[2]], want[3]) and torch.equal(out["image"], want[0])
    for p, w in zip(plist, want[4]):
        assert torch.equal(p.grad, w)
    # the same camera with the intrinsics in DEVICE memory (lnerf_march_rays_train_camera: what a captured trainer step
    # reads its view from): identical rays, spans, image and gradients
    for p in plist:
        p.grad = None
    intr_dev = torch.tensor([intr], dtype=torch.float32, device=dev)
    out = net.render(None, None, camera=(pose, intr_dev, HW, HW), bg_color=bg, perturb=False)
    out["image"].backward(gradient=g)
    assert torch.equal(net._ray_slots[0][1].view_as(ro), ro) and torch.equal(net._ray_slots[0][2].view_as(rd), rd)
    assert int(out["counter"][0]) == want[2] and torch.equal(out["rays"], want[1])
    assert torch.equal(out["image"], want[0])
    for p, w in zip(plist, want[4]):
        assert torch.equal(p.grad, w)
    # inference through the same keyword: plain ray generation first
    net.eval()
    with torch.no_grad():
        a = net.render(None, None, camera=(pose, intr, HW, HW), bg_color=bg)
        b = net.render(ro, rd, bg_color=bg)
        c = net.render(None, None, camera=(pose, intr_dev, HW, HW), bg_color=bg)
    assert torch.equal(a["image"], b["image"]) and torch.equal(c["image"], b["image"])


def test_prepared_rays_and_two_step_graph(dev):
    """prepare_rays() + render(prepared=...) is the same render as render(rays) (bitwise, both sample-buffer sets), and a
    captured graph of TWO steps whose marches alternate between the sets on a side stream (bench.py --prefetch-rays)
    replays: parameters finite and moving, device step counter + 2 per replay, jitter counter + 2 per replay."""
    from src.latent_nerf.training.graph_step import GraphedTrainStep
    from src.latent_nerf.training.optimizer import FusedAdam
    G, HW = 64, 32
    net, cfg, lv, table, params, grid = _make(dev, G, HW, 14, 16, seed=9, mlp_precision="bf16", table_dtype="bf16")
    net.train()
    ro, rd = _rays(HW, 62.0, 15.0, 1.3)
    ro, rd = ro.to(dev), rd.to(dev)
    bg = torch.rand(HW * HW, 4, device=dev)
    g = torch.randn(1, HW * HW, 4, device=dev) * 0.3
    plist = list(net.parameters())
    stream = torch.cuda.Stream()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        ref = net.render(ro, rd, bg_color=bg, perturb=False)
        ref["image"].backward(gradient=g)
        want = (ref["image"].detach().clone(), [p.grad.detach().clone() for p in plist], int(ref["counter"][0]))
        for slot in (0, 1):
            for p in plist:
                p.grad = None
            prep = net.prepare_rays(ro, rd, bg_color=bg, perturb=False, slot=slot)
            out = net.render(None, None, prepared=prep)
            out["image"].backward(gradient=g)
            assert int(out["counter"][0]) == want[2] and torch.equal(out["image"], want[0])
            for p, w in zip(plist, want[1]):
                assert torch.equal(p.grad, w)
        for p in plist:
            p.grad = None
        opt = FusedAdam(net.get_params(1e-3), encoder=net.encoder, capturable=True, fuse_table_update=True)
        state = {}

        def fwd_bwd():
            main = torch.cuda.current_stream()
            if "prep" not in state:
                state["prep"], state["slot"] = net.prepare_rays(ro, rd, bg_color=bg, perturb=True, slot=0), 0
            cur, slot = state["prep"], state["slot"]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                nxt = net.prepare_rays(ro, rd, bg_color=bg, perturb=True, slot=1 - slot)
            out = net.render(None, None, prepared=cur)
            opt.arm()
            out["image"].backward(gradient=g)
            main.wait_stream(side)
            state["prep"], state["slot"] = nxt, 1 - slot
            return out

        gs = GraphedTrainStep(fwd_bwd, lambda: opt.step(), plist, world=1, warmup=2, stream=stream, steps_per_graph=2)
        assert gs.steps_per_call == 2
        before = net.encoder.embeddings.detach().clone()
        c0, j0 = int(opt.step_dev[0].item()), int(net._noise_counter[0].item())
        for _ in range(3):
            gs()
        stream.synchronize()
        assert int(opt.step_dev[0].item()) == c0 + 6
        assert int(net._noise_counter[0].item()) == j0 + 6
        assert bool(torch.isfinite(net.encoder.embeddings).all()) and bool(torch.isfinite(net.w2).all())
        assert float((net.encoder.embeddings.detach() - before).abs().max()) > 0
    torch.cuda.synchronize()


def _run_steps(dev, fuse, precision, log2_T, steps=3, tail=None, capturable=False, full=False):
    from src.latent_nerf.training.optimizer import FusedAdam
    G, HW = 64, 32
    net, cfg, lv, table, params, grid = _make(dev, G, HW, log2_T, 16, seed=7, mlp_precision=precision,
                                              table_dtype=precision)
    net.train()
    opt = FusedAdam(net.get_params(1e-2), encoder=net.encoder, fuse_table_update=fuse, capturable=capturable,
                    mlp=net if tail is not None else None, tail=tail)
    gen = torch.Generator().manual_seed(3)
    bg = torch.rand(HW * HW, 4, generator=gen).to(dev)
    for k in range(steps):
        ro, rd = _rays(HW, 60.0, 25.0 * k, 1.3)
        g = (torch.randn(1, HW * HW, 4, generator=gen) * 0.3).to(dev)
        out = net.render(ro.to(dev), rd.to(dev), bg_color=bg, perturb=False)
        opt.arm()
        out["image"].backward(gradient=g)
        opt.step()
    torch.cuda.synchronize()
    emb = net.encoder.embeddings
    m, v = [(e[1], e[2]) for e in opt.big if e[0] is emb][0]
    sh = net.encoder.shadow()
    if full:   # every parameter, every moment, the MLP's bf16 weight fragments, the device step counter
        small = {k: (getattr(net, k).detach().clone(),) + tuple(t.clone() for e in opt.small if e[0] is getattr(net, k)
                                                                for t in e[1:3]) for k in ("w1", "b1", "w2", "b2", "w3", "b3")}
        frag = net.mlp_workspace(dev)[:30 * 1024].clone() if precision == "bf16" else None   # the 30 weight fragments (F_ALL)
        return {"table": emb.detach().clone(), "m": m.clone(), "v": v.clone(), "shadow": None if sh is None else sh.clone(),
                "small": small, "frag": frag, "step_dev": None if opt.step_dev is None else opt.step_dev.clone(),
                "frag_current": net.fragments_current(), "grad_w2": net.w2.grad,
                "inline_tail": bool(opt.fused is not None and opt.fused.inline_tail)}
    return (emb.detach().clone(), m.clone(), v.clone(), None if sh is None else sh.clone(), net.w2.detach().clone(),
            emb.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,log2_T", [("f32", 19), ("bf16", 19), ("bf16", 14)])
def test_fused_table_update_is_bit_identical(dev, precision, log2_T):
    """lnerf_grid_encode_backward_adam (Adam step of the hash table applied by the kernel that finishes a row's sum)
    against backward + FusedAdam.step(): same table, moments and bf16 shadow, BIT FOR BIT, after three steps -- the
    scatter sums in fixed point (order-independent) and both paths share one Adam definition.  2^19 rows per level:
    levels 5..15 are finished by their only workgroup, the sliced coarse levels by the last slice to arrive; 2^14: every
    level is sliced."""
    ref = _run_steps(dev, False, precision, log2_T)
    got = _run_steps(dev, True, precision, log2_T)
    assert got[5] is None            # the table never gets a .grad in fused mode
    for name, a, b in zip(("table", "exp_avg", "exp_avg_sq", "shadow", "w2"), got[:5], ref[:5]):
        if a is None and b is None:
            continue
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))
    assert float((got[0] - _make(dev, 64, 32, log2_T, 16, seed=7)[0].encoder.embeddings.detach()).abs().max()) > 0


@pytest.mark.gpu
def test_full_size_bf16_step_with_fused_adam_matches_oracle(dev):
    """BASELINE config 2 at full size (64x64 rays, 128^3 grid, L = 16, T = 2^19; bf16 table + features + MFMA MLP,
    8-byte scatter records, table Adam step fused into the scatter): ONE optimisation step against the oracle's
    render_frame (same bf16 roundings, forward and backward, and the record rounding of the scatter) + adam_step.
    First-step Adam moments are (1-b1) g and (1-b2) g^2, so they carry the gradient's tolerance (1e-2 of the maximum;
    2.1e-2 for v = g^2); the parameter moves by lr * sign(g) wherever |g| is clearly non-zero, and rows no sample touched
    keep their bits."""
    from src.latent_nerf.training.optimizer import FusedAdam
    G, HW, log2_T = 128, 64, 19
    net, cfg, lv, table, params, grid = _make(dev, G, HW, log2_T, 16, seed=11, mlp_precision="bf16", table_dtype="bf16")
    net.train()
    lr = 1e-2
    opt = FusedAdam(net.get_params(lr / 10.0), encoder=net.encoder, fuse_table_update=True)   # table lr = 10 x base
    assert net.encoder.scatter_variant == 3                                                    # 8-byte records
    ro, rd = _rays(HW, 65.0, 20.0, 1.25)
    N = HW * HW
    torch.manual_seed(5)
    bg = torch.rand(N, 4)
    g = torch.randn(1, N, 4) * 0.3
    t0 = net.encoder.embeddings.detach().cpu().clone()
    out = net.render(ro.to(dev), rd.to(dev), bg_color=bg.to(dev), perturb=False)
    opt.arm()
    out["image"].backward(g.to(dev))
    opt.step()
    torch.cuda.synchronize()
    assert net.encoder.embeddings.grad is None            # fused: the table gradient never existed in HBM
    ref = O.render_frame(ro[0], rd[0], table, params, lv, O.packbits(grid.reshape(-1), 0.01), G=G, bg_color=bg,
                         bf16_mlp=True, bf16_table=True)
    ref["image"].backward(g[0])
    M = ref["M"]
    assert int(out["counter"][0]) == M and M > 300000 and torch.equal(out["rays"].cpu(), ref["rays"])
    e, s = _err(out["image"][0], ref["image"])
    print("full-size bf16 step: image error %.3e (scale %.3f)" % (e, s))
    assert e <= 1e-3 * max(s, 1.0), ("image", e, s)
    gt = table.grad
    p1, m1, v1 = O.adam_step(t0, gt, torch.zeros_like(gt), torch.zeros_like(gt), 1, lr)
    emb = net.encoder.embeddings.detach().cpu()
    m_got, v_got = [(e_[1].cpu(), e_[2].cpu()) for e_ in opt.big if e_[0] is net.encoder.embeddings][0]
    gmax = float(gt.abs().max())
    e_m = float((m_got - m1).abs().max()) / (0.1 * gmax)
    e_v = float((v_got - v1).abs().max()) / (0.01 * gmax * gmax)
    print("full-size bf16 step: exp_avg error %.3e of (1-b1) max|g|, exp_avg_sq error %.3e of (1-b2) max|g|^2" % (e_m, e_v))
    assert e_m <= 1e-2, ("exp_avg", e_m)
    assert e_v <= 2.1e-2, ("exp_avg_sq", e_v)
    sure = gt.abs() > 2e-2 * gmax                        # sign of g beyond the stated tolerance: the step is lr * sign(g)
    assert int(sure.sum()) > 1000
    assert float((emb - p1)[sure].abs().max()) < 1e-4 * lr + 1e-7
    untouched = (gt == 0) & (m_got == 0)
    assert int(untouched.sum()) > 0 and torch.equal(emb[untouched], t0[untouched])
    moved = (emb != t0)
    assert float(((emb - t0)[moved]).abs().max()) <= lr * 1.001     # |Adam step| <= lr on the first step


@pytest.mark.gpu
@pytest.mark.parametrize("precision,log2_T,capturable", [("bf16", 19, True), ("bf16", 14, True), ("f32", 19, False)])
def test_step_tail_is_bit_identical_to_the_separate_launches(dev, precision, log2_T, capturable):
    """The step's tail -- the sum of the MLP's gradient slabs + the Adam step of its six tensors, the step-counter tick
    and the clearing of the scatter's level maxima -- in both of its forms: inside the armed scatter's pass 2
    (lnerf_grid_encode_backward_adam_tail, the capturable optimiser: no launch behind the scatter) and as one launch
    of its own (lnerf_step_tail, the optimiser without a device step counter), against the separate launches (slab sum
    inside the backward pass, multi-tensor Adam): table, every MLP tensor, every moment, the bf16 weight fragments and
    the device step counter agree BIT FOR BIT after four steps.  2^14 rows per level: every level's buckets are sliced,
    so the table's whole Adam step is done by the last slice of each bucket to arrive."""
    ref = _run_steps(dev, True, precision, log2_T, steps=4, tail=False, capturable=capturable, full=True)
    got = _run_steps(dev, True, precision, log2_T, steps=4, tail=True, capturable=capturable, full=True)
    assert got["grad_w2"] is None                      # tail mode: the weight gradients never exist
    assert got["inline_tail"] == bool(capturable)      # with the device counter the scatter closes the step itself
    for k in ("table", "m", "v", "shadow", "frag", "step_dev"):
        a, b = got[k], ref[k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert torch.equal(a, b), (k, float((a.float() - b.float()).abs().max()))
    for k, (p, m, v) in got["small"].items():
        rp, rm_, rv = ref["small"][k]
        assert torch.equal(p, rp) and torch.equal(m, rm_) and torch.equal(v, rv), k
    if precision == "bf16":
        assert got["frag_current"] and ref["frag_current"]
    if capturable:
        assert int(got["step_dev"][0]) == 5 and int(got["step_dev"][1]) == 0


@pytest.mark.gpu
def test_closing_scatter_refuses_what_it_cannot_do(dev):
    """lnerf_grid_encode_backward_adam_tail: without the device step counter (the closing arrival ticks it) and for an
    empty frame bound (m_host = 0) the call is refused with a message, nothing is launched; the optimiser picks the
    stand-alone tail for those cases by itself (inline_tail False without a device counter)."""
    from src.latent_nerf.models import encoding as E
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.training.optimizer import FusedAdam
    net, cfg, lv, table, params, grid = _make(dev, 64, 32, 14, 16, seed=3, mlp_precision="bf16", table_dtype="bf16")
    net.train()
    opt = FusedAdam(net.get_params(1e-3), encoder=net.encoder, fuse_table_update=True, mlp=net, tail=True, capturable=False)
    assert opt.fused.tail and not opt.fused.inline_tail            # no device counter: lnerf_step_tail closes the step
    M, stride = 1000, 1024
    xyzs = (torch.rand(stride, 3, device=dev) * 2 - 1)
    dfeat = torch.randn(16, stride, 2, device=dev)
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
    ws = net.mlp_workspace(dev)
    before = net.encoder.embeddings.detach().clone()
    with pytest.raises(B.LnerfError, match="device counter"):
        E.grid_encode_backward_adam_tail(xyzs, 1.0, dfeat, net.encoder, M, m_dev, stride, 3, ws, B.BF16, 5)
    opt2 = FusedAdam(net.get_params(1e-3), encoder=net.encoder, fuse_table_update=True, mlp=net, tail=True, capturable=True)
    assert opt2.fused.inline_tail
    with pytest.raises(B.LnerfError, match="m_host > 0"):
        E.grid_encode_backward_adam_tail(xyzs, 1.0, dfeat, net.encoder, 0, m_dev, stride, 3, ws, B.BF16, 5)
    torch.cuda.synchronize()
    assert torch.equal(net.encoder.embeddings.detach(), before) and int(opt2.step_dev[0]) == 1


@pytest.mark.gpu
def test_step_tail_without_an_mlp_still_ticks_and_clears(dev):
    """lnerf_step_tail with mlp_workspace = NULL (a C-ABI caller whose step has no fused MLP): the tick of the device step
    counter and the clearing of the scatter's level maxima it asked for still happen -- one workgroup arrives on its own.
    (Through ABI 5 the call returned LNERF_OK without a launch: a stalled counter, wrong Adam bias corrections.)  A call
    with nothing to do at all launches nothing and is not an error; flags without the scatter workspace are refused."""
    import ctypes
    from src.latent_nerf.models import encoding as E
    from src.latent_nerf.raymarching import backend as B
    from src.latent_nerf.raymarching.raymarching import _p, _stream
    levels = E.GridLevels(16, 2, 16, 2048, 14)
    M = 4096
    wst = E.scatter_workspace(levels, M, dev)
    n = levels.n_rows
    table = torch.zeros(n, 2, device=dev)
    m, v, zero = torch.zeros_like(table), torch.zeros_like(table), torch.zeros_like(table)
    step_dev = torch.tensor([7, 0], device=dev, dtype=torch.int32)
    wst[:4096].view(torch.int32)[::32][:16] = 0x3F800000          # the level maxima: something to clear
    torch.cuda.synchronize()

    def tail(flags, ws=wst, nl=levels.num_levels):
        B.call("lnerf_step_tail", nl, levels.level_dim, levels.c_offsets, levels.c_scales, levels.c_res, M, 3,
               _p(ws), 0 if ws is None else ws.numel(), _p(zero), _p(table), _p(m), _p(v), None, 1e-2, None, 0, B.BF16, 5,
               None, None, None, 1e-3, None, 0.9, 0.99, 1e-15, 1, _p(step_dev), 1.0, flags, _stream())

    tail(B.TAIL_TICK | B.TAIL_CLEAR_SCATTER)
    torch.cuda.synchronize()
    assert step_dev.tolist() == [8, 0]
    assert int(wst[:4096].view(torch.int32)[::32][:16].abs().sum()) == 0
    tail(B.TAIL_TICK)
    torch.cuda.synchronize()
    assert step_dev.tolist() == [9, 0]
    tail(0)                                                        # nothing to do: no launch, no error
    torch.cuda.synchronize()
    assert step_dev.tolist() == [9, 0]
    with pytest.raises(B.LnerfError):
        tail(B.TAIL_TICK, ws=None, nl=0)                           # the arrival counters live in the scatter workspace
    assert float(table.abs().max()) == 0.0
    E.ws_mark_dirty(dev)      # (this test wrote the shared workspace's header by hand)


@pytest.mark.gpu
def test_fused_table_update_arming(dev):
    """Only an armed backward applies the fused update; an unarmed one yields the ordinary table gradient, and a step
    that mixes both is refused."""
    from src.latent_nerf.training.optimizer import FusedAdam
    net, cfg, lv, table, params, grid = _make(dev, 64, 32, 14, 16, seed=1)
    net.train()
    opt = FusedAdam(net.get_params(1e-3), encoder=net.encoder, fuse_table_update=True)
    ro, rd = _rays(32)
    g = torch.randn(1, 32 * 32, 4, device=dev)
    emb = net.encoder.embeddings
    before = emb.detach().clone()
    net.render(ro.to(dev), rd.to(dev), perturb=False)["image"].backward(gradient=g)   # unarmed: plain gradient
    assert emb.grad is not None and torch.equal(emb.detach(), before)
    opt.step()                                                                         # ordinary Adam step
    assert float((emb.detach() - before).abs().max()) > 0 and emb.grad is None
    opt.arm()
    for _ in range(2):   # the first backward takes the arm, the second leaves a gradient behind
        net.render(ro.to(dev), rd.to(dev), perturb=False)["image"].backward(gradient=g)
    with pytest.raises(RuntimeError, match="exactly one backward"):
        opt.step()


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [False, True])
def test_empty_frame_trains_without_samples(dev, fuse):
    """A view that misses the occupied region: zero samples on the device (the host never learns the count), the
    image is the background, backward and the optimiser step run through every kernel and change nothing."""
    from src.latent_nerf.training.optimizer import FusedAdam
    net, cfg, lv, table, params, grid = _make(dev, 64, 32, 19, 16, seed=2, mlp_precision="bf16", table_dtype="bf16")
    net.train()
    opt = FusedAdam(net.get_params(1e-2), encoder=net.encoder, fuse_table_update=fuse)
    ro, rd = _rays(32, 60.0, 0.0, 1.3)
    rd = -rd                                   # look away from the sphere
    bg = torch.rand(32 * 32, 4, device=dev)
    before = net.encoder.embeddings.detach().clone()
    w_before = net.w1.detach().clone()
    out = net.render(ro.to(dev), rd.to(dev), bg_color=bg, perturb=True)
    assert int(out["counter"][0]) == 0
    assert torch.equal(out["image"][0], bg)
    opt.arm()
    out["image"].backward(gradient=torch.randn(1, 32 * 32, 4, device=dev))
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(net.encoder.embeddings.detach(), before) and torch.equal(net.w1.detach(), w_before)
    assert bool(torch.isfinite(net.encoder.shadow().float()).all())


def test_stale_backward_after_a_second_render_fails_loudly(dev):
    """The march reuses its sample buffers from render to render and the kernels write them through raw pointers;
    a backward pass of an EARLIER render must not silently use the later render's samples: autograd's version check
    fires because the reuse bumps the buffers' versions (raymarching.march_rays_train)."""
    net, cfg, lv, table, params, grid = _make(dev, 32, 16, 12, 16)
    net.train()
    ro, rd = _rays(16)
    ro2, rd2 = _rays(16, 80.0, 90.0, 1.4)
    first = net.render(ro.to(dev), rd.to(dev), bg_color=1.0, perturb=False)
    second = net.render(ro2.to(dev), rd2.to(dev), bg_color=1.0, perturb=False)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        first["image"].sum().backward()
    second["image"].sum().backward()          # the latest render is still valid
    assert float(net.encoder.embeddings.grad.abs().sum()) > 0


def test_importance_resampling_refines_the_uniform_render(dev):
    """`upsample_steps` (render.upsample_steps / run(upsample_steps=...)): 24 uniform + 24 importance samples stay
    within the quadrature error of the uniform render against a 512-sample reference on rays that cross the (sharpened)
    density blob, the inverse-CDF sampler puts its samples where the weights are, and gradients flow through the
    refined render."""
    G, HW = 32, 16
    net, cfg, lv, table, params, grid = _make(dev, G, HW, 12, 16, seed=9, cuda_ray=False, table_std=1e-4)
    net.eval()
    net.blob_std = 0.08                                      # a compact, sharp-edged density: where resampling pays
    net.blob_scale = 12.0
    ro, rd = _rays(HW)
    ro, rd = ro.to(dev), rd.to(dev)
    with torch.no_grad():
        fine = net.render(ro, rd, bg_color=1.0, num_steps=512)["weights_sum"][0]
        coarse = net.render(ro, rd, bg_color=1.0, num_steps=24)["weights_sum"][0]
        refined = net.render(ro, rd, bg_color=1.0, num_steps=24, upsample_steps=24)["weights_sum"][0]
    hit = fine > 0.05
    assert int(hit.sum()) > 8
    e_coarse = float((coarse - fine)[hit].abs().mean())
    e_refined = float((refined - fine)[hit].abs().mean())
    # (on this smooth blob both quadratures are within a percent of the reference; the refinement must not hurt)
    assert e_refined < 0.03 and e_refined < 1.5 * e_coarse + 1e-3, (e_refined, e_coarse)
    # the config field drives the default of render()
    cfg.upsample_steps, cfg.num_steps = 24, 24
    with torch.no_grad():
        again = net.render(ro, rd, bg_color=1.0)["weights_sum"][0]
    assert torch.equal(again, refined)                       # stratified (deterministic) when not training
    net.train()
    out = net.render(ro, rd, bg_color=1.0, perturb=True)
    out["image"].sum().backward()
    assert float(net.encoder.embeddings.grad.abs().sum()) > 0 and bool(torch.isfinite(out["image"]).all())
    from src.latent_nerf.models.renderer import _sample_pdf
    bins = torch.linspace(0, 1, 9, device=dev)[None].repeat(3, 1)
    w = torch.tensor([[0, 0, 0, 1, 1, 0, 0, 0.0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 1]], device=dev)
    z = _sample_pdf(bins, w, 64, stratified=True)
    assert bool((z[:, 1:] >= z[:, :-1]).all())
    assert float(((z[0] > 0.375) & (z[0] < 0.625)).float().mean()) > 0.95    # mass sits in the two middle bins
    assert float((z[1] < 0.125).float().mean()) > 0.95
    assert float((z[2] - torch.linspace(0.5 / 64, 1 - 0.5 / 64, 64, device=dev)).abs().max()) < 1e-5   # uniform -> identity


def test_sample_budget_follows_the_observed_march(dev):
    """Capacity of the sample buffers: worst case (rays x 256) until the first occupancy refresh has read the march
    counters back, then 1.5 x the largest M seen since the previous refresh, rounded up to 64 Ki -- and rays that do
    not fit a too-small budget are dropped and counted, never written out of bounds."""
    HW = 48
    net, cfg, lv, table, params, grid = _make(dev, 64, HW, 14, 16, seed=1)
    net.train()
    ro, rd = _rays(HW)
    ro, rd = ro.to(dev), rd.to(dev)
    out = net.render(ro, rd, bg_color=1.0, perturb=False)
    N = HW * HW
    assert out["xyzs"].shape[0] == N * 256
    M = int(out["counter"][0])
    img0 = out["image"].detach().clone()
    net.update_extra_state()                                  # the sync point: budget derived here
    want = -(-int(1.5 * M) // 65536) * 65536
    assert net._budget == ((N, 1024), want) and net.mean_count == M and want < N * 256, (net._budget, M)
    net.density_grid.copy_(grid.to(dev))                      # put the test scene back (the refresh re-estimated it)
    net.density_bitfield.copy_(O.packbits(grid.reshape(-1), 0.01).to(dev))
    out = net.render(ro, rd, bg_color=1.0, perturb=False)
    assert out["xyzs"].shape[0] == want and int(out["counter"][0]) == M and int(out["counter"][2]) == 0
    assert torch.equal(out["image"], img0)
    # an explicit (too small) budget: later rays are dropped and counted
    cfg.max_samples = (M // 2 // 64) * 64
    out = net.render(ro, rd, bg_color=1.0, perturb=False)
    c = out["counter"].cpu()
    assert int(c[0]) <= cfg.max_samples and int(c[2]) > 0 and out["xyzs"].shape[0] == cfg.max_samples
    out["image"].sum().backward()
    assert bool(torch.isfinite(net.encoder.embeddings.grad).all())


@pytest.mark.gpu
def test_optimizer_keeps_the_mlp_weight_fragments_current(dev):
    """FusedAdam(mlp=net) mirrors the updated w1 / w2 / w3 into the network's bf16 weight fragments, and the forward
    then skips its per-step fragment build: the field must be bit-identical to a forward that rebuilds them, before
    and after optimiser steps, and any other modification of a weight must bring the build back."""
    from src.latent_nerf.training.optimizer import FusedAdam
    net, *_ = _make(dev, 32, 16, 14, 16, seed=5, mlp_precision="bf16", table_dtype="bf16")
    opt = FusedAdam(net.get_params(1e-2), encoder=net.encoder, mlp=net)
    assert net._frag_owner is opt and not net.fragments_current()
    x = (torch.rand(5000, 3, device=dev) * 2 - 1) * 0.9

    def both():
        kept = net.fragments_current()
        with torch.no_grad():
            a = [t.clone() for t in net.field(x, x.shape[0])]
            net._frag_versions = None                    # force the build
            b = [t.clone() for t in net.field(x, x.shape[0])]
        return kept, a, b

    kept, a, b = both()                                  # first forward ever: builds either way
    assert not kept and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for it in range(3):
        sig, lat = net.field(x, x.shape[0])
        (sig.mean() + lat.square().mean()).backward()
        w_before = net.w2.detach().clone()
        opt.step()
        assert not torch.equal(net.w2.detach(), w_before)
        kept, a, b = both()
        assert kept, "the optimiser's mirror must leave the fragments current"
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with torch.no_grad():
        net.w1.mul_(1.5)                                 # somebody else's in-place change
    assert not net.fragments_current()
    kept, a, b = both()
    assert not kept and torch.equal(a[0], b[0])


@pytest.mark.gpu
def test_fragment_mirror_inside_a_replayed_graph(dev):
    """The mirroring optimiser captured in a hipGraph: replays move the weights AND their fragments (torch's version
    counters see nothing of it), so a later eager forward that skips the build must still equal one that rebuilds."""
    from src.latent_nerf.training.graph_step import GraphedTrainStep
    from src.latent_nerf.training.optimizer import FusedAdam
    G, HW = 64, 32
    net, *_ = _make(dev, G, HW, 14, 16, seed=6, mlp_precision="bf16", table_dtype="bf16")
    net.train()
    ro, rd = _rays(HW, 65.0, 10.0, 1.3)
    ro, rd = ro.to(dev), rd.to(dev)
    bg = torch.rand(HW * HW, 4, device=dev)
    g = torch.randn(1, HW * HW, 4, device=dev) * 0.3
    plist = list(net.parameters())
    x = (torch.rand(4000, 3, device=dev) * 2 - 1) * 0.9

    def fwd_bwd():
        out = net.render(ro, rd, bg_color=bg, perturb=False)
        out["image"].backward(gradient=g)
        return out

    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        opt = FusedAdam(net.get_params(1e-2), encoder=net.encoder, capturable=True, mlp=net)
        gs = GraphedTrainStep(fwd_bwd, lambda: opt.step(), plist, world=1, warmup=2, stream=stream)
        w0 = net.w2.detach().clone()
        for _ in range(4):
            gs()
        stream.synchronize()
        assert float((net.w2.detach() - w0).abs().max()) > 0 and net.fragments_current()
        with torch.no_grad():
            a = [t.clone() for t in net.field(x, x.shape[0])]      # skips the build
            net._frag_versions = None
            b = [t.clone() for t in net.field(x, x.shape[0])]      # rebuilds
        stream.synchronize()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    torch.cuda.synchronize()
