"""Python surface of the HIP ray-marching ops -- the counterpart of the `raymarching` CUDA
extension module that the reference's README lists under src/latent_nerf/raymarching
(README.md:152-156) but does not ship.  Names and argument meaning follow the upstream
torch-ngp / stable-dreamfusion module the reference says it is based on (README.md:163), as
enumerated in SURVEY.md §8(b); every op runs on liblnerf_hip.so through its C ABI.

Conventions that differ from a CUDA port, by design for MI355X:
  * `march_rays_train` is deterministic (per-ray count -> scan -> write) and never syncs with
    the host: it returns capacity-sized buffers plus a device counter; consumers take the
    counter (`m_dev`) instead of a Python int.
  * per-sample features are level-major ([L, capacity, 2]).
  * `deltas[:, 1]` is the absolute ray parameter t (depth = sum w_i t_i).
"""
import ctypes

import torch

from . import backend as _b

_NULL = None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current stream's handle.  torch.cuda.current_stream() builds a Stream object through four Python layers
    (~13 us; five calls per training step); the raw getter is what torch's own extension launchers use."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=torch.float32, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("%s: tensor is required" % name)
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got %s); there is no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------ H1 / H2 / H3
def get_rays(poses, intrinsics, H, W):
    """poses [B,4,4] (camera-to-world, columns right/down/forward/eye), intrinsics (fx,fy,cx,cy)
    -> rays_o, rays_d [B, H*W, 3]."""
    if poses.dim() == 2:
        poses = poses[None]
    poses = poses.contiguous()
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    B = poses.shape[0]
    rays_o = torch.empty(B, H * W, 3, device=poses.device, dtype=torch.float32)
    rays_d = torch.empty_like(rays_o)
    _b.call("lnerf_get_rays", _chk(poses, "poses"), B, H, W, fx, fy, cx, cy, _p(rays_o), _p(rays_d), _stream())
    return rays_o, rays_d


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """rays [N,3]; aabb: 6 floats (list/tuple/CPU or GPU tensor) -> nears, fars [N]."""
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N = rays_o.shape[0]
    a = [float(v) for v in (aabb.tolist() if isinstance(aabb, torch.Tensor) else aabb)]
    nears = torch.empty(N, device=rays_o.device, dtype=torch.float32)
    fars = torch.empty_like(nears)
    _b.call("lnerf_near_far_from_aabb", _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), N, a[0], a[1], a[2], a[3], a[4],
            a[5], float(min_near), _p(nears), _p(fars), _stream())
    return nears, fars


def morton3D(coords):
    """coords int32 [N,3] -> int32 [N] Morton index (x in bit 0)."""
    coords = coords.contiguous()
    N = coords.shape[0]
    out = torch.empty(N, device=coords.device, dtype=torch.int32)
    _b.call("lnerf_morton3d", _chk(coords, "coords", torch.int32), N, _p(out), _stream())
    return out


def morton3D_invert(indices):
    indices = indices.contiguous()
    N = indices.shape[0]
    out = torch.empty(N, 3, device=indices.device, dtype=torch.int32)
    _b.call("lnerf_morton3d_invert", _chk(indices, "indices", torch.int32), N, _p(out), _stream())
    return out


def packbits(grid, thresh, bitfield=None, mean_dev=None):
    """grid f32 [C, H^3] (Morton order) -> uint8 [C*H^3/8]; threshold = min(thresh, *mean_dev)."""
    grid = grid.contiguous()
    n = grid.numel()
    if bitfield is None:
        bitfield = torch.empty(n // 8, device=grid.device, dtype=torch.uint8)
    _b.call("lnerf_packbits", _chk(grid, "grid"), n, float(thresh), _chk(mean_dev, "mean_dev", allow_none=True),
            _chk(bitfield, "bitfield", torch.uint8), _stream())
    return bitfield


# ------------------------------------------------------------------------------ H4
class MarchResult:
    """Capacity-sized sample buffers of one training march.  `counter` (int32 [4 + scratch], device) holds
    [M, live rays, rays dropped for capacity, running peak of M | (dropped ? 2^30 : 0) over the marches into these
    buffers]; nothing here forces a host sync."""
    __slots__ = ("xyzs", "dirs", "deltas", "rays", "counter", "capacity")

    def __init__(self, xyzs, dirs, deltas, rays, counter, capacity):
        self.xyzs, self.dirs, self.deltas, self.rays, self.counter, self.capacity = (xyzs, dirs, deltas, rays, counter,
                                                                                   capacity)

    def take_peak(self):
        """(largest M, any ray dropped) of the marches into these buffers since the last call: ONE read-back
        (synchronises), then the word is cleared."""
        w = int(self.counter[3].item())
        self.counter[3:4].zero_()
        return w & ((1 << 30) - 1), bool(w >> 30)

    def num_samples(self) -> int:
        """Host read-back of M (synchronises)."""
        return int(self.counter[0].item())

    def trimmed(self):
        M = self.num_samples()
        return self.xyzs[:M], self.dirs[:M], self.deltas[:M], self.rays


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, perturb=False, dt_gamma=0.0,
                     max_steps=1024, capacity=None, noises=None, out=None, noise_state=None, aabb=None, min_near=0.0,
                     camera=None):
    """Occupancy-pruned march of N rays.  Returns a MarchResult.

    camera = (poses [B,4,4], (fx, fy, cx, cy), H_img, W_img) with rays_o / rays_d = PREALLOCATED [B*H*W, 3] outputs (and
    `aabb`): the rays are generated inside the march's count pass (lnerf_march_rays_train_pose: get_rays' arithmetic)
    and written to those tensors.

    nears / fars [N] from near_far_from_aabb -- or None with `aabb` = six host floats (+ `min_near`): the clip is
    then done inside the march passes (lnerf_march_rays_train_aabb: same arithmetic, one dispatch less).

    capacity: sample buffer size (default N * min(max_steps, 256)); rays that would overflow it
    are dropped and counted in counter[2].  `out` may pass a previous MarchResult to reuse its
    buffers.
    perturb: jitter of the march start.  `noises` [N] gives the values (upstream: torch.rand(N)); otherwise
    `noise_state` = (seed, int32 device counter [1]) selects the in-kernel counter-based generator (the call
    advances the counter; graph-capturable without host RNG state); with neither, torch.rand(N) is drawn here."""
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N = rays_o.shape[0]
    dev = rays_o.device
    if capacity is None:
        capacity = max(N * min(int(max_steps), 256), 64)
    seed, noise_counter = 0, None
    if perturb and noises is None:
        if noise_state is not None:
            seed, noise_counter = int(noise_state[0]) & 0xFFFFFFFF, noise_state[1]
        else:
            noises = torch.rand(N, device=dev, dtype=torch.float32)
    if out is not None and out.capacity == capacity and out.rays.shape[0] == N:
        xyzs, dirs, deltas, rays, counter = out.xyzs, out.dirs, out.deltas, out.rays, out.counter
        # The kernels write through raw pointers, which autograd cannot see: tell it that these buffers change, so
        # that a backward pass of an EARLIER render that saved them fails loudly ("modified by an inplace operation")
        # instead of silently using this march's samples.  Host-side bookkeeping only, no launch.
        for t in (xyzs, dirs, deltas, rays, counter):
            torch.autograd.graph.increment_version(t)
    else:
        xyzs = torch.empty(capacity, 3, device=dev, dtype=torch.float32)
        dirs = torch.empty(capacity, 3, device=dev, dtype=torch.float32)
        deltas = torch.empty(capacity, 2, device=dev, dtype=torch.float32)
        rays = torch.empty(N, 3, device=dev, dtype=torch.int32)
        # (four totals + the scratch the two-launch form of the march keeps its per-ray counts in; ZEROED: word 3 is a
        # running peak the library only ever raises -- MarchResult.take_peak() reads and clears it)
        counter = torch.zeros(int(_b.get_lib().lnerf_march_counter_len(N)), device=dev, dtype=torch.int32)
    if camera is not None:
        if aabb is None or nears is not None:
            raise ValueError("march_rays_train(camera=...) needs aabb= and no nears/fars")
        poses, intr, him, wim = camera
        poses = poses.contiguous()
        if poses.shape[0] * him * wim != N:
            raise ValueError("camera describes %d rays, the ray buffers hold %d" % (poses.shape[0] * him * wim, N))
        if torch.is_tensor(intr):
            # intrinsics as a device tensor [B,4]: nothing of the camera is a launch argument (graph replays render
            # whatever the caller copied into `poses` / `intr`)
            intr = intr.view(-1, 4)
            if intr.shape[0] != poses.shape[0]:
                raise ValueError("intrinsics tensor must be [B,4]")
            _b.call("lnerf_march_rays_train_camera", _chk(poses, "poses"), _chk(intr, "intrinsics"), int(poses.shape[0]),
                    int(him), int(wim), _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), *[float(v) for v in aabb],
                    float(min_near), _chk(density_bitfield, "density_bitfield", torch.uint8), float(bound), int(C), int(H),
                    int(max_steps), float(dt_gamma), _chk(noises, "noises", allow_none=True), seed,
                    _chk(noise_counter, "noise_counter", torch.int32, allow_none=True), int(capacity), _p(xyzs), _p(dirs),
                    _p(deltas), _p(rays), _p(counter), _stream())
            return MarchResult(xyzs, dirs, deltas, rays, counter, capacity)
        fx, fy, cx, cy = [float(v) for v in intr]
        _b.call("lnerf_march_rays_train_pose", _chk(poses, "poses"), int(poses.shape[0]), int(him), int(wim), fx, fy, cx,
                cy, _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), *[float(v) for v in aabb], float(min_near),
                _chk(density_bitfield, "density_bitfield", torch.uint8), float(bound), int(C), int(H), int(max_steps),
                float(dt_gamma), _chk(noises, "noises", allow_none=True), seed,
                _chk(noise_counter, "noise_counter", torch.int32, allow_none=True), int(capacity), _p(xyzs), _p(dirs),
                _p(deltas), _p(rays), _p(counter), _stream())
        return MarchResult(xyzs, dirs, deltas, rays, counter, capacity)
    if nears is None and aabb is not None:
        _b.call("lnerf_march_rays_train_aabb", _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"),
                *[float(v) for v in aabb], float(min_near), N, _chk(density_bitfield, "density_bitfield", torch.uint8),
                float(bound), int(C), int(H), int(max_steps), float(dt_gamma), _chk(noises, "noises", allow_none=True),
                seed, _chk(noise_counter, "noise_counter", torch.int32, allow_none=True), int(capacity), _p(xyzs),
                _p(dirs), _p(deltas), _p(rays), _p(counter), _stream())
        return MarchResult(xyzs, dirs, deltas, rays, counter, capacity)
    _b.call("lnerf_march_rays_train", _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), _chk(nears, "nears"),
            _chk(fars, "fars"), N, _chk(density_bitfield, "density_bitfield", torch.uint8), float(bound), int(C),
            int(H), int(max_steps), float(dt_gamma), _chk(noises, "noises", allow_none=True), seed,
            _chk(noise_counter, "noise_counter", torch.int32, allow_none=True), int(capacity), _p(xyzs),
            _p(dirs), _p(deltas), _p(rays), _p(counter), _stream())
    return MarchResult(xyzs, dirs, deltas, rays, counter, capacity)


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, fars,
               dt_gamma=0.0, max_steps=1024):
    """Inference march: up to n_step samples for each of the first n_alive entries of rays_alive."""
    dev = rays_o.device
    xyzs = torch.empty(n_alive * n_step, 3, device=dev, dtype=torch.float32)
    dirs = torch.empty(n_alive * n_step, 3, device=dev, dtype=torch.float32)
    deltas = torch.empty(n_alive * n_step, 2, device=dev, dtype=torch.float32)
    _b.call("lnerf_march_rays", int(n_alive), int(n_step), _chk(rays_alive, "rays_alive", torch.int32),
            _chk(rays_t, "rays_t"), _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d"), _chk(fars, "fars"),
            _chk(density_bitfield, "density_bitfield", torch.uint8), float(bound), int(C), int(H), int(max_steps),
            float(dt_gamma), _p(xyzs), _p(dirs), _p(deltas), _stream())
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                   transmittance, T_thresh=1e-4):
    C = image.shape[-1]
    _b.call("lnerf_composite_rays", int(n_alive), int(n_step), _chk(rays_alive, "rays_alive", torch.int32),
            _chk(rays_t, "rays_t"), _chk(sigmas, "sigmas"), _chk(rgbs, "rgbs"), _chk(deltas, "deltas"), int(C),
            float(T_thresh), _chk(weights_sum, "weights_sum"), _chk(depth, "depth"), _chk(image, "image"),
            _chk(transmittance, "transmittance"), _stream())


def compact_rays(rays_alive, n, out=None, n_alive_dev=None):
    """Keep entries >= 0 of rays_alive[:n] (order preserved).  Returns (compacted, count tensor)."""
    if out is None:
        out = torch.empty_like(rays_alive)
    if n_alive_dev is None:
        n_alive_dev = torch.empty(1, device=rays_alive.device, dtype=torch.int32)
    _b.call("lnerf_compact_rays", _chk(rays_alive, "rays_alive", torch.int32), int(n), _p(out), _p(n_alive_dev),
            _stream())
    return out, n_alive_dev


# ------------------------------------------------------------------------------ H8 / H9
class _CompositeRaysTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh, bg_color):
        sigmas = sigmas.contiguous()
        rgbs = rgbs.contiguous()
        N = rays.shape[0]
        C = rgbs.shape[1]
        dev = sigmas.device
        weights_sum = torch.empty(N, device=dev, dtype=torch.float32)
        depth = torch.empty(N, device=dev, dtype=torch.float32)
        image = torch.empty(N, C, device=dev, dtype=torch.float32)
        bg = None if bg_color is None else bg_color.contiguous()
        _b.call("lnerf_composite_rays_train_forward", _chk(sigmas, "sigmas"), _chk(rgbs, "rgbs"),
                _chk(deltas, "deltas"), _chk(rays, "rays", torch.int32), N, C, float(T_thresh),
                _chk(bg, "bg_color", allow_none=True), _p(weights_sum), _p(depth), _p(image), _stream())
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image, bg)
        ctx.set_materialize_grads(False)  # unused outputs (depth, weights_sum) arrive as None, not as zero fills
        ctx.T_thresh = float(T_thresh)
        ctx.bg_needs_grad = bg_color is not None and bg_color.requires_grad
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image):
        sigmas, rgbs, deltas, rays, weights_sum, depth, image, bg = ctx.saved_tensors
        N, C = rays.shape[0], rgbs.shape[1]
        g_image = torch.zeros_like(image) if g_image is None else g_image.contiguous()
        g_ws = None if g_ws is None else g_ws.contiguous()
        g_depth = None if g_depth is None else g_depth.contiguous()
        d_sigmas = torch.empty_like(sigmas)
        d_rgbs = torch.empty_like(rgbs)
        d_bg = torch.empty_like(bg) if ctx.bg_needs_grad else None
        _b.call("lnerf_composite_rays_train_backward", _chk(g_ws, "grad_weights_sum", allow_none=True),
                _chk(g_depth, "grad_depth", allow_none=True), _chk(g_image, "grad_image"), _p(sigmas), _p(rgbs),
                _p(deltas), _p(rays), _p(weights_sum), _p(depth), _p(image), _p(bg), N, C, ctx.T_thresh, _p(d_sigmas),
                _p(d_rgbs), _p(d_bg), _stream())
        return d_sigmas, d_rgbs, None, None, None, d_bg


def composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh=1e-4, bg_color=None):
    """sigmas [M], rgbs [M,C], deltas [M,2]=(dt,t), rays int32 [N,3]=(id,offset,count)
    -> weights_sum [N], depth [N], image [N,C] (+ (1-weights_sum)*bg_color)."""
    return _CompositeRaysTrain.apply(sigmas, rgbs, deltas, rays, T_thresh, bg_color)
