"""Guidance interface of the trainer: `train_step(text_z, latents) -> grad`, the contract of the
reference's StableDiffusion.train_step in this fork (src/stable_diffusion.py:248-334: it RETURNS the SDS
gradient w(t)(eps_hat - eps), :320-321,334, and the caller injects it with
`pred.backward(gradient=grad)`, src/latent_paint_mesh/training/trainer.py:657-658).

The diffusion model itself is out of scope (network-fetched weights, dense UNet).
`SyntheticGuidance` is a seeded stand-in with the same call shape; `StableDiffusionGuidance` adapts a
locally available diffusers checkpoint on machines that have one."""
import math

import torch

# approximate linear latent -> RGB map of the reference's previews (src/latent_paint/models/textured_mesh.py:34-40)
LATENT_TO_RGB = ((0.298, 0.207, 0.208), (0.187, 0.286, 0.173), (-0.158, 0.189, 0.264), (-0.184, -0.271, -0.473))
VAE_SCALE = 0.18215   # latent scaling of the SD-1.x autoencoder (src/stable_diffusion.py:462-489)


@torch.no_grad()
def linear_decode_latents(latents, upsample=8):
    """[B,4,h,w] latents -> [B,3,8h,8w] RGB in [0,1] WITHOUT a VAE: the linear latent->RGB estimate of
    src/latent_paint/models/textured_mesh.py:34-40, a bilinear 8x resize (the decoder's up-sampling factor) and the
    decoder's output mapping (x/2 + 0.5).clamp(0,1) (src/stable_diffusion.py:469).  What every trainer falls back to
    when its guidance object has no decoder."""
    m = torch.tensor(LATENT_TO_RGB, device=latents.device, dtype=latents.dtype)
    if latents.shape[1] != 4:
        rgb = latents[:, :3]
    else:
        rgb = torch.einsum("bchw,cr->brhw", latents, m)
    if upsample and upsample != 1:
        rgb = torch.nn.functional.interpolate(rgb, scale_factor=upsample, mode="bilinear", align_corners=False)
    return (rgb / 2 + 0.5).clamp(0, 1)


def decode_with(guidance, latents):
    """`guidance.decode_latents(latents)` when the guidance object has a decoder, the linear preview otherwise (a
    guidance object may be anything with `train_step`; a missing or abstract decoder must not end an evaluation)."""
    fn = getattr(guidance, "decode_latents", None)
    if fn is not None:
        try:
            return fn(latents)
        except NotImplementedError:
            pass
    return linear_decode_latents(latents)


class Guidance:
    latent_mode = True

    def get_text_embeds(self, prompt):
        raise NotImplementedError()

    def train_step(self, text_z, latents):
        """latents [B,4,H,W] -> gradient w.r.t. latents, same shape (no autograd through it)."""
        raise NotImplementedError()

    def decode_latents(self, latents):
        """latents [B,4,h,w] -> RGB [B,3,8h,8w] in [0,1] (evaluation renders, mesh export).  Default: the linear
        preview; a guidance object with a VAE overrides it."""
        return linear_decode_latents(latents)

    def encode_imgs(self, imgs):
        """imgs [B,3,H,W] in [0,1] -> latents [B,4,H/8,W/8] (src/stable_diffusion.py:482-489); needs a VAE."""
        raise NotImplementedError("%s has no image encoder" % type(self).__name__)


class SyntheticGuidance(Guidance):
    """Deterministic target-seeking gradient: for view bucket d (0..5) the target latent image is a fixed
    smooth pattern T_d; grad = w(t) * (latents - T_d + sigma_t * noise) with t ~ U{20..980}, w = sqrt(a_t)(1 - a_t)
    (the weighting form of src/stable_diffusion.py:274,320).  Converges like a denoising objective, needs no
    weights, and exercises the whole render/backward path."""

    def __init__(self, device, channels=4, size=64, seed=0, noise_scale=0.05):
        g = torch.Generator().manual_seed(seed)
        base = torch.randn(6, channels, 8, 8, generator=g)
        self.targets = torch.nn.functional.interpolate(base, size=(size, size), mode="bilinear",
                                                       align_corners=False).to(device) * 0.5
        self.device = device
        self.noise_scale = noise_scale
        self.num_train_timesteps = 1000
        self.min_step, self.max_step = 20, 980
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2   # SD "scaled_linear" schedule
        self.alphas = torch.cumprod(1.0 - betas, 0).to(device)
        self.weights = (torch.sqrt(self.alphas) * (1 - self.alphas)).contiguous()
        self.seed = int(seed)
        self._targets_rows = {}     # (H, W) -> targets as [6, H*W, C] rows (the renderer's image layout)

    def get_text_embeds(self, prompt):
        return torch.zeros(2, 77, 768, device=self.device)

    @torch.no_grad()
    def decode_latents(self, latents):
        """Stand-in for the VAE decoder: the linear preview (see linear_decode_latents)."""
        return linear_decode_latents(latents)

    # Every op of train_step_device is a device op on static shapes with device-side RNG: the trainer captures it INSIDE
    # its step graph (one graph launch per step instead of graph / eager launches / graph)
    capturable = True

    @torch.no_grad()
    def train_step_device(self, latents, dir_index):
        """train_step with the view bucket as a DEVICE int32 tensor [B] (no host value anywhere): same arithmetic, same
        random draws in the same order."""
        target = torch.index_select(self.targets, 0, dir_index)
        if target.shape[-2:] != latents.shape[-2:]:
            target = torch.nn.functional.interpolate(target, size=latents.shape[-2:], mode="bilinear",
                                                     align_corners=False)
        t = torch.randint(self.min_step, self.max_step + 1, [1], device=latents.device)
        w = self.weights[t]
        grad = torch.randn_like(latents).mul_(self.noise_scale)
        grad.add_(latents - target)
        return grad.mul_(w)

    @torch.no_grad()
    def train_step_image(self, image, dirs_dev, H, W, step_dev, weights_sum=None, sparsity_scale=0.0, eps=1e-5):
        """The same guidance as ONE HIP launch on the renderer's own layout (lnerf_synthetic_guidance): image
        [B, H*W, C] -> (gradient w.r.t. image [B, H*W, C], gradient of sparsity_scale * sparsity_loss(weights_sum) or
        None).  dirs_dev: int32 [B] on the device; step_dev: the optimiser's device step counter (noise and timestep are
        counter-based functions of (seed, *step_dev, element): capturable, fresh on every graph replay).  What the NeRF
        trainer calls, eager and captured alike; the [B,C,H,W] forms above keep the reference's call shape."""
        from ..raymarching import backend as _b
        from ..raymarching.raymarching import _chk, _p, _stream
        B, N, C = image.shape
        key = (int(H), int(W))
        rows = self._targets_rows.get(key)
        if rows is None:
            t = self.targets
            if t.shape[-2:] != key:
                t = torch.nn.functional.interpolate(t, size=key, mode="bilinear", align_corners=False)
            rows = t.permute(0, 2, 3, 1).reshape(t.shape[0], N, C).contiguous()
            self._targets_rows[key] = rows
        img = image.detach()
        grad = torch.empty_like(img)
        gws = None
        if weights_sum is not None and sparsity_scale > 0:
            weights_sum = weights_sum.detach().contiguous()
            gws = torch.empty_like(weights_sum)
        _b.call("lnerf_synthetic_guidance", _chk(img, "image"), _p(rows), _chk(dirs_dev, "dirs", torch.int32),
                _p(self.weights), int(B), int(N), int(C), int(rows.shape[0]), int(self.min_step), int(self.max_step),
                float(self.noise_scale), self.seed & 0xFFFFFFFF, _chk(step_dev, "step_dev", torch.int32), _p(grad),
                None if gws is None else _p(weights_sum), float(sparsity_scale), float(eps), None if gws is None else _p(gws),
                _stream())
        return grad, gws

    @torch.no_grad()
    def train_step(self, text_z, latents, dirs=None):
        """(few launches: the trainer's step is a graph replay either side of this call, every eager launch here is on
        the step's critical path)"""
        B = latents.shape[0]
        if dirs is None:
            target = self.targets[0:1].expand(B, -1, -1, -1)
        elif B == 1:
            target = self.targets[int(dirs[0]):int(dirs[0]) + 1]     # a view: no gather launch
        else:
            target = self.targets[dirs.to(latents.device)]
        if target.shape[-2:] != latents.shape[-2:]:
            target = torch.nn.functional.interpolate(target, size=latents.shape[-2:], mode="bilinear",
                                                     align_corners=False)
        t = torch.randint(self.min_step, self.max_step + 1, [1], device=latents.device)
        w = self.weights[t]                                           # sqrt(a_t) (1 - a_t), tabulated at construction
        grad = torch.randn_like(latents).mul_(self.noise_scale)      # noise
        grad.add_(latents - target)                                   # (latents - target) + noise
        return grad.mul_(w)


class StableDiffusionGuidance(Guidance):
    """Adapter for a machine that has `diffusers` + `transformers` and a LOCAL SD-1.x checkpoint directory (nothing is
    ever fetched: every loader runs with local_files_only=True).  Mirrors src/stable_diffusion.py in latent mode:
    get_text_embeds :226-246, train_step :248-334, decode_latents :462-470, encode_imgs :482-489.  Not used by the
    benchmark; tests drive it with stub model classes (`modules=`).

    modules: optional dict of the five classes {"AutoencoderKL", "UNet2DConditionModel", "PNDMScheduler",
    "CLIPTextModel", "CLIPTokenizer"} to use instead of importing diffusers / transformers."""

    def __init__(self, device, model_path, guidance_scale=100.0, modules=None):
        if modules is None:
            try:
                from diffusers import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
                from transformers import CLIPTextModel, CLIPTokenizer
            except Exception as e:  # pragma: no cover
                raise RuntimeError("StableDiffusionGuidance needs `diffusers` + `transformers` and a LOCAL checkpoint "
                                   "directory (no network access here): %s" % e)
            modules = {"AutoencoderKL": AutoencoderKL, "UNet2DConditionModel": UNet2DConditionModel,
                       "PNDMScheduler": PNDMScheduler, "CLIPTextModel": CLIPTextModel, "CLIPTokenizer": CLIPTokenizer}
        self.device = device
        self.guidance_scale = guidance_scale
        self.tokenizer = modules["CLIPTokenizer"].from_pretrained(model_path, subfolder="tokenizer",
                                                                  local_files_only=True)
        self.text_encoder = modules["CLIPTextModel"].from_pretrained(model_path, subfolder="text_encoder",
                                                                     local_files_only=True).to(device)
        self.unet = modules["UNet2DConditionModel"].from_pretrained(model_path, subfolder="unet",
                                                                    local_files_only=True).to(device)
        self.vae = None
        try:   # the decoder only serves evaluation renders: a checkpoint without it still trains
            self.vae = modules["AutoencoderKL"].from_pretrained(model_path, subfolder="vae",
                                                                local_files_only=True).to(device)
        except Exception as e:
            print("[guidance] no VAE under %s/vae (%s): previews use the linear latent->RGB map" % (model_path, e))
        self.scheduler = modules["PNDMScheduler"](beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                                  num_train_timesteps=1000)
        self.alphas = self.scheduler.alphas_cumprod.to(device)
        self.min_step, self.max_step = 20, 980

    @torch.no_grad()
    def get_text_embeds(self, prompt):
        tok = self.tokenizer([prompt], padding="max_length", max_length=self.tokenizer.model_max_length,
                             truncation=True, return_tensors="pt")
        unc = self.tokenizer([""], padding="max_length", max_length=self.tokenizer.model_max_length,
                             return_tensors="pt")
        return torch.cat([self.text_encoder(unc.input_ids.to(self.device))[0],
                          self.text_encoder(tok.input_ids.to(self.device))[0]])

    @torch.no_grad()
    def train_step(self, text_z, latents, dirs=None):
        t = torch.randint(self.min_step, self.max_step + 1, [1], dtype=torch.long, device=self.device)
        noise = torch.randn_like(latents)
        noisy = self.scheduler.add_noise(latents, noise, t)
        pred = self.unet(torch.cat([noisy] * 2), t, encoder_hidden_states=text_z).sample
        unc, txt = pred.chunk(2)
        pred = unc + self.guidance_scale * (txt - unc)
        w = self.alphas[t] ** 0.5 * (1 - self.alphas[t])
        return w * (pred - noise)

    @torch.no_grad()
    def decode_latents(self, latents):
        """[B,4,h,w] -> RGB [B,3,8h,8w] in [0,1]: vae.decode(latents / 0.18215), (x/2 + 0.5).clamp(0,1)
        (src/stable_diffusion.py:462-470)."""
        if self.vae is None:
            return linear_decode_latents(latents)
        imgs = self.vae.decode(latents.to(self.device) / VAE_SCALE).sample
        return (imgs / 2 + 0.5).clamp(0, 1)

    @torch.no_grad()
    def encode_imgs(self, imgs):
        """[B,3,H,W] in [0,1] -> latents [B,4,H/8,W/8]: posterior sample of vae.encode(2 imgs - 1) x 0.18215
        (src/stable_diffusion.py:482-489)."""
        if self.vae is None:
            raise NotImplementedError("this checkpoint has no VAE: encode_imgs is unavailable")
        posterior = self.vae.encode(2 * imgs.to(self.device) - 1).latent_dist
        return posterior.sample() * VAE_SCALE


def sparsity_loss_grad(weights_sum, scale, eps=1e-5):
    """d(scale * sparsity_loss(weights_sum)) / d(weights_sum) in ONE HIP launch (lnerf_opacity_entropy_grad): what the
    trainer hands to the compositing backward next to the guidance gradient.  (The autograd form of the same term is
    ten elementwise launches on the critical path of every step.)"""
    from ..raymarching import backend as _b
    from ..raymarching.raymarching import _chk, _p, _stream
    ws = weights_sum.detach().contiguous()
    grad = torch.empty_like(ws)
    _b.call("lnerf_opacity_entropy_grad", _chk(ws, "weights_sum"), ws.numel(), float(scale), float(eps), _p(grad),
            _stream())
    return grad


def sparsity_loss(weights_sum, eps=1e-5):
    """Entropy of the per-ray opacity (pushes rays to be fully empty or fully opaque)."""
    p = weights_sum.clamp(eps, 1.0 - eps)
    return (-p * torch.log2(p) - (1 - p) * torch.log2(1 - p)).mean()
