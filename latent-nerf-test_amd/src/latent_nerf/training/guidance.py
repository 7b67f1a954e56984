"""Guidance interface of the trainer: `train_step(text_z, latents) -> grad`, the contract of the
reference's StableDiffusion.train_step in this fork (src/stable_diffusion.py:248-334: it RETURNS the SDS
gradient w(t)(eps_hat - eps), :320-321,334, and the caller injects it with
`pred.backward(gradient=grad)`, src/latent_paint_mesh/training/trainer.py:657-658).

The diffusion model itself is out of scope (network-fetched weights, dense UNet).
`SyntheticGuidance` is a seeded stand-in with the same call shape; `StableDiffusionGuidance` adapts a
locally available diffusers checkpoint on machines that have one."""
import math

import torch


class Guidance:
    latent_mode = True

    def get_text_embeds(self, prompt):
        raise NotImplementedError()

    def train_step(self, text_z, latents):
        """latents [B,4,H,W] -> gradient w.r.t. latents, same shape (no autograd through it)."""
        raise NotImplementedError()

    def decode_latents(self, latents):
        """latents [B,4,h,w] -> RGB [B,3,8h,8w] in [0,1] (evaluation renders, mesh export)."""
        raise NotImplementedError()


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

    def get_text_embeds(self, prompt):
        return torch.zeros(2, 77, 768, device=self.device)

    @torch.no_grad()
    def decode_latents(self, latents):
        """[B,4,h,w] latents -> [B,3,8h,8w] RGB in [0,1]: stand-in for the VAE decoder (src/stable_diffusion.py
        decode_latents: 8x up-sampling decoder, output (x/2 + 0.5).clamp(0,1)) built from the linear latent->RGB
        estimate of src/latent_paint/models/textured_mesh.py:34-40 and a bilinear 8x resize."""
        m = torch.tensor([[0.298, 0.207, 0.208], [0.187, 0.286, 0.173], [-0.158, 0.189, 0.264],
                          [-0.184, -0.271, -0.473]], device=latents.device, dtype=latents.dtype)
        rgb = torch.einsum("bchw,cr->brhw", latents, m)
        rgb = torch.nn.functional.interpolate(rgb, scale_factor=8, mode="bilinear", align_corners=False)
        return (rgb / 2 + 0.5).clamp(0, 1)

    @torch.no_grad()
    def train_step(self, text_z, latents, dirs=None):
        B = latents.shape[0]
        d = torch.zeros(B, dtype=torch.long, device=latents.device) if dirs is None else dirs.to(latents.device)
        target = self.targets[d]
        if target.shape[-2:] != latents.shape[-2:]:
            target = torch.nn.functional.interpolate(target, size=latents.shape[-2:], mode="bilinear",
                                                     align_corners=False)
        t = torch.randint(self.min_step, self.max_step + 1, [1], device=latents.device)
        a = self.alphas[t]
        w = torch.sqrt(a) * (1 - a)
        noise = torch.randn_like(latents) * self.noise_scale
        return w * (latents - target + noise)


class StableDiffusionGuidance(Guidance):
    """Thin adapter for a machine that has `diffusers` and a local SD-1.x checkpoint directory.  Not used by
    tests or the benchmark.  Mirrors src/stable_diffusion.py:248-334 (latent mode)."""

    def __init__(self, device, model_path, guidance_scale=100.0):
        try:
            from diffusers import AutoencoderKL, PNDMScheduler, UNet2DConditionModel  # noqa: F401
            from transformers import CLIPTextModel, CLIPTokenizer  # noqa: F401
        except Exception as e:  # pragma: no cover
            raise RuntimeError("StableDiffusionGuidance needs `diffusers` + `transformers` and a LOCAL checkpoint "
                               "directory (no network access here): %s" % e)
        from diffusers import PNDMScheduler, UNet2DConditionModel
        from transformers import CLIPTextModel, CLIPTokenizer
        self.device = device
        self.guidance_scale = guidance_scale
        self.tokenizer = CLIPTokenizer.from_pretrained(model_path, subfolder="tokenizer", local_files_only=True)
        self.text_encoder = CLIPTextModel.from_pretrained(model_path, subfolder="text_encoder",
                                                          local_files_only=True).to(device)
        self.unet = UNet2DConditionModel.from_pretrained(model_path, subfolder="unet", local_files_only=True).to(device)
        self.scheduler = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
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


def sparsity_loss(weights_sum, eps=1e-5):
    """Entropy of the per-ray opacity (pushes rays to be fully empty or fully opaque)."""
    p = weights_sum.clamp(eps, 1.0 - eps)
    return (-p * torch.log2(p) - (1 - p) * torch.log2(1 - p)).mean()
