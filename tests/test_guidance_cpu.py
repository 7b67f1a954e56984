"""Guidance adapter: decode_latents / encode_imgs of the real-checkpoint adapter restated from
/root/reference/src/stable_diffusion.py:462-489, driven with STUB model classes (diffusers is not installed here and
nothing may be fetched), and the decoder fall-back both trainers use for a guidance object that has no decoder."""
import types

import pytest
import torch

from src.latent_nerf.training import guidance as G


class _Cfg:
    model_max_length = 77


class _Tok:
    model_max_length = 77

    @classmethod
    def from_pretrained(cls, path, subfolder=None, local_files_only=False):
        assert local_files_only, "the adapter must never fetch"
        return cls()

    def __call__(self, texts, **kw):
        return types.SimpleNamespace(input_ids=torch.zeros(len(texts), 77, dtype=torch.long))


class _TextEnc(torch.nn.Module):
    @classmethod
    def from_pretrained(cls, path, subfolder=None, local_files_only=False):
        assert local_files_only
        return cls()

    def forward(self, ids):
        return (torch.ones(ids.shape[0], 77, 768),)


class _UNet(torch.nn.Module):
    @classmethod
    def from_pretrained(cls, path, subfolder=None, local_files_only=False):
        assert local_files_only
        return cls()

    def forward(self, x, t, encoder_hidden_states=None):
        return types.SimpleNamespace(sample=0.5 * x)


class _Sched:
    def __init__(self, **kw):
        betas = torch.linspace(kw["beta_start"] ** 0.5, kw["beta_end"] ** 0.5, kw["num_train_timesteps"]) ** 2
        self.alphas_cumprod = torch.cumprod(1 - betas, 0)

    def add_noise(self, x, noise, t):
        a = self.alphas_cumprod[t]
        return a.sqrt() * x + (1 - a).sqrt() * noise


class _VAE(torch.nn.Module):
    """decode: nearest 8x up-sampling of the first three channels; encode: 8x average pooling, fourth channel zero."""
    seen = {}

    @classmethod
    def from_pretrained(cls, path, subfolder=None, local_files_only=False):
        assert local_files_only and subfolder == "vae"
        return cls()

    def decode(self, z):
        _VAE.seen["decode_in"] = z.clone()
        return types.SimpleNamespace(sample=torch.nn.functional.interpolate(z[:, :3], scale_factor=8, mode="nearest"))

    def encode(self, x):
        _VAE.seen["encode_in"] = x.clone()
        z = torch.nn.functional.avg_pool2d(x, 8)
        z = torch.cat([z, torch.zeros_like(z[:, :1])], 1)
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: z))


class _NoVAE:
    @classmethod
    def from_pretrained(cls, *a, **k):
        raise OSError("no vae directory")


def _modules(vae=_VAE):
    return {"AutoencoderKL": vae, "UNet2DConditionModel": _UNet, "PNDMScheduler": _Sched, "CLIPTextModel": _TextEnc,
            "CLIPTokenizer": _Tok}


def test_decode_and_encode_follow_the_reference_scaling():
    g = G.StableDiffusionGuidance(torch.device("cpu"), "/nonexistent", modules=_modules())
    lat = torch.randn(1, 4, 8, 8)
    rgb = g.decode_latents(lat)
    # src/stable_diffusion.py:462-470: latents / 0.18215 -> vae.decode -> (x / 2 + 0.5).clamp(0, 1)
    assert torch.allclose(_VAE.seen["decode_in"], lat / 0.18215)
    want = (torch.nn.functional.interpolate(lat[:, :3] / 0.18215, scale_factor=8, mode="nearest") / 2 + 0.5).clamp(0, 1)
    assert rgb.shape == (1, 3, 64, 64) and torch.equal(rgb, want)
    imgs = torch.rand(1, 3, 64, 64)
    z = g.encode_imgs(imgs)
    # :482-489: 2 imgs - 1 -> vae.encode -> posterior sample * 0.18215
    assert torch.allclose(_VAE.seen["encode_in"], 2 * imgs - 1)
    assert z.shape == (1, 4, 8, 8)
    assert torch.allclose(z[:, :3], torch.nn.functional.avg_pool2d(2 * imgs - 1, 8) * 0.18215)


def test_train_step_returns_weighted_noise_residual():
    torch.manual_seed(0)
    g = G.StableDiffusionGuidance(torch.device("cpu"), "/nonexistent", guidance_scale=7.5, modules=_modules())
    tz = g.get_text_embeds("a teddy bear")
    assert tz.shape == (2, 77, 768)
    lat = torch.randn(1, 4, 8, 8)
    grad = g.train_step(tz, lat)
    assert grad.shape == lat.shape and torch.isfinite(grad).all() and not grad.requires_grad


def test_checkpoint_without_vae_falls_back_to_the_linear_preview():
    g = G.StableDiffusionGuidance(torch.device("cpu"), "/nonexistent", modules=_modules(_NoVAE))
    lat = torch.randn(1, 4, 8, 8)
    assert g.vae is None
    assert torch.equal(g.decode_latents(lat), G.linear_decode_latents(lat))
    with pytest.raises(NotImplementedError):
        g.encode_imgs(torch.rand(1, 3, 64, 64))


def test_decode_with_handles_objects_without_a_decoder():
    class Bare:                      # anything with train_step is a guidance object
        def train_step(self, text_z, latents):
            return torch.zeros_like(latents)

    class Abstract(G.Guidance):
        def decode_latents(self, latents):
            raise NotImplementedError()

    lat = torch.randn(2, 4, 4, 4)
    ref = G.linear_decode_latents(lat)
    assert ref.shape == (2, 3, 32, 32) and float(ref.min()) >= 0 and float(ref.max()) <= 1
    for obj in (Bare(), Abstract(), G.Guidance()):
        assert torch.equal(G.decode_with(obj, lat), ref)
    # the linear map is the reference's preview matrix (src/latent_paint/models/textured_mesh.py:34-40)
    one = torch.zeros(1, 4, 1, 1)
    one[0, 0] = 1.0
    rgb = G.linear_decode_latents(one, upsample=1)[0, :, 0, 0]
    assert torch.allclose(rgb, torch.tensor([0.298, 0.207, 0.208]) / 2 + 0.5)
