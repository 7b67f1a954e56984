"""Host-side mirror of the reference's `src` package for the latent-NeRF path.

Only `src.latent_nerf` (absent from the reference checkout, imported by its
scripts/train_latent_nerf.py:3-4) and the `src.utils` helpers it needs are provided.
Put this directory's parent (`latent-nerf-test_amd/`) on sys.path, or copy `latent_nerf/`
into the reference's own `src/`."""
