"""MI355X-native counterpart of the reference's (absent) `src.latent_nerf` package."""
