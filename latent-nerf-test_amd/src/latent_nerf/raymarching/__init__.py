from .raymarching import (MarchResult, compact_rays, composite_rays, composite_rays_train, get_rays, march_rays,
                          march_rays_train, morton3D, morton3D_invert, near_far_from_aabb, packbits)

__all__ = ["MarchResult", "compact_rays", "composite_rays", "composite_rays_train", "get_rays", "march_rays",
           "march_rays_train", "morton3D", "morton3D_invert", "near_far_from_aabb", "packbits"]
