"""ctypes loader for liblnerf_hip.so (C ABI: include/lnerf_hip.h).

There is no CPU fallback: every op in this package runs on the HIP library or raises.
`get_lib()` raises `LnerfLibraryError` with build instructions when the library is missing."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.abspath(os.path.join(_HERE, "..", "..", ".."))           # latent-nerf-test_amd/
REPO_ROOT = os.path.abspath(os.path.join(PKG_ROOT, ".."))
LIB_PATH = os.environ.get("LNERF_HIP_LIB", os.path.join(PKG_ROOT, "lib", "liblnerf_hip.so"))
HEADER_PATH = os.path.join(REPO_ROOT, "include", "lnerf_hip.h")

LNERF_OK = 0
F32, BF16 = 0, 1
MLP_FRAGMENTS_READY = 0x100   # flag on lnerf_mlp_backward's precision tag (include/lnerf_hip.h)
SCATTER_ZERO_HEAD_BYTES = 64 * 1024  # LNERF_SCATTER_ZERO_HEAD_BYTES: head of a fresh scatter workspace that must be zero
MLP_FRAGMENT_BYTES = 36 * 1024  # LNERF_MLP_FRAGMENT_BYTES: the bf16 weight-fragment image at the head of the MLP workspace
SCATTER_DEFER_FINISH = 0x200  # flag on the scatter's variant: lnerf_step_tail runs the finishing pass
MLP_DEFER_REDUCE = 0x200      # flag on lnerf_mlp_backward's precision tag: lnerf_step_tail sums the slabs
TAIL_TICK, TAIL_CLEAR_SCATTER = 1, 2
GRID_BLOCKED = 0x400          # flag on the gather's / scatter's variant: blocked layout of the hashed levels
GRID_TILED = 0x800            # ... the upstream's `tiled` layout (dense index wrapped instead of hashed)
SCATTER_CLEARED = 0x100       # flag on the scatter's variant: the caller cleared the cursors (lnerf_grid_scatter_clear_bytes)


class LnerfLibraryError(RuntimeError):
    pass


class LnerfError(RuntimeError):
    pass


_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float
_Z = _c.c_size_t
_U = _c.c_uint32

ABI_VERSION = 7  # LNERF_ABI_VERSION of include/lnerf_hip.h this host side was written against

# name -> argtypes (return type int unless listed in _RESTYPES)
_SIGNATURES = {
    "lnerf_abi_version": [],
    "lnerf_last_error": [],
    "lnerf_build_info": [],
    "lnerf_set_tuning": [_c.c_char_p, _I],
    "lnerf_get_rays": [_P, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P],
    "lnerf_near_far_from_aabb": [_P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P],
    "lnerf_morton3d": [_P, _L, _P, _P],
    "lnerf_morton3d_invert": [_P, _L, _P, _P],
    "lnerf_packbits": [_P, _L, _F, _P, _P, _P],
    "lnerf_march_rays_train": [_P, _P, _P, _P, _L, _P, _F, _I, _I, _I, _F, _P, _U, _P, _L, _P, _P, _P, _P, _P, _P],
    "lnerf_march_rays_train_aabb": [_P, _P, _F, _F, _F, _F, _F, _F, _F, _L, _P, _F, _I, _I, _I, _F, _P, _U, _P, _L, _P, _P,
                                    _P, _P, _P, _P],
    "lnerf_march_rays_train_pose": [_P, _I, _I, _I, _F, _F, _F, _F, _P, _P, _F, _F, _F, _F, _F, _F, _F, _P, _F, _I, _I, _I,
                                    _F, _P, _U, _P, _L, _P, _P, _P, _P, _P, _P],
    "lnerf_march_rays_train_camera": [_P, _P, _I, _I, _I, _P, _P, _F, _F, _F, _F, _F, _F, _F, _P, _F, _I, _I, _I, _F, _P, _U,
                                      _P, _L, _P, _P, _P, _P, _P, _P],
    "lnerf_march_rays": [_L, _I, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _F, _P, _P, _P, _P],
    "lnerf_composite_rays": [_L, _I, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P],
    "lnerf_compact_rays": [_P, _L, _P, _P, _P],
    "lnerf_grid_encode_forward": [_P, _F, _P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P, _I, _I, _P],
    "lnerf_grid_encode_backward_workspace_bytes": [_I, _P, _L],
    "lnerf_grid_encode_backward": [_P, _F, _P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P, _I, _P, _Z, _P],
    "lnerf_grid_encode_backward_bf16": [_P, _F, _P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P, _I, _P, _Z, _P, _P],
    "lnerf_grid_scatter_bin": [_P, _F, _P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P, _I, _P, _Z, _P],
    "lnerf_grid_scatter_reduce_bf16": [_F, _I, _I, _P, _P, _P, _L, _L, _I, _I, _P, _I, _P, _Z, _P, _P],
    "lnerf_grid_encode_backward_adam": [_P, _F, _P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P, _I, _P, _Z, _P, _P, _P, _P, _F,
                                        _F, _F, _F, _I, _P, _F, _P],
    "lnerf_mlp_forward": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _L, _P, _P, _P, _I, _P, _Z, _P],
    "lnerf_mlp_backward_workspace_bytes": [_I],
    "lnerf_grid_scatter_clear_bytes": [_I, _P, _L],
    "lnerf_occ_sample_scratch_bytes": [_L],
    "lnerf_march_counter_len": [_L],
    "lnerf_occ_sample": [_P, _L, _I, _I, _F, _L, _U, _U, _P, _P, _P, _P],
    "lnerf_mlp_backward": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _L, _P, _P, _P, _P, _P, _P, _P, _P,
                           _P, _P, _P, _I, _P, _Z, _I, _P, _Z, _P],
    "lnerf_composite_rays_train_forward": [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P, _P],
    "lnerf_composite_rays_train_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P],
    "lnerf_opacity_entropy_grad": [_P, _L, _F, _F, _P, _P],
    "lnerf_synthetic_guidance": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _F, _U, _P, _P, _P, _F, _F, _P, _P],
    "lnerf_occ_cell_points": [_P, _L, _I, _I, _F, _P, _P, _P],
    "lnerf_occ_update": [_P, _P, _L, _P, _F, _P, _P],
    "lnerf_occ_mean": [_P, _L, _P, _P, _P],
    "lnerf_occ_update_mean": [_P, _L, _P, _L, _P, _F, _P, _P, _P, _P],
    "lnerf_bg_forward": [_P, _L, _P, _P, _P, _P, _I, _P, _P],
    "lnerf_bg_backward": [_P, _L, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "lnerf_mesh_winding_number": [_P, _L, _P, _I, _P, _P],
    "lnerf_mesh_distance": [_P, _L, _P, _I, _P, _P],
    "lnerf_raster_prepare": [_P, _I, _P, _I, _P, _P, _P, _P],
    "lnerf_rasterize": [_I, _I, _P, _P, _I, _P, _P, _P],
    "lnerf_interpolate_attributes": [_P, _P, _P, _I, _I, _P, _P],
    "lnerf_interpolate_attributes_backward": [_P, _P, _P, _I, _I, _P, _P],
    "lnerf_texture_map_forward": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "lnerf_texture_map_backward": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "lnerf_adam_step": [_P, _P, _I, _P, _P, _P, _L, _F, _F, _F, _F, _I, _P, _F, _I, _P],
    "lnerf_adam_tick": [_P, _P],
    "lnerf_adam_step_multi": [_I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _P, _F, _I, _P],
    "lnerf_adam_step_multi_shadow": [_I, _P, _P, _P, _P, _P, _P, _F, _F, _F, _I, _P, _F, _I, _P, _P, _P],
    "lnerf_mlp_fragment_maps": [_I, _P, _P, _P, _P],
    "lnerf_cast_f32_to_bf16": [_P, _P, _L, _P],
    "lnerf_mlp_backward_slabs": [_L, _I],
    "lnerf_grid_encode_backward_adam_tail": [_P, _F, _P, _I, _I, _I, _P, _P, _P, _L, _P, _L, _P, _I, _P, _Z, _P, _P, _P, _P,
                                             _F, _P, _Z, _I, _I, _P, _P, _P, _F, _P, _F, _F, _F, _I, _P, _F, _I, _P],
    "lnerf_step_tail": [_I, _I, _P, _P, _P, _L, _I, _P, _Z, _P, _P, _P, _P, _P, _F, _P, _Z, _I, _I, _P, _P, _P, _F, _P,
                        _F, _F, _F, _I, _P, _F, _I, _P],
}
_RESTYPES = {
    "lnerf_last_error": _c.c_char_p,
    "lnerf_build_info": _c.c_char_p,
    "lnerf_mlp_backward_workspace_bytes": _Z,
    "lnerf_grid_scatter_clear_bytes": _Z,
    "lnerf_occ_sample_scratch_bytes": _Z,
    "lnerf_march_counter_len": _L,
    "lnerf_grid_encode_backward_workspace_bytes": _Z,
}

_lib = None


def header_symbols(path: str = HEADER_PATH):
    """Function names declared in include/lnerf_hip.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lnerf_[a-z0-9_]+)\s*\(", text)))


def library_available() -> bool:
    return os.path.exists(LIB_PATH)


def get_lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LnerfLibraryError(
            "HIP library %s is missing: run `python latent-nerf-test_amd/build.py` (needs hipcc, no GPU) "
            "or `python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LnerfLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name)) from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _c.c_int)
    if lib.lnerf_abi_version() != ABI_VERSION:
        raise LnerfLibraryError("ABI version mismatch: library %d, host side %d" % (lib.lnerf_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_hook = None


def set_profile_hook(fn):
    """fn(name, 'pre'|'post') around every C-ABI call (bench.py records HIP events with it)."""
    global _hook
    _hook = fn


def call(name: str, *args):
    """Invoke an int-returning entry point; raise LnerfError with the library's message on failure."""
    lib = get_lib()
    if _hook is not None:
        _hook(name, "pre")
        rc = getattr(lib, name)(*args)
        _hook(name, "post")
    else:
        rc = getattr(lib, name)(*args)
    if rc != LNERF_OK:
        msg = lib.lnerf_last_error().decode("utf-8", "replace")
        raise LnerfError("%s failed (%d): %s" % (name, rc, msg))
    return rc
