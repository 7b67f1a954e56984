"""The C-ABI library builds for gfx950 without a GPU, loads, and exports exactly the entry
points include/lnerf_hip.h declares.  Only host-side argument validation is exercised here:
no kernel is launched on a CPU-only machine."""
import ctypes
import os
import re
import subprocess

import pytest

from src.latent_nerf.raymarching import backend as B


def test_header_symbols_all_exported(built_lib):
    declared = B.header_symbols()
    assert len(declared) >= 25
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True)
    exported = set(re.findall(r"\b(lnerf_[a-z0-9_]+)\b", out))
    missing = [s for s in declared if s not in exported]
    assert not missing, "declared in include/lnerf_hip.h but not exported: %s" % missing
    # and the ctypes table of the host side covers the whole header
    assert sorted(B._SIGNATURES) == declared


def test_library_loads_and_reports_gfx950(built_lib):
    lib = B.get_lib()
    assert lib.lnerf_abi_version() == B.ABI_VERSION == 7
    assert lib.lnerf_build_info().decode().startswith("gfx950;")
    assert lib.lnerf_mlp_backward_workspace_bytes(5) > 0


def test_code_object_targets_gfx950(built_lib):
    blob = open(built_lib, "rb").read()
    assert b"gfx950" in blob
    assert b"gfx90a" not in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_argument_validation_fails_loudly(built_lib):
    lib = B.get_lib()
    P = ctypes.c_void_p
    # packbits: n_cells not a multiple of 8
    rc = lib.lnerf_packbits(P(16), 7, 0.5, None, P(16), None)
    assert rc == -1 and b"multiple of 8" in lib.lnerf_last_error()
    # march: cascade out of range
    rc = lib.lnerf_march_rays_train(P(16), P(16), P(16), P(16), 4, P(16), 1.0, 0, 128, 1024, 0.0, None, 0, None, 64, P(16),
                                    P(16), P(16), P(16), P(16), None)
    assert rc == -1 and b"cascade" in lib.lnerf_last_error()
    # march: a noise table AND a noise counter
    rc = lib.lnerf_march_rays_train(P(16), P(16), P(16), P(16), 4, P(16), 1.0, 1, 128, 1024, 0.0, P(16), 7, P(16), 64,
                                    P(16), P(16), P(16), P(16), P(16), None)
    assert rc == -1 and b"either a noise table or a noise counter" in lib.lnerf_last_error()
    # composite: unsupported channel count
    rc = lib.lnerf_composite_rays_train_forward(P(16), P(16), P(16), P(16), 4, 7, 1e-4, None, P(16), P(16), P(16), None)
    assert rc == -1 and b"C must be 3 or 4" in lib.lnerf_last_error()
    # grid encode: level_dim other than 2
    offs = (ctypes.c_int32 * 3)(0, 8, 16)
    sc = (ctypes.c_float * 2)(1.0, 2.0)
    rs = (ctypes.c_int32 * 2)(2, 3)
    rc = lib.lnerf_grid_encode_forward(P(16), 1.0, P(16), 0, 2, 4, offs, sc, rs, 8, None, 8, P(16), 0, 0, None)
    assert rc == -1 and b"level_dim" in lib.lnerf_last_error()
    with pytest.raises(B.LnerfError):
        B.call("lnerf_adam_step", P(16), P(16), B.F32, P(16), P(16), None, 8, 1e-3, 0.9, 0.99, 1e-15, 0, None, 1.0, 1, None)


def test_ops_refuse_cpu_tensors(built_lib):
    import torch
    from src.latent_nerf.raymarching import raymarching as rm
    with pytest.raises(ValueError, match="no CPU path"):
        rm.near_far_from_aabb(torch.zeros(4, 3), torch.ones(4, 3), [-1, -1, -1, 1, 1, 1], 0.1)


def test_ctypes_signatures_match_header():
    """Every argtypes list of the host side agrees with the C declaration in include/lnerf_hip.h
    (count and kind of every parameter), so a drifted binding fails here, on the CPU."""
    import ctypes as C
    text = open(B.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    kinds = {C.c_void_p: "ptr", C.c_char_p: "ptr", C.c_int: "int", C.c_int64: "int64", C.c_float: "float",
             C.c_size_t: "size_t", C.c_uint32: "uint32"}
    n = 0
    for m in re.finditer(r"\b(lnerf_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        want = []
        if args not in ("void", ""):
            for a in args.split(","):
                a = a.strip()
                if "*" in a or "lnerf_stream_t" in a:
                    want.append("ptr")
                elif a.startswith("int64_t"):
                    want.append("int64")
                elif a.startswith("size_t"):
                    want.append("size_t")
                elif a.startswith("float"):
                    want.append("float")
                elif a.startswith("uint32_t"):
                    want.append("uint32")
                elif a.startswith("int ") or a.startswith("int32_t"):
                    want.append("int")
                else:
                    raise AssertionError("unparsed parameter %r of %s" % (a, name))
        got = [kinds[t] for t in B._SIGNATURES[name]]
        assert got == want, (name, got, want)
        n += 1
    assert n == len(B._SIGNATURES)


def test_scatter_planning_and_new_entry_points_validate(built_lib):
    """Host-side planning of the bucketed scatter and the argument checks of the fused backward+Adam entry."""
    from src.latent_nerf.models.encoding import GridLevels
    lib = B.get_lib()
    P = ctypes.c_void_p
    lv = GridLevels()
    ws = [lib.lnerf_grid_encode_backward_workspace_bytes(lv.num_levels, lv.c_offsets, m) for m in (1 << 14, 1 << 18, 1 << 20)]
    assert 0 < ws[0] < ws[1] < ws[2]
    # capacity 2^20: >= 8 records x 12 B per (sample, level) plus the partial-sum tiles of the sliced coarse levels
    assert ws[2] >= (1 << 20) * 16 * 8 * 12
    assert lib.lnerf_grid_encode_backward_workspace_bytes(0, lv.c_offsets, 1024) == 0
    args = [P(16), 1.0, P(16), B.F32, lv.num_levels, 2, lv.c_offsets, lv.c_scales, lv.c_res, 1024, None, 1024]
    # the fused form needs the bucketed scatter ...
    rc = lib.lnerf_grid_encode_backward_adam(*args, P(16), 0, P(16), 1 << 30, P(16), P(16), P(16), None, 1e-3, 0.9, 0.99,
                                             1e-15, 1, None, 1.0, None)
    assert rc == -1 and b"variant 2/3" in lib.lnerf_last_error()
    # ... optimiser state, and a step number
    rc = lib.lnerf_grid_encode_backward_adam(*args, P(16), 2, P(16), 1 << 30, None, P(16), P(16), None, 1e-3, 0.9, 0.99,
                                             1e-15, 1, None, 1.0, None)
    assert rc == -1 and b"optimiser state" in lib.lnerf_last_error()
    rc = lib.lnerf_grid_encode_backward_adam(*args, P(16), 2, P(16), 1 << 30, P(16), P(16), P(16), None, 1e-3, 0.9, 0.99,
                                             1e-15, 0, None, 1.0, None)
    assert rc == -1 and b"step must be >= 1" in lib.lnerf_last_error()
    # a workspace that is too small is refused before anything is launched
    rc = lib.lnerf_grid_encode_backward(*args, P(16), 2, P(16), 4096, None)
    assert rc == -1 and b"workspace too small" in lib.lnerf_last_error()
    # tuning keys
    assert lib.lnerf_set_tuning(b"scatter_bin_wgs", 768) == 0 and lib.lnerf_set_tuning(b"scatter_bin_wgs", 0) == 0
    assert lib.lnerf_set_tuning(b"scatter_bin_per_cu", 7) == -1 and b"scatter_bin_per_cu" in lib.lnerf_last_error()
    assert lib.lnerf_set_tuning(b"no_such_knob", 1) == -1 and b"unknown key" in lib.lnerf_last_error()
    # the round-1 switch that produced wrong sums for timing experiments is gone from the library
    assert lib.lnerf_set_tuning(b"scatter_bin_debug", 0) == -1 and b"unknown key" in lib.lnerf_last_error()


def _device_asm(src):
    """gfx950 assembly of one HIP source (device side only), as text."""
    import subprocess
    import tempfile
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "latent-nerf-test_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dev.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               '-DLNERF_BUILD_TAG="asm"', "--cuda-device-only", "-S", "-o", out, os.path.join(csrc, src)],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_cross_workgroup_handoffs_are_scoped_accesses():
    """The scatter's pass 2 hands partial tiles between workgroups on different XCDs with RELAXED device-scope accesses
    plus a hand-written s_waitcnt (csrc/grid.hip, "ORDERING"): correct on gfx950 because the stores are write-through
    (sc1), acknowledged before the arrival atomic is issued, and the last arriver's loads bypass its L2 (sc1) -- not
    because of the C++ memory orders.  This pins the ISA the argument rests on: the instructions are there, and no
    agent-scope fence (an L2 write-back / invalidate per workgroup: 2.6 x the pass's time) has crept in."""
    import re
    asm = _device_asm("grid.hip")
    kernels = re.findall(r"^(_ZN5lnerf16k_scatter_reduceILi1024E\w+):.*?\n(.*?)^\.Lfunc_end", asm, flags=re.S | re.M)
    fused = [(n, body) for n, body in kernels if "Lb1E" in n]          # FUSE = true: the closing form
    assert len(fused) == 2                                            # Rec8 and Rec12
    for name, body in fused:
        lines = [l.strip() for l in body.splitlines()]
        st = [i for i, l in enumerate(lines) if l.startswith("global_store_dwordx2") and l.endswith("sc1")]
        ld = [i for i, l in enumerate(lines) if l.startswith("global_load_dwordx2") and l.endswith("sc1")]
        at = [i for i, l in enumerate(lines) if l.startswith("global_atomic_add") and "sc0" in l]
        assert len(st) >= 4 and len(ld) >= 4 and len(at) >= 2, (name, len(st), len(ld), len(at))
        assert not any(l.startswith(("buffer_wbl2", "buffer_inv")) for l in lines), name
        # tile stores -> s_waitcnt vmcnt(0) -> barrier -> the slice's arrival atomic -> ... -> tile loads
        arrive = min(i for i in at if i > st[-1])
        between = lines[st[-1]:arrive]
        assert any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in between), name
        assert any(l.startswith("s_barrier") for l in between), name
        assert ld[0] > arrive, name
        # the level maximum is ONE device-scope load (the closing arrival zeroes it later in the same launch)
        assert any(l.startswith("global_load_dword ") and l.endswith("sc1") for l in lines), name


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_kernel_of_the_product_build_spills():
    """Scratch traffic shares the vector-memory queue with the loads a streaming loop waits for: a spilling kernel is a
    slow kernel on this path.  The product build (no -DLNERF_EXPERIMENTS) compiles every kernel without scratch."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("resource_usage", os.path.join(root, "tools", "resource_usage.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    rows = ru.usage()
    assert len(rows) > 50
    assert [r["kernel"] for r in rows if r["scratch"] > 0] == []
    by = {r["kernel"]: r for r in rows}
    assert by["k_scatter_reduce<1024, Rec8, true>"]["Occupancy [waves/SIMD]"] == 8      # two 1024-thread workgroups per CU
    assert by["k_scatter_reduce<1024, Rec12, true>"]["Occupancy [waves/SIMD]"] == 8
