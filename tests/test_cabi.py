"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports every symbol
include/movedepth_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "movedepth_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for s in ["md_costvol_fwd", "md_costvol_bwd", "md_schedule_depth_range", "md_fuse_fwd", "md_fuse_bwd",
              "md_warp_fwd", "md_warp_bwd", "md_reproj_loss_fwd", "md_reproj_loss_bwd", "md_masked_min_fwd",
              "md_masked_min_bwd", "md_smooth_fwd", "md_smooth_bwd", "md_last_error"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    from movedepth_amd import _lib

    path = _lib.LIB_PATH
    if not os.path.exists(path):
        import __graft_entry__ as g

        g.build()
    lib = ctypes.CDLL(path)
    for s in header_symbols():
        assert hasattr(lib, s), "libmovedepth_hip.so does not export %s" % s
    # the ctypes binding covers the whole header, and nothing else
    assert sorted(_lib.SIGNATURES) == header_symbols()
    assert _lib.load().md_abi_version() == 17


def test_invalid_arguments_fail_loudly_without_a_gpu():
    """Argument validation happens before any launch, so it can be exercised on CPU."""
    from movedepth_amd import _lib

    lib = _lib.load()
    rc = lib.md_costvol_fwd(None, None, None, None, None, None, None, None, 0.3, 0, 1, 32, 16, 8, 8, 4, 0, None, 0, 0, 0, 0, 0, None)
    assert rc == -1
    assert b"null" in lib.md_last_error()
    with pytest.raises(_lib.MovedepthHipError):
        _lib.call("md_schedule_depth_range", None, None, 1, 4, 4, 8, 0.3, 0, None, None)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under movedepth_amd/ may reference it."""
    pkg = os.path.join(ROOT, "movedepth_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "movedepth_oracle" not in src, f


def test_ops_refuse_cpu_tensors():
    import torch

    from movedepth_amd import _lib, ops

    with pytest.raises(_lib.MovedepthHipError):
        ops.reprojection_loss(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8))


def test_graft_entry_build():
    """__graft_entry__.build() is the driver's "does it build" check: make (a no-op on an up-to-date tree), the C oracle, the import
    and its own ABI assertion must pass here -- a stale version number in it fails the round's build check, not a test."""
    import __graft_entry__

    __graft_entry__.build()
