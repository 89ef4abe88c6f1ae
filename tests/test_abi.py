"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/mashmap_hip.h declares.  No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "mashmap_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from mashmap_amd import capi
    assert _declared() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    from mashmap_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    capi.load()
    assert capi.load().mm_abi_version() == 2      # include/mashmap_hip.h MM_ABI_VERSION (2: five pass counts, staged flag, mm_reads_prefetch_drop)


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mashmap_amd import capi
    with pytest.raises(capi.MashmapError):
        capi.Context()
