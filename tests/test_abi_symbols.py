"""CPU tier: the C-ABI library is built, loads, and exports exactly what include/pebblegpu.h declares.
No compute call is made here (there is no GPU in the build container and the library has no CPU path)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pebblegpu.h")


def header_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(pebblegpu_[a-z0-9_]+)\s*\(", txt)
    seen, out = set(), []
    for n in names:
        if n not in seen:
            seen.add(n)
            out.append(n)
    return out


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__ as g
    return g.build()


def test_every_declared_symbol_is_exported(lib_path):
    L = ctypes.CDLL(lib_path)
    fns = header_functions()
    assert len(fns) >= 40
    missing = [f for f in fns if not hasattr(L, f)]
    assert not missing, missing


def test_binding_symbol_list_matches_header():
    from pebblesdr_amd.binding import SYMBOLS
    assert sorted(SYMBOLS) == sorted(header_functions())


def test_abi_version_and_error_text(lib_path):
    L = ctypes.CDLL(lib_path)
    assert L.pebblegpu_abi_version() == 1
    L.pebblegpu_last_error.restype = ctypes.c_char_p
    assert isinstance(L.pebblegpu_last_error(), bytes)


def test_code_object_targets_gfx950_only(lib_path):
    """The fat binary section names its offload targets; gfx950 must be the only GPU ISA in the library."""
    blob = open(lib_path, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_device_is_a_loud_failure_not_a_fallback(lib_path):
    """In the GPU-less container create() must fail with PEBBLEGPU_E_NO_DEVICE; on a GPU box this test is moot."""
    import pebblesdr_amd as P
    L = P.load_library()
    if L.pebblegpu_device_count() > 0:
        pytest.skip("a device is visible")
    with pytest.raises(P.PebbleGpuError) as e:
        P.ReceiverBank(2048000, 1)
    assert e.value.code == -2
    with pytest.raises(P.PebbleGpuError):
        P.Mixer(2048000, 2048)


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under pebblesdr_amd/ or include/ may import, include or link it."""
    bad = []
    for base in ("pebblesdr_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"pebble_oracle|import oracle|from oracle|oracle/|hipemu|hip_emu", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
