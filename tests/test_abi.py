"""CPU-side checks of the C ABI: the library loads, exports every symbol include/bodyslam_hip.h declares,
and the product path fails loudly (no CPU fallback) without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from bodyslam_amd import _lib
    return _lib


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "bodyslam_hip.h")).read()
    declared = set(re.findall(r"\b(bs_[a-z0-9_]+)\s*\(", hdr))
    lib = built.load_library()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert set(built.EXPORTS) == declared, (sorted(set(built.EXPORTS) ^ declared))


def test_version_and_error_string(built):
    lib = built.load_library()
    assert lib.bs_version() >= 1
    assert isinstance(lib.bs_last_error(), bytes)


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built.BodySlamHipError):
        built.init(0)


def test_gemm_desc_layout_matches_header(built):
    """ctypes struct field order must mirror bs_gemm_desc."""
    hdr = open(os.path.join(ROOT, "include", "bodyslam_hip.h")).read()
    body = hdr[hdr.index("typedef struct bs_gemm_desc {"):hdr.index("} bs_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.findall(r"(\w+)\s*$", first)[0])
        names += [r.strip() for r in rest]
    assert names == [f[0] for f in built.GemmDesc._fields_]
