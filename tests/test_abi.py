"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mrgfe.h declares; with no GPU the
product path fails loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="mrgfe.h", testing=None):
    """Function names a header declares; testing=None: all of them, True / False: only those inside / outside `#ifdef MRGFE_TESTING`."""
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    m = re.search(r"#ifdef MRGFE_TESTING(.*?)#endif", txt, flags=re.S)
    inside = m.group(1) if m else ""
    if testing is True:
        txt = inside
    elif testing is False and m:
        txt = txt[: m.start()] + txt[m.end():]
    return sorted(set(re.findall(r"\b(mrgfe_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mrg_slam_amd import _lib

    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 45
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"libmrgfe.so lacks symbols declared in include/mrgfe.h: {missing}"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == declared
    assert b"gfx950" in L.mrgfe_version()
    # the integrator's header carries no diagnostic entry point (21 mrgfe_dbg_* declarations sat in it)
    assert not [s for s in declared if s.startswith("mrgfe_dbg_")]


def test_debug_header_and_the_testing_variant():
    """include/mrgfe_debug.h: the diagnostic entry points (exported by the shipped library, bound by the ctypes table) and, under MRGFE_TESTING, the two
    fault injectors — which the shipped libmrgfe.so must NOT export and mrg_slam_amd/libmrgfe_testing.so must."""
    from mrg_slam_amd import _lib

    _lib.build()
    diag, inject = _declared_symbols("mrgfe_debug.h", testing=False), _declared_symbols("mrgfe_debug.h", testing=True)
    assert sorted(_lib.DEBUG_SIGNATURES) == diag and all(s.startswith("mrgfe_dbg_") for s in diag) and len(diag) >= 15
    assert sorted(_lib.TESTING_SIGNATURES) == inject == ["mrgfe_dbg_fail_alloc_after", "mrgfe_dbg_node_fail_member"]
    shipped = C.CDLL(os.path.join(ROOT, "mrg_slam_amd", "libmrgfe.so"))
    testing = C.CDLL(_lib.TESTING_LIB_PATH)
    for s in diag:
        assert hasattr(shipped, s) and hasattr(testing, s), s
    for s in inject:
        assert not hasattr(shipped, s), f"the shipped library exports the fault injector {s}"
        assert hasattr(testing, s), s
    for s in _declared_symbols():
        assert hasattr(testing, s), s


def test_struct_layouts_match_the_header():
    from mrg_slam_amd import _lib

    assert C.sizeof(_lib.PairResult) == 384
    p = _lib.RegParams()
    _lib.lib().mrgfe_reg_default_params(_lib.NDT_HIP, C.byref(p))
    # defaults documented next to the get_parameter calls of registrations.cpp:34-43
    assert (p.transformation_epsilon, p.maximum_iterations, p.max_correspondence_distance, p.correspondence_randomness) == (0.01, 64, 2.0, 20)
    assert (p.resolution, p.nn_search_method, p.step_size, p.outlier_ratio, p.rotation_epsilon) == (1.0, _lib.SEARCH["DIRECT7"], 0.1, 0.55, 2e-3)


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mrg_slam_amd import Context, MrgfeError

    with pytest.raises(MrgfeError, match="no CPU fallback"):
        Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mrg_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src and '#include "../../oracle' not in src, f


def test_every_entry_point_is_placed_in_the_integration_table():
    """INTEGRATION.md §1 says for every entry point of include/mrgfe.h which reference call it stands for (or that it has none): by name, by a
    `prefix_*` row, or in a `name_a/b/c` row."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "mrgfe.h")).read()
    names = sorted(set(re.findall(r"\b(mrgfe_[a-z0-9_]+)\s*\(", header)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    wild = re.findall(r"`(mrgfe_[a-z0-9_]*)\*`", doc)
    slashed = set()
    for m in re.finditer(r"`(mrgfe_[a-z0-9_]+)((?:/[a-z0-9_]+)+)`", doc):  # `mrgfe_map_store_create/add/has`: the parts replace the first name's last word
        prefix = m.group(1)[: m.group(1).rindex("_") + 1]
        slashed.update(prefix + part for part in m.group(2).strip("/").split("/"))
    missing = [n for n in names if n not in doc and n not in slashed and not any(n.startswith(w) for w in wild)]
    assert not missing, missing
