"""CPU: libagl.so loads and exports exactly the entry points include/agl.h declares, and the ctypes table
in agl/lib.py covers them with matching arity.  No kernel is launched."""
import ctypes
import os
import re

from conftest import PKG, ROOT


def _header_decls():
    src = open(os.path.join(ROOT, "include", "agl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"struct AglSnLayer \{.*?\};", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"(?:const char\*|int|long|double)\s+(agl_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol():
    from agl import lib as L
    decls = _header_decls()
    assert len(decls) >= 40
    dll = ctypes.CDLL(L.LIB_PATH)
    for name in decls:
        assert hasattr(dll, name), f"{name} declared in include/agl.h but not exported by libagl.so"
    ver = int(re.search(r"#define\s+AGL_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "agl.h")).read()).group(1))
    assert dll.agl_version() == ver == L.ABI_VERSION, (dll.agl_version(), ver, L.ABI_VERSION)


def test_binding_table_matches_header():
    from agl import lib as L
    decls = _header_decls()
    assert set(L.SIGNATURES) == set(decls), set(L.SIGNATURES) ^ set(decls)
    for name, n in decls.items():
        assert len(L.SIGNATURES[name][1]) == n, (name, n, len(L.SIGNATURES[name][1]))
    lib = L.load()
    assert lib.agl_sn_layer_desc_bytes() == ctypes.sizeof(L.SnLayer)
    assert lib.agl_conv2d_bwd_weight_ws_bytes(4, 64, 128, 3, 32, 32) >= 0


def test_product_has_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of silently computing on the host."""
    import pytest
    import torch
    from agl import functional as F
    with pytest.raises(RuntimeError):
        F.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, 1, 1)
    from models.discriminator import ImageDiscriminator
    with pytest.raises(RuntimeError):
        ImageDiscriminator(conv_dim=8)(torch.zeros(1, 3, 64, 64))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dirpath, f)
