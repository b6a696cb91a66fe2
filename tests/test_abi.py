"""CPU-only: the C-ABI library loads, exports every symbol include/re2e.h declares, and the ctypes
signature table agrees with the header's argument counts.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, 'include', 're2e.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(re2e_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ('', 'void') else args.count(',') + 1
        decls[name] = n
    return decls


def test_library_exports_every_declared_symbol():
    from robust_e2e_gan_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    so = ctypes.CDLL(lib.LIB_PATH)
    decls = _header_decls()
    assert len(decls) >= 50
    for name in decls:
        assert hasattr(so, name), 'symbol %s declared in include/re2e.h is not exported' % name
    assert so.re2e_version() >= 100


def test_ctypes_table_matches_header():
    from robust_e2e_gan_amd import lib
    decls = _header_decls()
    assert set(decls) == set(lib.SIGNATURES), set(decls) ^ set(lib.SIGNATURES)
    for name, n in decls.items():
        assert len(lib.SIGNATURES[name][1]) == n, (name, n, len(lib.SIGNATURES[name][1]))


def test_missing_library_fails_loudly(monkeypatch):
    from robust_e2e_gan_amd import lib
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', '/nonexistent/libre2e_hip.so')
    with pytest.raises(lib.Re2eError):
        lib.load()


def test_ops_refuse_cpu_tensors():
    """The product path has no CPU fallback: CPU tensors raise instead of silently computing."""
    import torch
    from robust_e2e_gan_amd import ops, lib
    with pytest.raises(lib.Re2eError):
        ops.linear(torch.zeros(2, 3), torch.nn.Parameter(torch.zeros(4, 3)), None, None)
    with pytest.raises(lib.Re2eError):
        ops.mean_loss(torch.zeros(4), torch.zeros(4))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'robust_e2e_gan_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dp, f)
