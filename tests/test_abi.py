"""CPU-only: the C-ABI library loads, exports every symbol include/re2e.h declares, and the ctypes
signature table agrees with the header's argument TYPES.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctype_of(decl):
    """C parameter (or return) declaration -> the ctypes class that must stand for it."""
    d = re.sub(r'\bconst\b', '', decl).strip()
    if '*' in d:                      # every pointer parameter is passed as an address (device pointers, host length arrays)
        return ctypes.c_void_p
    base = re.sub(r'\s+\w+$', '', d).strip() if re.search(r'\s\w+$', d) else d          # drop the parameter name
    table = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'size_t': ctypes.c_size_t,
             're2e_stream_t': ctypes.c_void_p, 'unsigned': ctypes.c_uint, 'double': ctypes.c_double,
             'unsigned long long': ctypes.c_ulonglong}
    assert base in table, 'include/re2e.h: unhandled parameter type %r' % decl
    return table[base]


def _header_decls():
    """name -> (restype, [argtypes...]) parsed from include/re2e.h."""
    src = open(os.path.join(ROOT, 'include', 're2e.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'([\w\s\*]+?)\b(re2e_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith('typedef') or not ret:
            continue
        ret = ret.split('\n')[-1].strip()
        params = [] if args in ('', 'void') else [a.strip() for a in args.split(',')]
        res = ctypes.c_char_p if ret.replace(' ', '') == 'constchar*' else _ctype_of(ret + ' r')
        decls[name] = (res, [_ctype_of(a) for a in params])
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
    abi_macro = int(re.search(r'#define\s+RE2E_ABI_VERSION\s+(\d+)', open(os.path.join(ROOT, 'include', 're2e.h')).read()).group(1))
    assert so.re2e_version() == abi_macro == lib.ABI_VERSION, 'library, header and ctypes table must carry the same ABI version'


def test_ctypes_table_matches_header():
    """Argument by argument: the ctypes class in lib.SIGNATURES must be the one the header's C type maps to (pointer ->
    c_void_p, int -> c_int, long -> c_long, float -> c_float, size_t -> c_size_t), and so must the return type -- with
    up to 27 positional arguments a swapped int/long or float/int pair would otherwise corrupt a call silently.  The
    last header parameter of every launching entry point is ``re2e_stream_t stream``; lib.SIGNATURES lists it too."""
    from robust_e2e_gan_amd import lib
    decls = _header_decls()
    assert set(decls) == set(lib.SIGNATURES), set(decls) ^ set(lib.SIGNATURES)
    for name, (res, args) in decls.items():
        tres, targs = lib.SIGNATURES[name]
        assert tres is res, (name, 'return', tres, res)
        assert len(targs) == len(args), (name, len(args), len(targs))
        for i, (a, b) in enumerate(zip(targs, args)):
            assert a is b, '%s: argument %d is %s in lib.SIGNATURES but %s in include/re2e.h' % (name, i, a.__name__, b.__name__)


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
