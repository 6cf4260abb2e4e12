"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/lsfc.h declares, and the host mirror fails loudly (never silently falls back) when
there is no GPU.  No compute is attempted here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lsfc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsfc_[a-z0-9_]+)\s*\(", text)) - {"lsfc_precond_fn"})


def test_library_exports_every_declared_symbol():
    import fast_solver_lippmann_schwinger_amd._lib as L
    lib = ctypes.CDLL(L.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/lsfc.h but not exported"
    assert sorted(L.SIGNATURES) == names, "python binding and header disagree"


def test_binding_argument_counts_match_the_header():
    # every ctypes signature has exactly as many arguments as the C prototype it binds
    import fast_solver_lippmann_schwinger_amd._lib as L
    text = open(os.path.join(ROOT, "include", "lsfc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = dict(re.findall(r"\b(lsfc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    for name, (_, args) in L.SIGNATURES.items():
        params = protos[name].strip()
        n = 0 if params in ("", "void") else len([a for a in params.split(",") if a.strip()])
        assert n == len(args), f"{name}: header has {n} parameters, binding has {len(args)}"


def _ctype_class(t):
    """ctypes type -> one letter: what the C prototype must have in that position"""
    import ctypes as C
    if t is None:
        return "v"
    if t in (C.c_int,):
        return "i"
    if t in (C.c_int64,):
        return "l"
    if t in (C.c_double,):
        return "d"
    if t in (C.c_uint,):
        return "u"
    if t in (C.c_size_t,):
        return "z"
    return "p"              # c_void_p, c_char_p, POINTER(...), CFUNCTYPE(...): any object / function pointer


def test_binding_argument_types_and_struct_layouts_match_the_header(tmp_path):
    # A generated C++ translation unit static_asserts, for EVERY prototype of include/lsfc.h, that the return type and each
    # parameter belong to the class the ctypes binding passes (int / int64_t / double / unsigned / size_t / pointer), and
    # that offsetof / sizeof of lsfc_gmres_opts and lsfc_gmres_result equal the ctypes Structures' -- a c_int <-> int64_t slip
    # or a reordered struct field fails to compile.  The same layout table is quoted in julia/FastConvHIP.jl.
    import ctypes as C
    import fast_solver_lippmann_schwinger_amd._lib as L
    lines = ['#include <cstddef>', '#include <cstdint>', '#include <type_traits>', '#include "lsfc.h"',
             'template <class T> constexpr char cls() {',
             '  if constexpr (std::is_void<T>::value) return \'v\'; else if constexpr (std::is_pointer<T>::value) return \'p\';',
             '  else if constexpr (std::is_same<T, int>::value) return \'i\'; else if constexpr (std::is_same<T, int64_t>::value) return \'l\';',
             '  else if constexpr (std::is_same<T, double>::value) return \'d\'; else if constexpr (std::is_same<T, unsigned>::value) return \'u\';',
             '  else if constexpr (std::is_same<T, size_t>::value) return \'z\'; else return \'?\'; }',
             'template <class R, class... A> constexpr bool sig(R (*)(A...), const char* want) {',
             '  const char got[] = { cls<R>(), cls<A>()..., 0 };',
             '  for (int i = 0;; ++i) { if (got[i] != want[i]) return false; if (!got[i]) return true; } }']
    for name, (res, args) in sorted(L.SIGNATURES.items()):
        want = _ctype_class(res) + "".join(_ctype_class(a) for a in args)
        lines.append(f'static_assert(sig(&{name}, "{want}"), "{name}: binding passes {want}");')
    table = {}
    for cname, st in (("lsfc_gmres_opts", L.GmresOpts), ("lsfc_gmres_result", L.GmresResult)):
        lines.append(f'static_assert(sizeof({cname}) == {C.sizeof(st)}, "sizeof {cname}");')
        row = [f"size={C.sizeof(st)}"]
        for fname, _ in st._fields_:
            off = getattr(st, fname).offset
            lines.append(f'static_assert(offsetof({cname}, {fname}) == {off}, "{cname}.{fname}");')
            row.append(f"{fname}:{off}")
        table[cname] = " ".join(row)
    src = tmp_path / "abi_check.cpp"
    src.write_text("\n".join(lines) + "\nint main() { return 0; }\n")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    # the Julia binding's structs (same field order and types by eye) quote the same offsets
    jl = open(os.path.join(ROOT, "julia", "FastConvHIP.jl")).read()
    for cname, row in table.items():
        assert f"# ABI-LAYOUT {cname} {row}" in jl, f"julia/FastConvHIP.jl: layout comment of {cname} should read: {row}"
    # ... and its struct definitions list exactly those fields in that order
    for jname, st in (("GmresOpts", L.GmresOpts), ("GmresResult", L.GmresResult)):
        body = re.search(r"struct %s\n(.*?)\nend" % jname, jl, flags=re.S).group(1)
        assert re.findall(r"(\w+)::", body) == [f for f, _ in st._fields_], jname


def test_padded_length_rule():
    # host arithmetic only: smallest of 2^k, 3*2^k, 5*2^k in [32, 2048] that is >= 2n
    import fast_solver_lippmann_schwinger_amd._lib as L
    lib = L.load()
    lengths = sorted({v for k in range(12) for v in (2 ** k, 3 * 2 ** k, 5 * 2 ** k)
                      if (v >= 32 and v <= 2048) and not (v % 3 == 0 and v < 48) and not (v % 5 == 0 and v < 80)})
    assert lengths == [32, 48, 64, 80, 96, 128, 160, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 2048]
    for n in range(1, 1025):
        assert lib.lsfc_padded_length(n) == next(v for v in lengths if v >= max(2 * n, 32)), n
    assert lib.lsfc_padded_length(0) == 0 and lib.lsfc_padded_length(1025) == 0
    # from n = 24 on an axis is never padded by more than 4/3 beyond 2n (3*2^k -> 2^(k+2) is the widest gap)
    assert max(lib.lsfc_padded_length(n) / (2 * n) for n in range(24, 1025)) < 4 / 3


def test_header_compiles_as_c():
    src = '#include "lsfc.h"\nint main(void){ lsfc_gmres_opts o; (void)o; return LSFC_OK; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                       input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


def test_no_cpu_fallback_without_gpu():
    import fast_solver_lippmann_schwinger_amd as pkg
    pkg.load()
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.LsfcError) as ei:
        pkg.FastM(np.zeros((4, 4), complex), np.zeros(4), 4, 4, 2, 2, 1.0, quadRule="Greengard_Vico")
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_product_does_not_import_oracle():
    pk = os.path.join(ROOT, "fast_solver_lippmann_schwinger_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("CPU oracle", ""), f"{f} mentions the oracle"


def test_host_mirror_argument_errors():
    import fast_solver_lippmann_schwinger_amd as pkg
    with pytest.raises(NameError):       # reference: UndefVarError at src/FastConvolution.jl:106
        pkg.FastM(np.zeros((4, 4), complex), np.zeros(4), 4, 4, 2, 2, 1.0, quadRule="Simpson")
    with pytest.raises(ValueError):      # reference: DimensionMismatch from GFFT .* BFft
        pkg.FastM(np.zeros((5, 4), complex), np.zeros(4), 4, 4, 2, 2, 1.0, quadRule="Greengard_Vico")
    x, w = pkg.referenceValsTrapRule()
    assert x[0] == 1.0 and w[0] == 1 - 0.892j and len(w) == 6


def test_trapezoidal_table_index_out_of_range_raises():
    # D[round(Int, k*h)] (src/FastConvolution.jl:175-176): Julia throws BoundsError when k*h rounds to 0 (more than ~12
    # points per wavelength) or beyond the 6-entry table; a Python index of -1 would silently pick the last entry
    import fast_solver_lippmann_schwinger_amd as pkg
    from oracle import lsfc_oracle as o
    n = 21
    h = 1.0 / (n - 1)
    x = -0.5 + h * np.arange(n)
    for k in (0.3 / h, 6.6 / h):
        with pytest.raises(IndexError):
            pkg.buildFastConvolution(x, x, h, k, o.gaussian_bump, quadRule="trapezoidal")
        with pytest.raises(IndexError):
            o.build_fast_convolution(x, x, h, k, o.gaussian_bump, quadRule="trapezoidal")


def test_unique_id_exchange_needs_a_process_group():
    # more than one rank and no torch.distributed group: fail loudly instead of handing ncclCommInitRank an id nobody shares
    from fast_solver_lippmann_schwinger_amd.distributed import exchange_unique_id
    with pytest.raises(RuntimeError):
        exchange_unique_id(1, 2)
