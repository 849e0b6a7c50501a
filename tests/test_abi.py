"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/micformer_hip.h declares,
and the ctypes signatures in micformer_amd/_lib.py match the header argument by argument (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "micformer_hip.h")


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(micf_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        sig = ""
        for a in [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"]:
            if "*" in a or a.startswith("micf_stream_t"):
                sig += "p"
            elif a.startswith("int64_t"):
                sig += "l"
            elif a.startswith("int "):
                sig += "i"
            elif a.startswith("float "):
                sig += "f"
            elif a.startswith("double "):
                sig += "d"
            else:
                raise AssertionError(f"unparsed argument {a!r} in {name}")
        decls[name] = sig
    return decls


def test_header_declares_expected_entry_points():
    d = parse_header()
    assert len(d) == 69, sorted(d)
    assert all(sig.endswith("p") for n, sig in d.items()
               if n not in ("micf_abi_version", "micf_strerror", "micf_linear_bwd_weight_workspace",
                            "micf_linear_bwd_weight_grouped_workspace", "micf_conv3_bwd_data_workspace",
                            "micf_offset_sample_bwd_workspace", "micf_conv3_bwd_weight_workspace",
                            "micf_conv3_fwd_workspace", "micf_layernorm_bwd_partial_rows", "micf_block_tile_tokens",
                            "micf_offset_head_needs_zero", "micf_offset_head_bwd_workspace"))


def test_library_exports_every_declared_symbol():
    from micformer_amd.build import LIB
    assert os.path.exists(LIB), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(LIB)
    for name in parse_header():
        assert hasattr(lib, name), f"{name} declared in include/micformer_hip.h but not exported"
    assert lib.micf_abi_version() == 1
    lib.micf_strerror.restype = ctypes.c_char_p
    assert lib.micf_strerror(0) == b"ok" and b"invalid" in lib.micf_strerror(-1)


def test_ctypes_signatures_match_header():
    from micformer_amd import _lib
    d = parse_header()
    for name, sig in d.items():
        if name in ("micf_abi_version", "micf_strerror"):
            continue
        assert name in _lib.SIGNATURES, f"{name} missing from _lib.SIGNATURES"
        assert _lib.SIGNATURES[name] == sig, f"{name}: header {sig} vs ctypes {_lib.SIGNATURES[name]}"
    assert set(_lib.SIGNATURES) == set(d) - {"micf_abi_version", "micf_strerror"}


def test_invalid_arguments_return_error_codes_without_gpu():
    """Argument validation happens before any launch, so it is testable on a CPU-only box."""
    from micformer_amd import _lib
    assert _lib.lib.micf_layernorm_fwd(None, None, 4, None, None, None, None, None, 8, 4, 1e-5, None) == -1
    assert _lib.lib.micf_adam_tick(None, 1e-4, 0.0, 10, None) == -1
    assert _lib.lib.micf_conv_up_fwd(None, None, None, None, 1, 2, 2, 2, 4, 4, 3, None) == -1
