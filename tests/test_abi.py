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


_UNTABLED = ("micf_abi_version", "micf_strerror", "micf_set_option", "micf_get_option")     # bound by hand in _lib._load


def test_header_declares_expected_entry_points():
    d = parse_header()
    assert len(d) == 101, sorted(d)         # (97 product entry points + micf_set_option / micf_get_option (test hooks) + the round-5 measurement probe + the MFMA hazard probe)
    assert all(sig.endswith("p") for n, sig in d.items()
               if n not in ("micf_abi_version", "micf_strerror", "micf_set_option", "micf_get_option", "micf_linear_bwd_weight_workspace",
                            "micf_linear_bwd_weight_grouped_workspace", "micf_conv3_bwd_data_workspace",
                            "micf_conv3_bwd_weight_grouped_workspace", "micf_head_tail_fused_supported", "micf_head_tail_pack_bytes", "micf_head_tail_loss_parts", "micf_head_tail_bwd_weight_fused_workspace",
                            "micf_offset_sample_bwd_workspace", "micf_conv3_bwd_weight_workspace",
                            "micf_conv3_fwd_workspace", "micf_layernorm_bwd_partial_rows", "micf_block_tile_tokens",
                            "micf_offset_head_needs_zero", "micf_offset_head_bwd_workspace", "micf_block_saves_bf16", "micf_block_fuses_sampler", "micf_block_recomputes_h",
                            "micf_offset_head_finish_deferrable"))


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
        if name in _UNTABLED:
            continue
        assert name in _lib.SIGNATURES, f"{name} missing from _lib.SIGNATURES"
        assert _lib.SIGNATURES[name] == sig, f"{name}: header {sig} vs ctypes {_lib.SIGNATURES[name]}"
    assert set(_lib.SIGNATURES) == set(d) - set(_UNTABLED)


def test_option_hooks_round_trip_and_reject_unknown_names():
    """micf_set_option / micf_get_option: every documented hook reads back what was set and is restored; an unknown name is an error
    (no silent no-op for a mistyped hook)."""
    import pytest
    from micformer_amd import _lib
    defaults = {"block_wave": 1, "block_recompute_h": 0, "block_debug": 0, "sample_tile": 1, "sample_e": -1, "cell_cap": -1,
                "tile_cap_hits": -1, "tile_cap_cell": -1, "tile_cap_voxel": -1}
    for name, dflt in defaults.items():
        assert _lib.get_option(name) == dflt, name
        with _lib.option(name, 5):
            assert _lib.get_option(name) == 5
        assert _lib.get_option(name) == dflt
    with pytest.raises(KeyError):
        _lib.set_option("block_waev", 0)
    assert _lib.lib.micf_set_option(None, 0) == -1


def test_invalid_arguments_return_error_codes_without_gpu():
    """Argument validation happens before any launch, so it is testable on a CPU-only box."""
    from micformer_amd import _lib
    assert _lib.lib.micf_layernorm_fwd(None, None, 4, None, None, None, None, None, 8, 4, 1e-5, None) == -1
    assert _lib.lib.micf_adam_tick(None, 1e-4, 0.0, 10, None) == -1
    assert _lib.lib.micf_conv_up_fwd(None, None, None, None, 1, 2, 2, 2, 4, 4, 3, None) == -1


def test_round2_entry_points_validate_arguments_without_gpu():
    """The grouped / fused entry points added in round 2: NULL arrays, bad group counts, unknown dtypes and unsupported shapes
    are rejected before any launch; the pure shape queries answer on a CPU-only box."""
    import ctypes as C
    from micformer_amd import _lib
    L = _lib.lib
    EINVAL, EUNSUP = -1, -2
    assert L.micf_block_fwd(None, 2, 1, 4, 4, 4, 48, 3, 192, C.c_float(1e-5), C.c_float(0.25), 0, None) == EINVAL
    assert L.micf_block_bwd(None, 1, 1, 4, 4, 4, 48, 3, 192, C.c_float(0.25), 0, None) == EINVAL
    g = (_lib.BlockFwdGroup * 2)()
    assert L.micf_block_fwd(C.cast(g, C.c_void_p), 3, 1, 4, 4, 4, 48, 3, 192, C.c_float(1e-5), C.c_float(0.25), 0, None) == EINVAL
    assert L.micf_block_fwd(C.cast(g, C.c_void_p), 1, 1, 3, 4, 4, 48, 3, 192, C.c_float(1e-5), C.c_float(0.25), 0, None) == EUNSUP   # odd grid
    assert L.micf_block_fwd(C.cast(g, C.c_void_p), 1, 1, 4, 4, 4, 48, 3, 192, C.c_float(1e-5), C.c_float(0.25), 7, None) == EINVAL   # dtype
    assert L.micf_block_fwd(C.cast(g, C.c_void_p), 1, 1, 4, 4, 4, 48, 3, 192, C.c_float(1e-5), C.c_float(0.25), 0, None) == EINVAL   # NULL tensors
    # shape queries
    assert L.micf_block_tile_tokens(2, 32, 32, 32, 48, 3, 192, 0) == 32 and L.micf_block_tile_tokens(2, 4, 4, 4, 384, 24, 1536, 1) == 16
    assert L.micf_block_tile_tokens(1, 3, 4, 4, 48, 3, 192, 0) == 0 and L.micf_block_tile_tokens(1, 4, 4, 4, 768, 24, 3072, 0) == 0
    assert L.micf_offset_head_needs_zero(2, 8, 8, 8, 192) == 1 and L.micf_offset_head_needs_zero(2, 32, 32, 32, 48) == 0
    assert L.micf_offset_head_bwd_workspace(2, 2, 32, 32, 32) == 2 * L.micf_offset_sample_bwd_workspace(2, 32, 32, 32)
    assert L.micf_conv3_fwd_workspace(16, 48, 48) > 6 * 27 * 256      # fp32 + bf16 layouts
    # grouped helpers
    assert L.micf_offset_head_fwd(None, 2, 1, 4, 4, 4, 48, C.c_float(1e-5), 0, 0, 0, None) == EINVAL
    assert L.micf_offset_head_bwd(None, 2, 1, 4, 4, 4, 48, C.c_float(1e-5), 0, None, 0, 0, 0, None) == EINVAL
    assert L.micf_layernorm_fwd_pair(None, 2, 8, 48, C.c_float(1e-5), None, 0, None) == EINVAL
    assert L.micf_layernorm_bwd_pair(None, 2, 8, 48, None) == EINVAL
    assert L.micf_weight_prep_grouped(None, 3, None) == EINVAL and L.micf_weight_prep_grouped(None, 0, None) == 0
    assert L.micf_conv3_weight_prep_grouped(None, 1, None) == EINVAL
    # the sampler's adjoint addresses one sample's tap rows with 32-bit element offsets: D*H*W*C >= 2^31 is refused, not wrapped
    nul = [None] * 12
    assert L.micf_offset_sample_bwd(*nul, 1, 128, 512, 512, 64, C.c_float(1e-5), None, 0, None) == EUNSUP
    assert L.micf_offset_sample_bwd(*nul, 1, 128, 512, 512, 48, C.c_float(1e-5), None, 0, None) == EINVAL    # (in range: NULL tensors)
    assert L.micf_adam_step(None, None, None, None, 8, None, C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), C.c_float(1.0), None, None) == EINVAL


def _struct_fields(header_text, name):
    """Pointer field names of `typedef struct NAME { ... } NAME;` in declaration order (comments stripped)."""
    import re
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header_text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        # "const float *a, *b" / "float* y" / "const void *wq, *wkv" / "void* h"
        first, *rest = decl.split(",")
        names.append(first.replace("*", " ").split()[-1])
        names.extend(r.replace("*", " ").split()[-1] for r in rest)
    return names


def test_block_group_structs_mirror_the_header_field_by_field():
    """The ctypes mirrors of micf_block_fwd_group / micf_block_bwd_group (arrays of them are what micf_block_fwd / _bwd receive)
    list the header's fields in the header's order: a field added on one side only shifts every pointer behind it."""
    from micformer_amd import _lib
    text = open(HEADER).read()
    assert tuple(_struct_fields(text, "micf_block_fwd_group")) == _lib.BlockFwdGroup.FIELDS
    assert tuple(_struct_fields(text, "micf_block_bwd_group")) == _lib.BlockBwdGroup.FIELDS
