"""ctypes binding of libmicformer_hip.so (the C-ABI declared in include/micformer_hip.h).

There is NO fallback: if the library is missing or a symbol cannot be resolved the import raises, and every op
raises on non-CUDA tensors -- the product path is the HIP path or nothing.
"""
import ctypes
import os

# ROCm 7.2's HIP runtime replays a captured graph from AQL packets it records at the first launch ("graph packet capture").  With
# MORE THAN ONE instantiated graph of this step's size launched alternately, the second launch of a graph after another graph
# ran faults (memory access fault in the replayed kernels; found by bisection on MI355X: [forward | backward] as two graphs works
# once and faults at the next step, and works with the optimisation off; a single graph is not affected, its step time does not
# change either way).  The segmented step capture (functional.StepSegmenter) therefore needs the flag OFF, and the runtime reads
# it when it initialises -- i.e. before the first HIP call of the process.  GRAPH_SEGMENTS_OK records whether that was achieved;
# TrainEngine falls back to the one-graph layout otherwise.  The segmented layout is OPT-IN (MICF_SEGMENTED=1 / TrainEngine(
# segmented=True)): it measures the same step time as the best one-graph placement (the main chain alone replays in the step's
# time: the side work is fully hidden either way), and a suite that keeps several segmented engines alive in one process was
# seen to abort intermittently on this runtime.
_PKT = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_pkt_before = os.environ.get(_PKT)
if os.environ.get("MICF_SEGMENTED", "0") == "1":        # opt-in (bench.py --segmented sets both)
    os.environ.setdefault(_PKT, "0")

import torch  # noqa: E402

GRAPH_SEGMENTS_OK = os.environ.get(_PKT) == "0" and (_pkt_before == "0" or not torch.cuda.is_initialized())

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MICF_LIB: another build of the same library -- A/B runs of kernel variants on one box, tools/ab_build.sh)
LIB_PATH = os.environ.get("MICF_LIB") or os.path.join(_HERE, "libmicformer_hip.so")

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
_T = {"p": _P, "i": _I, "l": _L, "f": _F, "d": _D}

# name -> argument signature (p pointer, i int, l int64, f float, d double); every function returns int
# except micf_strerror.  Mirrors include/micformer_hip.h one to one (tests/test_abi.py checks the header).
SIGNATURES = {
    "micf_layernorm_fwd": "ppippppplifp",
    "micf_layernorm_bwd": "pppippppppplippp",
    "micf_layernorm_bwd_partial_rows": "lii",
    "micf_layernorm_bwd_finish": "pip",
    "micf_linear_fwd": "ppipppplppliiiip",
    "micf_linear_bwd_data": "pplppppiiliiip",
    "micf_linear_bwd_weight": "pplppiippliiplip",
    "micf_linear_bwd_weight_workspace": "lii",
    "micf_linear_bwd_weight_grouped": "piplip",
    "micf_linear_bwd_weight_grouped_workspace": "pi",
    "micf_head_tail_compose": "pppppiiiipp",
    "micf_head_tail_col2im": "pppiiiiiip",
    "micf_head_tail_im2col": "ppiiiiiip",
    "micf_head_tail_col2im_sw": "pppppiiiiiiiiiip",
    "micf_head_tail_decompose": "pppppppppiiiipp",
    "micf_head_tail_fused_supported": "iiiiiii",
    "micf_head_tail_pack_bytes": "ii",
    "micf_head_tail_pack": "pppppiiip",
    "micf_head_tail_fwd_fused": "pppiiiiiiip",
    "micf_head_tail_fwd_fused_sw": "pppppiiiiiiiiiiip",
    "micf_head_tail_loss_parts": "iiii",
    "micf_head_tail_fwd_loss_fused": "ppppipppiiiiiiip",
    "micf_head_tail_bwd_data_fused": "pppiiiiiiip",
    "micf_head_tail_bwd_weight_fused": "pppppliiiiiiip",
    "micf_head_tail_bwd_weight_fused_workspace": "iiiii",
    "micf_sw_window": "ppiiiiiiiiiip",
    "micf_sw_accumulate": "pppiiiiiiiiiip",
    "micf_sw_normalize": "ppilp",
    "micf_window_attn_fwd": "pippipiiiiiiiiiifp",
    "micf_window_attn_fwd_fp8": "pippipiiiiiiiiiifp",
    "micf_window_attn_bwd": "pippipipippiiiiiiiiiifp",
    "micf_conv3_fwd": "pipipppiiiiiipliip",
    "micf_conv3_fwd_workspace": "iii",
    "micf_conv3_bwd_data": "pippiipiiiiiiipliip",
    "micf_conv3_weight_prep_grouped": "pip",
    "micf_layernorm_fwd_pair": "pilifplp",
    "micf_layernorm_bwd_pair": "pilip",
    "micf_offset_head_needs_zero": "iiiii",
    "micf_offset_head_fwd": "piiiiiifiiip",
    "micf_offset_head_bwd_workspace": "iiiii",
    "micf_offset_head_bwd": "piiiiiifipliip",
    "micf_offset_head_finish_deferrable": "iiii",
    "micf_offset_head_bwd_finish": "piiiiiiplp",
    "micf_offset_head_bwd_finish_grouped": "pip",
    "micf_conv3_bwd_data_workspace": "iii",
    "micf_conv3_bwd_weight": "pipipippiiiiiplip",
    "micf_conv3_bwd_weight_workspace": "iiiiiii",
    "micf_conv3_bwd_weight_grouped": "piiiiiiiiplip",
    "micf_conv3_bwd_weight_grouped_workspace": "iiiiiiii",
    "micf_offset_sample_fwd": "pppppppiiiiifp",
    "micf_offset_sample_bwd": "ppppppppppppiiiiifplp",
    "micf_offset_sample_bwd_workspace": "iiii",
    "micf_stn_fwd": "pppiiiiip",
    "micf_stn_bwd": "pppppiiiiip",
    "micf_patch_embed_fwd": "piipppiiiiiip",
    "micf_patch_embed_bwd_weight": "ppiippiiiiiip",
    "micf_conv_down_fwd": "ppppiiiiiip",
    "micf_conv_down_bwd_data": "pppiiiiiip",
    "micf_conv_down_bwd_weight": "ppppiiiiiip",
    "micf_conv_up_fwd": "ppppiiiiiiip",
    "micf_conv_up_bwd_data": "pppiiiiiiip",
    "micf_conv_up_bwd_weight": "ppppiiiiiiip",
    "micf_space_to_depth": "ppiiiiiilp",
    "micf_depth_to_space": "pppiiiiiip",
    "micf_depth_to_space_add": "ppppiiiiiip",
    "micf_colsum": "pplip",
    "micf_pad3d": "ppiiiiiiiip",
    "micf_crop3d": "ppiiiiiiiiip",
    "micf_resize_trilinear_fwd": "ppiiiiiiiip",
    "micf_resize_trilinear_bwd": "ppiiiiiiiip",
    "micf_dice_bce_fwd": "ppppiilp",
    "micf_dice_bce_bwd": "pppppiilp",
    "micf_dice_bce_label_fwd": "ppppiilp",
    "micf_dice_bce_label_bwd": "pppppiilp",
    "micf_argmax_meandice": "pppppiilp",
    "micf_adam_tick": "pddlp",
    "micf_adam_step": "pppplpffffpp",
    "micf_block_tile_tokens": "iiiiiiii",
    "micf_block_saves_bf16": "iii",
    "micf_block_fuses_sampler": "ii",
    "micf_block_recomputes_h": "ii",
    "micf_weight_prep_grouped": "pip",
    "micf_grad_wire_pack": "plplp",
    "micf_grad_wire_sum": "pilpp",
    "micf_grad_wire_unpack": "pplp",
    "micf_block_fwd": "piiiiiiiiffip",
    "micf_block_bwd": "piiiiiiiifip",
    "micf_block_fwd_persistent_probe": "piiiiiiiiffiipp",
    "micf_probe_mfma_chain": "pppppip",
    "micf_dice_metric": "ppippiilp",
    "micf_sw_window_batch": "pppiiiiiiiiip",
    "micf_sw_accumulate_batch": "ppppiiiiiiiiip",
    "micf_intensity_stats": "pipiilp",
    "micf_input_prepare": "pipppppiiiiip",
    "micf_patch_rows_prepared": "pippppiiiiiip",
    "micf_zero": "plp",
    "micf_drop_path_draw": "pppiip",
}


class WgradItem(ctypes.Structure):
    """struct micf_wgrad_item (include/micformer_hip.h)."""
    _fields_ = [("a", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dp_scale", ctypes.c_void_p), ("dw", ctypes.c_void_p),
                ("dbias", ctypes.c_void_p), ("M", ctypes.c_int64), ("rows_per_sample", ctypes.c_int64),
                ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("operand_dtype", ctypes.c_int32), ("ldw", ctypes.c_int32)]


class Conv3WgradItem(ctypes.Structure):
    """struct micf_conv3_wgrad_item (include/micformer_hip.h)."""
    _fields_ = [("dy", ctypes.c_void_p), ("x1", ctypes.c_void_p), ("x2", ctypes.c_void_p), ("dw", ctypes.c_void_p),
                ("dbias", ctypes.c_void_p)]


class LnFinishItem(ctypes.Structure):
    """struct micf_ln_finish_item (include/micformer_hip.h)."""
    _fields_ = [("partials", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p),
                ("blocks", ctypes.c_int32), ("C", ctypes.c_int32)]


_VP = ctypes.c_void_p


class BlockFwdGroup(ctypes.Structure):
    """struct micf_block_fwd_group (include/micformer_hip.h)."""
    FIELDS = ("x", "kvsrc", "ln1_g", "ln1_b", "bq", "bkv", "bp", "ln2_g", "ln2_b", "b1", "b2", "wq", "wkv", "wp", "w1", "w2",
              "s1", "s2", "y", "xn", "q", "kv", "o", "x1", "xn2", "h", "g", "stats", "kvs16", "hid", "samp_src", "ln16_g", "ln16_b", "w1c", "flow", "xs32",
              "nln_g", "nln_b", "nln_y", "nln_mean", "nln_rstd", "zero16")
    _fields_ = [(n, _VP) for n in FIELDS]


class BlockBwdGroup(ctypes.Structure):
    """struct micf_block_bwd_group (include/micformer_hip.h)."""
    FIELDS = ("dy", "x", "x1", "stats", "q", "kv", "h", "ln1_g", "ln2_g", "wqt", "wkvt", "wpt", "w1t", "w2t", "s1", "s2",
              "dx", "dxs", "dx1", "dh", "dq", "dkv", "ln1_part", "ln2_part", "dx1_copy", "dy16", "xn2", "w1", "b1",
              "pre_d", "pre_x", "pre_mean", "pre_rstd", "pre_g", "pre_part")
    _fields_ = [(n, _VP) for n in FIELDS]


class WeightPrepItem(ctypes.Structure):
    """struct micf_weight_prep_item (include/micformer_hip.h)."""
    _fields_ = [("src", _VP), ("dst", _VP), ("dst_t", _VP), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("bf16", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class LnPairItem(ctypes.Structure):
    """struct micf_ln_pair_item (include/micformer_hip.h)."""
    FIELDS = ("x", "gamma", "beta", "y", "mean", "rstd")
    _fields_ = [(n, _VP) for n in FIELDS]


class LnBwdPairItem(ctypes.Structure):
    """struct micf_ln_bwd_pair_item (include/micformer_hip.h)."""
    FIELDS = ("dy", "x", "mean", "rstd", "gamma", "dx", "dgamma", "dbeta", "add", "partials")
    _fields_ = [(n, _VP) for n in FIELDS]


class OffsetHeadGroup(ctypes.Structure):
    """struct micf_offset_head_group (include/micformer_hip.h)."""
    FIELDS = ("xn", "xa", "conv_w", "conv_b", "conv_ws", "ln_g", "ln_b", "w1", "hid", "flow", "xs")
    _fields_ = [(n, _VP) for n in FIELDS]


class OffsetHeadBwdGroup(ctypes.Structure):
    """struct micf_offset_head_bwd_group (include/micformer_hip.h)."""
    FIELDS = ("dxs", "hid", "flow", "xa", "ln_g", "ln_b", "w1", "conv_w", "conv_ws", "dxa", "dxn", "dhid", "dln_g", "dln_b", "dw1")
    _fields_ = [(n, _VP) for n in FIELDS]


class OffsetHeadFinishCall(ctypes.Structure):
    """struct micf_offset_head_finish_call (include/micformer_hip.h)."""
    _fields_ = [("groups", _VP), ("ngroups", ctypes.c_int32), ("B", ctypes.c_int32), ("D", ctypes.c_int32), ("H", ctypes.c_int32),
                ("W", ctypes.c_int32), ("C", ctypes.c_int32), ("workspace", _VP), ("workspace_floats", ctypes.c_int64)]


class Conv3PrepItem(ctypes.Structure):
    """struct micf_conv3_prep_item (include/micformer_hip.h)."""
    _fields_ = [("w", _VP), ("fwd", _VP), ("bwd", _VP), ("N", ctypes.c_int32), ("Cin", ctypes.c_int32)]


class MicfError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python micformer_amd/build.py` (hipcc, gfx950). "
            "micformer_amd has no CPU / PyTorch fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = [_T[c] for c in sig]
        fn.restype = _I
    lib.micf_linear_bwd_weight_workspace.restype = _L
    lib.micf_linear_bwd_weight_grouped_workspace.restype = _L
    lib.micf_conv3_bwd_data_workspace.restype = _L
    lib.micf_offset_sample_bwd_workspace.restype = _L
    lib.micf_conv3_bwd_weight_workspace.restype = _L
    lib.micf_conv3_bwd_weight_grouped_workspace.restype = _L
    lib.micf_head_tail_pack_bytes.restype = _L
    lib.micf_head_tail_loss_parts.restype = _L
    lib.micf_head_tail_loss_parts.argtypes = [_I] * 4
    lib.micf_head_tail_bwd_weight_fused_workspace.restype = _L
    lib.micf_conv3_fwd_workspace.restype = _L
    lib.micf_offset_head_bwd_workspace.restype = _L
    lib.micf_block_tile_tokens.argtypes = [_I] * 8          # (no stream argument: a pure shape query)
    lib.micf_block_saves_bf16.argtypes = [_I] * 3
    lib.micf_offset_head_finish_deferrable.argtypes = [_I] * 4
    lib.micf_block_fuses_sampler.argtypes = [_I] * 2
    lib.micf_block_recomputes_h.argtypes = [_I] * 2
    lib.micf_strerror.argtypes = [_I]
    lib.micf_strerror.restype = ctypes.c_char_p
    lib.micf_set_option.argtypes = [ctypes.c_char_p, _I]
    lib.micf_set_option.restype = _I
    lib.micf_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(_I)]
    lib.micf_get_option.restype = _I
    lib.micf_abi_version.argtypes = []
    lib.micf_abi_version.restype = _I
    if lib.micf_abi_version() != 1:
        raise ImportError("libmicformer_hip.so ABI version mismatch")
    return lib


lib = _load()


def get_option(name):
    """Current value of a library test hook / probe (include/micformer_hip.h micf_set_option; csrc/common.h struct Options)."""
    v = _I(0)
    if lib.micf_get_option(name.encode(), ctypes.byref(v)) != 0:
        raise KeyError(f"unknown micformer option {name!r}")
    return v.value


def set_option(name, value):
    """Set a test hook / probe of the library; returns the previous value.  Never called by the product path."""
    prev = get_option(name)
    if lib.micf_set_option(name.encode(), int(value)) != 0:
        raise KeyError(f"unknown micformer option {name!r}")
    return prev


class option:
    """with option("block_wave", 0): ...   -- a hook for the duration of a block (tests, measurement scripts)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = set_option(self.name, self.value)

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False


def ptr(t):
    """Device pointer of a contiguous fp32 (or given dtype) CUDA tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MicfError("micformer_amd ops need CUDA (ROCm) tensors: there is no CPU path")
    if not t.is_contiguous():
        raise MicfError("micformer_amd ops need contiguous tensors")
    return t.data_ptr()


# The launch stream of a C-ABI call = torch's current stream of the current device.  torch.cuda.current_stream() builds a Stream
# object through three Python layers (8.5 us: a sixth of the engine-less loop's time per call); the raw handle is one C call.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


# Optional per-entry-point timing with HIP events on the launch stream (bench.py's roofline leg).
# PROFILE = None (off) or a dict  name -> [events [(start, end)], bytes, flops]
PROFILE = None
BLOCK_DEPTH = 0          # > 0 while a transformer-block Function is issuing launches (bench.py's path roofline)
UNIT = None              # SURVEY.md 8(d) unit the launches issued now belong to, e.g. "cross_bwd|2x65536x48" (bench.py's roofline leg)


def set_unit(kind, groups, tokens, channels):
    """Names the 8(d) unit -- the self / cross block pair of a depth slot, forward or backward -- whose launches follow (profiling
    only; cleared when the enclosing block_region closes)."""
    global UNIT
    if PROFILE is not None:
        UNIT = f"{kind}|{groups}x{tokens}x{channels}"


class block_region:
    """Marks the launches issued inside as transformer-block kernels (SURVEY.md 8(d) `t_attention_kernels`)."""

    def __enter__(self):
        global BLOCK_DEPTH
        BLOCK_DEPTH += 1

    def __exit__(self, *exc):
        global BLOCK_DEPTH, UNIT
        BLOCK_DEPTH -= 1
        if BLOCK_DEPTH == 0:
            UNIT = None
        return False


LAUNCHES = [0]           # C-ABI calls issued so far (functional.StepSegmenter: is the segment being captured empty?)


def call(name, *args, cost=None):
    """Launch one C-ABI entry point on torch's current stream.  cost = (bytes the launch itself moves with every tensor touched
    once, flops[, shape tag[, SURVEY 8(d) bytes]]) for the profiler; the 8(d) bytes are the ideal-fusion activation passes + block
    weights of a transformer-block launch (None for everything that is not one)."""
    LAUNCHES[0] += 1
    if PROFILE is None:
        rc = getattr(lib, name)(*args, stream())
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args, stream())
        e1.record()
        key = name if (cost is None or len(cost) < 3 or cost[2] is None) else f"{name}|{cost[2]}"
        rec = PROFILE.setdefault(key, [[], 0, 0, 0])
        rec[0].append((e0, e1, BLOCK_DEPTH > 0, UNIT, cost[3] if (cost is not None and len(cost) > 3 and cost[3]) else 0))
        if cost is not None:
            rec[1] += cost[0]
            rec[2] += cost[1]
            if len(cost) > 3 and cost[3] is not None:
                rec[3] += cost[3]
    if rc != 0:
        raise MicfError(f"{name} failed: {lib.micf_strerror(rc).decode()} (code {rc})")


def profile_start():
    global PROFILE
    PROFILE = {}


def profile_stop():
    """-> {name: dict(calls, ms, bytes, flops, block_ms, s8d_bytes, by_unit {unit: [calls, ms, 8(d) bytes]})}; synchronises."""
    global PROFILE, UNIT
    prof, PROFILE, UNIT = PROFILE, None, None
    torch.cuda.synchronize()
    out = {}
    for name, (evs, nbytes, flops, s8d) in (prof or {}).items():
        ms = [(a.elapsed_time(b), blk, unit, u8d) for a, b, blk, unit, u8d in evs]
        by_unit = {}
        for m, _, unit, u8d in ms:
            if unit is not None:
                u = by_unit.setdefault(unit, [0, 0.0, 0])
                u[0] += 1
                u[1] += m
                u[2] += u8d
        out[name] = dict(calls=len(evs), ms=sum(m[0] for m in ms), bytes=nbytes, flops=flops,
                         block_ms=sum(m[0] for m in ms if m[1]), s8d_bytes=s8d, by_unit=by_unit)
    return out


def f32(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise MicfError(f"expected float32 tensor, got {t.dtype}")
    return ptr(t)
