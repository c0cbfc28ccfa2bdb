"""ctypes binding of libssrhip.so (C-ABI declared in include/ssrhip.h).

The library is loaded lazily and loudly: any compute entry point of this package raises
`SsrHipUnavailable` when the shared object is missing — there is no CPU fallback (the CPU
restatement is `oracle/`, test-only).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libssrhip.so")

ABI_VERSION = 107          # include/ssrhip.h SSRHIP_VERSION
PAGE = 128
MAX_CODEBOOKS = 4
MAX_SILENCE = 8

PRO_NONE, PRO_LAYERNORM, PRO_ATTN_COMBINE = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU_ERF, ACT_ELU = 0, 1, 2, 3
EPI_STORE, EPI_RESIDUAL, EPI_QKV_APPEND = 0, 1, 2

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)


class SsrHipUnavailable(RuntimeError):
    pass


class KV(C.Structure):
    _fields_ = [("pool", C.c_void_p), ("table", C.c_void_p), ("max_pages", C.c_int32),
                ("n_layer", C.c_int32), ("n_head", C.c_int32), ("head_dim", C.c_int32)]


class GemvArgs(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p),
                ("B", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("groups", C.c_int32),
                ("x_stride", C.c_int32), ("y_stride", C.c_int32),
                ("pro", C.c_int32), ("act", C.c_int32), ("epi", C.c_int32),
                ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float),
                ("part_o", C.c_void_p), ("part_ml", C.c_void_p), ("max_splits", C.c_int32),
                ("row_len", C.c_void_p),
                ("kv", KV), ("layer", C.c_int32), ("kv_pos", C.c_void_p),
                ("x_tiled", C.c_int32), ("y_tiled", C.c_int32), ("w_tiled", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("q_stride", C.c_int32), ("kv", KV), ("layer", C.c_int32),
                ("row_seq", C.c_void_p), ("row_len", C.c_void_p), ("R", C.c_int32), ("max_splits", C.c_int32),
                ("scale", C.c_float), ("part_o", C.c_void_p), ("part_ml", C.c_void_p), ("out_tiled", C.c_int32),
                ("prefetch", C.c_void_p), ("prefetch_floats", C.c_int32)]


class EmbedArgs(C.Structure):
    _fields_ = [("text_emb", C.c_void_p), ("audio_emb", C.c_void_p), ("pe", C.c_void_p),
                ("alpha_text", C.c_float), ("alpha_audio", C.c_float),
                ("tok", C.c_void_p), ("pos", C.c_void_p), ("kind", C.c_void_p),
                ("R", C.c_int32), ("D", C.c_int32), ("K", C.c_int32), ("card", C.c_int32), ("out", C.c_void_p),
                ("out_tiled", C.c_int32)]


class SamplerCfg(C.Structure):
    _fields_ = [("top_k", C.c_int32), ("top_p", C.c_float), ("temperature", C.c_float), ("stop_repetition", C.c_int32),
                ("cfg_coef", C.c_float), ("cfg_one_minus", C.c_float), ("cfg_stride", C.c_int32), ("use_cfg", C.c_int32),
                ("n_silence", C.c_int32), ("silence", C.c_int32 * MAX_SILENCE),
                ("text_len", C.c_int32), ("n_spans", C.c_int32),
                ("empty_token", C.c_int32), ("eog", C.c_int32), ("eos", C.c_int32), ("sos", C.c_int32),
                ("mts", C.c_int32), ("max_n_spans", C.c_int32), ("max_steps", C.c_int32),
                ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("use_noise", C.c_int32)]


class SamplerState(C.Structure):
    _fields_ = [("span", C.c_int32), ("num_gen", C.c_int32), ("num_eog", C.c_int32), ("num_cfg_tag", C.c_int32),
                ("prev_token", C.c_int32), ("consec_silence", C.c_int32), ("audio_pos", C.c_int32),
                ("n_steps", C.c_int32), ("done", C.c_int32), ("span_end", C.c_int32 * 3), ("pad", C.c_int32 * 3)]


class SampleArgs(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("n_utt", C.c_int32), ("K", C.c_int32), ("card", C.c_int32),
                ("cfg", C.c_void_p), ("state", C.c_void_p), ("noise", C.c_void_p), ("generated", C.c_void_p),
                ("next_tok", C.c_void_p), ("next_pos", C.c_void_p), ("kv_pos", C.c_void_p), ("row_len", C.c_void_p),
                ("dbg_logits", C.c_void_p), ("embed", EmbedArgs)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldc", C.c_int32),
                ("act", C.c_int32), ("residual", C.c_int32),
                ("act_in", C.c_int32), ("R", C.c_void_p), ("ldr", C.c_int32), ("batch", C.c_int32),
                ("strideA", C.c_int64), ("strideC", C.c_int64), ("strideR", C.c_int64),
                ("tm_c", C.c_int32), ("tm_lo", C.c_int32), ("tm_hi", C.c_int32),
                ("rbias", C.c_void_p), ("rclass", C.c_void_p), ("rrep", C.c_int32), ("rclass_stride", C.c_int32),
                ("W_split", C.c_void_p), ("act_out", C.c_int32)]


class LstmArgs(C.Structure):
    _fields_ = [("gin", C.c_void_p), ("w_hh", C.c_void_p), ("out", C.c_void_p), ("skip", C.c_void_p),
                ("hbuf", C.c_void_p), ("cbuf", C.c_void_p), ("gates", C.c_void_p),
                ("B", C.c_int32), ("T", C.c_int32), ("C", C.c_int32),
                ("gin_bstride", C.c_int64), ("out_bstride", C.c_int64), ("skip_bstride", C.c_int64),
                ("t_begin", C.c_int32), ("t_end", C.c_int32), ("w_packed", C.c_int32), ("out_act", C.c_int32),
                ("w_split", C.c_void_p), ("hsplit", C.c_void_p)]


_PP = C.POINTER(C.c_void_p)


class ResblockArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("B", C.c_int32), ("T", C.c_int32), ("C", C.c_int32), ("x_bstride", C.c_int64), ("y_bstride", C.c_int64),
                ("out_act", C.c_int32), ("w3_split", C.c_void_p), ("w1_split", C.c_void_p)]


class LMWeights(C.Structure):
    _fields_ = [("text_emb", C.c_void_p), ("audio_emb", C.c_void_p), ("pe", C.c_void_p),
                ("alpha_text", C.c_float), ("alpha_audio", C.c_float),
                ("ln1_w", _PP), ("ln1_b", _PP), ("in_proj_w", _PP), ("in_proj_b", _PP),
                ("out_proj_w", _PP), ("out_proj_b", _PP), ("ln2_w", _PP), ("ln2_b", _PP),
                ("ffn1_w", _PP), ("ffn1_b", _PP), ("ffn2_w", _PP), ("ffn2_b", _PP),
                ("lnf_w", C.c_void_p), ("lnf_b", C.c_void_p),
                ("head1_w", C.c_void_p), ("head1_b", C.c_void_p), ("head2_w", C.c_void_p), ("head2_b", C.c_void_p),
                ("in_proj_wt", _PP), ("out_proj_wt", _PP), ("ffn1_wt", _PP), ("ffn2_wt", _PP),
                ("head1_wt", C.c_void_p), ("head2_wt", C.c_void_p),
                ("in_proj_ws", _PP), ("out_proj_ws", _PP), ("ffn1_ws", _PP), ("ffn2_ws", _PP)]


class LMDims(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_head", C.c_int32), ("n_layer", C.c_int32), ("d_ffn", C.c_int32),
                ("n_codebooks", C.c_int32), ("card", C.c_int32), ("head_hidden", C.c_int32), ("n_text", C.c_int32),
                ("max_pos", C.c_int32), ("ln_folded", C.c_int32)]


class LMBuffers(C.Structure):
    _fields_ = [("B", C.c_int32), ("n_utt", C.c_int32), ("max_splits", C.c_int32), ("pair_mode", C.c_int32),
                ("x", C.c_void_p), ("q", C.c_void_p), ("h", C.c_void_p), ("logits", C.c_void_p),
                ("part_o", C.c_void_p), ("part_ml", C.c_void_p),
                ("next_tok", C.c_void_p), ("next_pos", C.c_void_p), ("kv_pos", C.c_void_p), ("row_len", C.c_void_p),
                ("kv", KV), ("cfg", C.c_void_p), ("state", C.c_void_p),
                ("noise", C.c_void_p), ("generated", C.c_void_p), ("dbg_logits", C.c_void_p)]


class PrefillArgs(C.Structure):
    _fields_ = [("tok", C.c_void_p), ("pos", C.c_void_p), ("kind", C.c_void_p),
                ("row_seq", C.c_void_p), ("row_pos", C.c_void_p), ("row_len", C.c_void_p),
                ("R", C.c_int32), ("max_splits", C.c_int32),
                ("x", C.c_void_p), ("xn", C.c_void_p), ("qkv", C.c_void_p), ("o", C.c_void_p), ("h", C.c_void_p),
                ("part_o", C.c_void_p), ("part_ml", C.c_void_p),
                ("seq_start", C.c_void_p), ("n_seq", C.c_int32), ("max_len", C.c_int32),
                ("table", C.c_void_p), ("no_embed", C.c_int32), ("reserved_", C.c_int32)]


# every symbol include/ssrhip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("ssrhip_version", C.c_int, []),
    ("ssrhip_sizeof", C.c_int, [C.c_int]),
    ("ssrhip_last_error", C.c_char_p, []),
    ("ssrhip_gemv", C.c_int, [C.POINTER(GemvArgs), C.c_void_p]),
    ("ssrhip_pair_buffer", C.c_int, [C.c_int32, C.c_int32]),
    ("ssrhip_gemv_pair_applicable", C.c_int, [C.POINTER(GemvArgs), C.POINTER(GemvArgs)]),
    ("ssrhip_gemv_pair", C.c_int, [C.POINTER(GemvArgs), C.POINTER(GemvArgs), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("ssrhip_gemv_pair_status", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ssrhip_attn_decode", C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    ("ssrhip_attn_combine", C.c_int, [C.POINTER(AttnArgs), C.c_void_p, C.c_void_p]),
    ("ssrhip_attn_rows", C.c_int, [C.POINTER(AttnArgs), C.c_void_p, C.c_void_p]),
    ("ssrhip_attn_prefill", C.c_int, [C.POINTER(AttnArgs), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ("ssrhip_embed", C.c_int, [C.POINTER(EmbedArgs), C.c_void_p]),
    ("ssrhip_sample", C.c_int, [C.POINTER(SampleArgs), C.c_void_p]),
    ("ssrhip_gemm", C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    ("ssrhip_split_weights", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("ssrhip_conv_few_out", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    ("ssrhip_conv_cin1", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    ("ssrhip_pad_reflect", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    ("ssrhip_pad_ragged", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    ("ssrhip_lstm_layer", C.c_int, [C.POINTER(LstmArgs), C.c_void_p]),
    ("ssrhip_rvq_encode", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    ("ssrhip_rvq_decode", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    ("ssrhip_resblock", C.c_int, [C.POINTER(ResblockArgs), C.c_void_p]),
    ("ssrhip_layernorm", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("ssrhip_kv_scatter", C.c_int, [C.c_void_p, C.POINTER(KV), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("ssrhip_lm_create", C.c_int, [C.POINTER(LMDims), C.POINTER(LMWeights), C.POINTER(LMBuffers), C.POINTER(C.c_void_p)]),
    ("ssrhip_lm_destroy", None, [C.c_void_p]),
    ("ssrhip_lm_decode", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("ssrhip_lm_prefill", C.c_int, [C.c_void_p, C.POINTER(PrefillArgs), C.c_void_p]),
    ("ssrhip_lm_embed_pending", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ssrhip_lm_pairing", C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    ("ssrhip_lm_pair_status", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ssrhip_debug_occupy", C.c_int, [C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ("ssrhip_lm_time_steps", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, c_f32p, c_i32p, C.c_int32]),
    ("ssrhip_lm_time_category", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, c_f32p, c_i32p]),
]

ABI_STRUCTS = [KV, GemvArgs, AttnArgs, EmbedArgs, SamplerCfg, SamplerState, SampleArgs, GemmArgs, LMWeights, LMDims,
               LMBuffers, PrefillArgs, LstmArgs, ResblockArgs]

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libssrhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-8000:])
    if r.returncode != 0:
        raise RuntimeError("building libssrhip.so failed")
    return LIB_PATH


def lib():
    """The loaded library with typed entry points; raises SsrHipUnavailable if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SsrHipUnavailable(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                f"(or make -C {CSRC}). This package has no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 absent
        raise SsrHipUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.ssrhip_version() != ABI_VERSION:
        raise SsrHipUnavailable(f"{LIB_PATH} has ABI {L.ssrhip_version()}, this package was written against {ABI_VERSION}: rebuild it (make -C {CSRC})")
    for i, st in enumerate(ABI_STRUCTS):
        if L.ssrhip_sizeof(i) != C.sizeof(st):
            raise SsrHipUnavailable(f"ABI mismatch: sizeof({st.__name__}) = {C.sizeof(st)} but the library says {L.ssrhip_sizeof(i)}")
    _lib = L
    return L


def check(rc: int, what: str = "ssrhip"):
    if rc != 0:
        msg = lib().ssrhip_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Raw device (or host) pointer of a torch tensor / None."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
