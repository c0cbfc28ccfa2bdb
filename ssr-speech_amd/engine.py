"""Host-side driver of the HIP decode engine (libssrhip.so): weight arena, paged KV cache, prefill
rows, hipGraph decode loop, read-back.  PyTorch is used for device memory and streams only.

The arithmetic replaced is the device part of `SSR_Speech.inference` (reference `models/ssr.py:597-754`);
integer layout code lives in `layout.py`.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import PAGE, MAX_CODEBOOKS


def sine_pe_table(n: int, dim: int) -> torch.Tensor:
    """Sinusoidal table built on the CPU in fp32 exactly like the reference does
    (models/modules/embedding.py:67-92: built on CPU, then moved to the device)."""
    pe = torch.zeros(n, dim)
    position = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


@dataclass
class DecodeKnobs:
    """Per-utterance decode parameters (the `inference()` keyword surface, models/ssr.py:513-523)."""
    top_k: int = -100
    top_p: float = 1.0
    temperature: float = 1.0
    stop_repetition: int = -1
    silence_tokens: Sequence[int] = (1388, 1898, 131)
    cfg_coef: float = 1.5
    cfg_stride: int = 1
    use_cfg: bool = False
    text_len: int = 0
    n_spans: int = 1
    seed: int = 0


def to_streaming_order(W: torch.Tensor) -> torch.Tensor:
    """[.., N, K] row-major -> the matrix-core GEMV's streaming order (include/ssrhip.h SSRHIP_WTILED_INDEX): rows in 8-row
    units (zero-padded), K in 16-float steps, each (unit, k-step) a 512-byte block indexed [k-slot 0..3][row 0..7][4 floats]."""
    *lead, N, K = W.shape
    assert K % 16 == 0, K
    U = (N + 7) // 8
    if U * 8 != N:
        W = torch.cat([W, W.new_zeros(*lead, U * 8 - N, K)], dim=-2)
    return W.reshape(*lead, U, 8, K // 16, 4, 4).permute(*range(len(lead)), -5, -3, -2, -4, -1).contiguous()


class LMWeightsArena:
    """Device-resident fp32 weights in the layout the kernels want (one-time repack at load)."""

    def __init__(self, args, sd: dict, device, max_pos: int = 8192):
        f32 = dict(dtype=torch.float32, device=device)
        self.args = args
        self.device = device
        self.D = int(args.d_model)
        self.H = int(args.nhead)
        self.L = int(args.num_decoder_layers)
        self.K = int(args.n_codebooks)
        V = int(args.audio_vocab_size)
        self.card = V + int(args.n_special) + int(args.max_n_spans)
        self.Hh = V // 2
        self.F = 4 * self.D
        self.n_text = int(args.text_vocab_size) + 1
        self.max_pos = max_pos
        g = lambda k: sd[k].detach().to(**f32).contiguous()
        self.text_emb = g("text_embedding.word_embeddings.weight")
        self.audio_emb = torch.stack([g(f"audio_embedding.{k}.word_embeddings.weight") for k in range(self.K)]).contiguous()
        self.alpha_text = float(sd["text_positional_embedding.alpha"].reshape(-1)[0])
        self.alpha_audio = float(sd["audio_positional_embedding.alpha"].reshape(-1)[0])
        self.pe = sine_pe_table(max_pos, self.D).to(**f32).contiguous()
        self.generation = 0     # bumped whenever a device pointer handed to an engine changes (engines re-create their ctx)
        # LayerNorm's affine part is folded into the Linear that follows it (one-time repack at load):
        #   Linear(LN(x)) = W (gamma * xhat + beta) + b = (W diag(gamma)) xhat + (b + W beta),  xhat = (x - mean) * rstd
        # so the decode GEMV only standardises x (in registers, no gamma/beta traffic); ln*_w / ln*_b become ones / zeros.
        self.ln_folded = True
        ones, zeros = torch.ones(self.D, **f32), torch.zeros(self.D, **f32)

        def fold(wk, bk, gk, bek):
            Wm, bm, gam, bet = g(wk), g(bk), g(gk), g(bek)
            return (Wm * gam.unsqueeze(0)).contiguous(), (bm.double() + Wm.double() @ bet.double()).to(torch.float32).contiguous()

        self.layers = []
        for l in range(self.L):
            p = f"decoder.layers.{l}."
            in_w, in_b = fold(p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", p + "norm1.weight", p + "norm1.bias")
            f1_w, f1_b = fold(p + "linear1.weight", p + "linear1.bias", p + "norm2.weight", p + "norm2.bias")
            self.layers.append(dict(
                ln1_w=ones, ln1_b=zeros, in_proj_w=in_w, in_proj_b=in_b,
                out_proj_w=g(p + "self_attn.out_proj.weight"), out_proj_b=g(p + "self_attn.out_proj.bias"),
                ln2_w=ones, ln2_b=zeros, ffn1_w=f1_w, ffn1_b=f1_b,
                ffn2_w=g(p + "linear2.weight"), ffn2_b=g(p + "linear2.bias")))
        self.lnf_w, self.lnf_b = ones, zeros
        gam, bet = g("decoder.norm.weight"), g("decoder.norm.bias")
        h1w = torch.cat([g(f"predict_layer.{k}.0.weight") for k in range(self.K)], 0)
        h1b = torch.cat([g(f"predict_layer.{k}.0.bias") for k in range(self.K)], 0)
        self.head1_w = (h1w * gam.unsqueeze(0)).contiguous()
        self.head1_b = (h1b.double() + h1w.double() @ bet.double()).to(torch.float32).contiguous()
        self.head2_w = torch.stack([g(f"predict_layer.{k}.2.weight") for k in range(self.K)]).contiguous()
        self.head2_b = torch.stack([g(f"predict_layer.{k}.2.bias") for k in range(self.K)]).contiguous()

    def ensure_streaming_copies(self) -> bool:
        """Second copy of the six matrices of a decode step in the streaming order of the 5..16-row GEMV (one-time repack; used
        only by engines with more than 4 rows, +3.3 GB at 830M). Returns True when the copies were created now."""
        if getattr(self, "_wt_ready", False):
            return False
        for lay in self.layers:
            for name in ("in_proj", "out_proj", "ffn1", "ffn2"):
                lay[name + "_wt"] = to_streaming_order(lay[name + "_w"])
        self.head1_wt = to_streaming_order(self.head1_w)
        self.head2_wt = to_streaming_order(self.head2_w)
        self._wt_ready = True
        self.generation += 1
        return True

    def ensure_split_planes(self) -> bool:
        """The four matrices of every layer as three bf16 planes each ([3][N][K], `ssrhip_split_weights`) for the PREFILL GEMMs: fp32
        operands split exactly, six cross products on the bf16 matrix cores (csrc/gemm_split.hip; +6 bytes per weight = +4.8 GB at 830M,
        built once on first use). `SSRHIP_PREFILL_SPLIT=0` keeps the prefill on the fp32 FMA chain. Returns True when created now."""
        if getattr(self, "_ws_ready", False) or os.environ.get("SSRHIP_PREFILL_SPLIT", "1")[:1] == "0":      # the C side's rule (engine.hip): a value that starts with '0' 
            return False
        lib = _lib.lib()
        for lay in self.layers:
            for name in ("in_proj", "out_proj", "ffn1", "ffn2"):
                Wm = lay[name + "_w"]
                planes = torch.empty(3 * Wm.numel(), dtype=torch.int16, device=Wm.device)
                _lib.check(lib.ssrhip_split_weights(Wm.data_ptr(), planes.data_ptr(), Wm.numel(), _lib.stream_ptr()), "ssrhip_split_weights")
                lay[name + "_ws"] = planes
        self._ws_ready = True
        self.generation += 1
        return True

    def ensure_positions(self, n: int) -> bool:
        """Grow the sinusoidal table so that positions [0, n) exist, like `SinePositionalEmbedding.extend_pe` does on demand
        (models/modules/embedding.py:66-92: no length limit in the reference). Returns True when the table was rebuilt
        (its device pointer changed: engines built on the old one must be discarded)."""
        if n <= self.max_pos:
            return False
        self.max_pos = ((int(n) + 4095) // 4096) * 4096
        self.pe = sine_pe_table(self.max_pos, self.D).to(dtype=torch.float32, device=self.device).contiguous()
        self.generation += 1
        return True

    def nbytes_per_step(self) -> int:
        """Algorithmic weight bytes one decode step must stream."""
        n = 0
        for lay in self.layers:
            n += sum(t.numel() for k, t in lay.items() if not k.endswith("_wt") and not k.endswith("_ws"))     # incl. the (now constant) LayerNorm vectors, as SURVEY §8d counts them
        n += self.lnf_w.numel() + self.lnf_b.numel()
        n += self.head1_w.numel() + self.head1_b.numel() + self.head2_w.numel() + self.head2_b.numel()
        n += (self.K + 1) * self.D  # K embedding rows + one pe row
        return 4 * n

    def c_struct(self):
        w = _lib.LMWeights()
        w.text_emb, w.audio_emb, w.pe = self.text_emb.data_ptr(), self.audio_emb.data_ptr(), self.pe.data_ptr()
        w.alpha_text, w.alpha_audio = self.alpha_text, self.alpha_audio
        self._arrays = {}
        for name in ("ln1_w", "ln1_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln2_w", "ln2_b",
                     "ffn1_w", "ffn1_b", "ffn2_w", "ffn2_b"):
            arr = (C.c_void_p * self.L)(*[lay[name].data_ptr() for lay in self.layers])
            self._arrays[name] = arr
            setattr(w, name, C.cast(arr, C.POINTER(C.c_void_p)))
        w.lnf_w, w.lnf_b = self.lnf_w.data_ptr(), self.lnf_b.data_ptr()
        w.head1_w, w.head1_b = self.head1_w.data_ptr(), self.head1_b.data_ptr()
        w.head2_w, w.head2_b = self.head2_w.data_ptr(), self.head2_b.data_ptr()
        if getattr(self, "_wt_ready", False):
            for name in ("in_proj_wt", "out_proj_wt", "ffn1_wt", "ffn2_wt"):
                arr = (C.c_void_p * self.L)(*[lay[name].data_ptr() for lay in self.layers])
                self._arrays[name] = arr
                setattr(w, name, C.cast(arr, C.POINTER(C.c_void_p)))
            w.head1_wt, w.head2_wt = self.head1_wt.data_ptr(), self.head2_wt.data_ptr()
        if getattr(self, "_ws_ready", False):
            for name in ("in_proj_ws", "out_proj_ws", "ffn1_ws", "ffn2_ws"):
                arr = (C.c_void_p * self.L)(*[lay[name].data_ptr() for lay in self.layers])
                self._arrays[name] = arr
                setattr(w, name, C.cast(arr, C.POINTER(C.c_void_p)))
        return w

    def dims(self):
        return _lib.LMDims(self.D, self.H, self.L, self.F, self.K, self.card, self.Hh, self.n_text, self.max_pos, int(self.ln_folded))


MAX_ROWS = 16   # gemv_mfma.hip: one 16-column MFMA tile


class PagePool:
    """Host-side allocator of the paged KV cache's physical pages (replaces the dense `past` tensor the reference re-concatenates
    every step, models/ssr.py:685-686 / modules/activation.py:626-631). A free list; pages are handed to a row when its
    sequence is about to cross into a new 128-position page and go back when its utterance is done. `order` fixes the order
    in which an untouched pool hands pages out (tests pass a shuffled one; the kernels only ever see the table)."""

    def __init__(self, n_pages: int, order: Optional[Sequence[int]] = None):
        self.n_pages = int(n_pages)
        order = list(range(self.n_pages)) if order is None else [int(p) for p in order]
        if sorted(order) != list(range(self.n_pages)):
            raise ValueError("page order must be a permutation of range(n_pages)")
        self._order = order
        self.reset()

    def reset(self):
        self._free = self._order[::-1]          # pop() hands out order[0] first
        self._owner = {}
        self.handed_out = []                    # (page, owner) in hand-out order since the last reset (tests / debugging)

    @property
    def n_free(self) -> int:
        return len(self._free)

    def take(self, owner) -> int:
        if not self._free:
            raise RuntimeError(f"KV page pool exhausted ({self.n_pages} pages of {PAGE} positions): raise pool_pages")
        p = self._free.pop()
        self._owner[p] = owner
        self.handed_out.append((p, owner))
        return p

    def give_back(self, pages: Sequence[int]):
        for p in pages:
            if p not in self._owner:
                raise RuntimeError(f"page {p} returned twice (or never taken)")
            del self._owner[p]
            self._free.append(p)


class TorchCpuNoiseFeed:
    """The Exp(1) tensors `torch.multinomial(probs[K, card], 1)` draws on the CPU — one `exponential_` of the logits' shape per
    decode step (reference models/ssr.py:85 via :713/:732; checked in oracle/make_golden.py::make_sampler) — produced a chunk of
    steps at a time so that the host draws for steps [s, s+n) while the GPU decodes the previous chunk. One contiguous
    `[n, K, card].exponential_()` consumes the generator exactly like n successive per-step draws (tests/test_noise_feed.py).

    generators[u] is utterance u's `torch.Generator`, or None for the global CPU generator (what the reference uses).
    `finish(u, n_taken)` puts that generator in the state it has after exactly `n_taken` per-step draws, i.e. where the
    reference leaves it — chunks drawn ahead of the stop are un-drawn."""

    _chunk_ok: dict = {}             # (K, card) -> does ONE [n, K, card] draw consume the generator like n [K, card] draws on this build?

    @classmethod
    def chunked_draws_match(cls, K: int, card: int) -> bool:
        """Self-check of the assumption above, once per shape and process: torch builds whose CPU `exponential_` seeds a per-call
        stream (some MKL/VSL configurations) would give a chunked draw other numbers than per-step draws — sampled runs would then
        silently stop reproducing `torch.multinomial`. On a mismatch the feed draws step by step into the same staging buffer."""
        key = (int(K), int(card))
        if key not in cls._chunk_ok:
            g1, g2 = torch.Generator().manual_seed(987654321), torch.Generator().manual_seed(987654321)
            a = torch.empty(3, K, card).exponential_(1, generator=g1)
            b = torch.stack([torch.empty(K, card).exponential_(1, generator=g2) for _ in range(3)])
            cls._chunk_ok[key] = bool(torch.equal(a, b)) and bool(torch.equal(g1.get_state(), g2.get_state()))
        return cls._chunk_ok[key]

    def __init__(self, generators, K: int, card: int):
        self.gens = list(generators)
        self.K, self.card = K, card
        self.chunked = self.chunked_draws_match(K, card)
        self.marks = [[] for _ in self.gens]      # per utterance: (first step of the chunk, generator state before drawing it)
        self.drawn = [0] * len(self.gens)

    def _get(self, u):
        g = self.gens[u]
        return torch.get_rng_state() if g is None else g.get_state()

    def _set(self, u, st):
        g = self.gens[u]
        if g is None:
            torch.set_rng_state(st)
        else:
            g.set_state(st)

    def draw(self, u: int, out: torch.Tensor):
        """Fill `out` [n, K, card] (CPU, may be pinned) with the draws of the next n steps of utterance u."""
        self.marks[u].append((self.drawn[u], self._get(u)))
        if self.chunked:
            out.exponential_(1, generator=self.gens[u])
        else:
            for i in range(out.shape[0]):
                out[i].exponential_(1, generator=self.gens[u])
        self.drawn[u] += out.shape[0]

    def reset_slot(self, u: int, generator):
        """Slot u now serves another utterance with its own generator (continuous batching): forget the old stream's bookkeeping."""
        self.gens[u] = generator
        self.marks[u] = []
        self.drawn[u] = 0

    def finish(self, u: int, n_taken: int):
        if n_taken >= self.drawn[u]:
            return
        s0, st = [m for m in self.marks[u] if m[0] <= n_taken][-1]
        self._set(u, st)
        if n_taken > s0:
            if self.chunked:
                torch.empty(n_taken - s0, self.K, self.card).exponential_(1, generator=self.gens[u])
            else:
                for _ in range(n_taken - s0):
                    torch.empty(self.K, self.card).exponential_(1, generator=self.gens[u])
        self.drawn[u] = n_taken


class DecodeEngine:
    """B rows (= n_utt x (2 if CFG else 1)) decoded in lock-step; one captured hipGraph per engine."""

    def __init__(self, arena: LMWeightsArena, n_utt: int, use_cfg: bool, max_seq: int, max_steps: int, debug_logits: bool = False,
                 pool_pages: Optional[int] = None, page_order: Optional[Sequence[int]] = None, pair_mode: int = 0):
        """max_seq: longest sequence (text + audio positions) any ONE row may reach; pool_pages: physical KV pages shared by all
        rows (default rows x pages-per-row, the no-sharing worst case; a batch of short and long utterances needs only the sum
        of their own page counts). pair_mode (2-row engines; include/ssrhip.h ssrhip_lm_buffers): 0 = pair launches if this engine
        gets its device's pairing slot, 1 = never, 2 = always (tests of the give-up path)."""
        self.lib = _lib.lib()
        self.a = arena
        dev = arena.device
        self.device = dev
        self.n_utt = n_utt
        self.use_cfg = use_cfg
        self.rows_per_utt = 2 if use_cfg else 1
        self.B = n_utt * self.rows_per_utt
        if self.B not in (1, 2, 4) and not (5 <= self.B <= MAX_ROWS):
            raise ValueError(f"rows B={self.B} not supported by this build (1, 2, 4 or 5..{MAX_ROWS})")
        self.max_pages = (max_seq + PAGE - 1) // PAGE
        self.max_seq = self.max_pages * PAGE
        arena.ensure_positions(self.max_seq)      # every text / audio position of a row is < its sequence capacity
        if self.B > 4:
            arena.ensure_streaming_copies()       # the matrix-core GEMV streams W in its own order
        arena.ensure_split_planes()               # the prefill GEMMs run on the bf16 matrix cores with exactly split operands
        self.max_steps = max_steps
        D, H, L, K = arena.D, arena.H, arena.L, arena.K
        self.hd = D // H
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        n_pages = self.B * self.max_pages if pool_pages is None else int(pool_pages)
        self.pages = PagePool(n_pages, page_order)
        self.scratch_page = n_pages              # one extra page: where the rows of a FINISHED utterance keep (harmlessly) writing
        self.kv_pool = torch.empty((n_pages + 1) * L * 2 * H * PAGE * self.hd, **f32)
        self.page_table = torch.full((self.B, self.max_pages), self.scratch_page, **i32)
        self._table_host = np.full((self.B, self.max_pages), self.scratch_page, dtype=np.int32)
        self._row_pages: List[List[int]] = [[] for _ in range(self.B)]
        self._kv0 = [0] * self.B                 # sequence length of each row after the prefill
        self._steps_enqueued = 0
        self._admit_step = [0] * n_utt           # value of _steps_enqueued when the slot's current utterance was admitted
        self.n_admitted = 0                      # utterances admitted since the last start / run_queue (tests: refill happened)
        self.n_refills = 0                       # ... of which into a slot another utterance had used before
        self._utt_live = [False] * n_utt
        # two-phase admission (run_queue): a slot that is being prefilled on the side stream. Its rows own KV pages, but the table the decode
        # step reads still shows the scratch page for them — the pages are entered in `page_table_warm`, which only that prefill reads.
        self._utt_warm = [False] * n_utt
        self.page_table_warm = torch.full((self.B, self.max_pages), self.scratch_page, **i32)
        self._table_warm_host = np.full((self.B, self.max_pages), self.scratch_page, dtype=np.int32)
        self._side_stream = None
        self._warm_pinned = self._warm_dev = None
        self._warm_free = None                    # event: the last activation has read the warm staging buffers
        self.t_first_chunk = 0.0
        rows = self.B if self.B <= 4 else MAX_ROWS      # > 4 rows: x / q / h are 16-column tiled buffers (include/ssrhip.h SSRHIP_TILED)
        self.x = torch.zeros(rows, D, **f32)
        self.q = torch.zeros(rows, D, **f32)
        self.h = torch.zeros(rows, max(arena.F, K * arena.Hh), **f32)
        self.logits = torch.zeros(self.B, K, arena.card, **f32)
        self.part_o = torch.zeros(self.B * H * self.max_pages * self.hd, **f32)
        self.part_ml = torch.zeros(self.B * H * self.max_pages * 2, **f32)
        self.next_tok = torch.zeros(self.B, MAX_CODEBOOKS, **i32)
        self.next_pos = torch.zeros(self.B, **i32)
        self.kv_pos = torch.zeros(self.B, **i32)
        self.row_len = torch.zeros(self.B, **i32)
        self.cfg_dev = torch.zeros(n_utt * C.sizeof(_lib.SamplerCfg), dtype=torch.uint8, device=dev)
        self.state_dev = torch.zeros(n_utt * C.sizeof(_lib.SamplerState), dtype=torch.uint8, device=dev)
        self.generated = torch.zeros(n_utt, max_steps, K, **i32)
        # ONE persistent buffer for host-drawn sampling noise: its pointer is baked into the captured graph, so generations
        # with / without host noise reuse the same context (ssrhip_sampler_cfg.use_noise selects per utterance)
        self.noise = torch.empty(n_utt, max_steps, K, arena.card, **f32)
        self._pinned = None
        self._copy_stream = None
        self._admit_pinned = self._admit_dev = self._admit_sent = None      # one-copy staging of an admission's integer arrays
        self._prefill_ws = None
        self._arena_gen = arena.generation
        self.dbg_logits = torch.zeros(n_utt, K, arena.card, **f32) if debug_logits else None
        self._w = arena.c_struct()
        self._ctx = None
        self.pair_mode = int(pair_mode)
        self.pairing = False                      # set by _create_ctx: does this engine's step run the pair launches?
        self.pairing_why = ""

    # ------------------------------------------------------------------ C structs
    def kv_struct(self):
        return _lib.KV(self.kv_pool.data_ptr(), self.page_table.data_ptr(), self.max_pages, self.a.L, self.a.H, self.hd)

    def _create_ctx(self):
        if self._ctx is not None:
            self.lib.ssrhip_lm_destroy(self._ctx)
            self._ctx = None
        b = _lib.LMBuffers()
        b.B, b.n_utt, b.max_splits, b.pair_mode = self.B, self.n_utt, self.max_pages, self.pair_mode
        b.x, b.q, b.h, b.logits = self.x.data_ptr(), self.q.data_ptr(), self.h.data_ptr(), self.logits.data_ptr()
        b.part_o, b.part_ml = self.part_o.data_ptr(), self.part_ml.data_ptr()
        b.next_tok, b.next_pos = self.next_tok.data_ptr(), self.next_pos.data_ptr()
        b.kv_pos, b.row_len = self.kv_pos.data_ptr(), self.row_len.data_ptr()
        b.kv = self.kv_struct()
        b.cfg, b.state = self.cfg_dev.data_ptr(), self.state_dev.data_ptr()
        b.noise = self.noise.data_ptr()
        b.generated = self.generated.data_ptr()
        b.dbg_logits = self.dbg_logits.data_ptr() if self.dbg_logits is not None else 0
        d = self.a.dims()
        ctx = C.c_void_p()
        _lib.check(self.lib.ssrhip_lm_create(C.byref(d), C.byref(self._w), C.byref(b), C.byref(ctx)), "ssrhip_lm_create")
        self._ctx = ctx
        why = C.create_string_buffer(256)
        self.pairing = bool(self.lib.ssrhip_lm_pairing(ctx, why, 256))
        self.pairing_why = why.value.decode(errors="replace")

    def close(self):
        if self._ctx is not None:
            self.lib.ssrhip_lm_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ set-up of one generation
    def start(self, text_rows: List[np.ndarray], audio_cols: List[np.ndarray], knobs: List[DecodeKnobs],
              noise: Optional[torch.Tensor] = None, host_noise: bool = False):
        """text_rows[b]: int array [L_b] (row b's text ids); audio_cols[u]: int array [K, T0_u]
        (layout-built prompt columns, WITHOUT the mask token that starts generation);
        knobs[u]. Runs the prefill and arms the decode state.

        Sampling noise: `noise` [n_utt, steps, K, card] (any device) is copied into the engine's buffer now; `host_noise=True`
        promises that `run_to_completion(feed=...)` will stream it in chunk by chunk; neither = on-device RNG (knobs.seed)."""
        a = self.a
        K = a.K
        assert len(text_rows) == self.B and len(audio_cols) == self.n_utt and len(knobs) == self.n_utt
        # KV pages: everything back to the pool; every slot is (re)admitted
        self.pages.reset()
        self._table_host[:] = self.scratch_page
        self._row_pages = [[] for _ in range(self.B)]
        self._kv0 = [0] * self.B
        self._steps_enqueued = 0
        self._admit_step = [0] * self.n_utt
        self._utt_live = [False] * self.n_utt
        self.n_admitted = self.n_refills = 0
        if noise is not None:
            assert noise.dim() == 4 and noise.shape[0] == self.n_utt and tuple(noise.shape[2:]) == (K, a.card), noise.shape
            n = min(noise.shape[1], self.max_steps)
            self.noise[:, :n].copy_(noise[:, :n].to(torch.float32), non_blocking=True)
            if n < self.max_steps:
                self.noise[:, n:].fill_(1.0)          # steps past the supplied draws: a defined value, never uninitialised memory
        rpu = self.rows_per_utt
        return self.admit(list(range(self.n_utt)), [text_rows[u * rpu:(u + 1) * rpu] for u in range(self.n_utt)], list(audio_cols), list(knobs),
                          use_noise=(noise is not None or host_noise))

    def admit_begin(self, slots: Sequence[int], text_rows: Sequence[Sequence[np.ndarray]], audio_cols: Sequence[np.ndarray],
                    knobs: Sequence[DecodeKnobs], use_noise: bool = False) -> dict:
        """First half of a TWO-PHASE admission (continuous batching without stalling the live rows): KV pages for the new rows, the
        integer arrays and the PREFILL of just those rows on a SIDE stream — concurrently with the decode chunks of the rows that keep
        running; the decode step does not see the new rows yet (their slots stay parked: `done` set, table rows on the scratch page; the
        prefill reads `page_table_warm`). `admit_finish(handle)` later puts them into the lock-step batch. Returns the handle."""
        return self.admit(slots, text_rows, audio_cols, knobs, use_noise=use_noise, _two_phase=True)

    def admit_finish(self, h: dict) -> None:
        """Second half: on the caller's (decode) stream, behind the side stream's prefill — sampler configuration / state, pending token,
        cache position and the table rows of the new slots, then the embedding of the pending tokens (every row: for rows in mid-decode
        that re-writes what the sampler's fused embedding left). A few small device copies; the rows step with the next chunk."""
        dev = self.device
        main = torch.cuda.current_stream(dev)
        main.wait_event(h["prefill_done"])
        csz, ssz = C.sizeof(_lib.SamplerCfg), C.sizeof(_lib.SamplerState)
        for i, u in enumerate(h["slots"]):
            self.cfg_dev[u * csz:(u + 1) * csz].copy_(torch.frombuffer(bytearray(bytes(h["cfgs"][i])), dtype=torch.uint8))
            self.state_dev[u * ssz:(u + 1) * ssz].copy_(torch.frombuffer(bytearray(bytes(h["sts"][i])), dtype=torch.uint8))
        idx = h["rows_d"].long()
        self.next_tok.index_copy_(0, idx, h["nt_d"])
        self.next_pos.index_copy_(0, idx, h["t0s"])
        self.kv_pos.index_copy_(0, idx, h["kv0"])
        self.row_len.index_copy_(0, idx, h["kv0"] + 1)
        for u in h["slots"]:
            self._utt_warm[u] = False
            self._utt_live[u] = True
            self._admit_step[u] = self._steps_enqueued
            for b in range(u * self.rows_per_utt, (u + 1) * self.rows_per_utt):
                self._table_host[b, :] = self._table_warm_host[b, :]
                self._table_warm_host[b, :] = self.scratch_page
        self.page_table.copy_(torch.from_numpy(self._table_host))
        _lib.check(self.lib.ssrhip_lm_embed_pending(self._ctx, _lib.stream_ptr()), "ssrhip_lm_embed_pending")
        self._warm_free = torch.cuda.Event()
        self._warm_free.record(main)
        self._keep_warm = h                       # staging views stay alive until the stream has consumed them

    def admit(self, slots: Sequence[int], text_rows: Sequence[Sequence[np.ndarray]], audio_cols: Sequence[np.ndarray],
              knobs: Sequence[DecodeKnobs], use_noise: bool = False, _two_phase: bool = False):
        """Put new utterances into the utterance slots `slots` (free ones: never used, or released by `release_utterance`) while
        the other slots keep decoding: KV pages from the pool, sampler configuration / state, pending input token, and the
        PREFILL of just those rows (the reference prefills one utterance at a time anyway: models/ssr.py:627-642). Everything is
        ordered on the caller's stream between two decode chunks; the captured step graph is untouched (it only holds pointers).
        text_rows[i]: the 1 (or 2 with CFG) text rows of slot slots[i]; audio_cols[i]: [K, T0]. Returns the prefilled row count."""
        a, dev = self.a, self.device
        K = a.K
        rpu = self.rows_per_utt
        assert len(slots) == len(text_rows) == len(audio_cols) == len(knobs) and len(set(slots)) == len(slots)
        toks, poss, kinds, seqs, rposs, lens = [], [], [], [], [], []
        rows_b: List[int] = []
        for u, trs, au in zip(slots, text_rows, audio_cols):
            assert 0 <= u < self.n_utt and not self._utt_live[u] and len(trs) == rpu, (u, len(trs))
            au = np.asarray(au, dtype=np.int64)
            T0 = au.shape[1]
            for r in range(rpu):
                b = u * rpu + r
                tx = np.asarray(trs[r], dtype=np.int64).reshape(-1)
                Lb = tx.shape[0]
                if Lb + T0 + 1 > self.max_seq:
                    raise ValueError("sequence exceeds engine capacity")
                t = np.zeros((Lb + T0, MAX_CODEBOOKS), dtype=np.int32)
                t[:Lb, 0] = tx
                t[Lb:, :K] = au.T
                toks.append(t)
                poss.append(np.concatenate([np.arange(Lb), np.arange(T0)]).astype(np.int32))
                kinds.append(np.concatenate([np.zeros(Lb), np.ones(T0)]).astype(np.int32))
                seqs.append(np.full(Lb + T0, b, dtype=np.int32))
                rposs.append(np.arange(Lb + T0, dtype=np.int32))
                lens.append(Lb + T0)
                rows_b.append(b)
                assert not self._row_pages[b], "slot still owns KV pages"
                self._kv0[b] = Lb + T0
            if _two_phase:
                assert not self._utt_warm[u]
                self._utt_warm[u] = True
            else:
                self._utt_live[u] = True
                self._admit_step[u] = self._steps_enqueued
            self.n_admitted += 1
        if _two_phase:                           # pages for the prompts (+ the first decoded position), entered in the WARM table only
            for b in rows_b:
                for have in range(min(self._kv0[b] // PAGE + 1, self.max_pages)):
                    pg = self.pages.take(b)
                    self._row_pages[b].append(pg)
                    self._table_warm_host[b, have] = pg
        else:
            self._grow_pages(0)                  # pages for the prompts (+ the first decoded position) of the new rows
        # Every integer array of this admission goes to the device in ONE copy: packed into a pinned staging buffer, sent asynchronously
        # into a persistent int32 workspace, addressed by views (ten small synchronous pageable copies per admit stalled the
        # rows that were still decoding). Layout (int32): tok[R][4] | pos[R] | kind[R] | seq[R] | rpos[R] | rlen[R] | seq_start[n+1] |
        # next_tok[rows][4] | t0[rows] | kv0[rows] | row index[rows].
        tok_h = np.concatenate(toks)
        R = tok_h.shape[0]
        nrow = len(rows_b)
        nt_h = np.zeros((nrow, MAX_CODEBOOKS), dtype=np.int32)
        nt_h[:, :K] = int(a.args.mts)
        t0_h = np.asarray([int(np.asarray(audio_cols[i // rpu]).shape[1]) for i in range(nrow)], dtype=np.int32)
        kv0_h = np.asarray(lens, dtype=np.int32)
        starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rpos_h = np.concatenate(rposs)
        parts = [tok_h.reshape(-1), np.concatenate(poss), np.concatenate(kinds), np.concatenate(seqs), rpos_h, rpos_h + 1, starts,
                 nt_h.reshape(-1), t0_h, kv0_h, np.asarray(rows_b, dtype=np.int32)]
        total = sum(int(p_.size) for p_ in parts)
        side = None
        if _two_phase:
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(dev)
            side = self._side_stream
            if self._warm_pinned is None or self._warm_pinned.numel() < total:
                cap_n = max(total, 2 * (self._warm_pinned.numel() if self._warm_pinned is not None else 0), 4096)
                torch.cuda.synchronize(dev)       # growth is rare: nothing in flight may still read the old buffer, and the new block (allocated
                self._warm_pinned = torch.empty(cap_n, dtype=torch.int32).pin_memory()
                self._warm_dev = torch.empty(cap_n, dtype=torch.int32, device=dev)     # on this stream, used on the side stream) has no pending user
            side.synchronize()                    # the previous two-phase prefill is through (>= one 16-step chunk ago): its staging is free
            if self._warm_free is not None:
                side.wait_event(self._warm_free)  # ... and the activation that read the staging views has run
            stage_np = self._warm_pinned.numpy()
            offs, o = [], 0
            for p_ in parts:
                stage_np[o:o + p_.size] = p_.astype(np.int32, copy=False).reshape(-1)
                offs.append(o)
                o += int(p_.size)
            with torch.cuda.stream(side):
                self._warm_dev[:total].copy_(self._warm_pinned[:total], non_blocking=True)
                self.page_table_warm.copy_(torch.from_numpy(self._table_warm_host))
            stage_dev = self._warm_dev
        elif self._admit_pinned is None or self._admit_pinned.numel() < total:
            cap_n = max(total, 2 * (self._admit_pinned.numel() if self._admit_pinned is not None else 0), 4096)
            self._admit_pinned = torch.empty(cap_n, dtype=torch.int32).pin_memory()
            self._admit_dev = torch.empty(cap_n, dtype=torch.int32, device=dev)
            self._admit_sent = None
        if not _two_phase:
            if self._admit_sent is not None:
                self._admit_sent.synchronize()   # the previous admission's copy has left the staging buffer (admissions are >= 16 steps apart)
            stage_np = self._admit_pinned.numpy()
            offs, o = [], 0
            for p_ in parts:
                stage_np[o:o + p_.size] = p_.astype(np.int32, copy=False).reshape(-1)
                offs.append(o)
                o += int(p_.size)
            self._admit_dev[:total].copy_(self._admit_pinned[:total], non_blocking=True)
            self._admit_sent = torch.cuda.Event()
            self._admit_sent.record(torch.cuda.current_stream(dev))
            stage_dev = self._admit_dev
        view = lambda i, n: stage_dev[offs[i]: offs[i] + n]
        tok, pos, kind, seq, rpos, rlen = view(0, 4 * R).view(R, MAX_CODEBOOKS), view(1, R), view(2, R), view(3, R), view(4, R), view(5, R)
        seq_start = view(6, len(lens) + 1)
        nt_d, t0s, kv0, rows_d = view(7, 4 * nrow).view(nrow, MAX_CODEBOOKS), view(8, nrow), view(9, nrow), view(10, nrow)

        # sampler configuration / state of the slots
        args = a.args
        csz, ssz = C.sizeof(_lib.SamplerCfg), C.sizeof(_lib.SamplerState)
        whole = list(slots) == list(range(self.n_utt))
        cfgs = (_lib.SamplerCfg * len(slots))()
        sts = (_lib.SamplerState * len(slots))()
        for i, (u, kn, au) in enumerate(zip(slots, knobs, audio_cols)):
            c = cfgs[i]
            c.top_k, c.top_p, c.temperature, c.stop_repetition = int(kn.top_k), float(kn.top_p), float(kn.temperature), int(kn.stop_repetition)
            c.cfg_coef, c.cfg_one_minus = float(kn.cfg_coef), float(1 - kn.cfg_coef)
            c.cfg_stride, c.use_cfg = int(kn.cfg_stride), int(self.use_cfg)
            sil = list(kn.silence_tokens)[: _lib.MAX_SILENCE]
            c.n_silence = len(sil)
            for j, sv in enumerate(sil):
                c.silence[j] = int(sv)
            c.text_len, c.n_spans = int(kn.text_len), int(kn.n_spans)
            c.empty_token, c.eog, c.eos, c.sos = int(args.empty_token), int(args.eog), int(args.eos), int(args.sos)
            c.mts, c.max_n_spans, c.max_steps = int(args.mts), int(args.max_n_spans), int(self.max_steps)
            c.seed_lo, c.seed_hi = int(kn.seed) & 0xFFFFFFFF, (int(kn.seed) >> 32) & 0xFFFFFFFF
            c.use_noise = int(use_noise)
            st = sts[i]
            st.span, st.num_gen, st.num_eog, st.num_cfg_tag, st.prev_token, st.consec_silence = 0, 0, 0, 1, -1, 0
            st.audio_pos = int(np.asarray(au).shape[1])
            st.n_steps, st.done = 0, 0
        if _two_phase:
            pass                                 # the decode step must not see the new slots yet: admit_finish writes cfg / state / pending token
        elif whole:
            self.cfg_dev.copy_(torch.frombuffer(bytearray(bytes(cfgs)), dtype=torch.uint8))
            self.state_dev.copy_(torch.frombuffer(bytearray(bytes(sts)), dtype=torch.uint8))
        else:
            for i, u in enumerate(slots):
                self.cfg_dev[u * csz:(u + 1) * csz].copy_(torch.frombuffer(bytearray(bytes(cfgs[i])), dtype=torch.uint8))
                self.state_dev[u * ssz:(u + 1) * ssz].copy_(torch.frombuffer(bytearray(bytes(sts[i])), dtype=torch.uint8))
        if _two_phase:
            assert self._arena_gen == a.generation and self._ctx is not None, "two-phase admission needs the running engine's context"
        if self._arena_gen != a.generation:          # the arena re-allocated a table (position table grown): refresh the pointers
            self._w = a.c_struct()
            self._arena_gen = a.generation
            self.close()
        if self._ctx is None:
            self._create_ctx()

        # first decode input of the new rows: the span-0 mask token at audio position T0 (ssr.py:655-662)
        idx = rows_d.long() if (rows_b != list(range(self.B)) and not _two_phase) else None
        if _two_phase:
            pass                                 # admit_finish
        elif idx is None:                        # every row (the first fill): plain copies, no index tensor
            self.next_tok.copy_(nt_d)
            self.next_pos.copy_(t0s)
            self.kv_pos.copy_(kv0)
            self.row_len.copy_(kv0 + 1)
        else:
            self.next_tok.index_copy_(0, idx, nt_d)
            self.next_pos.index_copy_(0, idx, t0s)
            self.kv_pos.index_copy_(0, idx, kv0)
            self.row_len.index_copy_(0, idx, kv0 + 1)

        # prefill workspaces: kept across admissions, grown when a larger prompt arrives
        f32 = dict(dtype=torch.float32, device=dev)
        D, F, H = a.D, a.F, a.H
        ms = (max(lens) + PAGE - 1) // PAGE
        if self._prefill_ws is None or self._prefill_ws["x"].shape[0] < R:
            if _two_phase:
                torch.cuda.synchronize(dev)      # (rare) the workspace is allocated on this stream and used on the side stream
            Rc = max(R, int(1.25 * (self._prefill_ws["x"].shape[0] if self._prefill_ws is not None else 0)))
            self._prefill_ws = dict(x=torch.empty(Rc, D, **f32), xn=torch.empty(Rc, D, **f32), qkv=torch.empty(Rc, 3 * D, **f32),
                                    o=torch.empty(Rc, D, **f32), h=torch.empty(Rc, F, **f32))
        ws = dict(self._prefill_ws)
        if os.environ.get("SSRHIP_PREFILL_ATTN_ROWWISE", "0") not in ("", "0"):      # A/B knob: the round-1 per-row attention needs its partials
            ws.update(part_o=torch.empty(R * H * ms * self.hd, **f32), part_ml=torch.empty(R * H * ms * 2, **f32))
        # rows of one sequence are contiguous and in position order: the tiled prefill attention needs only where each starts (seq_start above)
        p = _lib.PrefillArgs()
        p.tok, p.pos, p.kind = tok.data_ptr(), pos.data_ptr(), kind.data_ptr()
        p.row_seq, p.row_pos, p.row_len = seq.data_ptr(), rpos.data_ptr(), rlen.data_ptr()
        p.R, p.max_splits = R, ms
        for k, v in ws.items():
            setattr(p, k, v.data_ptr())
        p.seq_start, p.n_seq, p.max_len = seq_start.data_ptr(), len(lens), int(max(lens))
        if _two_phase:
            # on the SIDE stream, against the warm table, without the closing embedding (x is live decode state): see admit_begin
            p.table, p.no_embed = self.page_table_warm.data_ptr(), 1
            with torch.cuda.stream(side):
                _lib.check(self.lib.ssrhip_lm_prefill(self._ctx, C.byref(p), _lib.stream_ptr()), "ssrhip_lm_prefill")
                done = torch.cuda.Event()
                done.record(side)
            return dict(slots=list(slots), rows=R, prefill_done=done, cfgs=cfgs, sts=sts, nt_d=nt_d, t0s=t0s, kv0=kv0, rows_d=rows_d, ws=ws)
        # (the prefill ends by embedding the pending input token of EVERY row into x: for rows in mid-decode that re-writes the
        # very values the sampler's fused embedding left there — same function, same inputs)
        _lib.check(self.lib.ssrhip_lm_prefill(self._ctx, C.byref(p), _lib.stream_ptr()), "ssrhip_lm_prefill")
        self._keep = (ws, idx)                   # alive until the stream has consumed them (the integer arrays live in the engine's workspace)
        return R

    # ------------------------------------------------------------------ KV page bookkeeping
    def _grow_pages(self, steps_ahead: int):
        """Make sure every row of a live utterance owns the pages for positions [0, kv0 + steps_ahead] (the decode step at
        sequence length S appends position S), round-robin over rows so that concurrent rows interleave in the pool, then
        push the table if it changed. Ordered on the caller's stream before the launches that need it."""
        want = [0] * self.B
        for b in range(self.B):
            u = b // self.rows_per_utt
            if self._utt_live[u]:       # steps this row has been through = steps enqueued since ITS utterance was admitted
                want[b] = min((self._kv0[b] + max(steps_ahead - self._admit_step[u], 0)) // PAGE + 1, self.max_pages)
        changed = False
        more = True
        while more:
            more = False
            for b in range(self.B):
                have = len(self._row_pages[b])
                if have < want[b]:
                    p = self.pages.take(b)
                    self._row_pages[b].append(p)
                    self._table_host[b, have] = p
                    changed = more = True
        if changed:
            self.page_table.copy_(torch.from_numpy(self._table_host))       # pageable source: staged before this call returns

    def release_utterance(self, u: int):
        """Utterance u is done: its rows' pages go back to the pool and the rows are pointed at the scratch page (they stay in
        the lock-step batch and keep appending at a frozen position; nothing reads what they produce)."""
        if not self._utt_live[u]:
            return
        self._utt_live[u] = False
        for b in range(u * self.rows_per_utt, (u + 1) * self.rows_per_utt):
            self.pages.give_back(self._row_pages[b])
            self._row_pages[b] = []
            self._table_host[b, :] = self.scratch_page
        self.page_table.copy_(torch.from_numpy(self._table_host))

    # ------------------------------------------------------------------ decode
    def decode(self, n_steps: int, use_graph: bool = True):
        self._steps_enqueued += int(n_steps)
        self._grow_pages(self._steps_enqueued)
        _lib.check(self.lib.ssrhip_lm_decode(self._ctx, int(n_steps), int(use_graph), _lib.stream_ptr()), "ssrhip_lm_decode")

    def check_pairs(self) -> None:
        """Raise if a pair launch of this engine's step gave up (csrc/gemv.hip gemv_pair_kernel: a workgroup that waited ~1 s for the other
        255 — something else held CUs for that long — flags it and the step's results are garbage from there on). EVERY way a token
        leaves the engine goes through here first (`states`, `tokens`). The library hands the pairing slot to one engine per device
        (ssrhip_lm_create), so this needs a foreign kernel that squats on CUs; when it happens the engine recreates its context WITHOUT
        pair launches — the utterances in flight are lost (their KV cache is garbage), the next `start` / `run_queue` is correct."""
        if not self.pairing:
            return
        rc = self.lib.ssrhip_lm_pair_status(self._ctx, _lib.stream_ptr())
        if rc == 1:
            self.pair_mode = 1
            self._create_ctx()                    # destroys the context (gives the slot back), new one steps with the ordinary launches
            raise RuntimeError("a paired GEMV launch of the decode step gave up waiting for its other workgroups (another kernel held CUs "
                               "for about a second): every token since the last poll is invalid and the utterances in flight must be "
                               "submitted again. This engine now steps without pair launches (same tokens, ~5 % slower).")
        _lib.check(rc, "ssrhip_lm_pair_status")

    def states(self) -> List[_lib.SamplerState]:
        raw = bytes(self.state_dev.cpu().numpy().tobytes())
        self.check_pairs()
        arr = (_lib.SamplerState * self.n_utt).from_buffer_copy(raw)
        return list(arr)

    def tokens(self, u: int, n: int) -> np.ndarray:
        """The first n generated steps of slot u as int64 [n, K] — after the pair-launch check (never read `generated` directly)."""
        out = self.generated[u, :n].cpu().numpy().astype(np.int64)
        self.check_pairs()
        return out

    def run_to_completion(self, chunk: int = 16, use_graph: bool = True, max_total: Optional[int] = None,
                          feed: Optional[TorchCpuNoiseFeed] = None):
        """Decode in chunks of `chunk` steps until every utterance reports done (the flags are polled once per chunk).
        With `feed`, the host draws the NEXT chunk's sampling noise into pinned memory and uploads it on a copy stream while
        the GPU runs the current chunk (a 16-step chunk is ~14 ms of GPU time at 830M; its draws ~3 ms of host time)."""
        total = 0
        limit = self.max_steps if max_total is None else min(max_total, self.max_steps)
        dev = self.device
        main = torch.cuda.current_stream(dev) if feed is not None else None
        live = [True] * self.n_utt
        ready = None
        if feed is not None:
            K, card = self.a.K, self.a.card
            if self._pinned is None or self._pinned[0].shape[1] != chunk:
                self._pinned = [torch.empty(self.n_utt, chunk, K, card, dtype=torch.float32).pin_memory() for _ in range(2)]
                self._copy_stream = torch.cuda.Stream(dev)
            slot_free = [None, None]

            def stage(ci: int, s0: int, s1: int):
                """draw + upload the noise of steps [s0, s1) (chunk index ci); returns the event the decode must wait for"""
                buf = self._pinned[ci & 1]
                if slot_free[ci & 1] is not None:
                    slot_free[ci & 1].synchronize()            # the upload that last used this staging slot has finished
                for u in range(self.n_utt):
                    if live[u]:
                        feed.draw(u, buf[u, : s1 - s0])
                with torch.cuda.stream(self._copy_stream):
                    self.noise[:, s0:s1].copy_(buf[:, : s1 - s0], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                slot_free[ci & 1] = ev
                return ev

            ready = stage(0, 0, min(chunk, limit))             # overlaps with the prefill `start()` enqueued
        ci = 0
        states = None
        while total < limit:
            n = min(chunk, limit - total)
            if ready is not None:
                main.wait_event(ready)
            self.decode(n, use_graph)
            total += n
            if feed is not None and total < limit:
                ready = stage(ci + 1, total, min(total + chunk, limit))      # host works while the GPU decodes chunk ci
            ci += 1
            states = self.states()                              # blocks until chunk ci-1 has finished
            if ci == 1:
                self.t_first_chunk = time.perf_counter()        # first `chunk` frames exist (bench: time to first frames)
            live = [not s_.done for s_ in states]
            for u in range(self.n_utt):
                if not live[u]:
                    self.release_utterance(u)
            if not any(live):
                break
        if states is None:
            states = self.states()
        if feed is not None:
            for u, s_ in enumerate(states):
                feed.finish(u, int(s_.n_steps))
        return states

    # ------------------------------------------------------------------ continuous batching
    def run_queue(self, jobs: Sequence[dict], chunk: int = 16, use_graph: bool = True, sampling: bool = False):
        """Decode a QUEUE of utterances through this engine's utterance slots with row refill: the slots run in lock-step; every
        `chunk` steps the done flags are read, a finished utterance's tokens are collected and its KV pages released, and the next
        pending utterance is admitted into those rows (`admit`: prefill of just those rows, sampler state reset, same graph). A
        lock-step group would instead idle its finished rows until its longest member ends. The reference runs utterances one
        after the other (inference_v2.py:331-333); the result of each job does not depend on what shares the engine with it.

        jobs[i]: {text_rows: [row0(, row1)], audio_cols: [K, T0], knobs: DecodeKnobs, gen: torch.Generator | None (sampling),
        cap: max steps of this utterance}. Returns a list of (SamplerState, generated int64 [n_steps, K]) in job order."""
        n_jobs = len(jobs)
        results: List[Optional[tuple]] = [None] * n_jobs
        if n_jobs == 0:
            return results
        dev = self.device
        K, card = self.a.K, self.a.card
        slot_job: List[Optional[int]] = [None] * self.n_utt
        local: List[int] = [0] * self.n_utt           # steps enqueued for the slot's current utterance
        # longest (by its step cap) first: the utterances that will run longest start earliest and the short ones fill the slots freed
        # towards the end — the tail with a few long utterances on a mostly idle engine shrinks (simulated on the bench's ragged queue:
        # 344 -> 327 chunks). Which utterance sits in which slot when does not change any result.
        pending = sorted(range(n_jobs), key=lambda j: -int(jobs[j]["cap"])) if n_jobs > self.n_utt else list(range(n_jobs))
        feed = TorchCpuNoiseFeed([None] * self.n_utt, K, card) if sampling else None
        main = torch.cuda.current_stream(dev)
        if sampling and (self._pinned is None or self._pinned[0].shape[1] != chunk or len(self._pinned) < 3):
            self._pinned = [torch.empty(self.n_utt, chunk, K, card, dtype=torch.float32).pin_memory() for _ in range(3)]
            self._copy_stream = torch.cuda.Stream(dev)
        slot_free = [None, None, None]
        waits: List[torch.cuda.Event] = []

        def upload(buf_i: int, slots: Sequence[int]):
            """draw the next `chunk` steps of `slots` into staging buffer buf_i and send them to noise[u, local[u] : +chunk]"""
            buf = self._pinned[buf_i]
            if slot_free[buf_i] is not None:
                slot_free[buf_i].synchronize()
            todo = []
            for u in slots:
                n = min(chunk, self.max_steps - feed.drawn[u])
                if n > 0:
                    off = feed.drawn[u]
                    feed.draw(u, buf[u, :n])
                    todo.append((u, off, n))
            with torch.cuda.stream(self._copy_stream):
                for u, off, n in todo:
                    self.noise[u, off:off + n].copy_(buf[u, :n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            slot_free[buf_i] = ev
            return ev

        # Two-phase admission is built, bit-exact and OFF by default: on the bench's ragged queue (64 utterances through 8
        # slots, 56 refills) it measured 19,785 codec-tokens/s against 19,930 for the blocking form, same token CRC
        # (profiles/r05_microbench/dp64_ragged_two_phase_ab.log). The 9 ms of prefill work do not disappear by moving to a side stream: the
        # prefill's workgroups take CU slots from the decode step's 83 dependent launches, each of which then starts late — the two
        # time-slice the GPU instead of overlapping HBM-bound with matrix-bound work — and the admitted slot idles one more 16-step chunk.
        two_phase = os.environ.get("SSRHIP_ADMIT_TWO_PHASE", "0") not in ("", "0")

        def take_jobs(slots: Sequence[int]):
            take = [pending.pop(0) for _ in slots[: len(pending)]]
            return list(slots[: len(take)]), take

        def admit_into(slots: Sequence[int]):
            """blocking admission (the first fill; refills with SSRHIP_ADMIT_TWO_PHASE=0): prefill on the decode stream"""
            slots, take = take_jobs(slots)
            if not take:
                return
            for u, j in zip(slots, take):
                slot_job[u] = j
                local[u] = 0
                if sampling:
                    feed.reset_slot(u, jobs[j].get("gen"))
            self.admit(slots, [jobs[j]["text_rows"] for j in take], [jobs[j]["audio_cols"] for j in take], [jobs[j]["knobs"] for j in take],
                       use_noise=sampling)
            if sampling:
                waits.append(upload(2, slots))                 # the new utterances' first chunk, before their first step

        def begin_into(slots: Sequence[int]):
            """two-phase admission, first half: the prefill of the new rows starts on the side stream NOW and overlaps the decode chunk that
            is already enqueued; the slots join the lock-step batch at the next poll (`finish_warm`)"""
            slots, take = take_jobs(slots)
            if not take:
                return None
            h = self.admit_begin(slots, [jobs[j]["text_rows"] for j in take], [jobs[j]["audio_cols"] for j in take],
                                 [jobs[j]["knobs"] for j in take], use_noise=sampling)
            h["jobs"] = take
            return h

        def finish_warm(h: dict):
            self.admit_finish(h)
            for u, j in zip(h["slots"], h["jobs"]):
                slot_job[u] = j
                local[u] = 0
                if sampling:
                    feed.reset_slot(u, jobs[j].get("gen"))
            if sampling:
                waits.append(upload(2, h["slots"]))             # the new utterances' first chunk, before their first step

        # first fill: as `start` (pool reset), but only as many slots as there are jobs; the others stay parked on the scratch page
        self.pages.reset()
        self._table_host[:] = self.scratch_page
        self._table_warm_host[:] = self.scratch_page
        self._row_pages = [[] for _ in range(self.B)]
        self._kv0 = [0] * self.B
        self._steps_enqueued = 0
        self._admit_step = [0] * self.n_utt
        self._utt_live = [False] * self.n_utt
        self._utt_warm = [False] * self.n_utt
        self.n_admitted = self.n_refills = 0
        # idle slots: done = 1 so that the sampler leaves them alone
        idle = (_lib.SamplerState * self.n_utt)()
        for st in idle:
            st.done = 1
        self.state_dev.copy_(torch.frombuffer(bytearray(bytes(idle)), dtype=torch.uint8))
        self.page_table.copy_(torch.from_numpy(self._table_host))
        admit_into(list(range(self.n_utt)))
        ci = 0
        ready = None
        warm = None                                             # handle of the slots being prefilled on the side stream
        in_flight: List[int] = []                               # slots stepping in the chunk that is enqueued and not yet polled

        def enqueue_chunk():
            nonlocal waits, ready, ci, in_flight
            for ev in waits:
                main.wait_event(ev)
            waits = []
            if ready is not None:
                main.wait_event(ready)
            in_flight = [u for u in range(self.n_utt) if slot_job[u] is not None]
            self.decode(chunk, use_graph)
            for u in in_flight:
                local[u] += chunk
            if sampling:
                ready = upload(ci & 1, in_flight)               # host draws the next chunk while the GPU runs this one
            ci += 1

        enqueue_chunk()
        while True:
            states = self.states()                              # blocks until the enqueued chunk has finished
            if ci == 1:
                self.t_first_chunk = time.perf_counter()
            freed = []
            for u in in_flight:
                st, j = states[u], slot_job[u]
                cap_j = min(int(jobs[j]["cap"]), self.max_steps)
                over = local[u] >= cap_j
                if st.done or over:
                    n = int(st.n_steps)
                    if st.done and n > cap_j:
                        # the flags are polled every `chunk` steps: an utterance may finish up to chunk - 1 steps past ITS OWN cap before the
                        # poll sees it. The fixed-group path bounded every member by the group's cap; here the bound is per utterance, and a
                        # result longer than its cap is the same failure as not finishing (the caller raises on done != 1).
                        st.done = 2
                    gen = self.tokens(u, n)
                    snap = _lib.SamplerState.from_buffer_copy(bytes(st))
                    results[j] = (snap, gen)
                    if sampling:
                        feed.finish(u, n)
                    if not st.done:                             # ran into its cap without finishing: park the slot (the caller raises)
                        parked = _lib.SamplerState.from_buffer_copy(bytes(st))
                        parked.done = 1
                        ssz = C.sizeof(_lib.SamplerState)
                        self.state_dev[u * ssz:(u + 1) * ssz].copy_(torch.frombuffer(bytearray(bytes(parked)), dtype=torch.uint8))
                    self.release_utterance(u)
                    slot_job[u] = None
                    freed.append(u)
            in_flight = []
            # the slots whose prefill was started one poll ago join the batch now (their prefill had a whole chunk to finish)
            if warm is not None:
                finish_warm(warm)
                warm = None
            free_slots = [u for u in range(self.n_utt) if slot_job[u] is None]
            any_live = any(j is not None for j in slot_job)
            if two_phase and any_live:
                # keep the GPU busy FIRST: the next chunk of the live rows is enqueued before the host prepares the admission, whose
                # prefill then overlaps that chunk on the side stream
                enqueue_chunk()
                if free_slots and pending:
                    self.n_refills += min(len(free_slots), len(pending))
                    warm = begin_into(free_slots)
                continue
            if free_slots and pending:                          # nothing is decoding (or the knob is off): the blocking form
                self.n_refills += min(len(free_slots), len(pending))
                admit_into(free_slots)
            if not any(j is not None for j in slot_job):
                break
            enqueue_chunk()
        return results

    def time_kernels(self, n_steps: int):
        """Event-timed eager steps: list of (kind, avg_us) per launch slot of one decode step
        (kind: 'gemv' | 'attn' | 'sample')."""
        self._steps_enqueued += int(n_steps)
        self._grow_pages(self._steps_enqueued)
        n_out = 8 * self.a.L + 16
        us = (C.c_float * n_out)()
        kind = (C.c_int32 * n_out)()
        n = self.lib.ssrhip_lm_time_steps(self._ctx, int(n_steps), _lib.stream_ptr(), us, kind, n_out)
        if n < 0:
            _lib.check(n, "ssrhip_lm_time_steps")
        names = ("gemv", "attn", "sample")
        return [(names[kind[i]], float(us[i])) for i in range(n)]

    def time_category(self, kind: str, n_replays: int = 50):
        """(avg_us_per_launch, launches_per_step) of one kernel category, graph-chained (no per-launch event overhead).
        Leaves the hidden-state buffers dirty: `start()` again before decoding."""
        us = C.c_float()
        n = C.c_int32()
        if kind == "sample":                     # the sampler advances the positions on every replay (3 warm-up replays inside)
            self._steps_enqueued += int(n_replays) + 3
            self._grow_pages(self._steps_enqueued)
        _lib.check(self.lib.ssrhip_lm_time_category(self._ctx, ("gemv", "attn", "sample").index(kind), int(n_replays),
                                                    _lib.stream_ptr(), C.byref(us), C.byref(n)), "ssrhip_lm_time_category")
        return float(us.value), int(n.value)
