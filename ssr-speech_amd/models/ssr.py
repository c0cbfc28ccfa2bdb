"""`SSR_Speech` — drop-in for the reference's `models.ssr.SSR_Speech` on the inference path
(reference `models/ssr.py:88-812`): same constructor (`args` Namespace or `config` dict), same
`state_dict` keys, same `inference(...)` signature and 4-tuple return.  The arithmetic runs in
libssrhip.so (hand-written HIP for gfx950); this file holds only host orchestration.

There is no CPU fallback: `inference` raises if the HIP library or a GPU is missing.
`forward(batch)` (the training loss, models/ssr.py:280-379) is outside this package's scope and raises.
"""
from __future__ import annotations

import copy
import logging
import time
from argparse import Namespace
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .. import layout as LY
from ..engine import MAX_ROWS, DecodeEngine, DecodeKnobs, LMWeightsArena, TorchCpuNoiseFeed
from ..weights import lm_param_specs


def _set_param(root: nn.Module, dotted: str, p: nn.Parameter):
    mod = root
    parts = dotted.split(".")
    for name in parts[:-1]:
        if not hasattr(mod, name):
            mod.add_module(name, nn.Module())
        mod = getattr(mod, name)
    mod.register_parameter(parts[-1], p)


class SSR_Speech(nn.Module):
    def __init__(self, args: Optional[Namespace] = None, config: Optional[Dict] = None):
        super().__init__()
        if args is not None and config is not None:
            raise ValueError("Cannot provide both `args` and `config`.")          # ssr.py:99-100
        if args is None:
            if config is None:
                raise ValueError("Either `args` or `config` must be provided.")  # ssr.py:109-110
            args = Namespace(**config)
        self.args = copy.copy(args)
        if not getattr(self.args, "n_special", False):                            # ssr.py:114-116
            self.args.n_special = 3
        self.args.eos = getattr(self.args, "eos", -1)
        if isinstance(self.args.audio_vocab_size, str):                           # ssr.py:118-119
            self.args.audio_vocab_size = eval(self.args.audio_vocab_size)
        a = self.args
        self.n_text_tokens = a.text_vocab_size + 1
        assert a.text_pad_token == a.text_vocab_size, f"self.args.text_vocab_size: {a.text_vocab_size}, self.args.text_pad_token: {a.text_pad_token}"
        self.n_audio_tokens = [int(a.audio_vocab_size) + a.n_special + a.max_n_spans] * a.n_codebooks
        assert a.audio_vocab_size == a.empty_token, a.empty_token                 # ssr.py:125-130
        assert a.eog == a.audio_vocab_size + 1, a.eog
        assert a.audio_pad_token == a.audio_vocab_size + 2, a.audio_pad_token
        assert a.eos == a.audio_vocab_size + 3, a.eos
        assert a.sos == a.audio_vocab_size + 4, a.sos
        assert a.mts == a.audio_vocab_size + 5, a.mts
        # parameters under the reference's names (values are placeholders until load_state_dict)
        for name, (shape, _kind) in lm_param_specs(a).items():
            _set_param(self, name, nn.Parameter(torch.zeros(shape, dtype=torch.float32), requires_grad=False))
        self._arena: Optional[LMWeightsArena] = None
        self._engines: Dict[tuple, DecodeEngine] = {}
        self.debug_logits = False          # tests: keep the per-step post-edit logits
        self.page_order = None             # tests: permutation deciding which physical KV pages the allocator hands out first
        self.last_run: dict = {}

    # ------------------------------------------------------------------ nn.Module plumbing
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        # tolerate the training-only torchmetrics states a checkpoint may carry
        sd = {k: v for k, v in state_dict.items() if not k.startswith("accuracy_metrics.")}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        return out

    def _invalidate(self):
        for e in getattr(self, "_engines", {}).values():
            e.close()
        self._engines = {}
        self._arena = None

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, batch):
        raise NotImplementedError("SSR_Speech.forward (training loss, reference models/ssr.py:280-379) is outside the "
                                  "scope of ssr_speech_amd: only the inference hot path is implemented.")

    # ------------------------------------------------------------------ engine management
    def _get_engine(self, n_utt: int, use_cfg: bool, need_seq: int, need_steps: int, need_pages: Optional[int] = None) -> DecodeEngine:
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("ssr_speech_amd.SSR_Speech.inference needs the model on a ROCm GPU (model.to('cuda')); "
                               "there is no CPU path in this package.")
        if self._arena is None:
            self._arena = LMWeightsArena(self.args, self.state_dict(), dev)
        cap_seq = ((need_seq + 1023) // 1024) * 1024
        cap_steps = ((need_steps + 255) // 256) * 256
        if self._arena.ensure_positions(cap_seq):      # long text / long utterances: grow the position table (reference: extend_pe)
            for e in self._engines.values():
                e.close()
            self._engines = {}
        rows = n_utt * (2 if use_cfg else 1)
        # KV pool: `need_pages` = sum over rows of the pages each row can reach (short and long utterances share one pool); never more
        # than rows x pages-per-row
        need = rows * (cap_seq // 128)
        if need_pages is not None:
            need = min(need, int(need_pages))
        order_key = tuple(self.page_order) if self.page_order is not None else None
        # reuse: any resident engine of the same shape whose capacities cover the request (a batch of other lengths must not re-allocate
        # the pool, the noise buffer and re-capture the graph); a caller-chosen page order (tests) pins the pool size exactly
        for (k_utt, k_cfg, k_seq, k_steps, k_dbg, k_pool, k_order), e in self._engines.items():
            if (k_utt, k_cfg, k_dbg, k_order) == (n_utt, use_cfg, bool(self.debug_logits), order_key) and k_seq >= cap_seq and k_steps >= cap_steps \
                    and (k_pool >= need if order_key is None else k_pool == min(rows * (cap_seq // 128), ((need + 7) // 8) * 8)):
                return e
        pool = min(rows * (cap_seq // 128), ((need + 7) // 8) * 8)
        key = (n_utt, use_cfg, cap_seq, cap_steps, bool(self.debug_logits), pool, order_key)
        for e in self._engines.values():         # one engine (KV pool) resident at a time
            e.close()
        self._engines = {}
        order = None
        if self.page_order is not None:          # tests: a caller-chosen hand-out order of the physical pages
            order = [p for p in self.page_order if p < pool] if len(self.page_order) >= pool else None
        eng = DecodeEngine(self._arena, n_utt, use_cfg, cap_seq, cap_steps, debug_logits=self.debug_logits, pool_pages=pool, page_order=order)
        self._engines[key] = eng
        return eng

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def inference(
        self,
        x: torch.Tensor,
        x_lens: torch.Tensor,
        prompt_x: torch.Tensor,
        prompt_x_lens: torch.Tensor,
        y: torch.Tensor,
        prompt: torch.Tensor,
        mask_interval: List[torch.Tensor],
        top_k: int = -100,
        top_p: float = 1.0,
        temperature: float = 1.0,
        stop_repetition: int = -1,
        kvcache: int = 1,
        silence_tokens: List[int] = [1388, 1898, 131],
        cfg_coef: float = 1.5,
        cfg_stride: int = 1,
        aug_text: bool = False,
        aug_context: bool = False,
        cfg_pretrained: bool = False,
        *,
        noise: Optional[torch.Tensor] = None,
        uncond_x: Optional[torch.Tensor] = None,
        max_new_steps: Optional[int] = None,
        use_graph: bool = True,
    ):
        """Same contract as the reference's `SSR_Speech.inference` (models/ssr.py:504-812).

        `kvcache` is accepted and ignored (the engine always keeps a paged KV cache; the reference
        produces identical tokens for kvcache 0/1).  Keyword-only extras (not in the reference):
        `noise` [steps,K,card] Exp(1) draws replacing the multinomial generator draw, `uncond_x`
        overriding the random CFG text (ssr.py:574), `max_new_steps` (tests), `use_graph`."""
        t_call = time.perf_counter()
        K = self.args.n_codebooks
        assert cfg_coef >= 1.0, cfg_coef
        assert x.ndim == 2, x.shape
        assert x_lens.ndim == 1, x_lens.shape
        assert y.ndim == 3, y.shape
        y = y.transpose(2, 1)
        assert prompt.ndim == 3, prompt.shape
        prompt = prompt.transpose(2, 1)
        assert y.shape[0] == 1 and y.shape[1] == K, y.shape
        assert prompt.shape[0] == 1 and prompt.shape[1] == K, prompt.shape
        assert mask_interval.shape == torch.Size((1, mask_interval.shape[1], 2)), mask_interval
        # ssr.py:563-568: the context is prepended only when the masked spans are short (< 2 s at 50 Hz)
        context_len = int(sum(int(item[1] - item[0]) for item in mask_interval[0]))
        aug_context = bool(aug_context and context_len < 2 * 50)

        dev = self.device
        x_np = x.detach().cpu().numpy().astype(np.int64)
        y_np = y[0].detach().cpu().numpy().astype(np.int64)
        mi = mask_interval[0].detach().cpu().numpy().astype(np.int64)
        out_len = 0
        if aug_context:
            # ssr.py:578-594, 607-608: [prompt text ‖ text], [prompt codes ‖ codes], spans shifted by the prompt length
            assert prompt_x.ndim == 2, prompt_x.shape
            out_len = int(prompt.shape[2])
            x_np = np.concatenate([prompt_x.detach().cpu().numpy().astype(np.int64), x_np], axis=1)
            y_np = np.concatenate([prompt[0].detach().cpu().numpy().astype(np.int64), y_np], axis=1)
            mi = mi + out_len
        L = x_np.shape[1]
        text_rows = [x_np[0]]
        if aug_text:
            if cfg_pretrained:
                # ssr.py:576/587 + :631-634: the unconditional row is `text_vocab_size-1` repeated L times with text keys 1..L-1
                # padded out for every query. Those positions are then attended by nothing and their own outputs are dropped
                # (:275-276), and text / audio positions are embedded independently (:599-600, :662), so the row is exactly
                # the length-1 text [text_vocab_size-1] — the engine's rows carry their own text length.
                text_rows.append(np.asarray([int(self.args.text_vocab_size) - 1], dtype=np.int64))
            else:
                if uncond_x is None:
                    # drawn from the global CPU generator, before any sampling draw — exactly ssr.py:574/585
                    uncond_x = torch.randint(0, self.n_text_tokens, (1, L))
                text_rows.append(uncond_x.detach().cpu().numpy().astype(np.int64)[0])
        cated, mask_position, num_task, nmi = LY.build_layout(y_np, mi, self.args)
        T0 = cated.shape[1]
        # upper bound on steps: every span stops at the latest when y_len > 10*L (ssr.py:739) + K eog steps
        cap = max(10 * L + 2 - T0, 1) + num_task * (K + 1)
        if max_new_steps is not None:
            cap = min(cap, max_new_steps)
        eng = self._get_engine(1, bool(aug_text), L + T0 + cap + 8, cap)
        knobs = DecodeKnobs(top_k=top_k, top_p=top_p, temperature=temperature, stop_repetition=stop_repetition,
                            silence_tokens=tuple(int(s) for s in silence_tokens), cfg_coef=cfg_coef, cfg_stride=cfg_stride,
                            use_cfg=bool(aug_text), text_len=L, n_spans=num_task, seed=int(torch.initial_seed()))
        greedy = top_k == 1
        feed = None
        if noise is not None:
            eng.start(text_rows, [cated], [knobs], noise=noise.to(torch.float32).unsqueeze(0))
        elif not greedy:
            # reproduce the reference's CPU sampling stream: torch.multinomial(probs[K,card], 1) draws one Exp(1) tensor of that
            # shape per step from the global generator (after the uncond_x draw above) — streamed in 16 steps at a time beside the decode
            feed = TorchCpuNoiseFeed([None], K, eng.a.card)
            eng.start(text_rows, [cated], [knobs], host_noise=True)
        else:
            eng.start(text_rows, [cated], [knobs])
        states = eng.run_to_completion(chunk=16, use_graph=use_graph, max_total=cap, feed=feed)
        st = states[0]
        gen = eng.tokens(0, int(st.n_steps))
        self.last_run = dict(steps=st.n_steps, done=st.done, span_end=list(st.span_end), prefill_rows=(L + T0) * (2 if aug_text else 1),
                             t_start=t_call, t_first_chunk=eng.t_first_chunk, t_end=time.perf_counter())
        if st.done != 1:
            if max_new_steps is not None:
                return None
            raise RuntimeError(f"generation did not finish within {cap} steps (done={st.done})")
        ends = [0] + [st.span_end[i] for i in range(num_task)]
        spans = [gen[ends[i]:ends[i + 1]] for i in range(num_task)]
        res, marks, masks, nmi_out = LY.assemble(y_np, spans, nmi, self.args)
        if aug_context:                                         # ssr.py:806-810
            res, marks = res[:, out_len:], marks[out_len:]
            masks = [(a - out_len, b - out_len) for a, b in masks]
            nmi_out = [(a - out_len, b - out_len) for a, b in nmi_out]
        res_t = torch.from_numpy(np.ascontiguousarray(res)).unsqueeze(0).to(dev)
        marks_t = torch.from_numpy(np.ascontiguousarray(marks)).unsqueeze(0)          # CPU tensor, as the reference (ssr.py:805)
        logging.info(f"ssr_speech_amd: generated {st.n_steps} steps")
        return res_t, marks_t, masks, nmi_out

    # ------------------------------------------------------------------ batched decode (new capability)
    @torch.no_grad()
    def inference_batch(self, utterances, top_k: int = -100, top_p: float = 1.0, temperature: float = 1.0, stop_repetition: int = -1,
                        silence_tokens=(1388, 1898, 131), cfg_coef: float = 1.5, cfg_stride: int = 1, aug_text: bool = False,
                        seed: int = 0, first_index: int = 0, group: Optional[int] = None, use_graph: bool = True, refill: bool = True,
                        indices: Optional[Sequence[int]] = None):
        """Several independent utterances decoded in lock-step so that one pass over the weights serves all of them
        (the reference is strictly batch-1: `assert y.shape[0] == 1`, ssr.py:559, and loops `--sample_batch_size`
        sequentially, inference_v2.py:331-333).

        utterances: list of dicts {x: LongTensor[1,L], y: LongTensor[1,T,K], mask_interval: LongTensor[1,M,2]}.
        Parity contract: result i == `inference()` of utterance i alone after `torch.manual_seed(seed + first_index + i)`
        (per-utterance RNG streams, independent of grouping and of the DP world size); `indices[i]` replaces `first_index + i`
        when the list is a cost-balanced (non-contiguous) shard of a larger job (`dp.generate`).
        `refill=False` (A/B knob, bench): fixed groups of `group` utterances, each decoded until its longest member ends (round 2).
        Returns a list of the same 4-tuples `inference` returns."""
        K = self.args.n_codebooks
        assert cfg_coef >= 1.0, cfg_coef
        rows = 2 if aug_text else 1
        if group is None:
            group = MAX_ROWS // rows               # 16 rows per engine pass: 8 utterances with CFG
        assert 1 <= group * rows <= MAX_ROWS, (group, rows)
        dev = self.device
        greedy = top_k == 1
        if not utterances:
            return []
        jobs, metas, page_need = [], [], []
        cap_max, seq_max = 1, 1
        for j, u in enumerate(utterances):
            gi = int(indices[j]) if indices is not None else first_index + j
            rng = torch.Generator().manual_seed(seed + gi)      # same stream as `torch.manual_seed(seed + gi)` + a batch-1 run
            x_np = u["x"].detach().cpu().numpy().astype(np.int64)
            L = x_np.shape[1]
            text_rows = [x_np[0]]
            if aug_text:
                text_rows.append(torch.randint(0, self.n_text_tokens, (1, L), generator=rng).numpy().astype(np.int64)[0])     # ssr.py:574
            y_np = u["y"][0].transpose(1, 0).detach().cpu().numpy().astype(np.int64)
            mi = u["mask_interval"][0].detach().cpu().numpy().astype(np.int64)
            cated, mask_position, num_task, nmi = LY.build_layout(y_np, mi, self.args)
            T0 = cated.shape[1]
            cap = max(10 * L + 2 - T0, 1) + num_task * (K + 1)
            cap_max, seq_max = max(cap_max, cap), max(seq_max, L + T0 + cap + 8)
            page_need.append(rows * ((L + T0 + cap + 16) // 128 + 1))      # + the 16-step chunk the allocator provisions ahead
            kn = DecodeKnobs(top_k=top_k, top_p=top_p, temperature=temperature, stop_repetition=stop_repetition,
                             silence_tokens=tuple(int(s) for s in silence_tokens), cfg_coef=cfg_coef, cfg_stride=cfg_stride,
                             use_cfg=bool(aug_text), text_len=L, n_spans=num_task, seed=seed + gi)
            jobs.append(dict(text_rows=text_rows, audio_cols=cated, knobs=kn, gen=rng, cap=cap))
            metas.append((y_np, nmi, num_task))
        # Utterance slots of ONE engine, refilled as utterances finish (continuous batching): a finished utterance's rows take the next
        # pending utterance at the following 16-step poll instead of idling until the longest member of a fixed group ends.
        n_slots = min(group, len(jobs))
        while n_slots * rows == 3:                 # 3 rows is the one unsupported count: one more slot (it stays idle when there is no job for it)
            n_slots += 1
        # KV pool: at most n_slots utterances are resident at a time -> the n_slots largest page demands bound the concurrent need
        pages_sum = sum(sorted(page_need, reverse=True)[:n_slots])
        eng = self._get_engine(n_slots, bool(aug_text), seq_max, cap_max + 16, need_pages=pages_sum)
        if refill:
            outs = eng.run_queue(jobs, chunk=16, use_graph=use_graph, sampling=not greedy)
        else:
            outs = []
            for g0 in range(0, len(jobs), n_slots):
                outs += eng.run_queue(jobs[g0: g0 + n_slots], chunk=16, use_graph=use_graph, sampling=not greedy)
        results = []
        for j, (st, gen) in enumerate(outs):
            if st.done != 1:
                raise RuntimeError(f"utterance {j} did not finish within {jobs[j]['cap']} steps (done={st.done})")
            y_np, nmi, num_task = metas[j]
            ends = [0] + [st.span_end[i] for i in range(num_task)]
            spans = [gen[ends[i]:ends[i + 1]] for i in range(num_task)]
            res, marks, masks, nmi_out = LY.assemble(y_np, spans, nmi, self.args)
            results.append((torch.from_numpy(res).unsqueeze(0).to(dev), torch.from_numpy(marks).unsqueeze(0), masks, nmi_out))
        return results
