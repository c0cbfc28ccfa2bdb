"""GPU: BASELINE.json's configurations at their STATED shapes, each against the oracle on this box's CPU
plus size-independent properties:

  config 1   the same prompt and text, greedy: the first 12 steps against the oracle (token ids identical).
  config 2   830M, L=130 phonemes, the 160-frame prompt cut from demo/5895_34622_000026_000002.wav (tests/golden/), cfg_stride=5, top_k=40 / top_p=0.8 sampling — tokens identical to the
             oracle's under the same seed (the sampler consumes torch's CPU stream), 20 steps.
  config 3   830M speech editing on the reference's demo prompt demo/84_121550_000074_000000.wav (tests/golden/, 126,880 samples =
             397 frames, oracle/make_golden_demo.py): wav -> wmencodec codes -> single-span edit [150, 250) -> watermarked wav; the first
             greedy steps against the oracle on the codes of the real file, the whole chain run to completion for its properties.
  config 4   830M, 8 utterances x CFG = 16 rows of different lengths in one engine (the per-GPU shard of the 64-utterance batch).
  config 5   wmencodec encode + decode of randn(256, 1, 480000) * 0.1 on ONE GPU: shapes, finiteness, code range, the
             decode_latent round trip, and three clips cut from the batch compared with the oracle.
"""
import os
import time

import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import weights as W
from ssr_speech_amd.models.ssr import SSR_Speech
from oracle import codec as OC, lm as O

pytestmark = pytest.mark.gpu


def _model_830m():
    args = W.lm_args_830m()
    sd_gpu = W.lm_state_dict(args, seed=0, device="cuda")
    m = SSR_Speech(args)
    m.load_state_dict({k: v.cpu() for k, v in sd_gpu.items()})
    sd_cpu = O.reference_params({k: v.cpu() for k, v in sd_gpu.items()})
    return args, m.to("cuda").eval(), sd_gpu, sd_cpu


def _demo_prompt_codes(N=160):
    """The prompt BASELINE configs 1-2 name: demo/5895_34622_000026_000002.wav, first 160 frames (tests/golden/, oracle/make_golden_demo.py),
    tokenised by wmencodec (synthetic weights: there are no pretrained ones) -> int64 [1, N, 4] on the CPU. The LM and the oracle are fed
    the SAME codes."""
    import json
    from ssr_speech_amd.data.tokenizer import AudioTokenizer, tokenize_audio
    gold = os.path.join(os.path.dirname(__file__), "golden")
    facts = json.load(open(os.path.join(gold, "demo_5895_34622_000026_000002_160f.json")))
    ccfg = W.codec_config_full()
    tok = AudioTokenizer(device="cuda", config=ccfg, state_dict=W.codec_state_dict(ccfg, seed=0))
    codes, _, _ = tokenize_audio(tok, os.path.join(gold, "demo_5895_34622_000026_000002_160f.wav"))
    assert tuple(codes.shape) == (1, 4, facts["frames_320"]) == (1, 4, N)
    y = codes.transpose(2, 1).cpu().contiguous()
    assert len(np.unique(y.numpy())) > 20                               # real audio through a (random-weight) RVQ: not a constant
    return y


def test_config1_830m_greedy_on_the_demo_prompt_matches_oracle():
    """BASELINE config 1 as SURVEY 8(d) spells it out: the 160-frame cut of demo/5895_34622_000026_000002.wav, L = 130 phoneme ids, empty
    span at the prompt's end, greedy (top_k = 1), cfg_stride 5, aug_text — the first 12 steps on the GPU against the oracle on this
    box's CPU: identical token ids."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    args, m, _, sd_cpu = _model_830m()
    gen = torch.Generator().manual_seed(2024)
    L, N, steps = 130, 160, 12
    x = torch.randint(0, 100, (1, L), generator=gen)
    y = _demo_prompt_codes(N)
    unc = torch.randint(0, 101, (1, L), generator=gen)
    mi = torch.LongTensor([[[N, N]]])
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, kvcache=1, cfg_coef=1.5, cfg_stride=5, aug_text=True)
    trace = {}
    O.inference(sd_cpu, args, x, y, mi, uncond_x=unc, max_steps=steps, trace=trace, **kw)
    ref_tok = torch.stack(trace["samples"]).numpy()
    out = m.inference(x.cuda(), torch.LongTensor([L]), x.cuda(), torch.LongTensor([L]), y.cuda(), y.cuda(), mi.cuda(), uncond_x=unc, max_new_steps=steps, **kw)
    assert out is None and m.last_run["steps"] == steps
    eng = next(iter(m._engines.values()))
    assert np.array_equal(eng.tokens(0, steps), ref_tok)
    eng.close()                                                          # (hands the device's pairing slot on)


def test_config2_830m_sampled_top_k40_top_p08_matches_oracle():
    import gc
    gc.collect()                                                         # engines of earlier tests give the pairing slot back when collected
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    args, m, _, sd_cpu = _model_830m()
    gen = torch.Generator().manual_seed(2024)
    L, N, steps = 130, 160, 20
    x = torch.randint(0, 100, (1, L), generator=gen)
    y = _demo_prompt_codes(N)
    unc = torch.randint(0, 101, (1, L), generator=gen)
    mi = torch.LongTensor([[[N, N]]])
    kw = dict(top_k=40, top_p=0.8, temperature=1.0, stop_repetition=2, kvcache=1, cfg_coef=1.5, cfg_stride=5, aug_text=True)
    torch.manual_seed(4242)
    trace = {}
    O.inference(sd_cpu, args, x, y, mi, uncond_x=unc, max_steps=steps, trace=trace, **kw)
    rng_after_ref = torch.get_rng_state()
    ref_tok = torch.stack(trace["samples"]).numpy()
    torch.manual_seed(4242)
    out = m.inference(x.cuda(), torch.LongTensor([L]), x.cuda(), torch.LongTensor([L]), y.cuda(), y.cuda(), mi.cuda(),
                      uncond_x=unc, max_new_steps=steps, **kw)
    assert out is None and m.last_run["steps"] == steps
    eng = next(iter(m._engines.values()))
    assert eng.pairing, f"the default 2-row step on a 256-CU part runs the pair launches: {eng.pairing_why}"
    got_tok = eng.generated[0, :steps].cpu().numpy()
    assert np.array_equal(got_tok, ref_tok), (got_tok, ref_tok)
    assert len({tuple(t) for t in ref_tok.tolist()}) > 5            # it really sampled (not one token repeated)
    # the global generator is left where the reference's loop leaves it: exactly `steps` multinomial draws consumed
    assert torch.equal(torch.get_rng_state(), rng_after_ref)
    # the run above paired FFN2 with the next launch (csrc/gemv.hip gemv_pair_kernel, default on a 256-CU part); the unpaired step
    # (SSRHIP_GEMV_PAIR=0, read when the step is enqueued) must sample the same tokens from bit-identical logits
    logits_paired = eng.dbg_logits.clone() if getattr(eng, "dbg_logits", None) is not None else None
    m._engines.clear()
    eng.close()
    os.environ["SSRHIP_GEMV_PAIR"] = "0"
    try:
        torch.manual_seed(4242)
        m.inference(x.cuda(), torch.LongTensor([L]), x.cuda(), torch.LongTensor([L]), y.cuda(), y.cuda(), mi.cuda(),
                    uncond_x=unc, max_new_steps=steps, **kw)
    finally:
        del os.environ["SSRHIP_GEMV_PAIR"]
    eng0 = next(iter(m._engines.values()))
    assert not eng0.pairing and "SSRHIP_GEMV_PAIR=0" in eng0.pairing_why
    assert np.array_equal(eng0.generated[0, :steps].cpu().numpy(), ref_tok)
    if logits_paired is not None and getattr(eng0, "dbg_logits", None) is not None:
        assert torch.equal(eng0.dbg_logits, logits_paired)


def test_config4_830m_sixteen_rows_match_oracle():
    """8 utterances x CFG = 16 rows (the matrix-core GEMV's full tile, two attention row groups) at the 830M shape, greedy,
    every utterance with its own text / prompt length; tokens equal the oracle's run one by one on the CPU."""
    from ssr_speech_amd import layout as LY
    from ssr_speech_amd.engine import DecodeEngine, DecodeKnobs, LMWeightsArena
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    args = W.lm_args_830m()
    sd_gpu = W.lm_state_dict(args, seed=0, device="cuda")
    arena = LMWeightsArena(args, sd_gpu, torch.device("cuda"))
    sd_cpu = O.reference_params({k: v.cpu() for k, v in sd_gpu.items()})
    gen = torch.Generator().manual_seed(78)
    n_utt, steps = 8, 5
    eng = DecodeEngine(arena, n_utt, True, 512, 64, debug_logits=True)
    rows, cols, knobs, utts = [], [], [], []
    for u in range(n_utt):
        L, N = 18 + 4 * u, 150 + 7 * u                       # prompts of ~3 s as in config 4: context crosses a page boundary
        x = torch.randint(0, 100, (1, L), generator=gen)
        y = torch.randint(0, 2048, (1, N, 4), generator=gen)
        unc = torch.randint(0, 101, (1, L), generator=gen)
        mi = torch.LongTensor([[[N, N]]])
        cated, _, num_task, _ = LY.build_layout(y[0].T.numpy(), mi[0].numpy(), args)
        rows += [x[0].numpy(), unc[0].numpy()]
        cols.append(cated)
        knobs.append(DecodeKnobs(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=2, use_cfg=True,
                                 text_len=L, n_spans=num_task, seed=u))
        utts.append((x, y, unc, mi))
    eng.start(rows, cols, knobs)
    eng.decode(steps, use_graph=True)
    torch.cuda.synchronize()
    got = eng.generated[:, :steps].cpu().numpy()
    last = eng.dbg_logits.cpu().numpy()
    worst = 0.0
    for u, (x, y, unc, mi) in enumerate(utts):
        trace = {}
        O.inference(sd_cpu, args, x, y, mi, uncond_x=unc, max_steps=steps, trace=trace, top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2,
                    kvcache=1, cfg_coef=1.5, cfg_stride=2, aug_text=True)
        ref_tok = torch.stack(trace["samples"]).numpy()
        assert np.array_equal(got[u], ref_tok), (u, got[u], ref_tok)
        err = float(np.abs(last[u] - torch.stack(trace["edited_logits"]).numpy()[steps - 1]).max())
        worst = max(worst, err)
        assert err < 5e-4, (u, err)
    print(f"830M x 16 rows: max |logit diff| at step {steps}: {worst:.2e}")


def test_config4_830m_sampled_queue_through_eight_slots_matches_oracle():
    """multi-utterance parity at the 830M shape had been greedy only. 9 short utterances, SAMPLED (top_k 20 /
    top_p 0.9, CFG), run to completion through `inference_batch` = 8 utterance slots x 2 rows (the 16-row matrix-core path) with one
    refill; every utterance's tokens equal the oracle's CPU run under `torch.manual_seed(seed + i)` (the per-utterance RNG contract)."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    args, m, _, sd_cpu = _model_830m()
    gen = torch.Generator().manual_seed(91)
    utts = []
    for i in range(9):
        L, T = 5 + i % 4, 10 + 3 * i
        utts.append(dict(x=torch.randint(0, 100, (1, L), generator=gen), y=torch.randint(0, 2048, (1, T, 4), generator=gen),
                         mask_interval=torch.LongTensor([[[T, T]]])))
    kw = dict(top_k=20, top_p=0.9, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=2, aug_text=True)
    got = m.inference_batch(utts, seed=900, **kw)
    eng = next(iter(m._engines.values()))
    assert eng.B == 16 and eng.n_admitted == 9 and eng.n_refills == 1
    n_steps = 0
    for i, u in enumerate(utts):
        torch.manual_seed(900 + i)
        res, marks, masks, nmi = O.inference(sd_cpu, args, u["x"], u["y"], u["mask_interval"], kvcache=1, **kw)
        assert torch.equal(got[i][0].cpu(), res) and torch.equal(got[i][1], marks) and got[i][2] == masks, i
        n_steps += res.shape[-1] - u["y"].shape[1]
    assert n_steps > 100


def test_config5_codec_256_clips_of_30s_on_one_gpu():
    from ssr_speech_amd.codec.wmencodec import WMEncodecModel
    from helpers_codec import code_margins
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = W.codec_config_full()
    sd = W.codec_state_dict(cfg, seed=0)
    m = WMEncodecModel(cfg, sd, "cuda")
    B, n = 256, 480000
    g = torch.Generator().manual_seed(0)
    wav = torch.randn(B, 1, n, generator=g) * 0.1
    wav[-1, :, n - 12345:] = 0.0                                   # a ragged clip inside the batch (zero-padded tail, data/encode.py)
    wav_gpu = wav.cuda()
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes, scale, emb = m.encode(wav_gpu)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dec = m.decode(codes)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    peak_gb = torch.cuda.max_memory_allocated() / 2 ** 30
    T = n // cfg.hop
    print(f"config 5: encode {1000 * (t1 - t0):.0f} ms, decode {1000 * (t2 - t1):.0f} ms, peak {peak_gb:.1f} GiB "
          f"({B * 30 / (t1 - t0):.0f} / {B * 30 / (t2 - t1):.0f} audio-s per s)")
    # ---- properties at the full size
    assert scale is None and tuple(codes.shape) == (B, cfg.n_q, T) and codes.dtype == torch.int64
    assert tuple(emb.shape) == (B, cfg.dimension, T) and tuple(dec.shape) == (B, 1, n)
    assert int(codes.min()) >= 0 and int(codes.max()) < cfg.bins
    assert bool(torch.isfinite(emb).all()) and bool(torch.isfinite(dec).all())
    lat = m.decode_latent(codes[:4])                                # round trip of the quantiser: latent = sum of the chosen code vectors
    want = sum(torch.nn.functional.embedding(codes[:4, q].cpu(), sd[f"quantizer.vq.layers.{q}._codebook.embed"]) for q in range(cfg.n_q)).permute(0, 2, 1)
    torch.testing.assert_close(lat.cpu(), want, rtol=0, atol=1e-6)
    # items are independent: the first clip alone gives the same result as inside the batch of 256
    c1, _, e1 = m.encode(wav_gpu[:1])
    torch.testing.assert_close(e1, emb[:1], rtol=0, atol=2e-5)
    del lat, c1, e1
    # ---- three clips against the oracle (first / middle / the ragged last one)
    for b in (0, 131, B - 1):
        o_codes, _, o_emb = OC.encode(sd, wav[b:b + 1], cfg)
        np.testing.assert_allclose(emb[b:b + 1].cpu().numpy(), o_emb.numpy(), rtol=0, atol=2e-4)
        diff = codes[b:b + 1].cpu() != o_codes
        if diff.any():                                              # fp32 near-ties of the RVQ argmax: judged by the oracle's own margin
            marg = code_margins(sd, cfg, o_emb, o_codes)
            first = diff.float().cumsum(1) == 1
            assert (marg[diff & first] < 1e-4).all(), (b, int(diff.sum()))
        assert diff.float().mean() < 0.01, b
        o_dec = OC.decode(sd, codes[b:b + 1].cpu(), cfg)            # decode of the GPU's own codes: isolates the decoder
        np.testing.assert_allclose(dec[b:b + 1].cpu().numpy(), o_dec.numpy(), rtol=0, atol=2e-4)


def test_config5_wmdecode_256_clips_of_30s_checked_where_it_is_timed():
    """`bench.py` times `wmdecode` at 256 x 30 s in 4 batch lanes (marks = second half ones), but the largest comparison
    with the oracle was 20 clips x ~12 frames. Here the SAME call (the `--use_watermark` product path, wmencodec.py:358-375 /
    seanet.py:555-600: skip encoder with four taps, three stacked LSTMs over 1,500 steps, label-conditioned projections, detector):
    shapes and finiteness for all 256 clips, and three clips — the first of lane 0, one in the middle lane, the last of the last lane —
    wav and detector output against oracle/codec.py within 2e-4."""
    from ssr_speech_amd.codec.wmencodec import WMEncodecModel
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = W.codec_config_full()
    sd = W.codec_state_dict(cfg, seed=0)
    m = WMEncodecModel(cfg, sd, "cuda")
    m.lanes = 4
    B, n = 256, 480000
    T = n // cfg.hop
    g = torch.Generator().manual_seed(5)
    wav = torch.randn(B, 1, n, generator=g) * 0.1
    codes = torch.randint(0, cfg.bins, (B, cfg.n_q, T), generator=g)
    marks = torch.zeros(B, T, dtype=torch.long)
    marks[:, T // 2:] = 1
    marks[B - 1, : T // 4] = 1                                     # the last clip carries its own label pattern
    wav_gpu, codes_gpu, marks_gpu = wav.cuda(), codes.cuda(), marks.cuda()
    out, mk = m.wmdecode(codes_gpu, marks_gpu, wav_gpu, with_mark=True)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (B, 1, n) and tuple(mk.shape) == (B, T, 2)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(mk).all())
    out_nomark, none = m.wmdecode(codes_gpu, marks_gpu, wav_gpu, with_mark=False)      # the CLI's call (the detector output is discarded there)
    assert none is None and torch.equal(out_nomark, out)
    del out_nomark
    for b in (0, 131, B - 1):
        o_wav, o_mk = OC.wmdecode(sd, codes[b:b + 1], marks[b:b + 1], wav[b:b + 1], cfg)
        np.testing.assert_allclose(out[b:b + 1].cpu().numpy(), o_wav.numpy(), rtol=0, atol=2e-4)
        np.testing.assert_allclose(mk[b:b + 1].cpu().numpy(), o_mk.numpy(), rtol=0, atol=2e-4)


def test_config3_edit_of_the_reference_demo_wav_end_to_end():
    """BASELINE config 3 on the file it names (the chain wav -> codes -> edit -> wav had only been run on random codes).
    (a) the demo wav tokenises to 397 frames; the 830M LM's first 8 greedy CFG steps of the span [150, 250) edit on THOSE codes equal
    the oracle's (token ids identical); (b) `inference_one_sample` (edit mode, --use_watermark) runs the whole chain to completion:
    head + generated span + tail frames, 320 samples each, finite."""
    import argparse
    import json
    from ssr_speech_amd.data.tokenizer import AudioTokenizer, tokenize_audio
    from ssr_speech_amd.inference_scale import inference_one_sample
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    gold = os.path.join(os.path.dirname(__file__), "golden")
    fn = os.path.join(gold, "demo_84_121550_000074_000000.wav")
    facts = json.load(open(os.path.join(gold, "demo_84_121550_000074_000000.json")))
    ccfg = W.codec_config_full()
    tok = AudioTokenizer(device="cuda", config=ccfg, state_dict=W.codec_state_dict(ccfg, seed=0))
    codes, _, _ = tokenize_audio(tok, fn)
    assert tuple(codes.shape) == (1, 4, facts["frames_320"]) == (1, 4, 397)
    args = W.lm_args_830m()
    sd_gpu = W.lm_state_dict(args, seed=0, device="cuda")
    for k in range(4):          # keep a random-weight LM off the special ids the RVQ decoder rejects (as the reference's F.embedding would)
        sd_gpu[f"predict_layer.{k}.2.bias"][2048:] = -30.0
    m = SSR_Speech(args)
    m.load_state_dict({k: v.cpu() for k, v in sd_gpu.items()})
    m = m.to("cuda").eval()
    sd_cpu = O.reference_params({k: v.cpu() for k, v in sd_gpu.items()})
    gen = torch.Generator().manual_seed(33)
    L, steps = 120, 8
    x = torch.randint(0, 52, (1, L), generator=gen)
    unc = torch.randint(0, 101, (1, L), generator=gen)
    mi = torch.LongTensor([[[150, 250]]])
    y = codes.transpose(2, 1).cpu()
    kw = dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, kvcache=1, cfg_coef=1.5, cfg_stride=5, aug_text=True)
    trace = {}
    O.inference(sd_cpu, args, x, y, mi, uncond_x=unc, max_steps=steps, trace=trace, **kw)
    ref_tok = torch.stack(trace["samples"]).numpy()
    out = m.inference(x.cuda(), torch.LongTensor([L]), x.cuda(), torch.LongTensor([L]), y.cuda(), y.cuda(), mi.cuda(), uncond_x=unc, max_new_steps=steps, **kw)
    assert out is None
    got_tok = next(iter(m._engines.values())).generated[0, :steps].cpu().numpy()
    assert np.array_equal(got_tok, ref_tok), (got_tok, ref_tok)
    # (b) the public per-utterance entry point, to completion (the span ends by the reference's 10 x L cap: ~800 steps)
    symbols = [chr(ord("a") + i) for i in range(26)] + [chr(ord("A") + i) for i in range(26)]
    phn2num = {c: i for i, c in enumerate(symbols)}
    text = "".join(symbols[int(i)] for i in x[0])

    class Chars:
        def __call__(self, texts):
            return [[c for c in t if c != " "] for t in texts]

    decode_config = {"top_k": 1, "top_p": 1.0, "temperature": 1, "stop_repetition": 2, "kvcache": 1, "codec_audio_sr": 16000, "codec_sr": 50}
    torch.manual_seed(1)
    wav = inference_one_sample(m, argparse.Namespace(**vars(args)), phn2num, Chars(), tok, fn, text, text, mi[0], 1.5, 5, True, False, True, False, "cuda", decode_config)
    lr = m.last_run
    assert lr["done"] == 1 and wav.dim() == 3 and tuple(wav.shape[:2]) == (1, 1) and wav.shape[-1] % 320 == 0 and bool(torch.isfinite(wav).all())
    # frames: kept head [0, 150) + the generated span (steps - 3 delay columns - 1 eog column) + kept tail [250, 397)
    assert wav.shape[-1] // 320 == 150 + (lr["steps"] - 4) + (397 - 250), (wav.shape, lr["steps"])
