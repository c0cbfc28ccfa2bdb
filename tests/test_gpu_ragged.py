"""GPU: the batched-TTS path end to end. The codec's ragged batch decode (`decode_ragged`, `wmdecode_ragged`:
items of different lengths in one dense pass, each with its own halo) must give, per item, what a batch-1 call gives — the SEANet
convolutions are not causal, so zero-padding a short item to the longest would change its tail; and `dp.synthesize` (shard ->
lock-step decode -> all-gather -> sharded ragged codec decode) must equal one `inference_one_sample` call per utterance
(reference: inference_v2.py:331-358, inference_scale.py:63-86, wmencodec.py:341-375).

Tolerance: the per-sample arithmetic is the same; only the LSTM step kernel differs with the batch size (B <= 4 vs 16-item
MFMA tiles: another summation order), so 2e-5 absolute on O(1) waveforms; against the oracle the codec's usual 2e-4."""
import argparse
import dataclasses

import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import dp, weights as W
from ssr_speech_amd.codec.wmencodec import WMEncodecModel
from ssr_speech_amd.data.tokenizer import AudioTokenizer, write_wav
from ssr_speech_amd.inference_scale import inference_one_sample
from ssr_speech_amd.models.ssr import SSR_Speech
from oracle import codec as OC

pytestmark = pytest.mark.gpu
LENS = [37, 5, 64, 36, 1, 12, 64, 23, 9]          # frames; includes a 1-frame item (shorter than every reflect pad) and two equal ones


def _codes(cfg, lens, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, cfg.bins, (1, cfg.n_q, n), generator=g) for n in lens]


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
@pytest.mark.parametrize("full", [False, True])
def test_decode_ragged_equals_batch1_decode(pad_mode, full):
    cfg = dataclasses.replace(W.codec_config_full() if full else W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64), pad_mode=pad_mode)
    sd = W.codec_state_dict(cfg, seed=21)
    m = WMEncodecModel(cfg, sd, "cuda")
    codes = _codes(cfg, LENS, 3)
    m.RAGGED_STEP_US = 1e9                      # one bucket: the 1-frame item rides with the 64-frame ones
    assert len(m._ragged_buckets(LENS)) == 1
    got = m.decode_ragged([c.cuda() for c in codes])
    for i, c in enumerate(codes):
        one = m.decode(c.cuda())
        assert got[i].shape == one.shape == (1, 1, LENS[i] * cfg.hop)
        torch.testing.assert_close(got[i], one, rtol=0, atol=2e-5)
    # and against the oracle directly (three items incl. the shortest)
    for i in (4, 1, 7):
        np.testing.assert_allclose(got[i].cpu().numpy(), OC.decode(sd, codes[i], cfg).numpy(), rtol=0, atol=2e-4)
    # a dense zero-padded batch is NOT the same thing: the test would be vacuous if it were
    dense = torch.zeros(len(LENS), cfg.n_q, max(LENS), dtype=torch.long)
    for i, c in enumerate(codes):
        dense[i, :, : LENS[i]] = c[0]
    naive = m.decode(dense.cuda())
    assert (naive[1, :, : LENS[1] * cfg.hop] - got[1][0]).abs().max() > 1e-3


def test_decode_ragged_buckets_and_order():
    """Default cost model: several buckets; results come back in the caller's order whatever the bucketing."""
    cfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    sd = W.codec_state_dict(cfg, seed=22)
    m = WMEncodecModel(cfg, sd, "cuda")
    lens = [300, 20, 280, 25, 22, 310, 18]
    m.RAGGED_STEP_US, m.RAGGED_ITEM_US = 1.0, 50.0          # padding is expensive here: short and long items must not share a pass
    b = m._ragged_buckets(lens)
    assert sorted(sum(b, [])) == list(range(len(lens))) and len(b) >= 2
    assert not any(set(x) & {0, 2, 5} and set(x) & {1, 3, 4, 6} for x in b)
    codes = _codes(cfg, lens, 4)
    got = m.decode_ragged([c.cuda() for c in codes])
    for i in (0, 1, 4, 5):
        torch.testing.assert_close(got[i], m.decode(codes[i].cuda()), rtol=0, atol=2e-5)
    with pytest.raises(IndexError):
        bad = [c.clone() for c in codes]
        bad[3][0, 1, 2] = cfg.bins
        m.decode_ragged([c.cuda() for c in bad])


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_wmdecode_ragged_equals_batch1_wmdecode(pad_mode):
    cfg = dataclasses.replace(W.codec_config_full(), pad_mode=pad_mode)
    sd = W.codec_state_dict(cfg, seed=23)
    m = WMEncodecModel(cfg, sd, "cuda")
    lens = [11, 3, 17, 8, 17, 1]
    codes = _codes(cfg, lens, 5)
    g = torch.Generator().manual_seed(6)
    labels = [torch.randint(0, 2, (1, n), generator=g) for n in lens]
    wavs = [torch.randn(1, 1, n * cfg.hop, generator=g) * 0.2 for n in lens]
    m.RAGGED_STEP_US = 1e9
    got_w, got_m = m.wmdecode_ragged([c.cuda() for c in codes], [l.cuda() for l in labels], [w.cuda() for w in wavs], with_mark=True)
    for i in range(len(lens)):
        w1, m1 = m.wmdecode(codes[i].cuda(), labels[i].cuda(), wavs[i].cuda())
        assert got_w[i].shape == w1.shape and got_m[i].shape == m1.shape
        torch.testing.assert_close(got_w[i], w1, rtol=0, atol=2e-5)
        torch.testing.assert_close(got_m[i], m1, rtol=0, atol=2e-5)
    o_w, o_m = OC.wmdecode(sd, codes[1], labels[1], wavs[1], cfg)
    np.testing.assert_allclose(got_w[1].cpu().numpy(), o_w.numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(got_m[1].cpu().numpy(), o_m.numpy(), rtol=0, atol=2e-4)
    no_mark, none = m.wmdecode_ragged([c.cuda() for c in codes], [l.cuda() for l in labels], [w.cuda() for w in wavs], with_mark=False)
    assert none is None and all(torch.equal(a, b) for a, b in zip(no_mark, got_w))
    with pytest.raises(IndexError):                                      # a label outside the embedding table (seanet.py:562 F.embedding)
        m.wmdecode(codes[0].cuda(), torch.full_like(labels[0], 2).cuda(), wavs[0].cuda())


class FakePhonemizer:
    def __call__(self, texts):
        return [[c for c in t if c != " "] for t in texts]


def _tiny_stack():
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=7)
    args = W.lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64)
    lsd = W.lm_state_dict(args, seed=8)
    for k in range(4):   # keep a random-weight LM away from the special ids RVQ decode rejects (as the reference's F.embedding would)
        lsd[f"predict_layer.{k}.2.bias"][64:] = -30.0
    m = SSR_Speech(args)
    m.load_state_dict(lsd)
    return m.to("cuda").eval(), AudioTokenizer(device="cuda", config=ccfg, state_dict=csd), args


@pytest.mark.parametrize("use_watermark", [False, True])
def test_dp_synthesize_equals_one_inference_one_sample_per_utterance(tmp_path, use_watermark):
    """8 utterances of different text / prompt lengths (=> different numbers of generated frames) through `dp.synthesize`
    == 8 `inference_one_sample` calls seeded seed + i (sampling, CFG), to 2e-4 on the waveforms; wav files are written."""
    m, tok, args = _tiny_stack()
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    texts = ["hello world again", "abc", "the quick brown fox jumps", "zzz yyy", "lorem ipsum dolor sit amet consectetur", "go", "speech editing", "a b c d e f g"]
    g = torch.Generator().manual_seed(11)
    decode_config = {"top_k": 20, "top_p": 0.9, "temperature": 1, "stop_repetition": 2, "kvcache": 1, "codec_audio_sr": 16000, "codec_sr": 50}
    utts, refs = [], []
    for i, t in enumerate(texts):
        n_frames = 8 + 3 * i
        wav = torch.randn(1, n_frames * 320, generator=g) * 0.2
        fn = str(tmp_path / f"p{i}.wav")
        write_wav(fn, wav, 16000)
        mi = torch.LongTensor([[n_frames, n_frames]])
        torch.manual_seed(100 + i)
        refs.append(inference_one_sample(m, argparse.Namespace(**vars(args)), phn2num, FakePhonemizer(), tok, fn, "", t, mi,
                                         1.5, 2, True, False, use_watermark, True, "cuda", decode_config))
        from ssr_speech_amd.data.tokenizer import tokenize_audio
        codes, _, _ = tokenize_audio(tok, fn)
        utts.append(dict(x=torch.LongTensor([[phn2num[c] for c in t if c != " "]]), y=codes.transpose(2, 1).cpu(), mask_interval=mi.unsqueeze(0), wav=fn))
    stats = {}
    waves, mine, tokens = dp.synthesize(m, tok, utts, seed=100, use_watermark=use_watermark, tts=True, output_dir=str(tmp_path / "out"), stats=stats,
                                            top_k=20, top_p=0.9, temperature=1, stop_repetition=2, cfg_coef=1.5, cfg_stride=2, aug_text=True)
    assert mine == list(range(8)) and len(waves) == 8 and len(tokens) == 8 and "codec_s" in stats
    assert len({w.shape[-1] for w in waves}) > 2                        # really ragged
    for i in range(8):
        assert waves[i].shape == refs[i].shape, (i, waves[i].shape, refs[i].shape)
        torch.testing.assert_close(waves[i], refs[i], rtol=0, atol=2e-4)
        assert (tmp_path / "out" / f"utt{i:05d}.wav").exists()


def test_rccl_one_rank_group_runs_the_real_collectives():
    """a 1-rank `nccl` (= RCCL) process group on cuda:0 takes `dp.gather_tokens` and `dp.generate` through the
    real `all_gather_into_tensor` / `all_reduce` on device tensors (the world-1 short-circuit is off under `force_collective`),
    so the first multi-GPU run is not also the first RCCL run."""
    import os
    import socket
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        local = [torch.randint(0, 2048, (4, 5 + 3 * i)) for i in range(5)]
        out = dp.gather_tokens(local, 5, 4, pad_token=2048, force_collective=True)
        assert dp._collective_device().type == "cuda"
        assert len(out) == 5 and all(o.is_cuda and torch.equal(o.cpu(), t) for o, t in zip(out, local))
        m, tok, args = _tiny_stack()
        g = torch.Generator().manual_seed(3)
        utts = [dict(x=torch.randint(0, 26, (1, 5 + i), generator=g), y=torch.randint(0, 64, (1, 10 + i, 4), generator=g),
                     mask_interval=torch.LongTensor([[[10 + i, 10 + i]]])) for i in range(3)]
        kw = dict(top_k=1, top_p=1.0, temperature=1, stop_repetition=2, cfg_coef=1.5, cfg_stride=2, aug_text=True)
        toks, (mine, outs) = dp.generate(m, utts, seed=9, force_collective=True, **kw)
        plain, _ = dp.generate(m, utts, seed=9, **kw)
        assert mine == [0, 1, 2] and all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(toks, plain))
        waves, _, _ = dp.synthesize(m, tok, utts, seed=9, force_collective=True, **kw)
        assert len(waves) == 3 and all(torch.isfinite(w).all() for w in waves)
    finally:
        dist.destroy_process_group()
