"""`WMEncodecModel` — drop-in for the reference's watermarked Encodec on the inference path
(`audiocraft/audiocraft/models/wmencodec.py:324-386`: `encode`, `decode`, `wmdecode`, `detect_watermark`,
`decode_latent`), built from a state-dict with the reference's key names. All arithmetic runs in libssrhip.so:
the SEANet convolutions / transposed convolutions / residual blocks as fp32 MFMA GEMMs over time-major
strided views (weight-norm folded and weights repacked once at load), the LSTM recurrence, RVQ search and
dequantisation, the watermark-label conditioning as dedicated HIP kernels. PyTorch only allocates buffers.

Layout: an activation is `[B][padL + T + padR][C]` fp32 (time-major, channels contiguous, halo rows zero or
reflected) so that the im2col matrix of a Conv1d is a *view* (row t = k consecutive time rows).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib
from ..weights import CodecConfig


def _extra_padding(length: int, k: int, s: int, pt: int) -> int:
    """audiocraft/modules/conv.py:47-53."""
    n_frames = (length - k + pt) / s + 1
    return (math.ceil(n_frames) - 1) * s + (k - pt) - length


def _empty(*shape, dtype, device):
    """torch.empty for the activation / state workspaces. SSRHIP_POISON_ALLOC=1 (debug; tools/poison_check.py, read at every call) fills them
    with NaN (float) / 0x7FC0 = bf16 NaN (int16) / a huge index (int32) instead: a kernel that READS a location no producer wrote turns its
    outputs NaN, where with plain torch.empty the result silently depends on what the allocator's block held before."""
    t = torch.empty(*shape, dtype=dtype, device=device)
    if os.environ.get("SSRHIP_POISON_ALLOC", "0") not in ("", "0"):
        if dtype == torch.float32:
            t.fill_(float("nan"))
        elif dtype == torch.int16:
            t.fill_(0x7FC0)
        else:
            t.fill_(0x3FFFFFFF)
    return t


class _NoLaunch:
    """Stands in for the library during a sizing pass (`WMEncodecModel._sized`): every entry point "succeeds" without launching anything."""

    def __getattr__(self, name):
        if not name.startswith("ssrhip_"):
            raise AttributeError(name)
        return lambda *args: 0


def _tensors_of(obj):
    """the tensors inside a (nested) tuple / list result"""
    if isinstance(obj, torch.Tensor):
        return [obj]
    if isinstance(obj, (tuple, list)):
        return [t for o in obj for t in _tensors_of(o)]
    return []


def _device_mallocs(device) -> int:
    """How many times the caching allocator has gone to the driver for memory on this device (hipMalloc calls)."""
    return int(torch.cuda.memory_stats(device).get("num_device_alloc", 0))


class TM:
    """Time-major activation buffer with halo rows."""

    def __init__(self, B: int, T: int, Cc: int, padL: int, padR: int, device, lens: "Optional[Ragged]" = None):
        self.B, self.T, self.C, self.padL, self.padR = B, T, Cc, padL, padR
        self.lens = lens                   # ragged batch: per-item valid rows (None = every item has T rows)
        self.elu = False                   # True: the producer stored ELU(y) (its only consumers read it through ELU: they skip theirs)
        self.rows = padL + T + padR
        # every producer writes the whole interior; only the halo rows need a defined value before the consumer reads them
        # (zero for constant / structural padding; reflect padding overwrites them). A torch.zeros of the whole buffer was 24
        # full-size memsets per encode+decode (6.7 ms at 32 clips x 30 s).
        self.data = _empty(B, self.rows, Cc, dtype=torch.float32, device=device)
        if padL:
            self.data[:, :padL].zero_()
        if padR:
            self.data[:, padL + T:].zero_()

    @property
    def base(self) -> int:
        return self.data.data_ptr()

    @property
    def interior(self) -> int:
        return self.data.data_ptr() + 4 * self.padL * self.C

    @property
    def bstride(self) -> int:
        return self.rows * self.C

    def interior_view(self) -> torch.Tensor:
        return self.data[:, self.padL: self.padL + self.T]


class Ragged:
    """Per-item lengths of a ragged batch at one time resolution: host list + the same numbers as a device int32 array (what
    `ssrhip_pad_ragged` reads). `scaled` derives the lengths after an up-/down-sampling layer; the device arrays are cached
    per resolution so that a decode creates four small tensors, not one per layer."""

    def __init__(self, lens: Sequence[int], device, _cache=None):
        self.host = [int(v) for v in lens]
        self.dev = torch.tensor(self.host, dtype=torch.int32, device=device)
        self._cache = _cache if _cache is not None else {}
        self._cache[tuple(self.host)] = self

    def scaled(self, num: int, den: int = 1) -> "Ragged":
        if any((v * num) % den for v in self.host):
            raise ValueError(f"ragged lengths {self.host} are not whole frames at a stride of {den}")
        new = tuple(v * num // den for v in self.host)
        hit = self._cache.get(new)
        return hit if hit is not None else Ragged(new, self.dev.device, self._cache)


def _fold_wn(sd, pfx: str, which: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """weight = g * v / ||v|| (old-style weight_norm, dim 0; conv.py:21-30) or a plain weight."""
    k = f"{pfx}{which}.{which}."
    if k + "weight" in sd:
        return sd[k + "weight"], sd[k + "bias"]
    return torch._weight_norm(sd[k + "weight_v"], sd[k + "weight_g"], 0), sd[k + "bias"]


class _Conv:
    def __init__(self, sd, pfx, stride, act_in, dev):
        w, b = _fold_wn(sd, pfx, "conv")
        self.Cout, self.Cin, self.k = w.shape
        self.s = stride
        self.act_in = act_in
        self.W = w.permute(0, 2, 1).contiguous().view(self.Cout, self.k * self.Cin).to(dev)     # [Cout][k][Cin]
        self.Wraw = w.contiguous().view(self.Cout, self.Cin * self.k).to(dev)                   # Cin == 1 path
        self.b = b.contiguous().to(dev)
        pt = self.k - self.s
        self.pr = pt // 2
        self.pl = pt - self.pr

    def pads(self, T):
        return self.pl, self.pr + _extra_padding(T, self.k, self.s, self.k - self.s)

    def out_len(self, T):
        pl, pr = self.pads(T)
        return (T + pl + pr - self.k) // self.s + 1


class _ConvTr:
    def __init__(self, sd, pfx, stride, dev):
        w, b = _fold_wn(sd, pfx, "convtr")          # [Cin][Cout][k]
        self.Cin, self.Cout, self.k = w.shape
        self.s = stride
        assert self.k == 2 * stride, "SEANet uses kernel = 2*stride for its transposed convolutions"
        s = stride
        # W_all[(p, co)][0:Cin] multiplies in[q-1] (tap p+s), [Cin:2Cin] multiplies in[q] (tap p)
        wa = torch.empty(s, self.Cout, 2 * self.Cin, dtype=torch.float32)
        for p in range(s):
            wa[p, :, : self.Cin] = w[:, :, p + s].t()
            wa[p, :, self.Cin:] = w[:, :, p].t()
        self.W = wa.view(s * self.Cout, 2 * self.Cin).contiguous().to(dev)
        self.b = b.repeat(s).contiguous().to(dev)
        pt = self.k - self.s
        self.trim_r = pt // 2
        self.trim_l = pt - self.trim_r


def pack_lstm_whh_planes(planes: torch.Tensor) -> torch.Tensor:
    """bf16 planes [3][4C][C] of a recurrent matrix (`ssrhip_split_weights` of `weight_hh`, torch's gate order i f g o) -> the fragment
    order `csrc/lstm_split.hip` streams them in (include/ssrhip.h ssrhip_lstm_args.w_split):
        out[ub][w][s][mb][q][lh][li][e] = planes[q][g C + 16 ub + u][w C/4 + 16 s + 8 lh + e]      with 32 mb + li = 16 g + u,
    ub = unit block (C/16), w = wave = K quarter, s = k-step of 16 inside the quarter (C/64 of them), lane = 32 lh + li. Needs C % 64 == 0."""
    q3, rows, Cc = planes.shape
    assert q3 == 3 and rows == 4 * Cc and Cc % 64 == 0, planes.shape
    ks = Cc // 64
    x = planes.reshape(3, 4, Cc // 16, 16, 4, ks, 2, 8)           # q, g, ub, u, w, s, lh, e
    x = x.permute(2, 4, 5, 1, 3, 0, 6, 7)                         # ub, w, s, g, u, q, lh, e
    x = x.reshape(Cc // 16, 4, ks, 2, 32, 3, 2, 8)                # (g, u) -> m = 16 g + u = 32 mb + li
    return x.permute(0, 1, 2, 3, 5, 6, 4, 7).contiguous()         # ub, w, s, mb, q, lh, li, e


class _Lstm:
    def __init__(self, sd, pfx, layers, dev):
        self.layers = []
        for l in range(layers):
            wih, whh = sd[pfx + f"lstm.weight_ih_l{l}"], sd[pfx + f"lstm.weight_hh_l{l}"]
            bias = sd[pfx + f"lstm.bias_ih_l{l}"] + sd[pfx + f"lstm.bias_hh_l{l}"]
            self.layers.append((wih.contiguous().to(dev), whh.contiguous().to(dev), bias.contiguous().to(dev)))
        self.C = self.layers[0][1].shape[1]
        # the recurrent matrix once more, in the order the matrix-core step kernel's lanes read it (include/ssrhip.h w_packed)
        Cc = self.C
        self.packed = []
        for _, whh, _ in self.layers:
            if Cc % 16 == 0:
                self.packed.append(whh.view(4, Cc // 4, 4, Cc // 16, 4, 4).permute(1, 3, 4, 2, 0, 5).contiguous())
            else:
                self.packed.append(None)
        self.split = [None] * layers     # W_hh planes in MFMA fragment order (csrc/lstm_split.hip), made by CodecModel._prepare_planes when enabled


class _SeaNet:
    """One nn.Sequential of the reference (SEANetEncoder.model / SEANetDecoder.model), as a list of fused nodes."""

    def __init__(self, sd, pfx: str, cfg: CodecConfig, decoder: bool, dev):
        self.cfg, self.dev = cfg, dev
        # (model index of the node's first module, kind, obj, model index of the module whose OUTPUT the node produces)
        self.nodes: List[tuple] = []
        i = 0
        if not decoder:                  # seanet.py:113-150
            self.nodes.append((i, "conv", _Conv(sd, f"{pfx}model.{i}.", 1, 0, dev), i))
            i += 1
            for r in reversed(cfg.ratios):
                self.nodes.append((i, "res", (_Conv(sd, f"{pfx}model.{i}.block.1.", 1, 1, dev), _Conv(sd, f"{pfx}model.{i}.block.3.", 1, 1, dev)), i))
                self.nodes.append((i + 1, "conv", _Conv(sd, f"{pfx}model.{i + 2}.", r, 1, dev), i + 2))      # ELU (i+1) folded in
                i += 3
            if cfg.lstm:
                self.nodes.append((i, "lstm", _Lstm(sd, f"{pfx}model.{i}.", cfg.lstm, dev), i))
                i += 1
            self.nodes.append((i, "conv", _Conv(sd, f"{pfx}model.{i + 1}.", 1, 1, dev), i + 1))
        else:                            # seanet.py:209-254
            self.nodes.append((i, "conv", _Conv(sd, f"{pfx}model.{i}.", 1, 0, dev), i))
            i += 1
            if cfg.lstm:
                self.nodes.append((i, "lstm", _Lstm(sd, f"{pfx}model.{i}.", cfg.lstm, dev), i))
                i += 1
            for r in cfg.ratios:
                self.nodes.append((i, "convtr", _ConvTr(sd, f"{pfx}model.{i + 1}.", r, dev), i + 1))      # ELU (i) folded in
                self.nodes.append((i + 2, "res", (_Conv(sd, f"{pfx}model.{i + 2}.block.1.", 1, 1, dev), _Conv(sd, f"{pfx}model.{i + 2}.block.3.", 1, 1, dev)), i + 2))
                i += 3
            self.nodes.append((i, "conv", _Conv(sd, f"{pfx}model.{i + 1}.", 1, 1, dev), i + 1))

    def slice_nodes(self, lo: int, hi: Optional[int]):
        return [n for n in self.nodes if n[0] >= lo and (hi is None or n[0] < hi)]


class WMEncodecModel:
    def __init__(self, cfg: CodecConfig, state_dict: dict, device):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ssr_speech_amd codec needs a ROCm GPU device; there is no CPU path in this package")
        self.lib = _lib.lib()
        self.fuse_resblock = True            # tests switch it off to compare with the two-GEMM path
        self.force_few_out = False           # tests: take the few-output-channel kernel also for short inputs
        self.lstm_packed = os.environ.get("SSRHIP_LSTM_PACKED", "1") != "0"      # A/B knob for the packed recurrent matrix
        # the recurrence on the bf16 matrix cores with split operands (csrc/lstm_split.hip). Written blind at the end of round 4; round 5's
        # first GPU call ran it: kernel test vs fp64 green, all codec fixtures green with it, 21.0 us per step alone / 35.6 with both layers'
        # chains sharing the GPU against 31.0 / 56.8 for the fp32-pipe kernel at 256 items, config 5 373.9 / 383.6 -> 335.7 / 344.3 ms
        # (profiles/r05_microbench/codec256_ab.log). Default for batches of `lstm_split_min_b` items and more; SSRHIP_LSTM_SPLIT=0 = fp32 pipe.
        self.lstm_split = os.environ.get("SSRHIP_LSTM_SPLIT", "1") not in ("", "0")
        self.lstm_split_min_b = int(os.environ.get("SSRHIP_LSTM_SPLIT_MIN_B", "128"))
        # channel counts whose residual block runs as one kernel (env knob for A/B runs: e.g. SSRHIP_RESBLOCK_FUSE=64,128,256,512).
        # Measured at 32 clips x 30 s (encode / decode ms): {64}: 86.9 / 88.8; {64,128}: 86.4 / 87.7 and 1.9 GB less memory;
        # adding 256 or 512 (short time axes, wide weights): 88.3 / 89.1-90.5 — the chained kernel's LDS footprint leaves one
        # workgroup per CU there and loses to two ordinary GEMM launches. So 64 and 128 are fused, 256 / 512 stay two GEMMs.
        # Batch lanes: the items of a batch are independent, so a batch can be cut into `lanes` groups that run one after the other
        # in host order on their own HIP streams: peak memory falls with the group size (256 clips x 30 s: 81 GB in one lane, 50 GB in
        # two, 34 GB in four) at 2-4 % of the throughput. It was built hoping that one group's LSTM (a latency-bound launch per time
        # step) would hide under the other group's convolutions; measured, the step launches queue behind the convolutions'
        # workgroups instead (32 clips: 85 -> 94 ms with two lanes, 122 ms with high-priority LSTM streams), so the default is ONE
        # lane and this is a memory knob (SSRHIP_CODEC_LANES, at least `lane_min_items` items per lane).
        # fp32 GEMMs on the bf16 matrix cores with exactly split operands (csrc/gemm_split.hip; SSRHIP_GEMM_SPLIT=0: the fp32 FMA chain)
        self.split_gemm = os.environ.get("SSRHIP_GEMM_SPLIT", "1") != "0"
        self._plane_cache = {}
        # ELU on store instead of ELU on load wherever a tensor is only read through ELU (SSRHIP_ELU_ON_STORE=0: every consumer applies it)
        self.elu_on_store = os.environ.get("SSRHIP_ELU_ON_STORE", "1") != "0"
        self.lanes = int(os.environ.get("SSRHIP_CODEC_LANES", "1"))
        self.lane_min_items = int(os.environ.get("SSRHIP_CODEC_LANE_MIN", "8"))
        self._side_streams, self._keep = {}, {}
        # the two-stream LSTM pipeline (`_lstm`) only where it pays: from this many items on (SSRHIP_LSTM_PIPE_MIN_B). Below it the two layers
        # run one after the other on the calling stream — at batch 1 the pipeline saves ~3 ms of a 10 s utterance's codec time, and a second
        # hardware queue in flight is one half of the trigger of DESIGN.md Part I.4: a single-utterance call keeps to ONE queue.
        self.lstm_pipe_min_b = int(os.environ.get("SSRHIP_LSTM_PIPE_MIN_B", "8"))
        # Round 6 — NO DEVICE MEMORY IS MAPPED WHILE CODEC KERNELS ARE IN FLIGHT (`_sized`; DESIGN.md "the multi-stream failure"):
        # SSRHIP_CODEC_PRESIZE=0 switches the sizing passes off (the A/B arm of tools/race_trials.py).
        self.presize = os.environ.get("SSRHIP_CODEC_PRESIZE", "1") not in ("", "0")
        self._envelopes = {}                 # (entry point, stream) -> [(items, samples-or-frames)] already sized
        self._small_reserved = set()
        self.passes_repeated = 0             # passes run again because the driver was asked for memory while they were in flight
        self.mallocs_in_flight = 0           # hipMallocs that happened during a call although it had been sized (tests assert 0)
        self.sizing_passes = 0
        env = os.environ.get("SSRHIP_RESBLOCK_FUSE")
        self.fuse_channels = tuple(int(v) for v in env.split(",") if v) if env is not None else (64, 128)
        sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items()}
        dev = self.device
        self.encoder = _SeaNet(sd, "encoder.", cfg, False, dev)
        self.decoder = _SeaNet(sd, "decoder.", cfg, True, dev)
        self.has_wm = "wmdecoder.wm_embed.weight" in sd
        if self.has_wm:
            self.wmdecoder = _SeaNet(sd, "wmdecoder.", cfg, True, dev)
            self.skip_encoder = _SeaNet(sd, "wmdecoder.skip_encoder.", cfg, False, dev)
            self.wm_encoder = _SeaNet(sd, "wmdecoder.wm_encoder.", cfg, False, dev)
            self.wm_proj = [_Conv(sd, f"wmdecoder.wm_proj{j}.1.", 1, 1, dev) for j in range(4)]
            self.wm_predictor = _Conv(sd, "wmdecoder.wm_predictor.1.", 1, 1, dev)
            w = sd["wmdecoder.wm_embed.weight"]                       # nn.Embedding(2, D/16, max_norm=True) (seanet.py:503)
            norm = w.norm(p=2, dim=1, keepdim=True)
            self.wm_table = torch.where(norm > 1.0, w * (1.0 / (norm + 1e-7)), w).contiguous().to(dev)
            # label conditioning folded: per projection (W_a [Cout][C_skip], class bias [n_labels][Cout] = W_b . ELU(embed(label)))
            self.wm_cls = []
            E = self.wm_table.shape[1]
            emb_act = torch.nn.functional.elu(self.wm_table.cpu().double())
            for c in self.wm_proj:
                assert c.k == 1, "wm_proj is a 1x1 convolution (seanet.py:513-539)"
                Wfull = c.W.cpu()                                                  # [Cout][C_skip + E]
                Cs = Wfull.shape[1] - E
                self.wm_cls.append((Wfull[:, :Cs].contiguous().to(dev), (emb_act @ Wfull[:, Cs:].double().t()).float().contiguous().to(dev)))
        self.codebooks = torch.stack([sd[f"quantizer.vq.layers.{q}._codebook.embed"] for q in range(cfg.n_q)]).contiguous().to(dev)
        self.e2 = self.codebooks.pow(2).sum(-1).contiguous()          # |e|^2 (core_vq.py:169)
        self.sample_rate, self.channels, self.frame_rate = cfg.sample_rate, cfg.channels, cfg.frame_rate
        self._prepare_planes()

    def _prepare_planes(self):
        """Split every GEMM weight once, now, on the current stream (the batch lanes run on their own streams later: nothing may be
        created lazily there), and wait for it."""
        nets = [self.encoder, self.decoder] + ([self.wmdecoder, self.skip_encoder, self.wm_encoder] if self.has_wm else [])
        for net in nets:
            for _, kind, obj, _ in net.nodes:
                if kind == "conv":
                    self._planes(obj.W)
                elif kind == "convtr":
                    self._planes(obj.W)
                elif kind == "res":
                    self._planes(obj[0].W)
                    self._planes(obj[1].W)
                    if obj[0].Cin in self.fuse_channels:
                        self._resblock_planes(obj[0], obj[1])
                elif kind == "lstm":
                    for l, (wih, whh, _) in enumerate(obj.layers):
                        self._planes(wih)
                        if self.lstm_split and obj.C % 128 == 0 and obj.split[l] is None:
                            obj.split[l] = pack_lstm_whh_planes(self._split_planes(whh))
        if self.has_wm:
            for Wa, _ in self.wm_cls:
                self._planes(Wa)
            self._planes(self.wm_predictor.W)
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ sizing passes
    def _sized(self, kind: str, B: int, T: int, run):
        """Run one dense codec pass `run()` so that the caching allocator never has to map device memory (hipMalloc) while the pass's
        kernels are in flight.

        Why (round 6, DESIGN.md "the multi-stream failure"): on this platform a kernel that runs while ANOTHER hardware queue of the
        process is busy and the host maps fresh device memory can come back with wrong values — one register of one quarter-wave, in
        the codec's plainest kernel; 36 of 206 fresh processes on round 5's code path against 0 of 260 with these sizing passes
        (and 0 of 147 with a warm allocator or a single hardware queue; tools/race_trials.py, profiles/r06_microbench/). A codec pass IS several queues — the two LSTM layers run on two streams —
        and a cold process maps memory for every layer, so the first call of every process was exposed, single-stream callers
        included. What the pass will allocate is a deterministic function of (entry point, items, length): the first time a shape is
        not covered by one already seen on this stream, the pass runs once DRY — the device idle (synchronised), every library launch
        a no-op, the same tensors allocated and freed in the same order — which leaves the allocator holding exactly the blocks the
        real pass then reuses. Later calls of that size or smaller find them there. `mallocs_in_flight` counts the driver allocations
        that still happened during real passes (a smaller shape can fall into another size class, the caller may hold more live tensors
        than the sizing pass saw): such a pass is REPEATED — its results are dropped, the device synchronised, and it runs again from the
        now sufficient pool (`passes_repeated`)."""
        if not self.presize:
            return run()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._keep.pop(stream, None)           # the previous call's LSTM state: its kernels are behind us on this stream
        env = self._envelopes.setdefault((kind, stream), [])
        if not any(b >= B and t >= T for b, t in env):
            torch.cuda.synchronize(self.device)                  # nothing in flight, on any stream, while memory is mapped
            real, self.lib = self.lib, _NoLaunch()
            try:
                dry_out = run()
                # room for ONE more generation of the results: a caller that still holds the previous call's outputs while this one runs
                # (`out = codec.decode(...)` in a loop) must not push the real pass to the driver either
                spare = [torch.empty_like(t) for t in _tensors_of(dry_out)]
                # ... and for a SMALLER shape later on: every large block can be split for a smaller request, but tensors that shrink below
                # the allocator's 1 MB class boundary move to its small pool (2 MB segments) — 32 MB of those per stream, once
                if stream not in self._small_reserved:
                    spare += [torch.empty((1 << 20) - 4096, dtype=torch.uint8, device=self.device) for _ in range(32)]
                    self._small_reserved.add(stream)
                del spare, dry_out
            finally:
                self.lib = real
                self._keep.pop(stream, None)
            torch.cuda.synchronize(self.device)                  # the dry pass's own fills and copies (torch kernels on garbage)
            self.sizing_passes += 1
            env[:] = [(b, t) for b, t in env if not (b <= B and t <= T)] + [(B, T)]
        for attempt in range(3):
            n0 = _device_mallocs(self.device)
            out = run()
            grown = _device_mallocs(self.device) - n0
            if not grown:
                return out
            # The driver was asked for memory while this pass was in flight after all (a live set the sizing pass did not see, a size-class
            # crossing, another thread's allocation): its results are SUSPECT on this platform — they are dropped and the pass runs again,
            # now from a pool that holds what it needs (the kernels are deterministic; inputs are untouched). Counted, so tests can assert 0.
            self.mallocs_in_flight += grown
            self.passes_repeated += 1
            del out
            torch.cuda.synchronize(self.device)
        return run()

    # ------------------------------------------------------------------ low-level launches
    def _s(self):
        return _lib.stream_ptr()

    def _planes(self, W: torch.Tensor) -> Optional[torch.Tensor]:
        """bf16 planes [3][N][K] of a weight matrix for the split GEMM (csrc/gemm_split.hip: the fp32 operand as the exact sum of three
        bf16 pieces), made once per matrix on the device; None when the split path is switched off or the shape can never take it."""
        if not self.split_gemm or W.dim() != 2 or W.shape[0] <= 64 or W.shape[1] % 8 != 0:
            return None
        key = W.data_ptr()
        hit = self._plane_cache.get(key)
        if hit is None:
            hit = torch.empty(3, W.shape[0], W.shape[1], dtype=torch.int16, device=W.device)
            _lib.check(self.lib.ssrhip_split_weights(W.data_ptr(), hit.data_ptr(), W.numel(), self._s()), "ssrhip_split_weights")
            self._plane_cache[key] = hit
        return hit

    def _split_planes(self, W: torch.Tensor) -> torch.Tensor:
        """bf16 planes [3][N][K] of any fp32 matrix (uncached: for callers that repack them, e.g. pack_lstm_whh_planes)"""
        out = torch.empty(3, W.shape[0], W.shape[1], dtype=torch.int16, device=W.device)
        _lib.check(self.lib.ssrhip_split_weights(W.data_ptr(), out.data_ptr(), W.numel(), self._s()), "ssrhip_split_weights")
        return out

    def _resblock_planes(self, c3, c1):
        """bf16 planes of a residual block's two matrices for csrc/resblock_split.hip (C in {64, 128}): W3 [C/2][3C] as it is, W1 [C][C/2]
        with its columns in the order the kernel's stage-1 accumulator hands the hidden channels on (include/ssrhip.h w1_split).
        Made once per block on the device; None when the split path is off."""
        if not self.split_gemm or c3.Cin not in (64, 128):
            return None
        key = ("res", c3.W.data_ptr(), c1.W.data_ptr())
        hit = self._plane_cache.get(key)
        if hit is None:
            Hh = c3.Cout
            kp = torch.arange(Hh, device=c1.W.device)
            j, g, i = kp // 16, (kp // 8) % 2, kp % 8
            perm = 16 * j + (i % 4) + 8 * (i // 4) + 4 * g
            w1p = c1.W.reshape(c1.Cout, Hh)[:, perm].contiguous()
            w3 = c3.W.reshape(Hh, -1).contiguous()
            p3 = torch.empty(3, w3.shape[0], w3.shape[1], dtype=torch.int16, device=w3.device)
            p1 = torch.empty(3, w1p.shape[0], w1p.shape[1], dtype=torch.int16, device=w3.device)
            _lib.check(self.lib.ssrhip_split_weights(w3.data_ptr(), p3.data_ptr(), w3.numel(), self._s()), "ssrhip_split_weights")
            _lib.check(self.lib.ssrhip_split_weights(w1p.data_ptr(), p1.data_ptr(), w1p.numel(), self._s()), "ssrhip_split_weights")
            torch.cuda.current_stream(w3.device).synchronize()       # w1p / w3 temporaries may be freed after this call
            hit = (p3, p1)
            self._plane_cache[key] = hit
        return hit

    def _gemm(self, A, W, bias, Cp, M, N, K, lda, ldc, act_in=0, R=0, ldr=0, batch=1, sA=0, sC=0, sR=0, tm=(0, 0, 0), rowcls=None, act_out=0):
        a = _lib.GemmArgs()
        a.act_out = act_out
        a.A, a.W, a.bias, a.C = A, W.data_ptr(), (bias.data_ptr() if bias is not None else 0), Cp
        planes = self._planes(W)
        a.W_split = planes.data_ptr() if planes is not None else 0
        a.M, a.N, a.K, a.lda, a.ldc = M, N, K, lda, ldc
        a.act_in, a.R, a.ldr, a.batch = act_in, R, ldr, batch
        a.strideA, a.strideC, a.strideR = sA, sC, sR
        a.tm_c, a.tm_lo, a.tm_hi = tm
        if rowcls is not None:           # (class bias [n_class][N], class ids int32 [batch][n], rows per id)
            a.rbias, a.rclass, a.rrep, a.rclass_stride = rowcls[0].data_ptr(), rowcls[1].data_ptr(), int(rowcls[2]), int(rowcls[1].shape[1])
        _lib.check(self.lib.ssrhip_gemm(C.byref(a), self._s()), "ssrhip_gemm")

    def _fill_pads(self, buf: TM, structural_zero: bool = False):
        if self.cfg.pad_mode not in ("constant", "reflect"):
            raise ValueError(self.cfg.pad_mode)
        if buf.lens is not None:
            # ragged batch: every item gets ITS OWN halo right behind its last valid row (and in front, for reflect)
            if buf.padL + buf.padR > 0:
                refl = int(self.cfg.pad_mode == "reflect" and not structural_zero)
                _lib.check(self.lib.ssrhip_pad_ragged(buf.base, buf.lens.dev.data_ptr(), buf.B, buf.T, buf.padL, buf.padR, buf.C, buf.bstride,
                                                      refl, self._s()), "ssrhip_pad_ragged")
            return
        if self.cfg.pad_mode == "reflect" and not structural_zero and (buf.padL + buf.padR) > 0:
            _lib.check(self.lib.ssrhip_pad_reflect(buf.base, buf.B, buf.T, buf.padL, buf.padR, buf.C, buf.bstride, self._s()), "ssrhip_pad_reflect")
        elif self.cfg.pad_mode not in ("constant", "reflect"):
            raise ValueError(self.cfg.pad_mode)

    def _need(self, node, T) -> Tuple[int, int, bool]:
        """(padL, padR, structural) the node wants on its input of interior length T."""
        if node is None:
            return 0, 0, False
        kind, obj = node[1], node[2]
        if kind == "conv":
            pl, pr = obj.pads(T)
            return pl, pr, False
        if kind == "res":
            pl, pr = obj[0].pads(T)
            return pl, pr, False
        if kind == "convtr":
            return 1, 1, True
        return 0, 0, False

    def _alloc_for(self, B, T, Cc, nxt, lens: Optional[Ragged] = None) -> TM:
        pl, pr, _ = self._need(nxt, T)
        return TM(B, T, Cc, pl, pr, self.device, lens)

    # ------------------------------------------------------------------ nodes
    def _act_in(self, wants_elu: bool, x: TM) -> int:
        """ELU-on-load flag of a consumer: needed unless the producer already stored ELU(x) (`x.elu`). A consumer that wants the RAW
        tensor must never be handed an ELU'd one."""
        if x.elu and not wants_elu:
            raise AssertionError("a layer that reads its input without ELU was given a tensor stored through ELU")
        return _lib.ACT_ELU if (wants_elu and not x.elu) else 0

    def _conv(self, c: _Conv, x: TM, nxt, R: Optional[TM] = None, post_elu: bool = False) -> TM:
        """`post_elu`: store ELU(result) (the caller knows that every consumer reads this output through ELU; `self.elu_on_store`)."""
        B = x.B
        T_out = (x.T + x.padL + x.padR - c.k) // c.s + 1
        out = self._alloc_for(B, T_out, c.Cout, nxt, x.lens.scaled(1, c.s) if x.lens is not None else None)
        act_in = self._act_in(bool(c.act_in), x)
        if c.Cin == 1:
            assert c.act_in == 0 and not post_elu
            _lib.check(self.lib.ssrhip_conv_cin1(x.base, c.Wraw.data_ptr(), c.b.data_ptr(), out.interior, B, T_out, c.k, c.s, c.Cout,
                                                 x.bstride, out.bstride, self._s()), "ssrhip_conv_cin1")
        elif c.Cout <= 4 and c.s == 1 and R is None and c.Cin % 8 == 0 and (T_out >= 4096 or self.force_few_out) and not post_elu:
            # the 1-channel output layer at the sample rate: a read-bound dot-product kernel instead of a GEMM tile with 1 useful column
            _lib.check(self.lib.ssrhip_conv_few_out(x.base, c.W.data_ptr(), c.b.data_ptr(), out.interior, B, T_out, c.k, c.Cin, c.Cout,
                                                    act_in, x.bstride, out.bstride, self._s()), "ssrhip_conv_few_out")
        else:
            self._gemm(x.base, c.W, c.b, out.interior, T_out, c.Cout, c.k * c.Cin, c.s * c.Cin, c.Cout, act_in=act_in,
                       R=(R.interior if R is not None else 0), ldr=(R.C if R is not None else 0), batch=B, sA=x.bstride, sC=out.bstride,
                       sR=(R.bstride if R is not None else 0), act_out=(_lib.ACT_ELU if post_elu else 0))
        out.elu = post_elu
        self._fill_pads(out, structural_zero=(nxt is not None and nxt[1] == "convtr"))
        return out

    def _convtr(self, c: _ConvTr, x: TM, nxt) -> TM:
        assert x.padL == 1 and x.padR == 1
        T_out = x.T * c.s
        out = self._alloc_for(x.B, T_out, c.Cout, nxt, x.lens.scaled(c.s) if x.lens is not None else None)
        Cp = out.interior - 4 * c.trim_l * c.Cout
        self._gemm(x.base, c.W, c.b, Cp, x.T + 1, c.s * c.Cout, 2 * c.Cin, c.Cin, c.s * c.Cout, act_in=self._act_in(True, x), batch=x.B,
                   sA=x.bstride, sC=out.bstride, tm=(c.Cout, c.trim_l, c.trim_l + T_out))
        self._fill_pads(out, structural_zero=(nxt is not None and nxt[1] == "convtr"))
        return out

    def _res(self, cs, x: TM, nxt, post_elu: bool = False) -> TM:
        c3, c1 = cs
        assert not x.elu, "a residual block adds its RAW input back (seanet.py:58-60)"
        if self.fuse_resblock and c3.Cin in self.fuse_channels and c3.Cout * 2 == c3.Cin and c3.k == 3 and c3.s == 1 and c1.k == 1 and c1.s == 1 \
                and x.padL == 1 and x.padR == 1:
            # the whole block as one kernel (csrc/resblock.hip): one read + one write of the activation, the C/2 intermediate stays on chip
            out = self._alloc_for(x.B, x.T, c1.Cout, nxt, x.lens)
            a = _lib.ResblockArgs()
            a.x, a.y = x.base, out.interior
            a.w3, a.b3, a.w1, a.b1 = c3.W.data_ptr(), c3.b.data_ptr(), c1.W.data_ptr(), c1.b.data_ptr()
            a.B, a.T, a.C = x.B, x.T, c3.Cin
            a.x_bstride, a.y_bstride = x.bstride, out.bstride
            a.out_act = _lib.ACT_ELU if post_elu else 0
            planes = self._resblock_planes(c3, c1)
            if planes is not None:
                a.w3_split, a.w1_split = planes[0].data_ptr(), planes[1].data_ptr()
            _lib.check(self.lib.ssrhip_resblock(C.byref(a), self._s()), "ssrhip_resblock")
            out.elu = post_elu
            self._fill_pads(out, structural_zero=(nxt is not None and nxt[1] == "convtr"))
            return out
        h = self._conv(c3, x, None, post_elu=self.elu_on_store)        # the intermediate feeds only `ELU -> 1x1 conv`
        return self._conv(c1, h, nxt, R=x, post_elu=post_elu)

    LSTM_CHUNK = 64          # time steps per pipeline stage of the two stacked LSTM layers

    def _lstm(self, L: _Lstm, x: TM, nxt, post_elu: bool = False) -> TM:
        """2-layer LSTM + skip (lstm.py:10-25). Each layer: input GEMM over time (MFMA) + one launch per time step. With two
        layers the recurrences are software-pipelined over chunks of LSTM_CHUNK steps on two streams: layer 2 works on chunk
        i (its input GEMM for that chunk, then its steps) while layer 1 already runs chunk i+1 — the step kernels are
        latency-bound and leave most of the GPU idle, so the two chains overlap almost completely."""
        B, T, Cc = x.B, x.T, x.C
        dev = self.device
        assert x.padL == 0 and x.padR == 0 and not x.elu
        nl = len(L.layers)
        rows = (B + 15) // 16 * 16                                     # include/ssrhip.h ssrhip_lstm_args
        gins = [_empty(B, T, 4 * Cc, dtype=torch.float32, device=dev) for _ in range(nl)]
        hbufs = [_empty(2, rows, Cc, dtype=torch.float32, device=dev) for _ in range(nl)]
        cbufs = [_empty(B, Cc, dtype=torch.float32, device=dev) for _ in range(nl)]
        outs = [self._alloc_for(B, T, Cc, nxt, x.lens) if l == nl - 1 else TM(B, T, Cc, 0, 0, dev) for l in range(nl)]
        # h of the split-operand path: two buffers of bf16 planes in fragment order (zeroed by the library at t = 0)
        use_split = self.lstm_split and B >= self.lstm_split_min_b and not (B <= 4 and Cc in (256, 512, 1024, 2048))
        hsplits = [_empty(2 * ((B + 63) // 64) * 64 * Cc * 3, dtype=torch.int16, device=dev) if (use_split and L.split[l] is not None) else None
                   for l in range(nl)]

        def in_gemm(l, t0, t1):                                        # gin_l[:, t0:t1] = in_l[:, t0:t1] W_ih^T + b
            src_ptr, src_bs = (x.interior, x.bstride) if l == 0 else (outs[l - 1].interior, outs[l - 1].bstride)
            wih, _, bias = L.layers[l]
            self._gemm(src_ptr + 4 * t0 * Cc, wih, bias, gins[l].data_ptr() + 4 * t0 * 4 * Cc, t1 - t0, 4 * Cc, Cc, Cc, 4 * Cc,
                       batch=B, sA=src_bs, sC=T * 4 * Cc)

        def steps(l, t0, t1):
            a = _lib.LstmArgs()
            small_b = B <= 4 and Cc in (256, 512, 1024, 2048)            # ssrhip_lstm_layer's path choice
            use_packed = (not small_b) and L.packed[l] is not None and self.lstm_packed
            a.gin, a.w_hh, a.out = gins[l].data_ptr(), (L.packed[l] if use_packed else L.layers[l][1]).data_ptr(), outs[l].interior
            a.w_packed = int(use_packed)
            if use_split and L.split[l] is not None:
                a.w_split, a.hsplit = L.split[l].data_ptr(), hsplits[l].data_ptr()
            a.skip = x.interior if l == nl - 1 else 0                   # y = lstm(x) + x (lstm.py:21-23)
            a.hbuf, a.cbuf, a.gates = hbufs[l].data_ptr(), cbufs[l].data_ptr(), 0
            a.B, a.T, a.C = B, T, Cc
            a.gin_bstride, a.out_bstride, a.skip_bstride = T * 4 * Cc, outs[l].bstride, x.bstride
            a.t_begin, a.t_end = t0, t1
            a.out_act = _lib.ACT_ELU if (post_elu and l == nl - 1) else 0
            _lib.check(self.lib.ssrhip_lstm_layer(C.byref(a), self._s()), "ssrhip_lstm_layer")

        if nl == 2 and T > self.LSTM_CHUNK and B >= self.lstm_pipe_min_b:
            main = torch.cuda.current_stream(dev)
            side = self._side_stream()
            # Every tensor the side stream touches was allocated on `main`, and `main` joins the side stream (`fin`) before this function
            # returns — also when a launch fails half way (the `finally`). A block freed later goes back to MAIN's pool and can only be handed
            # to an allocation whose kernels run on `main`, i.e. behind `fin`: the stream order alone makes reuse safe. Rounds 4-5 ALSO called
            # `record_stream(side)` on these tensors; that defers their reuse until the GPU has passed an event recorded at release time — in a
            # real pass the host is far ahead of the GPU, in a sizing pass (`_sized`) there is nothing to wait for, so the two would allocate
            # differently and the real pass would map memory while its kernels are in flight. SSRHIP_RECORD_STREAM=1 brings the calls back.
            if os.environ.get("SSRHIP_RECORD_STREAM", "0") not in ("", "0"):
                for tns in gins + hbufs + cbufs + [o.data for o in outs] + [x.data] + [h for h in hsplits if h is not None]:
                    tns.record_stream(side)
            try:
                in_gemm(0, 0, T)
                for t0 in range(0, T, self.LSTM_CHUNK):
                    t1 = min(t0 + self.LSTM_CHUNK, T)
                    steps(0, t0, t1)
                    ev = torch.cuda.Event()
                    ev.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        in_gemm(1, t0, t1)
                        steps(1, t0, t1)
            finally:
                fin = torch.cuda.Event()
                fin.record(side)
                main.wait_event(fin)
        else:
            for l in range(nl):
                in_gemm(l, 0, T)
                steps(l, 0, T)
        out = outs[-1]
        out.elu = post_elu
        self._fill_pads(out, structural_zero=(nxt is not None and nxt[1] == "convtr"))
        self._keep[torch.cuda.current_stream(dev).cuda_stream] = (gins, hbufs, cbufs, outs, hsplits)
        return out

    def _side_stream(self):
        """The second stream of the LSTM layer pipeline: one per stream the codec is called on (every batch lane has its own)."""
        key = torch.cuda.current_stream(self.device).cuda_stream
        if key not in self._side_streams:
            self._side_streams[key] = torch.cuda.Stream(self.device)
        return self._side_streams[key]

    def _lane_cuts(self, B: int):
        n = min(self.lanes, B // max(self.lane_min_items, 1))
        if n <= 1:
            return None
        edges = [B * i // n for i in range(n + 1)]
        return list(zip(edges[:-1], edges[1:]))

    def _in_lanes(self, B: int, body):
        """Run `body(lo, hi) -> tuple of tensors` for the batch lanes [lo, hi) ONE AFTER THE OTHER on the calling stream and return
        the list of per-lane results. Lanes are a memory knob: a lane's intermediates are freed before the next lane allocates its
        own, so the peak falls with the lane size (256 clips x 30 s: 81 GB in one lane, 34 GB in four). Round 2 ran the lanes
        concurrently on their own streams, hoping to hide one lane's LSTM under another's convolutions; measured, that was SLOWER
        (32 clips: 85 -> 94 ms with two lanes: the step launches queue behind the other lane's resident workgroups), and round 3 found
        it unsafe as well (with the faster split GEMM the last lane's LSTM state was intermittently corrupted: tests/test_gpu_codec.py
        ::test_batch_lanes_equal_one_lane failed one run in two) — so the streams are gone."""
        cuts = self._lane_cuts(B)
        if cuts is None:
            return [body(0, B)]
        return [body(lo, hi) for lo, hi in cuts]

    @staticmethod
    def _join(parts, i):
        vals = [p[i] for p in parts]
        if vals[0] is None:
            return None
        return vals[0] if len(vals) == 1 else torch.cat(vals, dim=0)

    def _run(self, nodes, x: TM, after=None) -> TM:
        """Run consecutive nodes; `after` = the node that will consume the result (decides its halo and whether the result is stored
        through ELU: a residual block's or the LSTM's output goes into `ELU -> conv / convtr` and nowhere else — the watermark
        decoder's skip taps are read through ELU as well, seanet.py:577-591 — so the producer applies the ELU once, on store)."""
        for idx, node in enumerate(nodes):
            nxt = nodes[idx + 1] if idx + 1 < len(nodes) else after
            kind, obj = node[1], node[2]
            post = self.elu_on_store and kind in ("res", "lstm") and nxt is not None and \
                ((nxt[1] == "conv" and bool(nxt[2].act_in)) or nxt[1] == "convtr")
            if kind == "conv":
                x = self._conv(obj, x, nxt)
            elif kind == "convtr":
                x = self._convtr(obj, x, nxt)
            elif kind == "res":
                x = self._res(obj, x, nxt, post_elu=post)
            elif kind == "lstm":
                x = self._lstm(obj, x, nxt, post_elu=post)
        return x

    def _input_tm(self, wav: torch.Tensor, first, lens: Optional[Ragged] = None) -> TM:
        """[B,1,T] -> time-major buffer with the first node's halo."""
        assert wav.dim() == 3 and wav.shape[1] == self.channels == 1, wav.shape
        B, _, T = wav.shape
        x = self._alloc_for(B, T, 1, first, lens)
        x.data[:, x.padL: x.padL + T, 0] = wav[:, 0].to(self.device, torch.float32)
        self._fill_pads(x)
        return x

    # ------------------------------------------------------------------ public API (wmencodec.py:324-386)
    @torch.no_grad()
    def encode(self, x: torch.Tensor):
        """-> (codes int64 [B,K,T'], scale=None (renormalize False), emb f32 [B,D,T'])."""
        assert x.dim() == 3

        def body(lo, hi):
            inp = self._input_tm(x[lo:hi], self.encoder.nodes[0])
            emb = self._run(self.encoder.nodes, inp)
            B, T, D = emb.B, emb.T, emb.C
            codes = _empty(B, self.cfg.n_q, T, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.ssrhip_rvq_encode(emb.interior, self.codebooks.data_ptr(), self.e2.data_ptr(), codes.data_ptr(), B, T, D,
                                                  self.cfg.n_q, self.cfg.bins, emb.bstride, self._s()), "ssrhip_rvq_encode")
            return codes.to(torch.int64), emb.interior_view().transpose(1, 2).contiguous()

        parts = self._in_lanes(x.shape[0], lambda lo, hi: self._sized("encode", hi - lo, int(x.shape[-1]), lambda: body(lo, hi)))
        return self._join(parts, 0), None, self._join(parts, 1)

    def _codes32(self, codes: torch.Tensor) -> torch.Tensor:
        """Code ids on the device as int32, range-checked once for the whole batch (before it is cut into lanes)."""
        assert codes.dim() == 3
        c32 = codes.to(self.device, torch.int32).contiguous()
        if c32.numel() > 0:
            lo, hi = torch.aminmax(c32)                               # one kernel; the two scalars come back in one wait
            lo_hi = torch.stack([lo, hi]).tolist()
            if lo_hi[1] >= self.cfg.bins or lo_hi[0] < 0:
                raise IndexError("index out of range in self")        # what F.embedding raises in the reference (core_vq.py:175)
        return c32

    def _dequant(self, c32: torch.Tensor, nxt, lens: Optional[Ragged] = None) -> TM:
        B, K, T = c32.shape
        c32 = c32.contiguous()
        out = self._alloc_for(B, T, self.cfg.dimension, nxt, lens)
        _lib.check(self.lib.ssrhip_rvq_decode(c32.data_ptr(), self.codebooks.data_ptr(), out.interior, B, T, self.cfg.dimension, K,
                                              self.cfg.bins, out.bstride, self._s()), "ssrhip_rvq_decode")
        self._fill_pads(out)
        return out

    @staticmethod
    def _channel_major(y: TM) -> torch.Tensor:
        """[B][T][C] time-major result -> the reference's [B, C, T]. A 1-channel waveform without halo rows is the same memory
        either way (a view, no copy); anything else is one transposing copy of a small tensor (latents, detector output)."""
        if y.C == 1 and y.padL == 0 and y.padR == 0:
            return y.data.view(y.B, 1, y.T)
        return y.interior_view().transpose(1, 2).contiguous()

    @torch.no_grad()
    def decode_latent(self, codes: torch.Tensor) -> torch.Tensor:
        c32 = self._codes32(codes)
        return self._sized("latent", int(c32.shape[0]), int(c32.shape[-1]), lambda: self._dequant(c32, None).interior_view().transpose(1, 2).contiguous())

    @torch.no_grad()
    def decode(self, codes: torch.Tensor, scale=None) -> torch.Tensor:
        assert scale is None, "renormalize=False codec: scale must be None (wmencodec.py:199-203)"
        c32 = self._codes32(codes)

        def body(lo, hi):
            z = self._dequant(c32[lo:hi], self.decoder.nodes[0])
            return (self._channel_major(self._run(self.decoder.nodes, z)),)

        return self._join(self._in_lanes(c32.shape[0], lambda lo, hi: self._sized("decode", hi - lo, int(c32.shape[-1]), lambda: body(lo, hi))), 0)

    # ------------------------------------------------------------------ ragged batches (items of different lengths in one pass)
    # cost model of one dense pass over a bucket, in microseconds per frame of its longest item: every frame is one LSTM time step
    # of each layer (a latency-bound launch whose cost hardly depends on the batch) + the convolutions' share per item
    # (profiles/r02_codec_b32_kernel_trace_summary.md: ~10 us per step and layer; 85 ms / (32 items x 1500 frames) per item)
    RAGGED_STEP_US = 20.0
    RAGGED_ITEM_US = 1.8
    RAGGED_MAX_FRAMES = 256 * 1500          # items x frames of one dense pass (memory: ~0.3 GB per 1500 frames in the decoder)

    def _ragged_buckets(self, lens: Sequence[int], per_item_us: Optional[float] = None) -> List[List[int]]:
        """Partition item indices into buckets that are decoded as ONE dense pass each (every item padded to its bucket's
        longest). Optimal contiguous partition of the items sorted by length under the cost model above (O(n^2)): short items
        ride along with long ones as long as the padding they add costs less than another pass's per-frame launch chain."""
        order = sorted(range(len(lens)), key=lambda i: -int(lens[i]))
        n = len(order)
        item_us = self.RAGGED_ITEM_US if per_item_us is None else per_item_us
        best = [0.0] + [float("inf")] * n           # best[j]: cost of the first j items (longest first)
        cut = [0] * (n + 1)
        for j in range(1, n + 1):
            for i in range(j):                      # bucket = sorted items i .. j-1, its longest is item i
                t_max, cnt = int(lens[order[i]]), j - i
                if cnt > 1 and cnt * t_max > self.RAGGED_MAX_FRAMES:
                    continue
                c = best[i] + t_max * (self.RAGGED_STEP_US + item_us * cnt)
                if c < best[j]:
                    best[j], cut[j] = c, i
        buckets, j = [], n
        while j > 0:
            buckets.append(order[cut[j]: j])
            j = cut[j]
        return buckets[::-1]

    def _stack_ragged(self, items: Sequence[torch.Tensor], idx: Sequence[int], lead: int, dtype, fill=0) -> Tuple[torch.Tensor, List[int]]:
        """items[i]: [1, *lead dims, T_i] (or without the leading 1) -> dense [len(idx), *lead, T_max] on the device + the lengths."""
        ts = [items[i].reshape(*items[i].shape[-(lead + 1):]) for i in idx]
        lens = [int(t.shape[-1]) for t in ts]
        out = torch.full((len(idx),) + tuple(ts[0].shape[:-1]) + (max(lens),), fill, dtype=dtype, device=self.device)
        for j, t in enumerate(ts):
            out[j, ..., : lens[j]] = t.to(self.device, dtype)
        return out, lens

    @torch.no_grad()
    def decode_ragged(self, codes: Sequence[torch.Tensor], scale=None) -> List[torch.Tensor]:
        """Batched `decode` of items of DIFFERENT lengths: codes[i] is [1, K, T_i] (or [K, T_i]); returns the list of waveforms
        [1, 1, T_i * hop]. Contract: result i == `decode(codes[i])` of that item alone (same arithmetic per output sample; the
        LSTM step kernel is chosen by batch size, so the last bits may differ) — the SEANet convolutions are not causal, so every
        layer gives each item its own halo right behind its own last row (`ssrhip_pad_ragged`) instead of padding to the longest.
        The reference decodes one utterance at a time (inference_v2.py:331-358, wmencodec.py:341-356)."""
        assert scale is None, "renormalize=False codec: scale must be None (wmencodec.py:199-203)"
        out: List[Optional[torch.Tensor]] = [None] * len(codes)
        hop = int(math.prod(self.cfg.ratios))
        for idx in self._ragged_buckets([int(c.shape[-1]) for c in codes]):
            dense, lens = self._stack_ragged(codes, idx, 1, torch.int64)
            c32 = self._codes32(dense)
            rg = Ragged(lens, self.device) if min(lens) != max(lens) else None
            # (a ragged pass allocates what the dense pass of its longest item allocates: same envelope family as `decode`)
            wav = self._sized("decode", len(idx), max(lens), lambda: self._channel_major(self._run(self.decoder.nodes, self._dequant(c32, self.decoder.nodes[0], rg))))
            for j, i in enumerate(idx):
                out[i] = wav[j: j + 1, :, : lens[j] * hop]
        return out

    @torch.no_grad()
    def wmdecode_ragged(self, codes: Sequence[torch.Tensor], labels: Sequence[torch.Tensor], wavforms: Sequence[torch.Tensor], scale=None,
                        with_mark: bool = True):
        """Batched `wmdecode` of items of different lengths (see `decode_ragged`): codes[i] [1,K,T_i], labels[i] [1,T_i],
        wavforms[i] [1,1,T_i*hop]. Returns (list of wav [1,1,T_i*hop], list of mark [1,T_i,2] or None)."""
        assert scale is None and self.has_wm
        hop = int(math.prod(self.cfg.ratios))
        n = len(codes)
        assert len(labels) == n and len(wavforms) == n
        wavs: List[Optional[torch.Tensor]] = [None] * n
        marks: List[Optional[torch.Tensor]] = [None] * n
        for i in range(n):
            T = int(codes[i].shape[-1])
            if int(labels[i].shape[-1]) != T or int(wavforms[i].shape[-1]) != T * hop:
                raise ValueError(f"item {i}: {T} frames need {T} labels and {T * hop} samples, got {tuple(labels[i].shape)} / {tuple(wavforms[i].shape)}")
        # 3x the decoder's work per item (skip encoder + decoder [+ detector]): the per-item share of the cost model scales with it
        for idx in self._ragged_buckets([int(c.shape[-1]) for c in codes], per_item_us=3 * self.RAGGED_ITEM_US):
            dense, lens = self._stack_ragged(codes, idx, 1, torch.int64)
            lab, _ = self._stack_ragged(labels, idx, 0, torch.int64)
            wv, _ = self._stack_ragged(wavforms, idx, 0, torch.float32)
            rg = Ragged(lens, self.device) if min(lens) != max(lens) else None
            c32, l32, wv1 = self._codes32(dense), self._labels32(lab), wv.unsqueeze(1)
            w, m = self._sized("wmdecode" + ("+mark" if with_mark else ""), len(idx), max(lens), lambda: self._wmdecode_dense(c32, l32, wv1, with_mark, rg))
            for j, i in enumerate(idx):
                wavs[i] = w[j: j + 1, :, : lens[j] * hop]
                if m is not None:
                    marks[i] = m[j: j + 1, : lens[j]]
        return wavs, (marks if with_mark else None)

    def _concat_proj(self, j: int, skip: TM, labels32: torch.Tensor, rep: int, x: TM, nxt) -> TM:
        """wm_proj_j(ELU(cat(skip, wm_embed(labels upsampled)))) + x   (seanet.py:577-591) without building the concatenation:
        the 1x1 convolution splits into W_a . ELU(skip) + W_b . ELU(embed(label)); the second term has one value per label
        (`self.wm_cls[j]`, computed at load) and enters the GEMM's epilogue as a per-row class bias (include/ssrhip.h rbias)."""
        c, (Wa, cls) = self.wm_proj[j], self.wm_cls[j]
        assert skip.T == x.T and skip.C == Wa.shape[1] and labels32.shape[1] * rep >= skip.T, (skip.T, x.T, skip.C, Wa.shape, labels32.shape, rep)
        out = self._alloc_for(skip.B, skip.T, c.Cout, nxt, skip.lens)
        assert not x.elu
        self._gemm(skip.interior, Wa, c.b, out.interior, skip.T, c.Cout, skip.C, skip.C, c.Cout, act_in=self._act_in(True, skip), R=x.interior, ldr=x.C,
                   batch=skip.B, sA=skip.bstride, sC=out.bstride, sR=x.bstride, rowcls=(cls, labels32, rep))
        self._fill_pads(out, structural_zero=(nxt is not None and nxt[1] == "convtr"))
        return out

    def _labels32(self, labels: torch.Tensor) -> torch.Tensor:
        """Watermark labels on the device as int32, range-checked like the code ids: they index the class-bias table in the GEMM
        epilogue, where the reference's F.embedding (seanet.py:562) raises IndexError for an id outside the table."""
        lab = labels.to(self.device, torch.int32).contiguous()
        if lab.numel() > 0:
            lo_hi = torch.stack(torch.aminmax(lab)).tolist()
            if lo_hi[1] >= self.wm_table.shape[0] or lo_hi[0] < 0:
                raise IndexError("index out of range in self")
        return lab

    @torch.no_grad()
    def wmdecode(self, codes: torch.Tensor, labels: torch.Tensor, wavform: torch.Tensor, scale=None, with_mark: bool = True):
        """-> (wav [B,1,T], mark [B,T',2]) as wmencodec.py:358-375. `with_mark=False` skips the detector pass whose
        output `AudioTokenizer.wmdecode` discards (data/tokenizer.py:133) and returns mark=None."""
        assert scale is None and self.has_wm
        lab_all = self._labels32(labels)
        c32 = self._codes32(codes)
        parts = self._in_lanes(c32.shape[0], lambda b0, b1: self._sized(
            "wmdecode" + ("+mark" if with_mark else ""), b1 - b0, int(c32.shape[-1]),
            lambda: self._wmdecode_dense(c32[b0:b1], lab_all[b0:b1].contiguous(), wavform[b0:b1], with_mark, None)))
        return self._join(parts, 0), self._join(parts, 1)

    def _wmdecode_dense(self, c32: torch.Tensor, lab: torch.Tensor, wavform: torch.Tensor, with_mark: bool, lens: Optional[Ragged]):
        """One dense pass of the watermark decoder (seanet.py:555-600) over a batch; `lens` = per-item frame counts of a ragged batch."""
        r = list(self.cfg.ratios)
        assert len(r) == 4, "the watermark decoder's slicing (seanet.py:560-591) is written for 4 ratios"
        dec, senc = self.wmdecoder, self.skip_encoder
        cuts = [(0, 4), (4, 7), (7, 10), (10, None)]
        dn = [dec.slice_nodes(lo, hi) for lo, hi in cuts]
        reps = [r[0] * r[1] * r[2], r[0] * r[1], r[0], 1]
        lens_s = lens.scaled(int(math.prod(r))) if lens is not None else None
        # skip features at 4 scales; every skip is consumed by a k=1 conv => no halo
        z = self._run(senc.slice_nodes(0, 2), self._input_tm(wavform, senc.nodes[0], lens_s), after=senc.slice_nodes(2, 5)[0])
        sk = []
        for lo, hi in [(2, 5), (5, 8), (8, 11), (11, None)]:
            nxt_nodes = senc.slice_nodes(hi, None) if hi is not None else []
            z = self._run(senc.slice_nodes(lo, hi), z, after=(nxt_nodes[0] if nxt_nodes else None))
            sk.append(z)
        x = self._dequant(c32, None, lens)
        for j in range(4):
            skip, rep = sk[3 - j], reps[3 - j]
            out = self._concat_proj(j, skip, lab, rep, x, dn[j][0])
            x = self._run(dn[j], out, after=None)
        wav = self._channel_major(x)
        if not with_mark:
            return wav, None
        m = self._run(self.wm_encoder.nodes, self._input_tm(wav, self.wm_encoder.nodes[0], lens_s), after=None)
        mk = self._conv(self.wm_predictor, m, None)
        return wav, mk.interior_view().contiguous()

    @torch.no_grad()
    def detect_watermark(self, x: torch.Tensor) -> torch.Tensor:
        """wmencodec.py:377-382, including its argmax over the LAST dim of [B,2,T'] (time)."""
        assert x.dim() == 3

        def body(lo, hi):
            m = self._run(self.wm_encoder.nodes, self._input_tm(x[lo:hi], self.wm_encoder.nodes[0]), after=None)
            mk = self._conv(self.wm_predictor, m, None).interior_view().transpose(1, 2)      # [B,2,T']
            return (torch.argmax(mk.squeeze(-1), dim=-1),)

        return self._join(self._in_lanes(x.shape[0], lambda lo, hi: self._sized("detect", hi - lo, int(x.shape[-1]), lambda: body(lo, hi))), 0)
