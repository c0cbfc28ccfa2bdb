"""CPU: the operand layouts of `csrc/lstm_split.hip` (LSTM recurrence on the bf16 matrix cores with split operands) replayed on the host.

The kernel streams both operands from global memory in MFMA FRAGMENT order — one wave-level 16-byte-per-lane load = one operand of
`v_mfma_f32_32x32x16_bf16` — so three index maps have to agree: the host packer of W_hh (`wmencodec.pack_lstm_whh_planes`), the producer
of h (the finishing thread of step t - 1 writes h[b][j] at its fragment position) and the consumer (block address + lane * 8 + element,
and the MFMA's own operand convention: lane l of the A operand = row l % 32, k = 8 (l / 32) + e; of the B operand = column l % 32, same
k; accumulator block [32 mb + m][32 nb + n]). This test replays the kernel's addressing (written out below exactly as the .hip file has
it) with three independent "planes" of small integers per operand and compares every gate pre-activation with the direct sum of the six
cross products. It cannot see arithmetic on the device — that is `tests/test_gpu_kernels.py::test_lstm_split_step_matches_fp64` — but it
pins the algebra: a wrong stride or a swapped lane half shows up here without a GPU."""
import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd.codec.wmencodec import pack_lstm_whh_planes

PAIRS = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]          # (W piece, h piece): the kernel's PA / PB


def frag_block(outer, wave, ks, s, blk, q):                        # lstm_split.hip frag_block(): in bf16 elements
    return ((((outer * 4 + wave) * ks + s) * 2 + blk) * 3 + q) * 512


def produce_h(hq, B, C):
    """the finishing phase's stores: piece q of h[b][j] -> its fragment position (one buffer)"""
    ks, nbg = C // 64, (B + 63) // 64
    out = np.zeros(nbg * 64 * C * 3)
    kq = C >> 2
    for b in range(B):
        bg, bl = divmod(b, 64)
        for j in range(C):
            w2, s2, lh2, e2 = j // kq, (j % kq) >> 4, (j >> 3) & 1, j & 7
            for q in range(3):
                out[frag_block(bg, w2, ks, s2, bl >> 5, q) + (lh2 * 32 + (bl & 31)) * 8 + e2] = hq[q][b, j]
    return out


def replay_kernel(wflat, hflat, B, C):
    """gates[b][g C + j] as the workgroups compute them (without gin)"""
    ks, nbg = C // 64, (B + 63) // 64
    gates = np.zeros((nbg * 64, 4 * C))
    for ub in range(C // 16):
        for bg in range(nbg):
            D = np.zeros((64, 64))
            for wave in range(4):
                for s in range(ks):
                    for qa, qb in PAIRS:
                        for mb in range(2):
                            for nb in range(2):
                                a0, b0 = frag_block(ub, wave, ks, s, mb, qa), frag_block(bg, wave, ks, s, nb, qb)
                                A = wflat[a0:a0 + 512].reshape(2, 32, 8)          # [lh][li][e]: lane = 32 lh + li, 8 elements per lane
                                Bm = hflat[b0:b0 + 512].reshape(2, 32, 8)
                                D[mb * 32:mb * 32 + 32, nb * 32:nb * 32 + 32] += np.einsum("hme,hne->mn", A, Bm)
            for m in range(64):
                g, u = divmod(m, 16)
                gates[bg * 64:bg * 64 + 64, g * C + ub * 16 + u] = D[m, :]
    return gates


@pytest.mark.parametrize("B,C", [(70, 128), (64, 256)])
def test_fragment_layouts_of_the_split_lstm_step_agree(B, C):
    rng = np.random.default_rng(B + C)
    wq = [rng.integers(-3, 4, size=(4 * C, C)).astype(np.float64) for _ in range(3)]
    hq = [rng.integers(-3, 4, size=(B, C)).astype(np.float64) for _ in range(3)]
    packed = pack_lstm_whh_planes(torch.from_numpy(np.stack(wq)))
    assert tuple(packed.shape) == (C // 16, 4, C // 64, 2, 3, 2, 32, 8) and packed.is_contiguous()
    gates = replay_kernel(packed.numpy().reshape(-1), produce_h(hq, B, C), B, C)
    want = sum(hq[qb] @ wq[qa].T for qa, qb in PAIRS)
    assert np.array_equal(gates[:B], want)
    assert not gates[B:].any()                                     # batch rows past B: their h planes stay zero


def test_packer_rejects_shapes_the_kernel_cannot_take():
    with pytest.raises(AssertionError):
        pack_lstm_whh_planes(torch.zeros(3, 4 * 48, 48, dtype=torch.int16))
