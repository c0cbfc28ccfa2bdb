"""Host-side integer code of the decode path: prompt layout before generation and span
re-assembly after it.  Pure NumPy (index arithmetic, no loops over time steps).

Restates, vectorised:
  build_layout  <- SSR_Speech.rearrange / get_pattern_sequence / shift / insert_mask / cat_y and the
                   interval arithmetic in inference()   (reference models/ssr.py:381-436, 466-502, 604-625)
  undelay       <- revert_pattern_sequence              (models/ssr.py:438-464)
  assemble      <- the tail of inference()              (models/ssr.py:776-812)
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def delay_pattern(seg: np.ndarray, fill: int) -> np.ndarray:
    """[K,T] -> [K,T+K-1]: codebook q delayed by q columns, gaps = `fill` (delays=[0..K-1])."""
    K, T = seg.shape
    out = np.full((K, T + K - 1), fill, dtype=np.int64)
    for q in range(K):
        out[q, q:q + T] = seg[q]
    return out


def undelay(pattern: np.ndarray, fill: int) -> np.ndarray:
    """[K,S] -> [K,S-K+1]: inverse of delay_pattern."""
    K, S = pattern.shape
    T = S - (K - 1)
    out = np.full((K, max(T, 0)), fill, dtype=np.int64)
    for q in range(K):
        out[q] = pattern[q, q:q + T]
    return out


def intervals(y_len: int, mask_interval: np.ndarray) -> Tuple[List[Tuple[int, int]], List[Tuple[int, int]]]:
    """(non_mask_intervals, mask_intervals) exactly as models/ssr.py:609-616."""
    mi = [(int(a), int(b)) for a, b in np.asarray(mask_interval).reshape(-1, 2)]
    starts = [a for a, _ in mi] + [y_len]
    ends = [0] + [b for _, b in mi]
    return list(zip(ends, starts)), mi


def build_layout(y: np.ndarray, mask_interval: np.ndarray, args):
    """y [K,T] int, mask_interval [M,2] -> (cated [K,T0], mask_position, num_task, non_mask_intervals).

    Column order: kept segments (first gets <sos> in front, last gets <eos> behind) each delayed
    separately and separated by their <mts+i> token, then for every masked span <mts+i> followed by
    its delayed content + <eog>; the result is cut right before the first generation-side <mts>."""
    y = np.asarray(y, dtype=np.int64)
    K, T = y.shape
    nmi, mi = intervals(T, mask_interval)
    col = lambda v: np.full((K, 1), v, dtype=np.int64)
    segs = []
    for i, (s, e) in enumerate(nmi):
        body = y[:, s:e]
        if i == 0:
            segs.append(np.concatenate([col(args.sos), body], 1))
        elif i == len(nmi) - 1:
            segs.append(np.concatenate([body, col(args.eos)], 1))
        else:
            segs.append(body)
    for s, e in mi:
        segs.append(np.concatenate([y[:, s:e], col(args.eog)], 1))
    shifted = [delay_pattern(s, args.empty_token) for s in segs]
    n_masks = (len(shifted) - 1) // 2
    assert 2 * n_masks == len(shifted) - 1 and n_masks <= args.max_n_spans, (len(shifted), args.max_n_spans)
    mask_value = list(range(args.mts, args.mts + n_masks)) * 2
    pieces, mask_position, run = [], [], 0
    for j in range(len(shifted) - 1):
        pieces.append(shifted[j])
        run += shifted[j].shape[1]
        mask_position.append(run)
        pieces.append(col(mask_value[j]))
        run += 1
    pieces.append(shifted[-1])
    cated = np.concatenate(pieces, 1)
    num_task = len(mask_position) // 2
    return cated[:, : mask_position[num_task]], mask_position, num_task, nmi


def assemble(y: np.ndarray, spans: Sequence[np.ndarray], non_mask_intervals, args):
    """y [K,T] original codes; spans[i] [S_i,K] generated rows of span i (incl. the eog cascade).
    -> (res [K,T'], marks [T'], masks, non_mask_intervals) as models/ssr.py:776-805."""
    K = y.shape[0]
    res, marks, masks, tmp = [], [], [], 0
    for (s, e), sp in zip(non_mask_intervals, spans):
        gen = undelay(np.asarray(sp, dtype=np.int64).T, args.empty_token)[:, :-1]   # drop the eog column
        res.append(y[:, s:e])
        masks.append((tmp, tmp + e - s))
        marks += [0] * (e - s)
        res.append(gen)
        tmp += (e - s) + gen.shape[1]
        marks += [1] * gen.shape[1]
    ls, le = non_mask_intervals[-1]
    if y.shape[1] != le + 1:            # reference quirk kept verbatim (ssr.py:799)
        res.append(y[:, ls:le])
        masks.append((tmp, tmp + le - ls))
        marks += [0] * (le - ls)
    return np.concatenate(res, 1), np.asarray(marks, dtype=np.int64), masks, list(non_mask_intervals)
