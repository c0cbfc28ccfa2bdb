"""CPU: the host-side KV page allocator (`engine.PagePool`) — the part of the paged cache that replaces the reference's
ever-growing dense `past` tensor (models/ssr.py:685-686). Pure bookkeeping; the kernels only see the table it fills."""
import pytest

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd.engine import PagePool


def test_pages_come_out_in_the_given_order_and_only_once():
    p = PagePool(6, order=[4, 1, 5, 0, 3, 2])
    got = [p.take("a") for _ in range(6)]
    assert got == [4, 1, 5, 0, 3, 2] and p.n_free == 0
    with pytest.raises(RuntimeError, match="exhausted"):
        p.take("a")


def test_returned_pages_are_reused_before_untouched_ones_and_double_free_is_an_error():
    p = PagePool(5)
    a = [p.take(0), p.take(0)]
    b = [p.take(1)]
    p.give_back(a)
    assert p.n_free == 4
    assert p.take(2) in a                       # LIFO reuse: the cache-warm page first
    with pytest.raises(RuntimeError, match="returned twice"):
        p.give_back([a[0], a[1], a[1]] if False else [b[0], b[0]])
    p.reset()
    assert p.n_free == 5 and p.take(0) == 0


def test_order_must_be_a_permutation():
    with pytest.raises(ValueError):
        PagePool(3, order=[0, 0, 1])
    with pytest.raises(ValueError):
        PagePool(3, order=[0, 1])
