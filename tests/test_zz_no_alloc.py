"""The entry point that allocates nothing (zkhip_prefix_product_noalloc_dev) against zkhip_prefix_product_dev, through the
C ABI: bit-identical outputs on two calls in a row, the second with a smaller input (the static block-total buffer keeps
the first call's contents), in place and out of place, and the size limit."""
import numpy as np
import pytest

from util import rand_fr

pytestmark = pytest.mark.gpu


def _both(ctx, m, init):
    n = len(m)
    d_m, d_a, d_b = ctx.upload(m), ctx.alloc(n * 32), ctx.alloc(n * 32)
    ctx.prefix_product_dev(d_m, d_a, n, init)
    ctx.prefix_product_noalloc_dev(d_m, d_b, n, init)
    a, b = ctx.download(d_a, (n, 4)), ctx.download(d_b, (n, 4))
    # in place as well
    ctx.prefix_product_noalloc_dev(d_m, d_m, n, init)
    c = ctx.download(d_m, (n, 4))
    for p in (d_m, d_a, d_b):
        ctx.free(p)
    return a, b, c


def test_no_alloc_prefix_product_matches_twice_in_a_row(ctx):
    rng = np.random.default_rng(77)
    for n in (300001, 4099):          # 147 blocks of 2048, then 3: the second call reuses a prefix of the same buffer
        m, init = rand_fr(rng, n), rand_fr(rng, 1)[0]
        a, b, c = _both(ctx, m, init)
        assert np.array_equal(a, b) and np.array_equal(a, c)
        assert np.array_equal(a[0], init)


def test_no_alloc_prefix_product_edges(ctx):
    from zkevm_circuits_amd import ffi
    rng = np.random.default_rng(78)
    init = rand_fr(rng, 1)[0]
    for n in (1, 2048, 2049):
        a, b, c = _both(ctx, rand_fr(rng, n), init)
        assert np.array_equal(a, b) and np.array_equal(a, c)
    ctx.prefix_product_noalloc_dev(0, 0, 0, init)            # empty input: nothing to do
    d = ctx.alloc(32)
    with pytest.raises(ffi.ZkhipError, match="at most"):
        ctx.prefix_product_noalloc_dev(d, d, (1 << 24) + 1, init)   # rejected before anything is launched
    ctx.free(d)
