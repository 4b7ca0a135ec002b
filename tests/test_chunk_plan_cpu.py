"""CPU: the ray-chunk plan of the secondary march (render.plan_secondary_chunks): coverage without gaps or overlaps, the serial plan's chunk
bound, and -- for several streams -- equal chunks whose number is a multiple of the streams (the static assignment chunk j -> thread j mod n
is balanced), none above 5 / 8 of the serial chunk, none below the minimum unless the batch itself is smaller."""
import pytest

from intrinsicavatar_amd.render import plan_secondary_chunks as plan


def _covers(chunks, M):
    assert chunks[0][0] == 0 and chunks[-1][1] == M
    assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
    assert all(c1 > c0 for c0, c1 in chunks)


@pytest.mark.parametrize("M", [1, 77, 1 << 22, 16_777_216, 16_777_217, 41_395_710, 114_825_021])
def test_serial_plan(M):
    ch = plan(M, 1 << 24, 1)
    _covers(ch, M)
    assert len(ch) == -(-M // (1 << 24)) and max(c1 - c0 for c0, c1 in ch) <= 1 << 24


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("M", [8_388_609, 28_581_756, 41_395_710, 114_825_021, 300_000_001])
def test_stream_plan_is_balanced(M, n):
    ch = plan(M, 1 << 24, n)
    _covers(ch, M)
    sizes = [c1 - c0 for c0, c1 in ch]
    assert len(ch) % n == 0
    assert max(sizes) - min(sizes) <= 1                            # bounds[i] = M i / n
    assert max(sizes) <= max((1 << 24) * 5 // 8, 1 << 22)
    per_thread = [sum(sizes[k::n]) for k in range(n)]
    assert max(per_thread) - min(per_thread) <= len(ch)


def test_stream_plan_small_batches_and_the_minimum_chunk():
    assert plan(0, 1 << 24, 2) == []
    assert plan(5_000_000, 1 << 24, 2) == [(0, 2_500_000), (2_500_000, 5_000_000)] or len(plan(5_000_000, 1 << 24, 2)) == 2
    # a batch smaller than two minimum chunks is split in two anyway only down to the minimum
    ch = plan(3_000_000, 1 << 24, 2, min_chunk=1 << 22)
    _covers(ch, 3_000_000)
    assert len(ch) in (1, 2)
    ch = plan(700_001, 240_000, 2, min_chunk=1000)                 # the GPU test's shape: six chunks of 116 667 rays
    _covers(ch, 700_001)
    assert len(ch) == 6 and ch[-1] == (583_334, 700_001)


def test_stream_plan_properties_on_random_small_inputs():
    """the fuzz of the round-4 review (M = 321, chunk 24, n = 4, min_chunk 16 gave 23 chunks): the number of chunks is a multiple of the
    streams whenever there are at least as many rays as chunks, sizes differ by at most one ray, coverage is exact."""
    import random
    rnd = random.Random(0)
    for _ in range(20000):
        M, chunk, n, mc = rnd.randint(1, 5000), rnd.randint(1, 400), rnd.randint(2, 5), rnd.randint(1, 64)
        ch = plan(M, chunk, n, mc)
        _covers(ch, M)
        sizes = [b - a for a, b in ch]
        assert max(sizes) - min(sizes) <= 1, (M, chunk, n, mc)
        cmax = max(min(chunk * 5 // 8, -(-M // n)), mc, 1)
        want = n * (-(-M // (n * cmax)))
        assert len(ch) == min(want, M) and (len(ch) % n == 0 or M < want), (M, chunk, n, mc, len(ch))
        assert max(sizes) <= cmax
