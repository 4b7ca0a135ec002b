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
    assert max(sizes) - min(sizes) <= len(ch)                      # equal up to the rounding of the last chunk
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
    assert len(ch) == 6 and ch[-1] == (583_335, 700_001)
