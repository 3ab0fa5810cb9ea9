"""CPU: the plan of a sharded capture (readsb_amd/shard.py) — ranges are whole buffers that tile the capture in order, and every
later range looks back at least two ICAO-filter generations (120 s of samples: an address is dropped by the second expiry after
its last add, icao_filter.c:96-130 with readsb.c:1227-1231) plus the 326 samples of history its first buffer starts from
(sdr_ifile.c:209-213).  The runs themselves are in tests/test_gpu_shard.py."""
import numpy as np
import pytest

from readsb_amd import shard


@pytest.mark.parametrize("n,world", [(8_639_873_024, 8), (8_639_873_024, 1), (696_000_000, 2), (131072 * 3 + 17, 4), (131072, 3), (5, 2)])
def test_ranges_tile_the_capture_in_whole_buffers(n, world):
    r = shard.shard_ranges(n, world)
    assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
    for (a, b), (c, d) in zip(r, r[1:]):
        assert b == c and a <= b                                    # in order, no gap, no overlap (empty ranges allowed)
    for a, b in r:
        assert a % shard.BUF == 0 and (b % shard.BUF == 0 or b == n)


def test_warmup_covers_two_filter_generations():
    two_generations = 2 * shard.FILTER_TTL_S * shard.SAMPLE_RATE
    # a generation lasts up to 60 s + two buffers (the expiry's clock is data-dependent within a buffer, and tested per buffer)
    assert shard.WARMUP % shard.BUF == 0 and two_generations + 4 * shard.BUF <= shard.WARMUP <= two_generations + 6 * shard.BUF
    for first in (0, shard.BUF, 100 * shard.BUF, shard.WARMUP, shard.WARMUP + shard.BUF, 40000 * shard.BUF):
        w = shard.warmup_start(first)
        assert w % shard.BUF == 0 and 0 <= w <= first
        assert w == 0 or first - w == shard.WARMUP                  # the whole look-back, or everything there is
        need = shard.needed_from(first)
        assert need == max(0, w - shard.TRAILING)                   # + the history the warm-up's first buffer starts from


def test_one_hour_on_eight_ranks():
    """BASELINE configs[4] on 8 GPUs: every rank's share of the work, in samples swept."""
    n = 8_640_000_000 - 8_640_000_000 % shard.BUF
    r = shard.shard_ranges(n, 8)
    swept = [(b - a) + (a - shard.warmup_start(a)) for a, b in r]
    assert swept[0] == r[0][1]                                      # rank 0: its range, nothing else
    assert max(swept) / (n / 8) < 1.28                              # the others: + 120 s of 450 s
    assert np.sum([b - a for a, b in r]) == n
