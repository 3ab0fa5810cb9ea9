"""-m gpu: deferred feeds (mgpu_set_deferred, include/modes_gpu.h): a stream handed over block after block with the next
block enqueued before the previous one is collected must give, feed by feed, exactly the messages of the synchronous
calls, and after the drain exactly the counters — which are the oracle's for the whole stream."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

B = 131072


def _blocks(iq, sizes):
    out, off = [], 0
    for n in sizes:
        out.append(iq[off * 2:(off + n) * 2])
        off += n
    assert off * 2 == len(iq)
    return out


@pytest.mark.parametrize("external", [False, True])
def test_deferred_feeds_equal_synchronous_feeds(built, external, monkeypatch):
    import readsb_amd
    monkeypatch.setenv("MGPU_CHUNK_BUFFERS", "16")             # several pipeline chunks per feed
    sizes = [40 * B, 56 * B, 24 * B, 33 * B + 4321]            # several 16-buffer chunks per feed; the last block ends the stream short
    iq = helpers.synth(nsamples=sum(sizes), seed=909, rate=3000.0)
    want, wst = helpers.oracle_run(iq, 0, 2, 1, 58)
    blocks = _blocks(iq, sizes)

    # synchronous reference run of the library itself: per-feed message lists
    d = readsb_amd.Demodulator(nfix_crc=2, startup_time_ms=helpers.STARTUP_MS, max_samples=16 * B)
    sync = []
    for blk in blocks:
        for off in range(0, len(blk) // 2, 16 * B):
            d.feed_iq(blk[off * 2:(off + 16 * B) * 2])
        m, _ = d.collect()
        sync.append(m.copy())
    d.close()

    d = readsb_amd.Demodulator(nfix_crc=2, startup_time_ms=helpers.STARTUP_MS, max_samples=64 * B)
    d.set_deferred(True)
    arrays = [np.empty(200000, dtype=readsb_amd.MSG_DTYPE) for _ in range(2)]
    got = []

    def feed(k):
        if external:
            d.set_message_buffer(arrays[k % 2])
        d.feed_iq(blocks[k])

    feed(0)
    for k in range(1, len(blocks)):
        feed(k)                                                 # enqueued before the previous feed is collected
        m, _ = d.collect_feed(arrays[(k - 1) % 2])
        got.append(m.copy())
    m, cnt = d.collect_feed(arrays[(len(blocks) - 1) % 2], want_counters=True)
    got.append(m.copy())
    d.finish()
    for k, (a, b) in enumerate(zip(got, sync)):
        assert len(a) == len(b) and a.tobytes() == b.tobytes(), f"feed {k}: deferred messages differ from the synchronous call's"
    helpers.assert_same_messages(np.concatenate(got), want)
    helpers.assert_same_counters(cnt, wst)
    tm = d.timing()
    assert tm["n_chunks"] >= 4
    d.set_deferred(False)
    d.close()


def test_deferred_mode_refuses_the_other_entries(built):
    import readsb_amd
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=16 * B)
    d.set_deferred(True)
    mag = np.zeros(B + 326, dtype=np.uint16)
    with pytest.raises(readsb_amd.MgpuError):
        d.demod_mag_buf(mag, B, 0, 0, 0.0)
    iq = helpers.synth(nsamples=4 * B, seed=3)
    for _ in range(4):
        d.feed_iq(iq)
    with pytest.raises(readsb_amd.MgpuError):                   # a fifth uncollected feed
        d.feed_iq(iq)
    d.close()
