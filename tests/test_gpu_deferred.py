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


def test_device_messages_equal_host_messages(built, monkeypatch):
    """mgpu_set_device_messages: the records k_build_messages leaves in HBM, feed by feed, are byte for byte the records the
    host builder writes (and therefore the oracle's messages); the field decoder and the beast encoder take them where they are."""
    import readsb_amd
    monkeypatch.setenv("MGPU_CHUNK_BUFFERS", "16")
    sizes = [48 * B, 40 * B, 19 * B + 999]
    iq = helpers.synth(nsamples=sum(sizes), seed=1234, rate=3500.0)
    want, wst = helpers.oracle_run(iq, 0, 2, 1, 58)
    blocks = _blocks(iq, sizes)
    host = []
    d = readsb_amd.Demodulator(nfix_crc=2, startup_time_ms=helpers.STARTUP_MS, max_samples=64 * B)
    d.set_deferred(True)
    buf = np.empty(300000, dtype=readsb_amd.MSG_DTYPE)
    for blk in blocks:
        d.feed_iq(blk)
        m, _ = d.collect_feed(buf)
        host.append(m.copy())
    d.set_deferred(False)
    d.close()

    d = readsb_amd.Demodulator(nfix_crc=2, startup_time_ms=helpers.STARTUP_MS, max_samples=64 * B)
    d.set_deferred(True)
    d.set_device_messages(True)
    got = []
    d.feed_iq(blocks[0])
    for k in range(1, len(blocks)):
        d.feed_iq(blocks[k])
        ptr, n, _ = d.collect_feed_device()
        assert n == len(host[k - 1]) and ptr
        got.append((ptr, n))
    # the last feed through the copying entry, with the settled counters
    m, cnt = d.collect_feed(buf, want_counters=True)
    assert m.tobytes() == host[-1].tobytes()
    # the device lists of the earlier feeds are still valid (three more feeds may start before they are reused)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for (ptr, n), h in zip(got, host):
        dev = np.empty(n, dtype=readsb_amd.MSG_DTYPE)
        assert hip.hipMemcpy(dev.ctypes.data, ptr, n * 64, 2) == 0          # hipMemcpyDeviceToHost
        assert dev.tobytes() == h.tobytes()
    d.finish()
    _, cnt = d.collect_feed(buf, want_counters=True)
    helpers.assert_same_messages(np.concatenate(host), want)
    helpers.assert_same_counters(cnt, wst)
    d.close()
