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
    monkeypatch.setattr(readsb_amd.binding, "DEFAULT_CHUNK_BUFFERS", 16)             # several pipeline chunks per feed
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
    monkeypatch.setattr(readsb_amd.binding, "DEFAULT_CHUNK_BUFFERS", 16)
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


def test_device_built_messages_delivered_to_a_page_locked_array(built, monkeypatch):
    """mgpu_set_device_messages(2): k_build_messages stores the records straight into the caller's page-locked array — the
    host has its list and builds nothing.  Byte for byte the oracle's messages, counters settled as ever; a pageable array is
    refused loudly (the kernel could not reach it)."""
    import readsb_amd
    monkeypatch.setattr(readsb_amd.binding, "DEFAULT_CHUNK_BUFFERS", 16)
    sizes = [48 * B, 40 * B, 19 * B + 999]
    iq = helpers.synth(nsamples=sum(sizes), seed=4321, rate=3500.0)
    want, wst = helpers.oracle_run(iq, 0, 1, 1, 58)
    blocks = _blocks(iq, sizes)
    d = readsb_amd.Demodulator(nfix_crc=1, startup_time_ms=helpers.STARTUP_MS, max_samples=64 * B)
    d.set_deferred(True)
    d.set_device_messages(2)
    bufs = [d.host_alloc(200000 * 64).view(readsb_amd.MSG_DTYPE) for _ in range(2)]
    got = []
    d.set_message_buffer(bufs[0])
    d.feed_iq(blocks[0])
    for k in range(1, len(blocks)):
        d.set_message_buffer(bufs[k % 2])
        d.feed_iq(blocks[k])
        m, _ = d.collect_feed(bufs[(k - 1) % 2])
        assert m.ctypes.data == bufs[(k - 1) % 2].ctypes.data                 # in place: nothing was copied
        got.append(m.copy())
    m, _ = d.collect_feed(bufs[(len(blocks) - 1) % 2])
    got.append(m.copy())
    d.finish()
    _, cnt = d.collect_feed(bufs[0], want_counters=True)
    helpers.assert_same_messages(np.concatenate(got), want)
    helpers.assert_same_counters(cnt, wst)
    assert d.timing()["build_wait_ms"] >= 0.0
    # a pageable array: the feed call says so
    d.reset()
    d.set_message_buffer(np.empty(1000, dtype=readsb_amd.MSG_DTYPE))
    with pytest.raises(readsb_amd.MgpuError, match="page-locked"):
        d.feed_iq(blocks[0][: 2 * B])
    d.set_deferred(False)
    d.close()


def test_device_messages_into_the_callers_device_buffer(built, monkeypatch):
    """mgpu_set_device_message_buffer: the feed's records are built into the caller's own device buffer (the buffer an RCCL gather
    sends from) — the pointer mgpu_collect_device hands back, the same bytes as ever; the next feed without one goes to the
    library's list again; too small a buffer fails loudly."""
    import ctypes as C
    import readsb_amd
    monkeypatch.setattr(readsb_amd.binding, "DEFAULT_CHUNK_BUFFERS", 16)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    sizes = [40 * B, 33 * B]
    iq = helpers.synth(nsamples=sum(sizes), seed=99, rate=3000.0)
    want, _ = helpers.oracle_run(iq)
    blocks = _blocks(iq, sizes)
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=64 * B)
    d_buf = C.c_void_p()
    cap = 100000
    assert hip.hipMalloc(C.byref(d_buf), cap * 64) == 0
    try:
        d.set_deferred(True)
        d.set_device_messages(True)
        d.set_device_message_buffer(d_buf.value, cap)
        d.feed_iq(blocks[0])
        d.feed_iq(blocks[1])                                   # (no buffer named: the library's list)
        p0, n0, _ = d.collect_feed_device()
        p1, n1, _ = d.collect_feed_device()
        assert p0 == d_buf.value and p1 != d_buf.value and n0 + n1 == len(want)
        got = np.empty(n0 + n1, dtype=readsb_amd.MSG_DTYPE)
        assert hip.hipMemcpy(got.ctypes.data, p0, n0 * 64, 2) == 0 and hip.hipMemcpy(got[n0:].ctypes.data, p1, n1 * 64, 2) == 0
        helpers.assert_same_messages(got, want)
        d.finish()
        d.reset()
        d.set_device_message_buffer(d_buf.value, 10)
        d.feed_iq(blocks[0])
        with pytest.raises(readsb_amd.MgpuError):
            d.collect_feed_device()
    finally:
        d.close()
        hip.hipFree(d_buf)


def test_one_chunk_host_feeds_on_a_busy_gpu(built, monkeypatch):
    """Deferred HOST feeds of one pipeline chunk each, all different, from one page-locked block that is overwritten as soon as
    the feed call returns, while a second context keeps the GPU's main queue full: every feed uploads into the same region of
    the library's staging buffer, so feed k+1's upload must wait for feed k's converter (round-2 advisor finding: it did not),
    and the caller's block must have been read when mgpu_feed_iq returns."""
    import threading
    import readsb_amd
    monkeypatch.setattr(readsb_amd.binding, "DEFAULT_CHUNK_BUFFERS", 64)
    nfeeds, per = 12, 48 * B                                   # one chunk per feed
    iq = helpers.synth(nsamples=nfeeds * per, seed=31337, rate=4000.0)
    want, wst = helpers.oracle_run(iq, 0, 1, 1, 58)

    stop = threading.Event()

    def hog():                                                  # a stream of its own, resident, looping: the GPU stays saturated
        h = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=256 * B)
        h.upload_iq(helpers.synth(nsamples=256 * B, seed=5, rate=6000.0))
        while not stop.is_set():
            h.reset()
            h.feed_resident(256 * B)
            h.collect(reuse=True)
        h.close()

    t = threading.Thread(target=hog)
    t.start()
    try:
        d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=64 * B)
        d.set_deferred(True)
        block = d.host_alloc(per * 2)                           # page-locked: the uploads are truly asynchronous
        out = np.empty(400000, dtype=readsb_amd.MSG_DTYPE)
        got = []
        for k in range(nfeeds):
            block[:] = iq[k * per * 2:(k + 1) * per * 2]
            d.feed_iq(block)
            block[:] = 0x55                                     # the block is the caller's again
            if k >= 2:
                m, _ = d.collect_feed(out)
                got.append(m.copy())
        for _ in range(2):
            m, _ = d.collect_feed(out)
            got.append(m.copy())
        d.finish()
        _, cnt = d.collect_feed(out, want_counters=True)
    finally:
        stop.set()
        t.join()
    helpers.assert_same_messages(np.concatenate(got), want)
    helpers.assert_same_counters(cnt, wst)
    d.host_free(block)
    d.close()


def test_a_stream_s_results_do_not_depend_on_the_chunking(built):
    """The same continuous stream — a block of 2304 buffers fed twelve times — through pipelines with chunks of 2048 / 1024 / 768 / 1536
    buffers and one to three feeds in flight: the kernels' partitioning, the host walk's rounds (a chunk of 2048 or 1536 buffers is
    walked as two), the slots' reuse and the deferred protocol all differ, every feed's messages and the final counters may not.
    (tools/stress_stream.py is the long form: 400 feeds x 5 variants.)"""
    import readsb_amd
    nbuf, feeds = 2304, 12
    n = nbuf * 131072
    iq = helpers.synth(nsamples=n, seed=777, rate=2500.0, threads=16)
    ref = None
    for chunk_buffers, ahead in ((2048, 2), (1024, 1), (768, 3), (1536, 2)):
        d = readsb_amd.Demodulator(max_samples=n, startup_time_ms=helpers.STARTUP_MS, chunk_buffers=chunk_buffers)
        try:
            d.upload_iq(iq)
            d.feed_resident(n)
            m0, _ = d.collect(reuse=True)
            bufs = [np.empty(len(m0) * 5 // 4 + 1024, dtype=readsb_amd.MSG_DTYPE) for _ in range(ahead + 1)]
            d.reset()
            d.set_deferred(True)
            got = []
            for k in range(feeds + ahead):
                if k < feeds:
                    d.set_message_buffer(bufs[k % (ahead + 1)])
                    d.feed_resident(n)
                if k >= ahead:
                    j = k - ahead
                    msgs, cnt = d.collect_feed(bufs[j % (ahead + 1)], want_counters=(j == feeds - 1))
                    got.append(msgs.tobytes())
            d.set_deferred(False)
        finally:
            d.close()
        cnt = repr({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in cnt.items()})
        if ref is None:
            ref = (got, cnt)
            assert sum(len(g) for g in got) // 64 > 100000 and got[0] != got[1]      # (the stream goes on: later feeds are not the first again)
        else:
            assert [i for i, (a, b) in enumerate(zip(ref[0], got)) if a != b] == [], (chunk_buffers, ahead)
            assert cnt == ref[1], (chunk_buffers, ahead)
