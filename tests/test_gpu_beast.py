"""-m gpu: the beast wire encoder on the GPU against the restated encoder (which tests/test_oracle.py pins against
streams written by the whole reference program): the byte stream of every accepted message, Mode S and Mode A/C."""
import ctypes as C

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _oracle_stream(msgs):
    lib = helpers.oracle_lib()
    lib.modes_oracle_beast_frame.restype = C.c_size_t
    lib.modes_oracle_beast_frame.argtypes = [C.c_void_p, C.c_void_p]
    buf = (C.c_uint8 * 64)()
    out = bytearray()
    for k in range(len(msgs)):
        nb = lib.modes_oracle_beast_frame(msgs[k:k + 1].ctypes.data, buf)
        out += bytes(buf[:nb])
    return bytes(out)


@pytest.mark.parametrize("seconds,rate,dense,nfix,mode_ac,seed", [(2.0, 1500.0, 0, 1, 0, 99), (3.0, 700.0, 2, 2, 1, 98), (20.0, 3000.0, 0, 1, 0, 5)])
def test_beast_stream(built, seconds, rate, dense, nfix, mode_ac, seed):
    import readsb_amd
    iq = helpers.synth(seconds=seconds, seed=seed, rate=rate, dense=dense, threads=8)
    want_msgs, _ = helpers.oracle_run(iq, 0, nfix, 1, 58, mode_ac=mode_ac)
    want = _oracle_stream(want_msgs)
    d = readsb_amd.Demodulator(nfix_crc=nfix, mode_ac=mode_ac, startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2)
    try:
        got_msgs, _ = d.demodulate_capture(iq)
        got = d.beast_encode(got_msgs)
        assert d.beast_encode(got_msgs[:0]) == b""
    finally:
        d.close()
    assert len(want) > 20000 and b"\x1a\x1a" in want[2:]
    assert got == want


def _splice(stream, deferred, frames):
    """The gated stream with the frames of the deferred messages the tracker forwards put in at their offsets (frames: index -> bytes)."""
    out, at = bytearray(), 0
    for e in deferred:
        out += stream[at:int(e["offset"])]
        at = int(e["offset"])
        out += frames.get(int(e["index"]), b"")
    out += stream[at:]
    return bytes(out)


@pytest.mark.parametrize("name", ["uc8_fix_2s", "uc8_aggressive_modeac_3s"])
def test_gated_stream_is_the_reference_programs(built, name):
    """§8(f).4 as a component: IQ -> messages -> tracking gate -> beast encoder on the GPU.  The stream of the certainly-forwarded
    messages, with the few deferred ones settled — here by what the WHOLE reference program did (tests/golden/gate_*.npz) — and
    spliced in at the offsets the call reports, is byte for byte the file the reference program wrote with --dump-beast
    (tests/golden/beast_*.bin): first messages of an aircraft suppressed (net_io.c:5846-5849), everything else in stream order."""
    import os
    import gate_util as gu
    import readsb_amd
    kw, opt = gu.CASES[name]
    iq = helpers.synth(threads=8, **kw)
    gold = open(os.path.join(helpers.GOLDEN_DIR, f"beast_{name}.bin"), "rb").read()
    fwd = gu.golden_forwarded(name)
    d = readsb_amd.Demodulator(nfix_crc=opt["nfix"], mode_ac=opt["mode_ac"], startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2)
    try:
        msgs, _ = d.demodulate_capture(iq)
        assert len(msgs) == len(fwd)
        stream, deferred = d.beast_encode_gated(msgs)
        assert len(deferred) <= 0.05 * len(msgs) and (np.diff(deferred["index"].astype(np.int64)) > 0).all() and (np.diff(deferred["offset"].astype(np.int64)) >= 0).all()
        frames = {int(i): d.beast_encode(msgs[int(i):int(i) + 1]) for i in deferred["index"] if fwd[int(i)]}
        assert _splice(stream, deferred, frames) == gold
        # the ungated encoder writes every accepted message: more than the reference does
        assert len(d.beast_encode(msgs)) > len(gold)
        # the network outputs' rule on top (correctedbits < 2, net_io.c:5863-5872): the same stream without those frames
        d.track_gate_reset()
        net_stream, net_def = d.beast_encode_gated(msgs, net_rule=True)
        keep = msgs["correctedbits"] < 2
        v = gu.oracle_gate(*gu.oracle_messages(name)[1:])
        want = b"".join(d.beast_encode(msgs[k:k + 1]) for k in np.nonzero(((v & 3) == 1) & keep)[0])
        assert net_stream == want and set(net_def["index"]) == set(np.nonzero(((v & 3) == 2) & keep)[0])
        # in calls of a few buffers each the aircraft table carries over: the same bytes
        d.track_gate_reset()
        buf = ((msgs["timestamp"].astype(np.int64) - 772) // 5) // gu.BUF
        cuts = [int(np.searchsorted(buf, b)) for b in range(0, int(buf[-1]) + 1, 5)] + [len(msgs)]
        parts = [d.beast_encode_gated(msgs[a:b])[0] for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        assert b"".join(parts) == stream
        with pytest.raises(readsb_amd.MgpuError):
            d.beast_encode_gated(msgs, deferred_cap=0 if len(deferred) else None) if len(deferred) else (_ for _ in ()).throw(readsb_amd.MgpuError("no deferred message in this capture"))
    finally:
        d.close()


def test_beast_stream_in_device_memory(built):
    """The aggregator's case: records already in HBM, stream written to HBM (plain HIP allocations through the
    runtime the library itself is linked against)."""
    import readsb_amd
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    iq = helpers.synth(seconds=2.0, seed=99, rate=1500.0)
    want_msgs, _ = helpers.oracle_run(iq)
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2)
    d_in, d_out = C.c_void_p(), C.c_void_p()
    try:
        msgs, _ = d.demodulate_capture(iq)
        msgs = np.ascontiguousarray(msgs)
        cap = len(msgs) * 44
        assert hip.hipMalloc(C.byref(d_in), msgs.nbytes) == 0 and hip.hipMalloc(C.byref(d_out), cap) == 0
        assert hip.hipMemcpy(d_in, msgs.ctypes.data, msgs.nbytes, 1) == 0
        nb = d.beast_encode_device(d_in.value, len(msgs), d_out.value, cap)
        got = np.empty(nb, dtype=np.uint8)
        assert hip.hipMemcpy(got.ctypes.data, d_out, nb, 2) == 0
        # a buffer that is too small is reported, with the size it would have needed
        with pytest.raises(RuntimeError):
            d.beast_encode_device(d_in.value, len(msgs), d_out.value, nb - 1)
    finally:
        hip.hipFree(d_in), hip.hipFree(d_out)
        d.close()
    assert got.tobytes() == _oracle_stream(want_msgs)
