"""-m gpu: the first-stage tracking gate on the device (kernels/gate.inc) — every verdict it calls certain equals
what the WHOLE reference program forwarded (tests/golden/gate_*.npz, written from its --dump-beast streams), the deferred share is
bounded, and all of it equals the CPU restatement (oracle/modes_oracle_gate.c) byte for byte, also when the message list arrives in
several calls."""
import numpy as np
import pytest

import gate_util as gu
import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,max_deferred", [("uc8_fix_2s", 0.05), ("uc8_aggressive_modeac_3s", 0.05), ("uc8_fix_200ac_60s", 0.005),
                                               ("uc8_fix_30000ac_130s", 0.12)])
def test_gate_against_the_reference_program(built, name, max_deferred):
    import readsb_amd
    kw, opt = gu.CASES[name]
    iq, want_msgs, want_fields = gu.oracle_messages(name)
    fwd = gu.golden_forwarded(name)
    d = readsb_amd.Demodulator(nfix_crc=opt["nfix"], mode_ac=opt["mode_ac"], startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2)
    try:
        msgs, _ = d.demodulate_capture(iq)
        helpers.assert_same_messages(msgs, want_msgs)
        assert len(msgs) == len(fwd)
        v = d.track_gate(msgs)
        share = gu.check_against_golden(v, fwd, max_deferred)
        assert np.array_equal(v, gu.oracle_gate(want_msgs, want_fields)), "device verdicts differ from the CPU restatement's"
        # the same list in calls of a few buffers each (the table carries the aircraft over), and everything in device memory
        d.track_gate_reset()
        buf = ((msgs["timestamp"].astype(np.int64) - 772) // 5) // gu.BUF
        cuts = [int(np.searchsorted(buf, b)) for b in range(0, int(buf[-1]) + 1, 7)] + [len(msgs)]
        parts = [d.track_gate(msgs[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        assert np.array_equal(np.concatenate(parts), v)
    finally:
        d.close()
    print(f"{name}: {len(msgs)} messages, deferred share {share:.5f}")


def test_gate_in_device_memory(built):
    """Messages and field records already in HBM (the pipeline's device-resident forms), verdicts written to HBM (plain HIP
    allocations through the runtime the library itself is linked against)."""
    import ctypes as C
    import readsb_amd
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    name = "uc8_aggressive_modeac_3s"
    kw, opt = gu.CASES[name]
    iq, want_msgs, want_fields = gu.oracle_messages(name)
    d = readsb_amd.Demodulator(nfix_crc=opt["nfix"], mode_ac=opt["mode_ac"], startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2)
    d_msgs, d_fields, d_v = C.c_void_p(), C.c_void_p(), C.c_void_p()
    try:
        msgs, _ = d.demodulate_capture(iq)
        msgs = np.ascontiguousarray(msgs)
        n = len(msgs)
        assert hip.hipMalloc(C.byref(d_msgs), msgs.nbytes) == 0 and hip.hipMalloc(C.byref(d_fields), n * readsb_amd.FIELDS_DTYPE.itemsize) == 0
        assert hip.hipMalloc(C.byref(d_v), n) == 0
        assert hip.hipMemcpy(d_msgs, msgs.ctypes.data, msgs.nbytes, 1) == 0
        d.decode_fields_device(d_msgs.value, n, d_fields.value)
        d.track_gate_device(d_msgs.value, d_fields.value, n, d_v.value)
        v = np.empty(n, dtype=np.uint8)
        assert hip.hipMemcpy(v.ctypes.data, d_v, n, 2) == 0
    finally:
        for p in (d_msgs, d_fields, d_v):
            if p.value:
                hip.hipFree(p)
        d.close()
    assert np.array_equal(v, gu.oracle_gate(want_msgs, want_fields))
    gu.check_against_golden(v, gu.golden_forwarded(name), 0.05)


def test_gate_edge_cases(built):
    """An empty list, a lone message, a list of Mode A/C replies only, and the reset: same answers as the CPU restatement."""
    import readsb_amd
    name = "uc8_aggressive_modeac_3s"
    kw, opt = gu.CASES[name]
    iq, want_msgs, want_fields = gu.oracle_messages(name)
    d = readsb_amd.Demodulator(nfix_crc=opt["nfix"], mode_ac=opt["mode_ac"], startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2)
    try:
        msgs, _ = d.demodulate_capture(iq)
        assert len(d.track_gate(msgs[:0])) == 0
        first = d.track_gate(msgs[:1])
        assert np.array_equal(first, gu.oracle_gate(want_msgs[:1], want_fields[:1]))
        d.track_gate_reset()
        ac = msgs["msgtype"] == 77
        assert ac.sum() > 10
        assert (d.track_gate(msgs[ac]) & 3 == 1).all()                       # Mode A/C replies are always forwarded, no aircraft
        d.track_gate_reset()
        a = d.track_gate(msgs)
        b = d.track_gate(msgs)                                                # the same list again: every aircraft is known by now
        d.track_gate_reset()
        c = d.track_gate(msgs)
        assert np.array_equal(a, c) and not np.array_equal(a, b)
        assert ((b & 3) == 2).sum() < ((a & 3) == 2).sum()                    # ... so fewer verdicts depend on the tracker
    finally:
        d.close()
