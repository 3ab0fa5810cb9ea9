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
