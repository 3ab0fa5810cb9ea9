"""-m gpu: the whole chain on the device — IQ in HBM -> magnitudes -> sweep -> slicer -> (host: ordered walk only) -> message
records built in HBM (k_build_messages) -> per-message field decode (k_decode_fields) and beast wire frames (k_beast_*) over
those records where they are.  No per-message loop runs on the host (the reference's: netUseMessage / decodeModesMessage /
modesSendBeastOutput, net_io.c:1655-1714, mode_s.c:598-803); field records and beast stream must equal the oracle's, feed by feed."""
import ctypes as C

import numpy as np
import pytest

import fields_util as fu
import helpers

pytestmark = pytest.mark.gpu
B = 131072


def test_messages_fields_and_beast_frames_without_leaving_the_device(built, monkeypatch):
    import readsb_amd
    monkeypatch.setattr(readsb_amd.binding, "DEFAULT_CHUNK_BUFFERS", 16)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    sizes = [40 * B, 35 * B + 17]
    iq = helpers.synth(nsamples=sum(sizes), seed=777, rate=3000.0)
    want, _ = helpers.oracle_run(iq, 0, 1, 1, 58)
    lib = helpers.oracle_lib()
    lib.modes_oracle_beast_frame.restype = C.c_size_t
    lib.modes_oracle_beast_frame.argtypes = [C.c_void_p, C.c_void_p]

    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=64 * B)
    d.set_deferred(True)
    d.set_device_messages(True)
    off, first = 0, 0
    for n in sizes:
        d.feed_iq(iq[off * 2:(off + n) * 2])
        off += n
    for _ in sizes:
        ptr, cnt, _ = d.collect_feed_device()
        ref = want[first:first + cnt]
        first += cnt
        # field records
        d_fields = C.c_void_p()
        assert hip.hipMalloc(C.byref(d_fields), max(1, cnt) * 176) == 0
        d.decode_fields_device(ptr, cnt, d_fields.value)
        got_fields = np.empty(cnt, dtype=readsb_amd.FIELDS_DTYPE)
        assert hip.hipMemcpy(got_fields.ctypes.data, d_fields, cnt * 176, 2) == 0
        want_fields = fu.oracle_fields(np.ascontiguousarray(ref["msg"]), ref["msgbits"].astype(np.int32))
        assert got_fields.tobytes() == want_fields.tobytes()
        # beast frames
        cap = cnt * 48 + 64
        d_out = C.c_void_p()
        assert hip.hipMalloc(C.byref(d_out), cap) == 0
        nb = d.beast_encode_device(ptr, cnt, d_out.value, cap)
        got = np.empty(nb, dtype=np.uint8)
        assert hip.hipMemcpy(got.ctypes.data, d_out, nb, 2) == 0
        frame, stream = (C.c_uint8 * 64)(), bytearray()
        for k in range(len(ref)):
            stream += bytes(frame[: lib.modes_oracle_beast_frame(ref[k:k + 1].ctypes.data, frame)])
        assert got.tobytes() == bytes(stream)
        hip.hipFree(d_fields)
        hip.hipFree(d_out)
    assert first == len(want)
    d.close()
