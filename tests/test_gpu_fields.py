"""-m gpu: the per-message field decode on the GPU (k_decode_fields through mgpu_decode_fields) against the restated
decode, which tests/test_oracle_fields.py pins against the reference's decodeModesMessage.  Byte-identical 176-byte
records: fuzz over every downlink format / ME type, every altitude / identity / movement / Mode A code, EVERY
velocity pair (the only transcendental on the path: atan2), and the messages of a demodulated stream."""
import numpy as np
import pytest

import fields_util as fu
import helpers

pytestmark = pytest.mark.gpu


def _records(frames, bits):
    import readsb_amd
    m = np.zeros(len(frames), dtype=readsb_amd.MSG_DTYPE)
    m["msg"] = frames
    m["msgbits"] = bits
    df = frames[:, 0] >> 3
    m["msgtype"] = np.where(bits == 16, 77, df)
    aa = (frames[:, 1].astype(np.uint32) << 16) | (frames[:, 2].astype(np.uint32) << 8) | frames[:, 3]
    syn = np.where(bits == 56, fu.crc24_vec(frames, 7), fu.crc24_vec(frames, 14))
    has_aa = (df == 11) | (df == 17) | (df == 18)
    m["addr"] = np.where(bits == 16, 0, np.where(has_aa, aa, syn))
    return m


@pytest.fixture(scope="module")
def demod(built):
    import readsb_amd
    d = readsb_amd.Demodulator(startup_time_ms=helpers.STARTUP_MS, max_samples=1 << 20)
    yield d
    d.close()


def _check(demod, frames, bits, what):
    got = demod.decode_fields(_records(frames, bits)).view(fu.FIELDS)
    want = fu.oracle_fields(frames, bits)
    a, b = got.view(np.uint8).reshape(len(got), -1), want.view(np.uint8).reshape(len(want), -1)
    bad = np.nonzero((a != b).any(axis=1))[0]
    if len(bad):
        k = bad[0]
        diff = [n for n in fu.FIELDS.names if not np.array_equal(got[k][n], want[k][n])]
        raise AssertionError(f"{what}: {len(bad)} of {len(a)} records differ; first at {k} frame {bytes(frames[k]).hex()} bits {bits[k]}: "
                             + ", ".join(f"{n}: got {got[k][n]!r} want {want[k][n]!r}" for n in diff))
    return want


def test_fuzz_every_format(demod):
    frames, bits = fu.fuzz_frames(1000000, 77)
    want = _check(demod, frames, bits, "fuzz")
    assert set(np.unique(want[want["msgtype"] == 17]["metype"])) == set(range(32))
    assert demod.decode_fields(_records(frames[:0], bits[:0])).size == 0
    for n in (1, 255, 256, 257):                       # ragged tails of the 256-message workgroups
        _check(demod, frames[:n], bits[:n], f"ragged {n}")


def test_comm_b_registers(demod):
    frames, bits = fu.commb_frames(1000000, 41)
    want = _check(demod, frames, bits, "comm-b")
    assert set(np.unique(want["commb_format"])) == set(range(11))


def test_comm_b_turn_rate_threshold(demod):
    """Every roll code x every track-rate code on a grid of airspeeds: the tan() behind BDS5,0's consistency penalty."""
    frames, bits = fu.turn_rate_frames()
    want = _check(demod, frames, bits, "bds5,0 turn rate")
    assert (want["commb_format"] == 8).sum() > 1000000


def test_every_code(demod):
    frames, bits = fu.altitude_id_frames()
    want = _check(demod, frames, bits, "codes")
    assert (want["msgtype"] == 77).sum() == 65536


@pytest.mark.parametrize("subtype", [1, 2])
def test_every_velocity_pair(demod, subtype):
    frames, bits = fu.velocity_frames(subtype)
    want = _check(demod, frames, bits, f"velocity subtype {subtype}")
    assert (want["flags"] & 4).all() and len(np.unique(want["heading"])) > 1000000


def test_golden_records(demod):
    import os
    g = np.load(os.path.join(helpers.GOLDEN_DIR, "fields_fuzz_2000.npz"))
    frames, bits, want = g["frames"], g["bits"], g["fields"].view(fu.FIELDS).reshape(-1)
    got = demod.decode_fields(_records(frames, bits)).view(fu.FIELDS)
    fu.assert_same_fields(got, want, frames, "golden (reference-written)")


def test_stream_messages(built):
    import readsb_amd
    iq = helpers.synth(seconds=3.0, seed=98, rate=700.0, dense=2)
    d = readsb_amd.Demodulator(nfix_crc=2, mode_ac=1, startup_time_ms=helpers.STARTUP_MS, max_samples=len(iq) // 2)
    try:
        msgs, _ = d.demodulate_capture(iq)
        got = d.decode_fields(msgs).view(fu.FIELDS)
    finally:
        d.close()
    want_msgs, _ = helpers.oracle_run(iq, 0, 2, 1, 58, mode_ac=1)
    assert len(msgs) == len(want_msgs) and (msgs["msgbits"] == 16).any()
    want = fu.oracle_fields(np.ascontiguousarray(want_msgs["msg"]), want_msgs["msgbits"].astype(np.int32))
    assert got.tobytes() == want.tobytes()
