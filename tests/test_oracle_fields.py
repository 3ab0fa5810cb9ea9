"""CPU: the restated field decode (oracle/modes_oracle_fields.c) pinned against the reference's own decodeModesMessage /
decodeModeAMessage (oracle/_ref, compiled in place from /root/reference) and against the committed golden records."""
import os

import numpy as np
import pytest

import fields_util as fu
import helpers

needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(helpers.ORACLE_DIR, "_ref", "libreadsb_ref.so")), reason="oracle/_ref not built")
GOLDEN = os.path.join(helpers.GOLDEN_DIR, "fields_fuzz_2000.npz")


def test_struct_sizes():
    assert fu.FIELDS.itemsize == 176


def test_vector_crc_is_the_oracle_crc():
    frames, bits = fu.fuzz_frames(500, 3)
    want = fu.checksum(frames, bits)
    got = np.where(bits == 56, fu.crc24_vec(frames, 7), fu.crc24_vec(frames, 14))
    assert np.array_equal(got, want)


@needs_ref
def test_fuzz_every_format_against_reference():
    frames, bits = fu.fuzz_frames(400000, 11)
    want, rc = fu.ref_fields(frames, bits)
    assert (rc == 0).all()
    got = fu.oracle_fields(frames, bits)
    fu.assert_same_fields(got, want, frames, "fuzz")
    # the fuzz reaches every ME decoder
    es = want[(want["msgtype"] == 17)]
    assert set(np.unique(es["metype"])) == set(range(32))
    assert (es["flags"] & (1 << 9)).any() and (es["nav_flags"] != 0).any() and (es["op_flags"] != 0).any()


@needs_ref
def test_comm_b_registers_against_reference():
    frames, bits = fu.commb_frames(600000, 23)
    want, rc = fu.ref_fields(frames, bits)
    assert (rc == 0).all()
    fu.assert_same_fields(fu.oracle_fields(frames, bits), want, frames, "comm-b")
    # every register hypothesis wins somewhere, and so do "ambiguous" and "unknown" (commb_format_t, readsb.h:249)
    assert set(np.unique(want["commb_format"])) == set(range(11))


@needs_ref
def test_comm_b_turn_rate_threshold_against_reference():
    frames, bits = fu.turn_rate_frames()
    want, rc = fu.ref_fields(frames, bits)
    assert (rc == 0).all()
    fu.assert_same_fields(fu.oracle_fields(frames, bits), want, frames, "bds5,0 turn rate")
    assert (want["commb_format"] == 8).sum() > 1000000


@needs_ref
def test_exhaustive_codes_against_reference():
    frames, bits = fu.altitude_id_frames()
    want, rc = fu.ref_fields(frames, bits)
    assert (rc == 0).all()
    fu.assert_same_fields(fu.oracle_fields(frames, bits), want, frames, "codes")


@needs_ref
@pytest.mark.parametrize("subtype", [1, 2])
def test_every_velocity_pair_against_reference(subtype):
    frames, bits = fu.velocity_frames(subtype)
    want, rc = fu.ref_fields(frames, bits)
    assert (rc == 0).all() and (want["flags"] & 4).all()
    fu.assert_same_fields(fu.oracle_fields(frames, bits), want, frames, f"velocity subtype {subtype}")


@needs_ref
def test_stream_messages_against_reference():
    iq = helpers.synth(seconds=3.0, seed=98, rate=700.0, dense=2)
    msgs, _ = helpers.oracle_run(iq, 0, 2, 1, 58, mode_ac=1)
    frames = np.ascontiguousarray(msgs["msg"])
    bits = msgs["msgbits"].astype(np.int32)
    assert (bits == 16).any() and (bits == 112).any()
    want, rc = fu.ref_fields(frames, bits)
    assert (rc == 0).all()
    got = fu.oracle_fields(frames, bits)
    fu.assert_same_fields(got, want, frames, "stream")
    assert np.array_equal(got["addr"] & 0xFFFFFF, msgs["addr"] & 0xFFFFFF)


def test_golden_records():
    """Records written by the reference (tests/golden/make_fields_golden.py) — runs without oracle/_ref."""
    g = np.load(GOLDEN)
    frames, bits, want = g["frames"], g["bits"], g["fields"].view(fu.FIELDS).reshape(-1)
    fu.assert_same_fields(fu.oracle_fields(frames, bits), want, frames, "golden")
