"""CPU tests: the plain-C restatement (oracle/modes_oracle.c) is pinned against
  (1) the golden fixtures in tests/golden/, produced by the reference's own C files, and
  (2) the reference itself (oracle/_ref), live, wherever it has been built (dev container and,
      because the binaries travel, the GPU box)."""
import hashlib
import json
import os

import numpy as np
import pytest

import helpers

INDEX = json.load(open(os.path.join(helpers.GOLDEN_DIR, "index.json")))
TABLES = np.load(os.path.join(helpers.GOLDEN_DIR, "tables.npz"))


def _stats_match(st, want):
    for f in helpers.COUNTER_FIELDS:
        assert np.asarray(st[f]).tolist() == want[f], f
    for f in ("signal_power_sum", "peak_signal_power", "noise_power_sum"):
        a, b = float(st[f]), float.fromhex(want[f])
        assert (np.isnan(a) and np.isnan(b)) or a == b, f


@pytest.mark.parametrize("name", sorted(INDEX))
def test_restatement_matches_golden(built, name):
    c = INDEX[name]
    iq = helpers.synth(**c["synth"])
    assert hashlib.sha256(iq.tobytes()).hexdigest() == c["iq_sha256"], "synthetic generator drifted: regenerate goldens"
    msgs, st = helpers.oracle_run(iq, c["fmt"], c["nfix"], c["fixdf"], c["thr"])
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, name + ".msgs.npy"))
    assert len(msgs) == c["nmsgs"] == len(gold)
    assert msgs.tobytes() == gold.tobytes()
    _stats_match(st, c["stats"])


@pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("fmt,nfix,fixdf,thr,skw", [
    (0, 1, 1, 58, dict(seconds=2.0, seed=31)),
    (0, 2, 1, 58, dict(seconds=2.0, seed=32, rate=6000.0)),
    (0, 1, 0, 40, dict(seconds=1.0, seed=33)),
    (2, 2, 1, 58, dict(seconds=1.0, seed=34, fmt=2)),
    (1, 1, 1, 58, dict(seconds=1.0, seed=35, fmt=1)),
    (0, 1, 1, 58, dict(nsamples=131072, seed=36)),          # exact multiple: zero-length EOF buffer
    (0, 1, 1, 58, dict(nsamples=1000, seed=37)),
])
def test_restatement_matches_reference_live(built, fmt, nfix, fixdf, thr, skw):
    iq = helpers.synth(**skw)
    a, sa, ma = helpers.oracle_run(iq, fmt, nfix, fixdf, thr, want_mag=True)
    b, sb, mb = helpers.ref_run(iq, fmt, nfix, fixdf, thr, want_mag=True)
    assert a.tobytes() == b.tobytes()
    assert np.array_equal(ma, mb)
    for f in helpers.COUNTER_FIELDS:
        assert np.array_equal(np.asarray(sa[f]), np.asarray(sb[f])), f
    for f in ("signal_power_sum", "peak_signal_power", "noise_power_sum"):
        x, y = float(sa[f]), float(sb[f])
        assert (np.isnan(x) and np.isnan(y)) or x == y


@pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref not built")
def test_filter_expiry_two_minutes(built):
    """130 s stream: three filter flips, aircraft fall silent and are forgotten."""
    iq = helpers.synth(seconds=130.0, seed=11, rate=1500.0, naircraft=300)
    a, sa = helpers.oracle_run(iq)
    b, sb = helpers.ref_run(iq)
    assert sa["nflips"] >= 3
    assert a.tobytes() == b.tobytes()
    for f in helpers.COUNTER_FIELDS:
        assert np.array_equal(np.asarray(sa[f]), np.asarray(sb[f])), f


def test_crc_table_sizes_are_crctests_numbers(built):
    """The only KATs the reference has for crc.c: `crctests 1 1` -> 51/107, `crctests 2 4` -> 1326/3831."""
    lib = helpers.oracle_lib()
    lib.modes_oracle_crc_init(1)
    assert (lib.modes_oracle_table_size(56), lib.modes_oracle_table_size(112)) == (51, 107)
    lib.modes_oracle_crc_init(2)
    assert (lib.modes_oracle_table_size(56), lib.modes_oracle_table_size(112)) == (1326, 3831)


@pytest.mark.parametrize("nfix", [1, 2])
def test_crc_tables_match_reference_dump(built, nfix):
    import ctypes as C
    lib = helpers.oracle_lib()
    lib.modes_oracle_crc_init(nfix)
    for bits in (56, 112):
        gold = TABLES[f"nfix{nfix}_{bits}"]
        assert lib.modes_oracle_table_size(bits) == len(gold)
        for syn, n, b0, b1 in gold[:: max(1, len(gold) // 400)]:
            x, y = C.c_int(), C.c_int()
            assert lib.modes_oracle_diagnose(int(syn), bits, C.byref(x), C.byref(y)) == n
            assert (x.value, y.value) == (b0, b1)
    x, y = C.c_int(), C.c_int()
    assert lib.modes_oracle_diagnose(0x123457, 112, C.byref(x), C.byref(y)) in (-1, 1, 2)


def test_uc8_table_matches_reference(built):
    lut = np.ctypeslib.as_array(helpers.oracle_lib().modes_oracle_uc8_lut(), shape=(65536,))
    gold = TABLES["uc8_mag_by_i_q"]     # [I][Q] from the reference converter
    # table index = I | Q<<8
    assert np.array_equal(lut.reshape(256, 256).T, gold)
    assert np.array_equal(gold, gold.T)  # symmetric


@pytest.mark.skipif(not helpers.have_ref(), reason="reference harness not built (no /root/reference here)")
def test_mode_ac_restatement_equals_reference():
    """demodulate2400AC (demod_2400.c:575-761), enabled after demodulate2400 on every buffer: the restatement and the
    reference's own code agree on Mode S + Mode A/C messages in netUseMessage order, on a stream that carries replies."""
    iq = helpers.synth(seconds=4.0, seed=404, rate=800.0, dense=2)   # dense Mode S traffic would raise the A/C noise floor
    want, wst = helpers.ref_run(iq, mode_ac=1)
    got, gst = helpers.oracle_run(iq, mode_ac=1)
    nac = int((want["msgtype"] == 77).sum())
    assert nac > 1000 and int(wst["demod_modeac"]) == nac
    assert got.tobytes() == want.tobytes()
    assert int(gst["demod_modeac"]) == nac
    # and switching it off gives exactly the Mode S subset
    only_s, _ = helpers.oracle_run(iq, mode_ac=0)
    assert only_s.tobytes() == want[want["msgtype"] != 77].tobytes()


def _beast_frames(stream):
    """Split a beast byte stream into raw frames and their unescaped fields (type, 48-bit timestamp, sig, message)."""
    out, i, n = [], 0, len(stream)
    while i < n:
        assert stream[i] == 0x1A
        j = i + 2
        while j < n and not (stream[j] == 0x1A and (j + 1 >= n or stream[j + 1] != 0x1A)):
            j += 2 if stream[j] == 0x1A else 1
        raw = bytes(stream[i:j])
        body = raw[2:].replace(b"\x1a\x1a", b"\x1a")
        out.append((raw, raw[1], int.from_bytes(body[:6], "big"), body[6], body[7:]))
        i = j
    return out


GOLDEN_BEAST = [("uc8_fix_2s", dict(seconds=2.0, seed=99, rate=1500.0), dict(nfix=1, mode_ac=0)),
                ("uc8_aggressive_modeac_3s", dict(seconds=3.0, seed=98, rate=700.0, dense=2), dict(nfix=2, mode_ac=1))]


@pytest.mark.parametrize("name,synth_kw,opt", GOLDEN_BEAST)
def test_beast_frames_equal_the_reference_programs(name, synth_kw, opt):
    """The golden streams were written by the whole reference program (`--dump-beast`, tests/golden/make_beast_golden.py).
    Every frame in them must be byte-identical to the restated encoder's frame for the message with that timestamp —
    which pins the encoder (timestamp bytes, signal byte from signalLevel, 0x1a escaping) and, on the way, the whole
    message list against the complete reference program rather than the harness."""
    import ctypes as C
    gold = open(os.path.join(helpers.GOLDEN_DIR, f"beast_{name}.bin"), "rb").read()
    frames = _beast_frames(gold)
    iq = helpers.synth(**synth_kw)
    msgs, _ = helpers.oracle_run(iq, 0, opt["nfix"], 1, 58, mode_ac=opt["mode_ac"])
    lib = helpers.oracle_lib()
    lib.modes_oracle_beast_frame.restype = C.c_size_t
    lib.modes_oracle_beast_frame.argtypes = [C.c_void_p, C.c_void_p]
    by_ts = {}
    buf = (C.c_uint8 * 64)()
    for k in range(len(msgs)):
        m = msgs[k:k + 1]
        nb = lib.modes_oracle_beast_frame(m.ctypes.data, buf)
        by_ts.setdefault(int(m["timestamp"][0]) & ((1 << 48) - 1), []).append(bytes(buf[:nb]))
    assert len(frames) > 0.5 * len(msgs) > 500                     # the reference forwards most, not all (first message of an aircraft, ...)
    for raw, typ, ts, sig, body in frames:
        assert ts in by_ts, f"frame at timestamp {ts} has no message in the oracle's list"
        assert raw in by_ts[ts], (raw.hex(), [x.hex() for x in by_ts[ts]])
    assert any(f[1] == ord("1") for f in frames) == bool(opt["mode_ac"])
    assert any(b"\x1a\x1a" in f[0][2:] for f in frames)                 # escaping was exercised
