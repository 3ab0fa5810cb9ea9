"""Frames and wrappers for the field-decode tests (oracle vs reference on the CPU, GPU vs oracle under -m gpu)."""
import ctypes as C
import os

import numpy as np

import helpers

FIELDS = np.dtype([
    ("addr", "<u4"), ("AA", "<u4"), ("flags", "<u4"), ("acc_flags", "<u2"), ("nav_flags", "u1"), ("msgtype", "u1"),
    ("addrtype", "u1"), ("source", "u1"), ("airground", "u1"), ("metype", "u1"),
    ("mesub", "u1"), ("CA", "u1"), ("CC", "u1"), ("CF", "u1"),
    ("DR", "u1"), ("FS", "u1"), ("KE", "u1"), ("ND", "u1"),
    ("RI", "u1"), ("SL", "u1"), ("UM", "u1"), ("VS", "u1"),
    ("IID", "u1"), ("category", "u1"), ("emergency", "u1"), ("cpr_type", "u1"),
    ("AC", "<u2"), ("ID", "<u2"), ("squawkHex", "<u2"), ("squawkDec", "<u2"),
    ("baro_alt", "<i4"), ("geom_alt", "<i4"), ("geom_delta", "<i4"), ("baro_rate", "<i4"), ("geom_rate", "<i4"),
    ("ias", "<u2"), ("tas", "<u2"),
    ("heading", "<f4"), ("gs_v0", "<f4"), ("gs_v2", "<f4"), ("gs_selected", "<f4"),
    ("cpr_lat", "<u4"), ("cpr_lon", "<u4"), ("callsign", "S8"),
    ("baro_alt_unit", "u1"), ("geom_alt_unit", "u1"), ("heading_type", "u1"), ("sil_type", "u1"),
    ("nac_p", "u1"), ("nac_v", "u1"), ("sil", "u1"), ("gva", "u1"),
    ("sda", "u1"), ("op_version", "u1"), ("op_hrd", "u1"), ("op_tah", "u1"),
    ("op_flags", "<u2"), ("op_cc_lw", "u1"), ("op_cc_antenna_offset", "u1"),
    ("op_cc_tc", "u1"), ("nav_heading_type", "u1"), ("nav_altitude_source", "u1"), ("nav_modes", "u1"),
    ("nav_fms_altitude", "<u4"), ("nav_mcp_altitude", "<u4"), ("nav_qnh", "<f4"), ("nav_heading", "<f4"),
    ("roll", "<f4"), ("track_rate", "<f4"), ("mach", "<f4"), ("oat", "<f4"), ("humidity", "<f4"), ("wind_direction", "<f4"),
    ("wind_speed", "<u2"), ("static_pressure", "<u2"), ("commb_format", "u1"), ("met_source", "u1"), ("turbulence", "u1"), ("pad0", "u1"),
    ("reserved", "u1", 8),
])
assert FIELDS.itemsize == 176

def _raw(a):
    return np.ascontiguousarray(a).view(np.uint8).reshape(len(a), -1)


def checksum(frames, bits):
    lib = helpers.oracle_lib()
    out = np.empty(len(frames), dtype=np.uint32)
    for i in range(len(frames)):
        out[i] = lib.modes_oracle_checksum(frames[i].ctypes.data, int(bits[i]))
    return out


def crc24_vec(frames, nbytes):
    """CRC-24 (Mode S generator 0x1FFF409) of the first nbytes-3 bytes XOR the last three, vectorised: the syndrome
    modesChecksum returns (crc.c:67-82).  Own bitwise implementation, checked against the oracle's table version in the tests."""
    rem = np.zeros(len(frames), dtype=np.uint32)
    for k in range(nbytes - 3):
        byte = frames[:, k].astype(np.uint32)
        for b in range(7, -1, -1):
            top = ((rem >> 23) & 1) ^ ((byte >> b) & 1)
            rem = ((rem << 1) & 0xFFFFFF) ^ (top * np.uint32(0xFFF409))
    tail = (frames[:, nbytes - 3].astype(np.uint32) << 16) | (frames[:, nbytes - 2].astype(np.uint32) << 8) | frames[:, nbytes - 1]
    return rem ^ tail


def seal(frames, iid=None):
    """Make the PI field of DF11/17/18 frames consistent (syndrome 0, or the interrogator id for DF11)."""
    df = frames[:, 0] >> 3
    for nbytes, sel in ((7, df == 11), (14, (df == 17) | (df == 18))):
        idx = np.nonzero(sel)[0]
        if not len(idx):
            continue
        sub = frames[idx].copy()
        sub[:, nbytes - 3:nbytes] = 0
        syn = crc24_vec(sub, nbytes)
        if iid is not None and nbytes == 7:
            syn = syn ^ iid[idx].astype(np.uint32)
        sub[:, nbytes - 3] = syn >> 16
        sub[:, nbytes - 2] = (syn >> 8) & 0xFF
        sub[:, nbytes - 1] = syn & 0xFF
        frames[idx] = sub
    return frames


def fuzz_frames(n, seed, dfs=(0, 4, 5, 11, 16, 17, 18, 20, 21, 24, 27, 31)):
    """Random frames of every downlink format the decoder handles, ME type and CF uniform, parity sealed.
    Half of the ES frames get the sparse payload real traffic has (zero fields switch whole branches)."""
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, size=(n, 14), dtype=np.uint8)
    df = rng.choice(np.array(dfs, dtype=np.uint8), size=n)
    low3 = rng.integers(0, 8, size=n, dtype=np.uint8)
    frames[:, 0] = (df << 3) | low3
    es = (df == 17) | (df == 18)
    metype = rng.integers(0, 32, size=n, dtype=np.uint8)
    frames[es, 4] = (metype[es] << 3) | (frames[es, 4] & 7)
    sparse = es & (rng.random(n) < 0.5)
    mask = rng.integers(0, 256, size=(n, 6), dtype=np.uint8) & rng.integers(0, 256, size=(n, 6), dtype=np.uint8)
    frames[sparse, 5:11] &= mask[sparse]
    short = (df < 16)
    frames[short, 7:] = 0
    iid = np.where(rng.random(n) < 0.5, 0, rng.integers(0, 128, size=n)).astype(np.uint8)
    seal(frames, iid)
    bits = np.where(short, 56, 112).astype(np.int32)
    return frames, bits


def velocity_frames(subtype):
    """Every (E/W, N/S) velocity pair of ES type 19 subtype 1 or 2: 4 sign combinations x 1023^2 magnitudes."""
    mags = np.arange(1, 1024, dtype=np.uint64)
    ew, ns = np.meshgrid(mags, mags, indexing="ij")
    ew, ns = ew.reshape(-1), ns.reshape(-1)
    out = []
    for sew in (0, 1):
        for sns in (0, 1):
            me = (np.uint64(19) << np.uint64(51)) | (np.uint64(subtype) << np.uint64(48)) | (np.uint64(sew) << np.uint64(42)) | (ew << np.uint64(32)) \
                | (np.uint64(sns) << np.uint64(31)) | (ns << np.uint64(21))
            fr = np.zeros((len(me), 14), dtype=np.uint8)
            fr[:, 0] = 17 << 3 | 5
            fr[:, 1:4] = (0x4B, 0x17, 0x2A)
            for k in range(7):
                fr[:, 4 + k] = (me >> np.uint64(8 * (6 - k))) & np.uint64(0xFF)
            out.append(fr)
    frames = np.concatenate(out)
    return seal(frames), np.full(len(frames), 112, dtype=np.int32)


def altitude_id_frames():
    """Every 13-bit AC / ID code (DF4 / DF5), every 12-bit ES altitude code (DF17 type 11) and every surface movement code."""
    parts, bits = [], []
    codes = np.arange(8192, dtype=np.uint32)
    for df in (4, 5):
        fr = np.zeros((8192, 14), dtype=np.uint8)
        fr[:, 0] = df << 3
        fr[:, 2] = codes >> 8
        fr[:, 3] = codes & 0xFF
        fr[:, 4:7] = (0x12, 0x34, 0x56)
        parts.append(fr); bits.append(np.full(8192, 56, dtype=np.int32))
    ac12 = np.arange(4096, dtype=np.uint32)
    fr = np.zeros((4096, 14), dtype=np.uint8)
    fr[:, 0] = 17 << 3 | 5
    fr[:, 1:4] = (0x40, 0x62, 0x1D)
    fr[:, 4] = 11 << 3
    fr[:, 5] = ac12 >> 4
    fr[:, 6] = (ac12 & 0xF) << 4 | 0x5
    fr[:, 7:11] = (0x12, 0x34, 0x56, 0x78)
    parts.append(seal(fr)); bits.append(np.full(4096, 112, dtype=np.int32))
    mv = np.arange(128, dtype=np.uint32)
    fr = np.zeros((128, 14), dtype=np.uint8)
    fr[:, 0] = 17 << 3 | 4
    fr[:, 1:4] = (0x40, 0x62, 0x1D)
    fr[:, 4] = (6 << 3) | (mv >> 4)
    fr[:, 5] = ((mv & 0xF) << 4) | 0xA
    fr[:, 6:11] = (0x55, 0x12, 0x34, 0x56, 0x78)
    parts.append(seal(fr)); bits.append(np.full(128, 112, dtype=np.int32))
    modea = np.arange(65536, dtype=np.uint32)
    fr = np.zeros((65536, 14), dtype=np.uint8)
    fr[:, 0] = modea >> 8
    fr[:, 1] = modea & 0xFF
    parts.append(fr); bits.append(np.full(65536, 16, dtype=np.int32))
    return np.concatenate(parts), np.concatenate(bits)


def _put(mb, first, last, value):
    """value into bits first..last (1-based from the MSB) of the 56-bit MB words."""
    width = last - first + 1
    v = (np.asarray(value).astype(np.uint64) & np.uint64((1 << width) - 1)) << np.uint64(56 - last)
    return mb | v


def commb_payloads(n, rng):
    """MB fields shaped like the registers decodeCommB recognises (comm_b.c): empty, BDS 1,0 1,7 2,0 3,0 4,0 4,4 5,0 6,0 with
    plausible values, plus noise; a sixth of them get one bit flipped so the reject branches and ties are exercised."""
    kind = rng.integers(0, 10, size=n)
    mb = rng.integers(0, 1 << 56, size=n, dtype=np.uint64)
    U = lambda lo, hi: rng.integers(lo, hi, size=n).astype(np.uint64)      # noqa: E731
    B = lambda p=0.5: (rng.random(n) < p).astype(np.uint64)                 # noqa: E731
    z = np.zeros(n, dtype=np.uint64)
    out = {}
    out[1] = z
    out[2] = _put(_put(rng.integers(0, 1 << 56, size=n, dtype=np.uint64) & np.uint64((1 << 42) - 1), 1, 8, 0x10), 9, 9, B())
    es = np.where(rng.random(n) < 0.45, 0x3F, np.where(rng.random(n) < 0.5, 0x3E, np.where(rng.random(n) < 0.7, 0, U(0, 64))))
    m17 = _put(z, 1, 6, es)
    m17 = _put(m17, 7, 7, B(0.8))
    tt = B(0.6)
    m17 = _put(_put(_put(m17, 16, 16, tt), 24, 24, np.where(rng.random(n) < 0.85, tt, 1 - tt)), 9, 9, B(0.4))
    for b in (10, 11, 12, 13, 14, 20, 21, 22):
        m17 = _put(m17, b, b, B(0.08))
    out[4] = m17
    chars = np.array([1, 2, 3, 4, 11, 12, 19, 20, 26, 32, 48, 49, 50, 51, 55, 57, 45, 46, 0, 33], dtype=np.uint64)
    m20 = _put(z, 1, 8, 0x20)
    for k in range(8):
        c = np.where(rng.random(n) < 0.97, chars[rng.integers(0, len(chars), size=n)], U(0, 64))
        m20 = _put(m20, 9 + 6 * k, 14 + 6 * k, c)
    out[3] = m20
    out[5] = _put(rng.integers(0, 1 << 56, size=n, dtype=np.uint64) & np.uint64((1 << 48) - 1), 1, 8, 0x30)
    mcp_v, fms_v, baro_v = B(0.8), B(0.5), B(0.7)
    alt = (U(2, 90) * np.uint64(500) + np.where(rng.random(n) < 0.8, 0, U(0, 500))) // np.uint64(16)
    alt = np.where(rng.random(n) < 0.9, alt, U(0, 4096))
    alt2 = np.where(rng.random(n) < 0.7, alt, (U(2, 90) * np.uint64(500)) // np.uint64(16))
    m40 = _put(_put(z, 1, 1, mcp_v), 2, 13, np.where((mcp_v == 1) | (rng.random(n) < 0.05), alt, 0))
    m40 = _put(_put(m40, 14, 14, fms_v), 15, 26, np.where((fms_v == 1) | (rng.random(n) < 0.05), alt2, 0))
    m40 = _put(_put(m40, 27, 27, baro_v), 28, 39, np.where(baro_v == 1, np.where(rng.random(n) < 0.9, U(900, 3100), U(0, 4096)), 0))
    mode_v, src_v = B(), B()
    m40 = _put(_put(m40, 48, 48, mode_v), 49, 51, np.where((mode_v == 1) | (rng.random(n) < 0.05), U(0, 8), 0))
    m40 = _put(_put(m40, 54, 54, src_v), 55, 56, np.where((src_v == 1) | (rng.random(n) < 0.05), U(0, 4), 0))
    m40 = _put(m40, 40, 47, np.where(rng.random(n) < 0.95, 0, U(0, 256)))
    out[6] = m40
    m50 = _put(_put(_put(z, 1, 1, B(0.95)), 2, 2, B()), 3, 11, np.where(rng.random(n) < 0.85, np.where(rng.random(n) < 0.5, U(0, 228), U(285, 512)), U(0, 512)))
    m50 = _put(_put(_put(m50, 12, 12, B(0.95)), 13, 13, B()), 14, 23, U(0, 1024))
    m50 = _put(_put(m50, 24, 24, B(0.95)), 25, 34, np.where(rng.random(n) < 0.9, U(20, 360), U(0, 1024)))
    rate_v = B(0.8)
    m50 = _put(_put(_put(m50, 35, 35, rate_v), 36, 36, np.where(rate_v == 1, B(), B(0.05))), 37, 45,
               np.where(rate_v == 1, np.where(rng.random(n) < 0.9, np.where(rng.random(n) < 0.5, U(0, 200), U(330, 512)), U(0, 512)), np.where(rng.random(n) < 0.05, U(0, 512), 0)))
    m50 = _put(_put(m50, 46, 46, B(0.95)), 47, 56, np.where(rng.random(n) < 0.9, U(20, 360), U(0, 1024)))
    out[7] = m50
    m60 = _put(_put(_put(z, 1, 1, B(0.95)), 2, 2, B()), 3, 12, U(0, 1024))
    m60 = _put(_put(m60, 13, 13, B(0.95)), 14, 23, np.where(rng.random(n) < 0.9, U(45, 710), U(0, 1024)))
    m60 = _put(_put(m60, 24, 24, B(0.95)), 25, 34, np.where(rng.random(n) < 0.9, U(20, 230), U(0, 1024)))
    for v, sgn, lo, hi in ((35, 36, 37, 45), (46, 47, 48, 56)):
        rv = B(0.7)
        m60 = _put(_put(_put(m60, v, v, rv), sgn, sgn, B()), lo, hi,
                   np.where(rv == 1, np.where(rng.random(n) < 0.85, np.where(rng.random(n) < 0.5, U(0, 190), U(322, 512)), U(0, 512)), np.where(rng.random(n) < 0.05, U(0, 512), 0)))
    out[8] = m60
    m44 = _put(_put(_put(z, 1, 4, np.where(rng.random(n) < 0.9, U(0, 7), U(0, 16))), 5, 5, B(0.7)), 6, 14, U(0, 512))
    m44 = _put(_put(_put(m44, 15, 23, U(0, 512)), 24, 24, B()), 25, 34, np.where(rng.random(n) < 0.8, np.where(rng.random(n) < 0.5, U(0, 513), U(512, 1024)), U(0, 1024)))
    m44 = _put(_put(m44, 35, 35, B(0.1)), 36, 46, U(0, 2048))
    m44 = _put(_put(_put(_put(m44, 47, 47, B()), 48, 49, U(0, 4)), 50, 50, B()), 51, 56, U(0, 64))
    out[9] = m44
    for k, v in out.items():
        mb = np.where(kind == k, v, mb)
    flip = rng.random(n) < 0.16
    mb = np.where(flip, mb ^ (np.uint64(1) << rng.integers(0, 56, size=n).astype(np.uint64)), mb)
    return mb


def commb_frames(n, seed):
    """DF20/21 frames around commb_payloads (DR = UM = 0 for most: decodeCommB ignores the others)."""
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, size=(n, 14), dtype=np.uint8)
    df = rng.choice(np.array([20, 21], dtype=np.uint8), size=n)
    frames[:, 0] = (df << 3) | rng.integers(0, 8, size=n, dtype=np.uint8)
    clean = rng.random(n) < 0.9
    frames[clean, 1] = 0
    frames[clean, 2] &= 0x1F
    mb = commb_payloads(n, rng)
    for k in range(7):
        frames[:, 4 + k] = (mb >> np.uint64(8 * (6 - k))) & np.uint64(0xFF)
    return frames, np.full(n, 112, dtype=np.int32)


def turn_rate_frames():
    """BDS5,0 reports around the one transcendental of comm_b.c (tan of the roll angle, comm_b.c:632): every roll code,
    a grid of airspeeds and every track-rate code, so that both sides of the 2 deg/s consistency threshold are hit."""
    roll = np.arange(1024, dtype=np.uint64)                      # sign + 9 bits
    tas = np.array([25, 26, 40, 60, 90, 125, 170, 220, 280, 350], dtype=np.uint64)
    rate = np.arange(1024, dtype=np.uint64)                      # sign + 9 bits
    r, t, q = np.meshgrid(roll, tas, rate, indexing="ij")
    r, t, q = r.reshape(-1), t.reshape(-1), q.reshape(-1)
    z = np.zeros(len(r), dtype=np.uint64)
    mb = _put(_put(z, 1, 1, 1), 2, 11, r)
    mb = _put(_put(mb, 12, 12, 1), 13, 23, 300)
    mb = _put(_put(mb, 24, 24, 1), 25, 34, t)
    mb = _put(_put(mb, 35, 35, 1), 36, 45, q)
    mb = _put(_put(mb, 46, 46, 1), 47, 56, t)
    frames = np.zeros((len(mb), 14), dtype=np.uint8)
    frames[:, 0] = 20 << 3
    frames[:, 2:4] = (0x01, 0x23)
    for k in range(7):
        frames[:, 4 + k] = (mb >> np.uint64(8 * (6 - k))) & np.uint64(0xFF)
    frames[:, 11:14] = (0x11, 0x22, 0x33)
    return frames, np.full(len(frames), 112, dtype=np.int32)


def oracle_fields(frames, bits):
    lib = helpers.oracle_lib()
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    bits = np.ascontiguousarray(bits, dtype=np.int32)
    out = np.zeros(len(frames), dtype=FIELDS)
    lib.modes_oracle_decode_fields_batch(C.c_void_p(frames.ctypes.data), C.c_void_p(bits.ctypes.data), C.c_uint64(len(frames)), C.c_void_p(out.ctypes.data))
    return out


_ref = None


def ref_fields(frames, bits, nfix=1):
    """The reference's own decodeModesMessage / decodeModeAMessage (oracle/_ref/libreadsb_ref.so)."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(helpers.ORACLE_DIR, "_ref", "libreadsb_ref.so"))
        assert _ref.ref_fields_init(nfix) == 0
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    bits = np.ascontiguousarray(bits, dtype=np.int32)
    out = np.zeros(len(frames), dtype=FIELDS)
    rc = np.zeros(len(frames), dtype=np.int32)
    _ref.ref_decode_fields_batch(C.c_void_p(frames.ctypes.data), C.c_void_p(bits.ctypes.data), C.c_uint64(len(frames)),
                                 C.c_void_p(out.ctypes.data), C.c_void_p(rc.ctypes.data))
    return out, rc


def assert_same_fields(got, want, frames, what=""):
    """Byte-identical records."""
    assert len(got) == len(want)
    a, b = _raw(got), _raw(want)
    bad = np.nonzero((a != b).any(axis=1))[0]
    if len(bad):
        idx = bad[0]
        diff = [n for n in FIELDS.names if not np.array_equal(got[idx][n], want[idx][n])]
        raise AssertionError(f"{what}: {len(bad)} of {len(a)} records differ; first at {idx} frame {bytes(frames[idx]).hex()} fields {diff}: "
                             + ", ".join(f"{n}: got {got[idx][n]!r} want {want[idx][n]!r}" for n in diff))
