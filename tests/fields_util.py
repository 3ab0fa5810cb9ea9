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
    ("reserved", "u1", 8),
])
assert FIELDS.itemsize == 144

# what mode_s.c itself sets for DF20/21; decodeCommB (comm_b.c, not restated) may touch everything else
DF20_21_SUBSET = ["addr", "AA", "msgtype", "addrtype", "source", "DR", "FS", "UM", "AC", "ID", "IID", "CA", "CF", "metype"]
F_COMMB_SAFE = (1 << 14) | (1 << 15) | (1 << 16) | (1 << 17)     # spi / alert bits from FS


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
    """Byte-identical records, except DF20/21 where only the fields mode_s.c itself sets are compared."""
    assert len(got) == len(want)
    commb = (want["msgtype"] == 20) | (want["msgtype"] == 21)
    a, b = _raw(got[~commb]), _raw(want[~commb])
    bad = np.nonzero((a != b).any(axis=1))[0]
    if len(bad):
        k = bad[0]
        idx = np.nonzero(~commb)[0][k]
        diff = [n for n in FIELDS.names if not np.array_equal(got[idx][n], want[idx][n])]
        raise AssertionError(f"{what}: {len(bad)} of {len(a)} records differ; first at {idx} frame {bytes(frames[idx]).hex()} fields {diff}: "
                             + ", ".join(f"{n}: got {got[idx][n]!r} want {want[idx][n]!r}" for n in diff))
    for n in DF20_21_SUBSET:
        assert np.array_equal(got[commb][n], want[commb][n]), f"{what}: DF20/21 field {n}"
