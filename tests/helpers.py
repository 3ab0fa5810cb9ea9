"""Test infrastructure: builds, the synthetic IQ generator, and the two CPU checkers under
oracle/ (our plain-C restatement, and oracle/_ref = the reference's own C files)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
TOOLS_DIR = os.path.join(ROOT, "tools")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "ref_demod")
STARTUP_MS = 1000000  # ORACLE_STARTUP_MS

FMT_NAMES = {0: "UC8", 1: "SC16", 2: "SC16Q11"}
FMT_BYTES = {0: 2, 1: 4, 2: 4}

# struct oracle_msg (72 bytes), struct oracle_stats
ORACLE_MSG = np.dtype([
    ("timestamp", "<i8"), ("sys_rel_ms", "<i8"), ("score", "<i4"), ("correctedbits", "<i4"), ("msgbits", "<i4"),
    ("msgtype", "<i4"), ("addr", "<u4"), ("msg", "u1", 14), ("raw", "u1", 14), ("signalLevel", "<f8"),
])
assert ORACLE_MSG.itemsize == 72
ORACLE_STATS = np.dtype([
    ("demod_preambles", "<u8"), ("demod_rejected_bad", "<u8"), ("demod_rejected_unknown_icao", "<u8"),
    ("demod_accepted", "<u8", 3), ("demod_preamblePhase", "<u8", 5), ("demod_bestPhase", "<u8", 5),
    ("strong_signal_count", "<u8"), ("signal_power_count", "<u8"), ("noise_power_count", "<u8"),
    ("samples_processed", "<u8"), ("samples_lost", "<u8"), ("nbuffers", "<u8"), ("nflips", "<u8"),
    ("signal_power_sum", "<f8"), ("noise_power_sum", "<f8"), ("peak_signal_power", "<f8"),
    ("t_convert_s", "<f8"), ("t_demod_s", "<f8"), ("demod_modeac", "<u8"),
])
COUNTER_FIELDS = ["demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted",
                  "demod_preamblePhase", "demod_bestPhase", "strong_signal_count", "signal_power_count",
                  "noise_power_count", "samples_processed", "samples_lost", "nbuffers", "nflips"]


def ensure_built():
    """Build whatever is missing (everything is normally built by __graft_entry__.build())."""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build(quiet=True)


_synth = None


def synth(seconds=None, nsamples=None, fmt=0, seed=88172645463325252, rate=2000.0, naircraft=200, dense=0,
          noise=3.0, first=0, threads=8):
    """Seeded synthetic capture (tools/synth_iq.c) as a uint8 array of raw IQ bytes."""
    global _synth
    if _synth is None:
        _synth = C.CDLL(os.path.join(TOOLS_DIR, "libsynth_iq.so"))
        _synth.synth_iq_generate.argtypes = [C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double,
                                             C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
    n = int(nsamples if nsamples is not None else round(seconds * 2400000))
    out = np.empty(n * FMT_BYTES[fmt], dtype=np.uint8)
    _synth.synth_iq_generate(seed, fmt, rate, naircraft, dense, noise, first, n, out.ctypes.data, threads)
    return out


_oracle = None


class OracleCfg(C.Structure):
    _fields_ = [("format", C.c_int), ("nfix_crc", C.c_int), ("fixDF", C.c_int), ("preamble_threshold", C.c_int)]


def oracle_lib():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(os.path.join(ORACLE_DIR, "libmodes_oracle.so"))
        lib.modes_oracle_run.argtypes = [C.POINTER(OracleCfg), C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
        lib.modes_oracle_free.argtypes = [C.c_void_p]
        lib.modes_oracle_convert.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double)]
        lib.modes_oracle_uc8_lut.restype = C.POINTER(C.c_uint16)
        lib.modes_oracle_checksum.argtypes = [C.c_void_p, C.c_int]
        lib.modes_oracle_checksum.restype = C.c_uint32
        lib.modes_oracle_crc_init.argtypes = [C.c_int]
        lib.modes_oracle_diagnose.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.modes_oracle_table_size.argtypes = [C.c_int]
        _oracle = lib
    return _oracle


def oracle_run(iq, fmt=0, nfix=1, fixdf=1, thr=58, want_mag=False, mode_ac=0, filter_clock=0):
    """Our CPU restatement (oracle/modes_oracle.c) on an in-memory capture.  filter_clock 1 = the reference program's other
    start-up order (first ICAO filter flip before buffer 0)."""
    lib = oracle_lib()
    lib.modes_oracle_set_mode_ac(int(mode_ac))
    lib.modes_oracle_set_filter_clock(int(filter_clock))
    iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = iq.size // FMT_BYTES[fmt]
    cfg = OracleCfg(fmt, nfix, fixdf, thr)
    out, nout = C.c_void_p(), C.c_uint64()
    st = np.zeros(1, dtype=ORACLE_STATS)
    mag = np.zeros(n + 326, dtype=np.uint16) if want_mag else None
    lib.modes_oracle_run(C.byref(cfg), iq.ctypes.data, n, C.byref(out), C.byref(nout), st.ctypes.data,
                         mag.ctypes.data if want_mag else None)
    msgs = np.zeros(nout.value, dtype=ORACLE_MSG)
    if nout.value:
        C.memmove(msgs.ctypes.data, out.value, nout.value * ORACLE_MSG.itemsize)
    lib.modes_oracle_free(out)
    return (msgs, st[0], mag) if want_mag else (msgs, st[0])


def reference_run(iq, fmt=0, nfix=1, fixdf=1, thr=58):
    """The checker of the -m gpu parity tests on the BASELINE configurations: the reference's OWN objects (oracle/_ref, compiled from
    /root/reference in place; the prebuilt files travel to the GPU box) where they are there, the restatement otherwise — one hop
    less between the HIP path and the reference (tests/test_oracle.py pins the restatement to _ref either way)."""
    if have_ref():
        return ref_run(iq, fmt, nfix, fixdf, thr)
    return oracle_run(iq, fmt, nfix, fixdf, thr)


def oracle_convert(iq, fmt):
    lib = oracle_lib()
    iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = iq.size // FMT_BYTES[fmt]
    mag = np.empty(n, dtype=np.uint16)
    ml, mp = C.c_double(), C.c_double()
    lib.modes_oracle_convert(fmt, iq.ctypes.data, mag.ctypes.data, n, C.byref(ml), C.byref(mp))
    return mag, ml.value, mp.value


def have_ref():
    return os.path.exists(REF_BIN)


def ref_run(iq, fmt=0, nfix=1, fixdf=1, thr=58, want_mag=False, mode_ac=0, flip_before=False):
    """The reference's own objects (oracle/_ref/ref_demod), one process per run."""
    iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    with tempfile.TemporaryDirectory() as d:
        fin, fm, fs, fg = (os.path.join(d, x) for x in ("in.iq", "out.msgs", "out.stats", "out.mag"))
        iq.tofile(fin)
        cmd = [REF_BIN, FMT_NAMES[fmt], str(nfix), str(fixdf), str(thr), fin, fm, fs] + ([fg] if want_mag else [])
        env = dict(os.environ)
        if mode_ac:
            env["ORACLE_MODE_AC"] = "1"
        else:
            env.pop("ORACLE_MODE_AC", None)
        env.pop("ORACLE_FLIP_BEFORE", None)
        if flip_before:
            env["ORACLE_FLIP_BEFORE"] = "1"
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, env=env)
        msgs = np.fromfile(fm, dtype=ORACLE_MSG)
        st = np.fromfile(fs, dtype=ORACLE_STATS)[0]
        if want_mag:
            return msgs, st, np.fromfile(fg, dtype=np.uint16)
    return msgs, st


def assert_same_messages(gpu_msgs, oracle_msgs, startup_ms=STARTUP_MS):
    """Bit-exact comparison of the product's mgpu_msg list with the checker's oracle_msg list."""
    assert len(gpu_msgs) == len(oracle_msgs), f"message count {len(gpu_msgs)} != oracle {len(oracle_msgs)}"
    if len(gpu_msgs) == 0:
        return
    g, o = gpu_msgs, oracle_msgs
    for name, a, b in [
        ("timestamp", g["timestamp"], o["timestamp"]),
        ("sysTimestamp", g["sysTimestamp"] - startup_ms, o["sys_rel_ms"]),
        ("score", g["score"].astype(np.int64), o["score"].astype(np.int64)),
        ("correctedbits", g["correctedbits"].astype(np.int64), o["correctedbits"].astype(np.int64)),
        ("msgbits", g["msgbits"].astype(np.int64), o["msgbits"].astype(np.int64)),
        ("msgtype", g["msgtype"].astype(np.int64), o["msgtype"].astype(np.int64)),
        ("addr", g["addr"].astype(np.int64), o["addr"].astype(np.int64)),
    ]:
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, f"{name}: {bad.size} mismatches, first at message {bad[0]}: {a[bad[0]]} vs {b[bad[0]]}"
    for name in ("msg", "raw"):
        bad = np.nonzero((g[name] != o[name]).any(axis=1))[0]
        assert bad.size == 0, f"{name}: {bad.size} mismatches, first at {bad[0]}: {g[name][bad[0]]} vs {o[name][bad[0]]}"
    sig = g["sig_sumsq"].astype(np.float64) / 65535.0 / 65535.0 / g["sig_len"].astype(np.float64)
    bad = np.nonzero(sig != o["signalLevel"])[0]
    assert bad.size == 0, f"signalLevel: {bad.size} mismatches, first at {bad[0]}: {sig[bad[0]]} vs {o['signalLevel'][bad[0]]}"


def assert_same_counters(gpu_counters, oracle_stats, float_tol=0.0):
    for f in COUNTER_FIELDS:
        a = np.asarray(gpu_counters[f], dtype=np.uint64)
        b = np.asarray(oracle_stats[f], dtype=np.uint64)
        assert (a == b).all(), f"counter {f}: gpu {a} != oracle {b}"
    for f in ("signal_power_sum", "peak_signal_power", "noise_power_sum"):
        a, b = float(gpu_counters[f]), float(oracle_stats[f])
        if np.isnan(a) and np.isnan(b):
            continue
        if float_tol == 0.0:
            assert a == b, f"{f}: gpu {a!r} != oracle {b!r}"
        else:
            assert abs(a - b) <= float_tol * max(1.0, abs(b)), f"{f}: gpu {a!r} vs oracle {b!r}"
