"""ctypes binding of libmodes_gpu.so (include/modes_gpu.h).

Mirrors the reference's interface for this path: a converter (`iq_convert_fn`, convert.h:34-39),
`demodulate2400(struct mag_buf *)` (demod_2400.h:38) and the ifile reader's feed loop
(sdr_ifile.c:169-270).  There is no CPU fallback: if the HIP library or a GPU is missing the
constructor raises.
"""
import ctypes as C
import os

import numpy as np

FMT_UC8, FMT_SC16, FMT_SC16Q11 = 0, 1, 2
_FMT_BYTES = {FMT_UC8: 2, FMT_SC16: 4, FMT_SC16Q11: 4}

# struct mgpu_msg (64 bytes)
MSG_DTYPE = np.dtype([
    ("timestamp", "<i8"), ("sysTimestamp", "<i8"), ("sig_sumsq", "<u8"), ("sig_len", "<u2"),
    ("score", "<i2"), ("phase", "u1"), ("correctedbits", "u1"), ("msgtype", "u1"), ("msgbits", "u1"),
    ("addr", "<u4"), ("msg", "u1", 14), ("raw", "u1", 14),
])
assert MSG_DTYPE.itemsize == 64

# struct mgpu_fields (include/modes_gpu.h): the per-message field decode
FIELDS_DTYPE = np.dtype([
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
assert FIELDS_DTYPE.itemsize == 176


FILTER_CLOCK_AFTER_FIRST, FILTER_CLOCK_BEFORE_FIRST, FILTER_CLOCK_EXTERNAL = 0, 1, 2


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("format", C.c_int32), ("nfix_crc", C.c_int32), ("fixDF", C.c_int32),
        ("preamble_threshold", C.c_int32), ("buf_samples", C.c_uint32), ("trailing_samples", C.c_uint32),
        ("mode_ac", C.c_uint32), ("max_samples", C.c_uint64), ("startup_time_ms", C.c_int64),
        ("record_pool_records", C.c_uint64), ("max_messages", C.c_uint64),
        ("filter_clock", C.c_uint32), ("streams_on_device", C.c_uint32), ("chunk_buffers", C.c_uint32), ("abi_version", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("demod_preambles", C.c_uint64), ("demod_rejected_bad", C.c_uint64),
        ("demod_rejected_unknown_icao", C.c_uint64), ("demod_accepted", C.c_uint64 * 3),
        ("demod_preamblePhase", C.c_uint64 * 5), ("demod_bestPhase", C.c_uint64 * 5),
        ("strong_signal_count", C.c_uint64), ("signal_power_count", C.c_uint64),
        ("noise_power_count", C.c_uint64), ("samples_processed", C.c_uint64), ("samples_lost", C.c_uint64),
        ("nbuffers", C.c_uint64), ("nflips", C.c_uint64), ("signal_power_sum", C.c_double),
        ("noise_power_sum", C.c_double), ("peak_signal_power", C.c_double), ("demod_modeac", C.c_uint64),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out

    @classmethod
    def from_dict(cls, d):
        """The struct of an as_dict() result (the wire form of the counters between ranks is the struct's bytes, as in the C host)."""
        k = cls()
        for name, ctype in cls._fields_:
            v = d[name]
            if hasattr(ctype, "_length_"):
                for i in range(ctype._length_):
                    getattr(k, name)[i] = v[i]
            else:
                setattr(k, name, v)
        return k


class Timing(C.Structure):
    _fields_ = [
        ("h2d_ms", C.c_float), ("convert_ms", C.c_float), ("sweep_ms", C.c_float), ("prescreen_ms", C.c_float),
        ("resolve_ms", C.c_float), ("sigpower_ms", C.c_float), ("d2h_ms", C.c_float), ("total_ms", C.c_float),
        ("n_candidates", C.c_uint64), ("n_records", C.c_uint64), ("n_live_records", C.c_uint64),
        ("n_messages", C.c_uint64), ("n_chunks", C.c_uint64), ("slice_ms", C.c_float), ("build_ms", C.c_float), ("n_timed_chunks", C.c_uint64),
        ("build_wait_ms", C.c_float), ("sweep_fused_chunks", C.c_float),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class ShardWalkArgs(C.Structure):
    _fields_ = [
        ("own_first", C.c_uint64), ("flip_after", C.c_void_p), ("nflips", C.c_uint64), ("start_state", C.c_void_p),
        ("start_state_bytes", C.c_uint64), ("check_records", C.c_int32), ("reserved", C.c_int32),
    ]


DEFAULT_CHUNK_BUFFERS = 0                   # mgpu_config.chunk_buffers of Demodulators created without one (tests lower it: more chunks per feed)


class ShardStreamArgs(C.Structure):
    _fields_ = [
        ("first_sample", C.c_uint64), ("history_iq", C.c_void_p), ("own_first", C.c_uint64), ("flip_after", C.c_void_p), ("nflips", C.c_uint64),
        ("start_state", C.c_void_p), ("start_state_bytes", C.c_uint64),
    ]


DEFERRED_DTYPE = np.dtype([("index", "<u8"), ("offset", "<u8")])     # struct mgpu_deferred
ABI_VERSION = 6                             # MGPU_ABI_VERSION of the include/modes_gpu.h these ctypes mirrors were written against


class MgpuError(RuntimeError):
    pass


def flip_schedule(end_clocks, startup_ms, filter_clock=0):
    """Indices of the buffers the ICAO filter expires after, from every buffer's end clock (mgpu_flip_schedule: the reference's
    rule, readsb.c:1227-1231)."""
    lib = load_library()
    clocks = np.ascontiguousarray(end_clocks, dtype=np.int64)
    out = np.empty(clocks.size // 1000 + 16, dtype=np.uint64)      # an expiry per 60 s = per ~1099 buffers at least
    n = int(lib.mgpu_flip_schedule(C.c_void_p(clocks.ctypes.data), clocks.size, int(startup_ms), int(filter_clock), C.c_void_p(out.ctypes.data), out.size))
    if n > out.size:
        out = np.empty(n, dtype=np.uint64)
        n = int(lib.mgpu_flip_schedule(C.c_void_p(clocks.ctypes.data), clocks.size, int(startup_ms), int(filter_clock), C.c_void_p(out.ctypes.data), out.size))
    return out[:n].copy()


def expiry_windows(nbuf_total, startup_ms, filter_clock=0, buf_samples=131072):
    """Boolean mask of the buffers an expiry of the ICAO filter can follow (mgpu_expiry_windows)."""
    mask = np.zeros(int(nbuf_total), dtype=np.uint8)
    if nbuf_total:
        load_library().mgpu_expiry_windows(int(nbuf_total), int(buf_samples), int(startup_ms), int(filter_clock), C.c_void_p(mask.ctypes.data))
    return mask.astype(bool)


def shard_round(sched_ts, gathered, nsamples, startup_ms, filter_clock=0, buf_samples=131072):
    """mgpu_shard_round: gathered[r] = (clocks, state_first, state_end) -> (done, next schedule, {rank: rank to import the end state of})."""
    lib = load_library()
    world = len(gathered)
    sched = np.ascontiguousarray(sched_ts, dtype=np.int64)
    clocks = [np.ascontiguousarray(g[0], dtype=np.int64) for g in gathered]
    sf = [np.frombuffer(bytes(g[1]) or b"\0", dtype=np.uint8) for g in gathered]
    se = [np.frombuffer(bytes(g[2]) or b"\0", dtype=np.uint8) for g in gathered]
    ptrs = lambda arrs: (C.c_void_p * world)(*[a.ctypes.data for a in arrs])
    lens = lambda xs: np.array(xs, dtype=np.uint64)
    ncl, nsf, nse = lens([c.size for c in clocks]), lens([len(g[1]) for g in gathered]), lens([len(g[2]) for g in gathered])
    cap = sum(c.size for c in clocks) // 1000 + 64
    nxt = np.empty(cap, dtype=np.int64)
    n_next, done = C.c_uint64(0), C.c_int32(0)
    imp = np.empty(world, dtype=np.int32)
    rc = lib.mgpu_shard_round(sched.ctypes.data if sched.size else None, sched.size, world, ptrs(clocks), C.c_void_p(ncl.ctypes.data), ptrs(sf),
                              C.c_void_p(nsf.ctypes.data), ptrs(se), C.c_void_p(nse.ctypes.data), int(nsamples), int(buf_samples), int(startup_ms),
                              int(filter_clock), C.c_void_p(nxt.ctypes.data), cap, C.byref(n_next), C.c_void_p(imp.ctypes.data), C.byref(done))
    if rc != 0:
        raise MgpuError("mgpu_shard_round failed")
    return bool(done.value), nxt[: n_next.value].copy(), {r: int(imp[r]) for r in range(world) if imp[r] >= 0}


def seqsum(start, terms):
    """((start + t0) + t1) + ... in doubles, in order (mgpu_seqsum)."""
    terms = np.ascontiguousarray(terms, dtype=np.float64)
    return float(load_library().mgpu_seqsum(float(start), C.c_void_p(terms.ctypes.data), terms.size))


SUM_BLOCK_DTYPE = np.dtype([("total", "<u8"), ("e", "<i4"), ("flags", "<u4")])
SUM_BLOCK = 1024                            # messages per block of the block-wise sequential sum


def seqsum_blocks(approx_start, msgs, block=SUM_BLOCK):
    """A range's messages prepared for the block-wise sequential sum of their signal powers (mgpu_seqsum_blocks)."""
    if msgs.dtype != MSG_DTYPE or not msgs.flags["C_CONTIGUOUS"]:
        raise ValueError("seqsum_blocks: need a contiguous mgpu_msg record array")
    out = np.zeros((msgs.size + block - 1) // block, dtype=SUM_BLOCK_DTYPE)
    rc = load_library().mgpu_seqsum_blocks(float(approx_start), C.c_void_p(msgs.ctypes.data), msgs.size, block, C.c_void_p(out.ctypes.data))
    if rc != 0:
        raise MgpuError("mgpu_seqsum_blocks failed")
    return out


def seqsum_blocks_terms(approx_start, sumsq, block=SUM_BLOCK):
    """The same blocks from the messages' 8-byte signal-power numerators (Demodulator.shard_signal_terms)."""
    sumsq = np.ascontiguousarray(sumsq, dtype=np.uint64)
    out = np.zeros((sumsq.size + block - 1) // block, dtype=SUM_BLOCK_DTYPE)
    rc = load_library().mgpu_seqsum_blocks_terms(float(approx_start), C.c_void_p(sumsq.ctypes.data), sumsq.size, block, C.c_void_p(out.ctypes.data))
    if rc != 0:
        raise MgpuError("mgpu_seqsum_blocks_terms failed")
    return out


def seqsum_apply(start, msgs, blocks, block=SUM_BLOCK):
    """-> (the exact sequential sum continued over this range, blocks that had to be re-added message by message)."""
    blocks = np.ascontiguousarray(blocks, dtype=SUM_BLOCK_DTYPE)
    if msgs.dtype != MSG_DTYPE or not msgs.flags["C_CONTIGUOUS"] or blocks.size != (msgs.size + block - 1) // block:
        raise ValueError("seqsum_apply: messages and blocks do not match")
    fb = C.c_uint64(0)
    s = load_library().mgpu_seqsum_apply(float(start), C.c_void_p(msgs.ctypes.data), msgs.size, block, C.c_void_p(blocks.ctypes.data), C.byref(fb))
    return float(s), int(fb.value)


def seqsum_signal_power(start, msgs):
    """... of the messages' signal powers sig_sumsq / 65535^2, in order (mgpu_seqsum_signal_power; demod_2400.c:445-447)."""
    if msgs.dtype != MSG_DTYPE or not msgs.flags["C_CONTIGUOUS"]:
        raise ValueError("seqsum_signal_power: need a contiguous mgpu_msg record array")
    return float(load_library().mgpu_seqsum_signal_power(float(start), C.c_void_p(msgs.ctypes.data), msgs.size))


def lib_path():
    """The product library; MGPU_LIBRARY=libmodes_gpu_exp.so selects the cross-check build (`make -C readsb_amd/csrc exp`:
    the same library plus the superseded fused sweep kernel) — tests and tools only."""
    name = os.path.basename(os.environ.get("MGPU_LIBRARY", "libmodes_gpu.so"))
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", name)


_lib = None


def load_library():
    """dlopen libmodes_gpu.so (built in-tree by __graft_entry__.build()).  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise MgpuError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    vp, u64, u32, i32, i64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int64
    lib.mgpu_config_defaults_abi.argtypes = [C.POINTER(Config), u32, u32]
    lib.mgpu_config_defaults_abi.restype = None
    lib.mgpu_abi_version.restype = u32
    # this file mirrors the structs of include/modes_gpu.h at ABI_VERSION: a library of another version is refused here, and
    # mgpu_create checks the version the defaults call wrote into the struct (never more than sizeof(Config) bytes of it)
    if lib.mgpu_abi_version() != ABI_VERSION:
        raise MgpuError(f"{path} is ABI version {lib.mgpu_abi_version()}, readsb_amd/binding.py mirrors version {ABI_VERSION}: rebuild the library")
    lib.mgpu_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.mgpu_destroy.argtypes = [vp]
    lib.mgpu_destroy.restype = None
    lib.mgpu_reset.argtypes = [vp]
    lib.mgpu_strerror.argtypes = [i32]
    lib.mgpu_strerror.restype = C.c_char_p
    lib.mgpu_last_error.argtypes = [vp]
    lib.mgpu_last_error.restype = C.c_char_p
    lib.mgpu_device_count.restype = i32
    lib.mgpu_set_deferred.argtypes = [vp, i32]
    lib.mgpu_set_device_messages.argtypes = [vp, i32]
    lib.mgpu_collect_device.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), vp]
    lib.mgpu_feed_iq.argtypes = [vp, vp, u64]
    lib.mgpu_feed_iq_device.argtypes = [vp, vp, u64]
    lib.mgpu_host_cpus.argtypes = [vp, vp, i32]
    lib.mgpu_host_alloc.argtypes = [vp, u64]
    lib.mgpu_host_alloc.restype = vp
    lib.mgpu_host_free.argtypes = [vp, vp]
    lib.mgpu_host_free.restype = None
    lib.mgpu_host_register.argtypes = [vp, vp, u64]
    lib.mgpu_host_unregister.argtypes = [vp, vp]
    lib.mgpu_upload_iq.argtypes = [vp, vp, u64]
    lib.mgpu_device_iq_buffer.argtypes = [vp]
    lib.mgpu_device_iq_buffer.restype = vp
    lib.mgpu_finish.argtypes = [vp]
    lib.mgpu_collect.argtypes = [vp, vp, u64, C.POINTER(u64), C.POINTER(Counters)]
    lib.mgpu_pending_messages.argtypes = [vp]
    lib.mgpu_filter_expire.argtypes = [vp]
    lib.mgpu_filter_add.argtypes = [vp, u32]
    lib.mgpu_set_message_buffer.argtypes = [vp, vp, u64]
    lib.mgpu_set_device_message_buffer.argtypes = [vp, vp, u64]
    lib.mgpu_decode_fields.argtypes = [vp, vp, u64, vp]
    lib.mgpu_decode_fields_device.argtypes = [vp, vp, u64, vp]
    lib.mgpu_track_gate.argtypes = [vp, vp, u64, vp]
    lib.mgpu_track_gate_device.argtypes = [vp, vp, vp, u64, vp]
    lib.mgpu_track_gate_reset.argtypes = [vp]
    lib.mgpu_beast_encode.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
    lib.mgpu_beast_encode_gated.argtypes = [vp, vp, u64, u32, vp, u64, C.POINTER(u64), vp, u64, C.POINTER(u64)]
    lib.mgpu_beast_encode_gated_device.argtypes = [vp, vp, vp, u64, u32, vp, u64, C.POINTER(u64), vp, u64, C.POINTER(u64)]
    lib.mgpu_beast_encode_device.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
    lib.mgpu_shard_begin.argtypes = [vp, u64, vp, i32]
    lib.mgpu_adder_bitmap_get.argtypes = [vp, vp]
    lib.mgpu_adder_bitmap_set.argtypes = [vp, vp]
    lib.mgpu_shard_packets.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.mgpu_walk_packets.argtypes = [vp, vp, u64]
    lib.mgpu_shard_clock_estimate.argtypes = [vp, vp, u64, u64, vp, u64, C.POINTER(u64)]
    lib.mgpu_shard_walk.argtypes = [vp, vp, u64, C.POINTER(ShardWalkArgs), vp, u64, C.POINTER(u64)]
    lib.mgpu_shard_state.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(u64)]
    lib.mgpu_shard_stream_begin.argtypes = [vp, C.POINTER(ShardStreamArgs)]
    lib.mgpu_shard_stream_mark.argtypes = [vp]
    lib.mgpu_shard_stream_end.argtypes = [vp, vp, u64, C.POINTER(u64)]
    lib.mgpu_shard_noise_terms.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.mgpu_shard_signal_terms.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.mgpu_seqsum_blocks_terms.argtypes = [C.c_double, vp, u64, u32, vp]
    lib.mgpu_flip_schedule.argtypes = [vp, u64, i64, i32, vp, u64]
    lib.mgpu_flip_schedule.restype = u64
    lib.mgpu_expiry_windows.argtypes = [u64, u32, i64, i32, vp]
    lib.mgpu_expiry_windows.restype = u64
    lib.mgpu_shard_round.argtypes = [vp, u64, u32, vp, vp, vp, vp, vp, vp, u64, u32, i64, i32, vp, u64, C.POINTER(u64), vp, C.POINTER(i32)]
    lib.mgpu_seqsum.argtypes = [C.c_double, vp, u64]
    lib.mgpu_seqsum.restype = C.c_double
    lib.mgpu_seqsum_signal_power.argtypes = [C.c_double, vp, u64]
    lib.mgpu_seqsum_signal_power.restype = C.c_double
    lib.mgpu_seqsum_blocks.argtypes = [C.c_double, vp, u64, u32, vp]
    lib.mgpu_seqsum_apply.argtypes = [C.c_double, vp, u64, u32, vp, C.POINTER(u64)]
    lib.mgpu_seqsum_apply.restype = C.c_double
    lib.mgpu_pending_messages.restype = u64
    lib.mgpu_last_timing.argtypes = [vp, C.POINTER(Timing)]
    lib.mgpu_debug_device_walk.argtypes = [vp, C.POINTER(u64)]
    lib.mgpu_debug_last_magnitudes.argtypes = [vp, vp, u64]
    lib.mgpu_event_bracket_us.argtypes = [vp, C.POINTER(C.c_float)]
    lib.mgpu_convert.argtypes = [vp, vp, vp, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mgpu_demod_mag_buf.argtypes = [vp, vp, u32, i64, i64, C.c_double, u32]
    lib.mgpu_demod_mag_buf_ac.argtypes = [vp, vp, u32, i64, i64, C.c_double, C.c_double, u32]
    lib.mgpu_crc_checksum.argtypes = [vp, i32]
    lib.mgpu_crc_checksum.restype = u32
    lib.mgpu_crc_diagnose.argtypes = [i32, u32, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.mgpu_crc_table_size.argtypes = [i32, i32]
    lib.mgpu_uc8_table.restype = C.POINTER(C.c_uint16)
    _lib = lib
    return lib


class Demodulator:
    """One SDR stream on one GPU (struct mgpu_ctx)."""
    shard_counters_complete = True      # mgpu_walk_packets rebuilds every statistic of an unsharded run (shard packets carry what it needs)

    def __init__(self, fmt=FMT_UC8, nfix_crc=1, fix_df=1, preamble_threshold=58, max_samples=64 * 131072,
                 device=0, startup_time_ms=0, record_pool_records=0, max_messages=0, buf_samples=131072, mode_ac=0,
                 filter_clock=0, chunk_buffers=None):
        self.lib = load_library()
        cfg = Config()
        self.lib.mgpu_config_defaults_abi(C.byref(cfg), C.sizeof(Config), ABI_VERSION)
        cfg.device, cfg.format, cfg.nfix_crc, cfg.fixDF = device, fmt, nfix_crc, fix_df
        cfg.preamble_threshold, cfg.max_samples, cfg.startup_time_ms = preamble_threshold, max_samples, startup_time_ms
        cfg.record_pool_records, cfg.max_messages, cfg.buf_samples = record_pool_records, max_messages, buf_samples
        cfg.mode_ac = 1 if mode_ac else 0
        cfg.filter_clock = filter_clock          # FILTER_CLOCK_*: who runs icaoFilterExpire (modes_gpu.h)
        cfg.chunk_buffers = DEFAULT_CHUNK_BUFFERS if chunk_buffers is None else chunk_buffers   # buffers per pipeline chunk (0 = the library's 1024)
        self.cfg = cfg
        self.fmt = fmt
        self._collect_buf = None
        self._msgbuf = None
        self.ctx = C.c_void_p()
        rc = self.lib.mgpu_create(C.byref(cfg), C.byref(self.ctx))
        if rc != 0:
            raise MgpuError(f"mgpu_create: {self.lib.mgpu_strerror(rc).decode()}")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mgpu_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise MgpuError(f"{what}: {self.lib.mgpu_strerror(rc).decode()} ({self.lib.mgpu_last_error(self.ctx).decode()})")

    def _nsamples(self, iq):
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
        return iq, iq.size // _FMT_BYTES[self.fmt]

    def reset(self):
        self._chk(self.lib.mgpu_reset(self.ctx), "mgpu_reset")

    def filter_expire(self):
        """icaoFilterExpire() forwarded by the host (filter_clock=FILTER_CLOCK_EXTERNAL only)."""
        self._chk(self.lib.mgpu_filter_expire(self.ctx), "mgpu_filter_expire")

    def filter_add(self, addr):
        """icaoFilterAdd(addr) made by the host outside the demodulator."""
        self._chk(self.lib.mgpu_filter_add(self.ctx, int(addr)), "mgpu_filter_add")

    def feed_iq(self, iq):
        iq, n = self._nsamples(iq)
        self._chk(self.lib.mgpu_feed_iq(self.ctx, iq.ctypes.data, n), "mgpu_feed_iq")

    def upload_iq(self, iq):
        iq, n = self._nsamples(iq)
        self._chk(self.lib.mgpu_upload_iq(self.ctx, iq.ctypes.data, n), "mgpu_upload_iq")
        return n

    def device_iq_buffer(self):
        return self.lib.mgpu_device_iq_buffer(self.ctx)

    def feed_resident(self, nsamples, dptr=None):
        self._chk(self.lib.mgpu_feed_iq_device(self.ctx, dptr if dptr is not None else self.device_iq_buffer(), nsamples),
                  "mgpu_feed_iq_device")

    def finish(self):
        self._chk(self.lib.mgpu_finish(self.ctx), "mgpu_finish")

    def set_message_buffer(self, arr):
        """Build the messages of the following feeds straight into `arr` (a contiguous mgpu_msg record array);
        collect(out=arr) then only reports how many there are.  None returns to the internal list."""
        if arr is None:
            self._chk(self.lib.mgpu_set_message_buffer(self.ctx, None, 0), "mgpu_set_message_buffer")
            self._msgbuf = None
            return
        if arr.dtype != MSG_DTYPE or not arr.flags["C_CONTIGUOUS"]:
            raise ValueError("set_message_buffer: need a contiguous mgpu_msg record array")
        self._chk(self.lib.mgpu_set_message_buffer(self.ctx, C.c_void_p(arr.ctypes.data), C.c_uint64(arr.size)), "mgpu_set_message_buffer")
        self._msgbuf = arr                   # keep it alive

    # ---- one capture sharded by buffer ranges (include/modes_gpu.h, "sharded" section; readsb_amd/shard.py) ----
    def shard_begin(self, first_sample, history_iq, mode):
        hist = None if history_iq is None else np.ascontiguousarray(history_iq, dtype=np.uint8)
        self._hist_keep = hist
        self._chk(self.lib.mgpu_shard_begin(self.ctx, C.c_uint64(first_sample), None if hist is None else C.c_void_p(hist.ctypes.data), mode),
                  "mgpu_shard_begin")

    def adder_bitmap(self):
        words = np.empty(1 << 19, dtype=np.uint32)
        self._chk(self.lib.mgpu_adder_bitmap_get(self.ctx, C.c_void_p(words.ctypes.data)), "mgpu_adder_bitmap_get")
        return words

    def set_adder_bitmap(self, words):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        assert words.size == 1 << 19
        self._chk(self.lib.mgpu_adder_bitmap_set(self.ctx, C.c_void_p(words.ctypes.data)), "mgpu_adder_bitmap_set")

    def shard_packets(self):
        p, n = C.c_void_p(), C.c_uint64(0)
        self._chk(self.lib.mgpu_shard_packets(self.ctx, C.byref(p), C.byref(n)), "mgpu_shard_packets")
        if not n.value:
            return np.zeros(0, dtype=np.uint8)
        # (a view of the library's own buffer: valid until the context's next shard pass or reset)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,))

    def walk_packets(self, packets):
        packets = np.ascontiguousarray(packets, dtype=np.uint8)
        self._chk(self.lib.mgpu_walk_packets(self.ctx, C.c_void_p(packets.ctypes.data), C.c_uint64(packets.size)), "mgpu_walk_packets")

    # ---- ... with the ordered walk itself sharded over the ranks (include/modes_gpu.h, "round 4"; shard.py: ShardWalkRank) ----
    @staticmethod
    def _packets_arg(packets):
        if packets is None:
            return None, 0, None
        packets = np.ascontiguousarray(packets, dtype=np.uint8)
        return C.c_void_p(packets.ctypes.data), packets.size, packets

    def shard_clock_estimate(self, own_first, nbuffers, packets=None):
        """The range's buffers' end clocks estimated from the records alone (packets=None: the context's own packets)."""
        ptr, size, keep = self._packets_arg(packets)
        out = np.empty(int(nbuffers) + 1, dtype=np.int64)
        n = C.c_uint64(0)
        self._chk(self.lib.mgpu_shard_clock_estimate(self.ctx, ptr, size, int(own_first), C.c_void_p(out.ctypes.data), out.size, C.byref(n)),
                  "mgpu_shard_clock_estimate")
        return out[: n.value]

    def shard_walk(self, own_first, nbuffers, flip_after_ts, start_state=None, packets=None, check_records=False):
        """Walk warm-up + range with the expiry schedule imposed, build the range's messages (mgpu_shard_walk).
        -> (end clocks of the range's buffers, state at own_first, state at the range's end) — the states as bytes."""
        ptr, size, keep = self._packets_arg(packets)
        sched = np.ascontiguousarray(flip_after_ts, dtype=np.int64)
        a = ShardWalkArgs()
        a.own_first = int(own_first)
        a.flip_after = sched.ctypes.data if sched.size else None
        a.nflips = sched.size
        st = None
        if start_state is not None:
            st = np.frombuffer(start_state, dtype=np.uint8)
            a.start_state, a.start_state_bytes = st.ctypes.data, st.size
        a.check_records = 1 if check_records else 0
        out = np.empty(int(nbuffers) + 1, dtype=np.int64)
        n = C.c_uint64(0)
        self._chk(self.lib.mgpu_shard_walk(self.ctx, ptr, size, C.byref(a), C.c_void_p(out.ctypes.data), out.size, C.byref(n)), "mgpu_shard_walk")
        states = []
        for which in (0, 1):
            bp, bn = C.c_void_p(), C.c_uint64(0)
            self._chk(self.lib.mgpu_shard_state(self.ctx, which, C.byref(bp), C.byref(bn)), "mgpu_shard_state")
            states.append(C.string_at(bp, bn.value) if bn.value else b"")
        return out[: n.value].copy(), states[0], states[1]

    def _shard_states(self):
        states = []
        for which in (0, 1):
            bp, bn = C.c_void_p(), C.c_uint64(0)
            self._chk(self.lib.mgpu_shard_state(self.ctx, which, C.byref(bp), C.byref(bn)), "mgpu_shard_state")
            states.append(C.string_at(bp, bn.value) if bn.value else b"")
        return states

    def shard_stream_begin(self, first_sample, history_iq, own_first, flip_after_ts, start_state=None):
        """A rank's pass through the ordinary pipeline (mgpu_shard_stream_*): resets the context, cold start or imported state."""
        hist = None if history_iq is None else np.ascontiguousarray(history_iq, dtype=np.uint8)
        sched = np.ascontiguousarray(flip_after_ts, dtype=np.int64)
        a = ShardStreamArgs()
        a.first_sample, a.own_first = int(first_sample), int(own_first)
        a.history_iq = None if hist is None else hist.ctypes.data
        a.flip_after, a.nflips = (sched.ctypes.data if sched.size else None), sched.size
        st = None
        if start_state is not None:
            st = np.frombuffer(start_state, dtype=np.uint8)
            a.start_state, a.start_state_bytes = st.ctypes.data, st.size
        self._chk(self.lib.mgpu_shard_stream_begin(self.ctx, C.byref(a)), "mgpu_shard_stream_begin")

    def shard_stream_mark(self):
        self._chk(self.lib.mgpu_shard_stream_mark(self.ctx), "mgpu_shard_stream_mark")

    def shard_stream_end(self, nbuffers):
        """-> (the range's true end clocks, state at own_first, state at the range's end)."""
        out = np.empty(int(nbuffers) + 1, dtype=np.int64)
        n = C.c_uint64(0)
        self._chk(self.lib.mgpu_shard_stream_end(self.ctx, C.c_void_p(out.ctypes.data), out.size, C.byref(n)), "mgpu_shard_stream_end")
        s0, s1 = self._shard_states()
        return out[: n.value].copy(), s0, s1

    def shard_noise_terms(self):
        """What each buffer of the walked range adds to noise_power_sum, in order (a copy)."""
        tp, tn = C.c_void_p(), C.c_uint64(0)
        self._chk(self.lib.mgpu_shard_noise_terms(self.ctx, C.byref(tp), C.byref(tn)), "mgpu_shard_noise_terms")
        if not tn.value:
            return np.zeros(0, dtype=np.float64)
        return np.ctypeslib.as_array(C.cast(tp, C.POINTER(C.c_double)), shape=(tn.value,)).copy()

    def shard_signal_terms(self):
        """sig_sumsq of every accepted message of the walked range, in order (a view of the library's own array: valid until the
        context's next shard pass); empty with Mode A/C."""
        tp, tn = C.c_void_p(), C.c_uint64(0)
        self._chk(self.lib.mgpu_shard_signal_terms(self.ctx, C.byref(tp), C.byref(tn)), "mgpu_shard_signal_terms")
        if not tn.value:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(C.cast(tp, C.POINTER(C.c_uint64)), shape=(tn.value,))

    def decode_fields(self, msgs):
        """Per-message field records (FIELDS_DTYPE) of a message record array, decoded on the GPU."""
        msgs = np.ascontiguousarray(msgs)
        assert msgs.dtype == MSG_DTYPE
        out = np.empty(len(msgs), dtype=FIELDS_DTYPE)
        self._chk(self.lib.mgpu_decode_fields(self.ctx, C.c_void_p(msgs.ctypes.data), len(msgs), C.c_void_p(out.ctypes.data)), "mgpu_decode_fields")
        return out

    def decode_fields_device(self, d_msgs_ptr, n, d_out_ptr):
        self._chk(self.lib.mgpu_decode_fields_device(self.ctx, C.c_void_p(d_msgs_ptr), n, C.c_void_p(d_out_ptr)), "mgpu_decode_fields_device")

    def track_gate(self, msgs):
        """First stage of the tracker + the forwarding rule on the GPU (include/modes_gpu.h): one verdict byte per message — bits 0-1
        0 not forwarded / 1 forwarded / 2 deferred to the host's tracker.  msgs: accepted messages in order, whole sample buffers
        per call; the aircraft table lives on the device from call to call (track_gate_reset)."""
        msgs = np.ascontiguousarray(msgs)
        assert msgs.dtype == MSG_DTYPE
        out = np.empty(len(msgs), dtype=np.uint8)
        self._chk(self.lib.mgpu_track_gate(self.ctx, C.c_void_p(msgs.ctypes.data), len(msgs), C.c_void_p(out.ctypes.data)), "mgpu_track_gate")
        return out

    def beast_encode_gated(self, msgs, net_rule=False, deferred_cap=None):
        """mgpu_beast_encode_gated: -> (the beast stream of the certainly-forwarded messages, deferred[] records {index, offset})."""
        msgs = np.ascontiguousarray(msgs)
        n = len(msgs)
        cap = n * 44 + 64
        dcap = int(deferred_cap if deferred_cap is not None else n)
        out = np.empty(cap, dtype=np.uint8)
        deferred = np.zeros(max(dcap, 1), dtype=DEFERRED_DTYPE)
        nb, nd = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.lib.mgpu_beast_encode_gated(self.ctx, C.c_void_p(msgs.ctypes.data), C.c_uint64(n), C.c_uint32(1 if net_rule else 0),
                                                   C.c_void_p(out.ctypes.data), C.c_uint64(cap), C.byref(nb), C.c_void_p(deferred.ctypes.data),
                                                   C.c_uint64(dcap), C.byref(nd)), "mgpu_beast_encode_gated")
        return out[: nb.value].tobytes(), deferred[: nd.value].copy()

    def track_gate_device(self, d_msgs_ptr, d_fields_ptr, n, d_verdict_ptr):
        self._chk(self.lib.mgpu_track_gate_device(self.ctx, C.c_void_p(d_msgs_ptr), C.c_void_p(d_fields_ptr), n, C.c_void_p(d_verdict_ptr)), "mgpu_track_gate_device")

    def track_gate_reset(self):
        self._chk(self.lib.mgpu_track_gate_reset(self.ctx), "mgpu_track_gate_reset")

    def beast_encode(self, msgs):
        """Beast wire stream (bytes) of a record array (host memory in, host memory out, encoded on the GPU)."""
        msgs = np.ascontiguousarray(msgs)
        assert msgs.dtype == MSG_DTYPE
        out = np.empty(len(msgs) * 44 + 64, dtype=np.uint8)
        nb = C.c_uint64(0)
        self._chk(self.lib.mgpu_beast_encode(self.ctx, C.c_void_p(msgs.ctypes.data), len(msgs), C.c_void_p(out.ctypes.data), out.size, C.byref(nb)),
                  "mgpu_beast_encode")
        return out[: nb.value].tobytes()

    def beast_encode_device(self, d_msgs_ptr, n, d_out_ptr, cap):
        """Same on device pointers (ints, e.g. torch tensor.data_ptr()); returns the stream's size in bytes."""
        nb = C.c_uint64(0)
        self._chk(self.lib.mgpu_beast_encode_device(self.ctx, C.c_void_p(d_msgs_ptr), n, C.c_void_p(d_out_ptr), cap, C.byref(nb)), "mgpu_beast_encode_device")
        return int(nb.value)

    def host_cpus(self):
        """CPUs the context's host threads are pinned to (empty list: not pinned)."""
        buf = (C.c_int32 * 64)()
        n = int(self.lib.mgpu_host_cpus(self.ctx, buf, 64))
        return [int(buf[i]) for i in range(max(0, min(n, 64)))]

    def keep_other_threads_away(self, confine_to_own_l3=False):
        """Move every thread of this process that is not one of the context's pinned threads off the pinned cores and
        their SMT siblings (the application side of mgpu_host_cpus' advice; Linux only, best effort).
        confine_to_own_l3=True (several ranks on one node): the other threads go to the free SMT siblings of the
        context's own L3 group instead, so that they cannot wander onto another rank's pipeline cores either."""
        cpus = self.host_cpus()
        if not cpus:
            return 0

        def cpulist(path):
            out = set()
            try:
                for part in open(path).read().strip().split(","):
                    lo, _, hi = part.partition("-")
                    out.update(range(int(lo), int(hi or lo) + 1))
            except (OSError, ValueError):
                pass
            return out

        siblings, l3 = set(), set()
        for c in cpus:
            siblings |= cpulist(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") or {c}
            l3 |= cpulist(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list")
        moved = 0
        for tid in os.listdir("/proc/self/task"):
            try:
                cur = os.sched_getaffinity(int(tid))
                if len(cur) == 1 and next(iter(cur)) in cpus:
                    continue                       # one of the pipeline's own threads
                target = ((l3 - set(cpus)) & cur) if confine_to_own_l3 else (cur - siblings)
                if target and target != cur:
                    os.sched_setaffinity(int(tid), target)
                    moved += 1
            except OSError:
                pass
        return moved

    def host_register(self, arr):
        """Page-lock a numpy array the caller keeps feeding from (mgpu_host_register)."""
        self._chk(self.lib.mgpu_host_register(self.ctx, C.c_void_p(arr.ctypes.data), C.c_uint64(arr.nbytes)), "mgpu_host_register")

    def host_unregister(self, arr):
        self._chk(self.lib.mgpu_host_unregister(self.ctx, C.c_void_p(arr.ctypes.data)), "mgpu_host_unregister")

    def collect(self, reuse=False, out=None):
        """Drain the decoded messages (stream order) and read the counters.

        reuse=True returns a view of a buffer the Demodulator keeps and overwrites on the next
        collect(reuse=True): a steady consumer then pays one copy and no page faults per call.
        out=<record array> collects into the caller's buffer (e.g. pinned staging memory).
        """
        n = int(self.lib.mgpu_pending_messages(self.ctx))
        if out is not None:
            if out.dtype != MSG_DTYPE or not out.flags["C_CONTIGUOUS"] or out.size < n:
                raise ValueError("collect(out=...): need a contiguous mgpu_msg array with room for %d records" % n)
        elif reuse:
            if self._collect_buf is None or self._collect_buf.size < n:
                self._collect_buf = np.empty(max(n, 1024) * 5 // 4, dtype=MSG_DTYPE)
            out = self._collect_buf
        else:
            out = np.empty(n, dtype=MSG_DTYPE)
        got = C.c_uint64(0)
        cnt = Counters()
        self._chk(self.lib.mgpu_collect(self.ctx, out.ctypes.data, n, C.byref(got), C.byref(cnt)), "mgpu_collect")
        return out[: got.value], cnt.as_dict()

    def host_alloc(self, nbytes):
        """Page-locked host memory on the device's NUMA node as a uint8 numpy array (mgpu_host_alloc); host_free(arr) releases it."""
        p = self.lib.mgpu_host_alloc(self.ctx, int(nbytes))
        if not p:
            raise MgpuError("mgpu_host_alloc failed")
        arr = np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(p))
        self._host_allocs = getattr(self, "_host_allocs", {})
        self._host_allocs[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_host_allocs", {}).pop(arr.ctypes.data, None)
        if p:
            self.lib.mgpu_host_free(self.ctx, p)

    def set_deferred(self, on=True):
        """mgpu_set_deferred: feeds return once enqueued; the loop is feed(k+1); collect_feed(k)."""
        self._chk(self.lib.mgpu_set_deferred(self.ctx, 1 if on else 0), "mgpu_set_deferred")

    def set_device_messages(self, on=True):
        """mgpu_set_device_messages (deferred mode): the messages of a feed are built on the GPU and stay there (True / 1), or are
        stored by the GPU into the page-locked array set_message_buffer named for the feed (2)."""
        self._chk(self.lib.mgpu_set_device_messages(self.ctx, int(on)), "mgpu_set_device_messages")

    def set_device_message_buffer(self, dptr, capacity):
        """mgpu_set_device_message_buffer: the next feed's records go to device address dptr (capacity records); None: the library's list."""
        self._chk(self.lib.mgpu_set_device_message_buffer(self.ctx, C.c_void_p(dptr) if dptr else None, C.c_uint64(capacity if dptr else 0)),
                  "mgpu_set_device_message_buffer")

    def collect_feed_device(self, want_counters=False):
        """Device-messages mode: wait for the oldest uncollected feed; returns (device pointer of its mgpu_msg records, count,
        counters or None).  The pointer stays valid until three more feeds have been started."""
        ptr, n = C.c_void_p(), C.c_uint64(0)
        cnt = Counters() if want_counters else None
        self._chk(self.lib.mgpu_collect_device(self.ctx, C.byref(ptr), C.byref(n), C.byref(cnt) if cnt is not None else None),
                  "mgpu_collect_device")
        return int(ptr.value or 0), int(n.value), (cnt.as_dict() if cnt is not None else None)

    def collect_feed(self, out, want_counters=False):
        """Deferred mode: wait for the oldest uncollected feed and take its messages into `out` (a contiguous mgpu_msg array,
        possibly the one named by set_message_buffer for that feed: then nothing is copied).  want_counters=True also drains
        everything in flight and returns the settled counters."""
        if out.dtype != MSG_DTYPE or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("collect_feed(out): need a contiguous mgpu_msg array")
        got = C.c_uint64(0)
        cnt = Counters() if want_counters else None
        self._chk(self.lib.mgpu_collect(self.ctx, out.ctypes.data, out.size, C.byref(got), C.byref(cnt) if cnt is not None else None),
                  "mgpu_collect")
        return out[: got.value], (cnt.as_dict() if cnt is not None else None)

    def last_magnitudes(self, n):
        """mgpu_debug_last_magnitudes: the first n magnitudes of the chunk the pipeline demodulated last, out of HBM."""
        out = np.empty(int(n), dtype=np.uint16)
        self._chk(self.lib.mgpu_debug_last_magnitudes(self.ctx, out.ctypes.data, int(n)), "mgpu_debug_last_magnitudes")
        return out

    def device_walk_stats(self):
        """mgpu_debug_device_walk: what the walk on the device (MGPU_DEVICE_WALK) did so far."""
        out = (C.c_uint64 * 8)()
        self._chk(self.lib.mgpu_debug_device_walk(self.ctx, out), "mgpu_debug_device_walk")
        keys = ("chunks", "device", "unsettled", "premises_failed", "not_modelled", "walks", "differences")
        return dict(zip(keys, [int(v) for v in out]))

    def event_bracket_us(self):
        """mgpu_event_bracket_us: what a pair of timing events around one kernel reports beyond the kernel itself (us)."""
        v = C.c_float(0)
        self._chk(self.lib.mgpu_event_bracket_us(self.ctx, C.byref(v)), "mgpu_event_bracket_us")
        return float(v.value)

    def timing(self):
        t = Timing()
        self._chk(self.lib.mgpu_last_timing(self.ctx, C.byref(t)), "mgpu_last_timing")
        return t.as_dict()

    def convert(self, iq):
        """iq_convert_fn: returns (mag u16[n], mean_level, mean_power)."""
        iq, n = self._nsamples(iq)
        mag = np.empty(n, dtype=np.uint16)
        ml, mp = C.c_double(), C.c_double()
        self._chk(self.lib.mgpu_convert(self.ctx, iq.ctypes.data, mag.ctypes.data, n, C.byref(ml), C.byref(mp)), "mgpu_convert")
        return mag, ml.value, mp.value

    def demod_mag_buf(self, data, length, sample_timestamp, sys_timestamp, mean_power, dropped=0):
        """demodulate2400(struct mag_buf *): data = u16[326 + length]."""
        data = np.ascontiguousarray(data, dtype=np.uint16)
        assert data.size >= 326 + length
        self._chk(self.lib.mgpu_demod_mag_buf(self.ctx, data.ctypes.data, length, sample_timestamp, sys_timestamp,
                                              float(mean_power), dropped), "mgpu_demod_mag_buf")

    def demod_mag_buf_ac(self, data, length, sample_timestamp, sys_timestamp, mean_level, mean_power, dropped=0):
        """demodulate2400 + demodulate2400AC on one struct mag_buf (cfg.mode_ac)."""
        data = np.ascontiguousarray(data, dtype=np.uint16)
        assert data.size >= 326 + length
        self._chk(self.lib.mgpu_demod_mag_buf_ac(self.ctx, data.ctypes.data, length, sample_timestamp, sys_timestamp,
                                                 float(mean_level), float(mean_power), dropped), "mgpu_demod_mag_buf_ac")

    def demodulate_capture(self, iq, chunk_samples=None):
        """Whole capture: feed in chunks that are multiples of the buffer size, finish, collect."""
        iq, n = self._nsamples(iq)
        bps = _FMT_BYTES[self.fmt]
        chunk = chunk_samples or int(self.cfg.max_samples)
        chunk -= chunk % int(self.cfg.buf_samples)
        assert chunk > 0
        off = 0
        while off < n:
            k = min(chunk, n - off)
            self.feed_iq(iq[off * bps:(off + k) * bps])
            off += k
        self.finish()
        return self.collect()
