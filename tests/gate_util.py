"""The first-stage tracking gate: inputs from a message list, the oracle's verdicts, the goldens."""
import ctypes as C
import os

import numpy as np

import fields_util as fu
import helpers

BUF = 131072
F_CPR_VALID = 1 << 10
CASES = {   # name -> (synth kwargs, oracle options): tests/golden/make_gate_golden.py
    "uc8_fix_2s": (dict(seconds=2.0, seed=99, rate=1500.0), dict(nfix=1, mode_ac=0)),
    "uc8_aggressive_modeac_3s": (dict(seconds=3.0, seed=98, rate=700.0, dense=2), dict(nfix=2, mode_ac=1)),
    "uc8_fix_200ac_60s": (dict(seconds=60.0, seed=4242, rate=2000.0), dict(nfix=1, mode_ac=0)),
    "uc8_fix_30000ac_130s": (dict(seconds=130.0, seed=777, rate=2500.0, naircraft=30000), dict(nfix=1, mode_ac=0)),
}


def golden_forwarded(name):
    z = np.load(os.path.join(helpers.GOLDEN_DIR, f"gate_{name}.npz"))
    n = int(z["n"])
    return np.unpackbits(z["forwarded"])[:n].astype(bool)


def oracle_messages(name):
    kw, opt = CASES[name]
    iq = helpers.synth(threads=8, **kw)
    msgs, _ = helpers.oracle_run(iq, 0, opt["nfix"], 1, 58, mode_ac=opt["mode_ac"])
    fields = fu.oracle_fields(np.ascontiguousarray(msgs["msg"]), msgs["msgbits"].astype(np.int32))
    return iq, msgs, fields


def oracle_gate(msgs, fields, state=None):
    """modes_oracle_gate_run on an oracle message list (+ its field records) -> verdict bytes."""
    lib = helpers.oracle_lib()
    lib.modes_oracle_gate_new.restype = C.c_void_p
    lib.modes_oracle_gate_free.argtypes = [C.c_void_p]
    lib.modes_oracle_gate_run.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 8
    n = len(msgs)
    own = state is None
    g = lib.modes_oracle_gate_new() if own else state
    arr = dict(msgtype=msgs["msgtype"].astype(np.uint8), addr=np.ascontiguousarray(fields["addr"], dtype=np.uint32), iid=np.ascontiguousarray(fields["IID"], dtype=np.uint8),
               cb=msgs["correctedbits"].astype(np.uint8), cpr=((fields["flags"] & F_CPR_VALID) != 0).astype(np.uint8),
               now=(msgs["sys_rel_ms"].astype(np.int64) + helpers.STARTUP_MS), buffer=(((msgs["timestamp"].astype(np.int64) - 772) // 5) // BUF).astype(np.uint64))
    out = np.zeros(n, dtype=np.uint8)
    lib.modes_oracle_gate_run(g, n, *[C.c_void_p(arr[k].ctypes.data) for k in ("msgtype", "addr", "iid", "cb", "cpr", "now", "buffer")], C.c_void_p(out.ctypes.data))
    if own:
        lib.modes_oracle_gate_free(g)
    return out


def oracle_gate_raw(msgtype, addr, iid, cb, cpr, now_ms, buffer):
    """modes_oracle_gate_run on plain arrays (a fresh table)."""
    lib = helpers.oracle_lib()
    lib.modes_oracle_gate_new.restype = C.c_void_p
    lib.modes_oracle_gate_free.argtypes = [C.c_void_p]
    lib.modes_oracle_gate_run.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 8
    arrs = [np.ascontiguousarray(msgtype, dtype=np.uint8), np.ascontiguousarray(addr, dtype=np.uint32), np.ascontiguousarray(iid, dtype=np.uint8),
            np.ascontiguousarray(cb, dtype=np.uint8), np.ascontiguousarray(cpr, dtype=np.uint8), np.ascontiguousarray(now_ms, dtype=np.int64),
            np.ascontiguousarray(buffer, dtype=np.uint64)]
    n = len(arrs[0])
    out = np.zeros(n, dtype=np.uint8)
    g = lib.modes_oracle_gate_new()
    lib.modes_oracle_gate_run(g, n, *[C.c_void_p(a.ctypes.data) for a in arrs], C.c_void_p(out.ctypes.data))
    lib.modes_oracle_gate_free(g)
    return out


def synthetic_list(n, seconds, naircraft, seed, startup_ms=helpers.STARTUP_MS if hasattr(helpers, "STARTUP_MS") else 0):
    """A message list that never saw a sample: what the gate reads of mgpu_msg / mgpu_fields (timestamp on the ifile grid, sysTimestamp,
    msgtype, correctedbits; addr, IID, the CPR flag) for `naircraft` aircraft over `seconds` — long gaps, hours of life, non-ICAO
    addresses, Address/Parity formats with address 0, bursts of more than 256 messages in a buffer."""
    import readsb_amd
    rng = np.random.default_rng(seed)
    pos = np.sort(rng.integers(0, int(seconds * 2.4e6), size=n)).astype(np.int64)
    pos[n // 3: n // 3 + 700] = pos[n // 3]                                           # one buffer with three batches and a bit
    pos = np.sort(pos)
    ac = rng.integers(0, naircraft, size=n)
    quiet = rng.random(naircraft) < 0.3                                              # aircraft that go silent for long stretches
    keep = ~(quiet[ac] & ((pos // int(400 * 2.4e6)) % 2 == 1))
    pos, ac = pos[keep], ac[keep]
    n = len(pos)
    addr = (0x400000 + ac).astype(np.uint32)
    addr[ac % 17 == 3] |= 1 << 24                                                     # MODES_NON_ICAO_ADDRESS
    addr[ac == 5] = 0
    kind = rng.integers(0, 100, size=n)
    msgtype = np.select([kind < 45, kind < 60, kind < 75, kind < 90], [17, 11, 4, 20], 0).astype(np.uint8)
    iid = np.where((msgtype == 11) & (rng.random(n) < 0.3), rng.integers(1, 16, size=n), 0).astype(np.uint8)
    cpr = ((msgtype == 17) & (rng.random(n) < 0.6)).astype(np.uint8)
    cpr[(ac % 5 == 0) & (msgtype == 17)] = 1                                          # aircraft whose every DF17 is a position message
    cb = np.select([rng.random(n) < 0.85, rng.random(n) < 0.7], [0, 1], 2).astype(np.uint8)
    msgs = np.zeros(n, dtype=readsb_amd.MSG_DTYPE)
    msgs["timestamp"] = pos * 5 + 768 + rng.integers(4, 9, size=n)
    msgs["sysTimestamp"] = startup_ms + msgs["timestamp"] // 12000
    msgs["msgtype"], msgs["correctedbits"], msgs["msgbits"], msgs["addr"] = msgtype, cb, np.where(msgtype >= 16, 112, 56), addr
    fields = np.zeros(n, dtype=readsb_amd.FIELDS_DTYPE)
    fields["addr"], fields["IID"], fields["flags"], fields["msgtype"] = addr, iid, cpr.astype(np.uint32) * F_CPR_VALID, msgtype
    raw = dict(msgtype=msgtype, addr=addr, iid=iid, cb=cb, cpr=cpr, now_ms=msgs["sysTimestamp"].astype(np.int64), buffer=(pos // BUF).astype(np.uint64))
    return msgs, fields, raw


def check_against_golden(verdict, forwarded, max_deferred_share):
    """Every verdict that is not `deferred` must be what the whole reference program did."""
    v = verdict & 3
    certain = v != 2
    wrong = np.nonzero(certain & ((v == 1) != forwarded))[0]
    assert len(wrong) == 0, f"{len(wrong)} certain verdicts differ from the reference program's, first at message {wrong[:5]}"
    share = float((~certain).sum()) / max(1, len(v))
    assert share <= max_deferred_share, f"deferred share {share:.4f}"
    return share
