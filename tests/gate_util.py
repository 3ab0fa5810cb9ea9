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


def check_against_golden(verdict, forwarded, max_deferred_share):
    """Every verdict that is not `deferred` must be what the whole reference program did."""
    v = verdict & 3
    certain = v != 2
    wrong = np.nonzero(certain & ((v == 1) != forwarded))[0]
    assert len(wrong) == 0, f"{len(wrong)} certain verdicts differ from the reference program's, first at message {wrong[:5]}"
    share = float((~certain).sum()) / max(1, len(v))
    assert share <= max_deferred_share, f"deferred share {share:.4f}"
    return share
