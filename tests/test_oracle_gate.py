"""CPU: the restated first-stage tracking gate (oracle/modes_oracle_gate.c) against the whole reference program's --dump-beast
streams (tests/golden/gate_*.npz): every verdict it calls certain is what the program did; what depends on the position tracker
is deferred, and that share is bounded."""
import numpy as np
import pytest

import gate_util as gu


@pytest.mark.parametrize("name,max_deferred", [("uc8_fix_2s", 0.05), ("uc8_aggressive_modeac_3s", 0.05), ("uc8_fix_200ac_60s", 0.005),
                                               ("uc8_fix_30000ac_130s", 0.12)])
def test_certain_verdicts_equal_the_reference_programs(name, max_deferred):
    _, msgs, fields = gu.oracle_messages(name)
    fwd = gu.golden_forwarded(name)
    assert len(fwd) == len(msgs)
    v = gu.oracle_gate(msgs, fields)
    share = gu.check_against_golden(v, fwd, max_deferred)
    print(f"{name}: {len(msgs)} messages, {int(fwd.sum())} forwarded, deferred share {share:.5f}, not forwarded (certain) {int(((v & 3) == 0).sum())}")
    assert ((v & 3) == 0).sum() > 0 or name == "uc8_fix_200ac_60s"


def test_state_carries_over_calls():
    """Two calls on the halves of a capture (cut at a buffer boundary) = one call on the whole."""
    import ctypes as C
    import helpers
    _, msgs, fields = gu.oracle_messages("uc8_fix_2s")
    whole = gu.oracle_gate(msgs, fields)
    buf = ((msgs["timestamp"].astype(np.int64) - 772) // 5) // gu.BUF
    cut = int(np.searchsorted(buf, buf[len(buf) // 2]))
    lib = helpers.oracle_lib()
    lib.modes_oracle_gate_new.restype = C.c_void_p
    lib.modes_oracle_gate_free.argtypes = [C.c_void_p]
    g = lib.modes_oracle_gate_new()
    a = gu.oracle_gate(msgs[:cut], fields[:cut], state=g)
    b = gu.oracle_gate(msgs[cut:], fields[cut:], state=g)
    lib.modes_oracle_gate_free(g)
    assert np.array_equal(np.concatenate([a, b]), whole)
