"""Golden vectors for the first-stage tracking gate (which accepted messages the reference forwards), generated
by the WHOLE reference program: `make -C oracle full`, run with `--dump-beast` on a seeded synthetic capture, exactly as
make_beast_golden.py does; every frame of the dump is then matched to the oracle's message list (timestamp + frame bytes), and what
is committed is ONE BIT PER ACCEPTED MESSAGE — forwarded or not — packed (tests/golden/gate_<name>.npz: `forwarded`, `n`).
Only runs in the development container (needs /root/reference).

The reference program has a start-up race (make_beast_golden.py): a capture is run until a stream comes out whose every frame is
one of the oracle's messages (the "flip after the first buffer" order the oracle implements), at most 12 times.

    python tests/golden/make_gate_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import helpers  # noqa: E402
from make_beast_golden import reference_frames  # noqa: E402

CASES = [  # name, synth kwargs, reference options, oracle options
    ("uc8_fix_2s", dict(seconds=2.0, seed=99, rate=1500.0), [], dict(nfix=1, mode_ac=0)),
    ("uc8_aggressive_modeac_3s", dict(seconds=3.0, seed=98, rate=700.0, dense=2), ["--aggressive", "--modeac"], dict(nfix=2, mode_ac=1)),
    # 200 aircraft for a minute: everybody is past its first two messages after the first second
    ("uc8_fix_200ac_60s", dict(seconds=60.0, seed=4242, rate=2000.0), [], dict(nfix=1, mode_ac=0)),
    # 30 000 aircraft that transmit once in 12 s on average: most of an aircraft's life IS its first messages, unreliable formats
    # arrive more than 45 s behind the last reliable one (track.c:1933), the ICAO filter's table grows and expires
    ("uc8_fix_30000ac_130s", dict(seconds=130.0, seed=777, rate=2500.0, naircraft=30000), [], dict(nfix=1, mode_ac=0)),
]


def split_frames(stream):
    """beast byte stream -> [(48-bit timestamp, raw frame bytes)]"""
    out, i, n = [], 0, len(stream)
    while i < n:
        j = i + 2
        while j < n and not (stream[j] == 0x1A and (j + 1 >= n or stream[j + 1] != 0x1A)):
            j += 2 if stream[j] == 0x1A else 1
        raw = bytes(stream[i:j])
        body = raw[2:].replace(b"\x1a\x1a", b"\x1a")
        out.append((int.from_bytes(body[:6], "big"), raw))
        i = j
    return out


def forwarded_bits(msgs, frames):
    """Which of the oracle's messages the stream holds; None if a frame has no message (the other start-up order)."""
    lib = helpers.oracle_lib()
    lib.modes_oracle_beast_frame.restype = C.c_size_t
    lib.modes_oracle_beast_frame.argtypes = [C.c_void_p, C.c_void_p]
    buf = (C.c_uint8 * 64)()
    want = {}
    for ts, raw in frames:
        want[(ts, raw)] = want.get((ts, raw), 0) + 1
    fwd = np.zeros(len(msgs), dtype=bool)
    for k in range(len(msgs)):
        m = msgs[k:k + 1]
        nb = lib.modes_oracle_beast_frame(m.ctypes.data, buf)
        key = (int(m["timestamp"][0]) & ((1 << 48) - 1), bytes(buf[:nb]))
        if want.get(key, 0) > 0:
            fwd[k] = True
            want[key] -= 1
    return fwd if sum(want.values()) == 0 else None


def main():
    only = sys.argv[1:]
    for name, kw, opts, opt in CASES:
        if only and name not in only:
            continue
        iq = helpers.synth(threads=8, **kw)
        msgs, _ = helpers.oracle_run(iq, 0, opt["nfix"], 1, 58, mode_ac=opt["mode_ac"])
        fwd = None
        for attempt in range(12):
            fwd = forwarded_bits(msgs, split_frames(reference_frames(iq, opts)))
            if fwd is not None:
                break
        assert fwd is not None, f"{name}: no run of the reference program gave the oracle's start-up order"
        np.savez_compressed(os.path.join(HERE, f"gate_{name}.npz"), forwarded=np.packbits(fwd), n=np.int64(len(msgs)))
        print(name, kw, opts, len(msgs), "messages,", int(fwd.sum()), "forwarded, attempt", attempt + 1)


if __name__ == "__main__":
    main()
