"""Writes tests/golden/fields_fuzz_2000.npz: 2000 fuzzed frames (every downlink format / ME type, tests/fields_util.py)
and the field records the REFERENCE's decodeModesMessage / decodeModeAMessage produce for them (oracle/_ref/libreadsb_ref.so,
built in place from /root/reference by `make -C oracle ref`).  Run from the repo root in the dev container."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import fields_util as fu  # noqa: E402

frames, bits = fu.fuzz_frames(1200, 20240917)
fc, bc = fu.commb_frames(600, 20240918)
frames, bits = np.concatenate([frames, fc]), np.concatenate([bits, bc])
rng = np.random.default_rng(5)
modea = rng.integers(0, 65536, size=200)
fa = np.zeros((200, 14), dtype=np.uint8)
fa[:, 0], fa[:, 1] = modea >> 8, modea & 0xFF
frames = np.concatenate([frames, fa])
bits = np.concatenate([bits, np.full(200, 16, dtype=np.int32)])
want, rc = fu.ref_fields(frames, bits)
assert (rc == 0).all()
out = os.path.join(os.path.dirname(__file__), "fields_fuzz_2000.npz")
np.savez_compressed(out, frames=frames, bits=bits, fields=want.view(np.uint8).reshape(len(want), -1))
print(out, os.path.getsize(out))
