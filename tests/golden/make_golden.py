#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by running the REFERENCE ITSELF
(oracle/_ref/ref_demod = readsb's own convert.c/demod_2400.c/mode_s.c/crc.c/icao_filter.c compiled
unmodified from /root/reference) on seeded synthetic captures (tools/synth_iq.c).

The reference ships no IQ fixtures or golden vectors for this path, so these are the
pins: oracle/modes_oracle.c and the HIP path are both checked against them.  Only runs in the dev
container (needs /root/reference to build oracle/_ref); the fixtures it writes are committed.

    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

# name -> (synth kwargs, fmt, nfix, fixdf, thr)
CASES = {
    "uc8_fix_1s": (dict(seconds=1.0, seed=1001), 0, 1, 1, 58),
    "uc8_aggressive_dense_1s": (dict(seconds=1.0, seed=1002, rate=8000.0, dense=1), 0, 2, 1, 58),
    "uc8_nofix_nofixdf_1s": (dict(seconds=1.0, seed=1003), 0, 0, 0, 58),
    "uc8_thr75_exact_4buf": (dict(nsamples=4 * 131072, seed=1004, rate=3000.0), 0, 1, 1, 75),
    "sc16q11_aggressive_1s": (dict(seconds=1.0, seed=1005, fmt=2, rate=2500.0), 2, 2, 1, 58),
    "sc16_fix_1s": (dict(seconds=1.0, seed=1006, fmt=1), 1, 1, 1, 58),
}


def main():
    helpers.ensure_built()
    assert helpers.have_ref(), "oracle/_ref missing: build it in the dev container (make -C oracle ref)"
    index = {}
    for name, (skw, fmt, nfix, fixdf, thr) in CASES.items():
        iq = helpers.synth(**skw)
        msgs, st = helpers.ref_run(iq, fmt, nfix, fixdf, thr)
        np.save(os.path.join(HERE, name + ".msgs.npy"), msgs)
        stats = {f: np.asarray(st[f]).tolist() for f in helpers.COUNTER_FIELDS}
        for f in ("signal_power_sum", "noise_power_sum", "peak_signal_power"):
            stats[f] = float(st[f]).hex()
        index[name] = {"synth": skw, "fmt": fmt, "nfix": nfix, "fixdf": fixdf, "thr": thr,
                       "iq_sha256": hashlib.sha256(iq.tobytes()).hexdigest(), "nmsgs": int(len(msgs)), "stats": stats}
        print(name, len(msgs), "messages")
    # CRC tables of the reference (crc.c) for nfix 1 and 2, and its UC8 magnitude table (convert.c)
    ref = C.CDLL(os.path.join(helpers.ORACLE_DIR, "_ref", "libreadsb_ref.so"))
    ref.ref_diagnose.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    ref.ref_modesChecksum.argtypes = [C.c_void_p, C.c_int]
    ref.ref_modesChecksum.restype = C.c_uint32
    tables = {}
    for nfix in (1, 2):
        ref.ref_crc_init(nfix)
        for bits in (56, 112):
            # enumerate every 1- and 2-bit pattern over bits 5..bits-1 and ask the reference
            single = []
            for k in range(bits):
                m = np.zeros(bits // 8, dtype=np.uint8)
                m[k >> 3] = 0x80 >> (k & 7)
                single.append(ref.ref_modesChecksum(m.ctypes.data, bits))
            rows = []
            seen = set()
            cands = [(single[a],) for a in range(5, bits)]
            if nfix == 2:
                cands += [(single[a] ^ single[b],) for a in range(5, bits) for b in range(a + 1, bits)]
            for (syn,) in cands:
                if syn in seen:
                    continue
                seen.add(syn)
                b0, b1 = C.c_int(), C.c_int()
                n = ref.ref_diagnose(syn, bits, C.byref(b0), C.byref(b1))
                if n > 0:
                    rows.append((syn, n, b0.value, b1.value))
            tables[f"nfix{nfix}_{bits}"] = np.array(sorted(rows), dtype=np.int64)
            print(f"nfix {nfix} bits {bits}: {len(rows)} correctable syndromes")
    tables["single_bit_syndrome_112"] = np.array(single, dtype=np.int64)
    ref.ref_convert.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
    i, q = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    iq = np.stack([i.ravel(), q.ravel()], axis=1).ravel()
    mag = np.zeros(65536, dtype=np.uint16)
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2); os.dup2(devnull, 2)
    ref.ref_convert(0, iq.ctypes.data, mag.ctypes.data, 65536, None, None)
    os.dup2(saved, 2)
    tables["uc8_mag_by_i_q"] = mag.reshape(256, 256)   # [I][Q]
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **tables)
    json.dump(index, open(os.path.join(HERE, "index.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
