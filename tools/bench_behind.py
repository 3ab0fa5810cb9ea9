"""Throughput of the two kernels behind the message list — k_decode_fields and k_beast_size/k_beast_write — on records
resident in HBM (python tools/bench_behind.py [--messages N]).  Prints one JSON line per kernel: messages/s, algorithmic
GB/s (64 + 176 B per message for the field decode; 64 B in + the frame bytes out for the encoder) against the
8 TB/s HBM peak.  Wall clock around the C-ABI `_device` calls (launch + stream sync included), so run it under
`rocprofv3 --kernel-trace --stats` for the kernels' own durations (profiles/r01_behind_*)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import readsb_amd  # noqa: E402
import fields_util as fu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--messages", type=int, default=8 << 20)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    frames, bits = fu.fuzz_frames(1 << 18, 7)
    m = np.zeros(len(frames), dtype=readsb_amd.MSG_DTYPE)
    m["msg"], m["msgbits"], m["msgtype"] = frames, bits, frames[:, 0] >> 3
    m["timestamp"] = np.arange(len(m)) * 977 + 0x1A00
    m["sig_sumsq"], m["sig_len"] = np.random.default_rng(1).integers(1 << 20, 1 << 36, size=len(m)), 268
    aa = (frames[:, 1].astype(np.uint32) << 16) | (frames[:, 2].astype(np.uint32) << 8) | frames[:, 3]
    m["addr"] = aa
    reps_in = a.messages // len(m)
    n = reps_in * len(m)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    d = readsb_amd.Demodulator(max_samples=1 << 20)
    d_in, d_f, d_b = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_in), n * 64) == 0 and hip.hipMalloc(C.byref(d_f), n * readsb_amd.FIELDS_DTYPE.itemsize) == 0 and hip.hipMalloc(C.byref(d_b), n * 44) == 0
    for k in range(reps_in):
        assert hip.hipMemcpy(C.c_void_p(d_in.value + k * m.nbytes), m.ctypes.data, m.nbytes, 1) == 0
    for _ in range(3):
        d.decode_fields_device(d_in.value, n, d_f.value)
        nbytes = d.beast_encode_device(d_in.value, n, d_b.value, n * 44)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        d.decode_fields_device(d_in.value, n, d_f.value)
    t_f = (time.perf_counter() - t0) / a.reps
    t0 = time.perf_counter()
    for _ in range(a.reps):
        d.beast_encode_device(d_in.value, n, d_b.value, n * 44)
    t_b = (time.perf_counter() - t0) / a.reps
    for name, t, algo in (("k_decode_fields", t_f, n * (64 + readsb_amd.FIELDS_DTYPE.itemsize)), ("k_beast_size+k_beast_write", t_b, n * 64 + nbytes)):
        print(json.dumps({"kernel": name, "messages": n, "ms": round(t * 1e3, 4), "messages_per_s": round(n / t),
                          "algorithmic_GBps": round(algo / t / 1e9, 1), "frac_of_hbm_peak": round(algo / t / 8e12, 4),
                          "timing": "wall clock around the C-ABI call, launch + sync included"}))
    d.close()


if __name__ == "__main__":
    main()
