#!/usr/bin/env python3
"""Development aid (GPU box): several captures through the library (MGPU_LIBRARY selects the build), compared with the CPU oracle WITHOUT stopping at the first difference:
prints which counters differ and where the message lists part, then the device timing.  Exit code = number of failing cases."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import readsb_amd  # noqa: E402

CASES = [
    # name, synth kwargs, fmt, nfix, fixdf, thr
    ("uc8_fix_2s", dict(seconds=2.0, seed=11, rate=2500.0), 0, 1, 1, 58),
    ("uc8_aggr_2s", dict(seconds=2.0, seed=12, rate=3000.0), 0, 2, 1, 58),
    ("uc8_nofix_thr40", dict(seconds=1.0, seed=13, rate=2000.0), 0, 0, 0, 40),
    ("uc8_dense", dict(seconds=1.5, seed=14, rate=8000.0, dense=1), 0, 2, 1, 58),
    ("sc16q11_aggr", dict(seconds=1.0, seed=15, rate=3000.0, fmt=2), 2, 2, 1, 58),
    ("uc8_ragged", dict(nsamples=3 * 131072 + 777, seed=16, rate=4000.0), 0, 1, 1, 75),
    ("uc8_fix_30s", dict(seconds=30.0, seed=17, rate=2000.0), 0, 1, 1, 58),
]


def main():
    only = sys.argv[1:] or None
    fails = 0
    for name, kw, fmt, nfix, fixdf, thr in CASES:
        if only and name not in only:
            continue
        iq = helpers.synth(**kw)
        want, wst = helpers.oracle_run(iq, fmt, nfix, fixdf, thr)
        n = iq.size // helpers.FMT_BYTES[fmt]
        d = readsb_amd.Demodulator(fmt=fmt, nfix_crc=nfix, fix_df=fixdf, preamble_threshold=thr, startup_time_ms=helpers.STARTUP_MS,
                                   max_samples=max(n, 131072))
        got, cnt = d.demodulate_capture(iq)
        tm = d.timing()
        d.close()
        ok = True
        for f in helpers.COUNTER_FIELDS:
            a, b = np.asarray(cnt[f], dtype=np.uint64), np.asarray(wst[f], dtype=np.uint64)
            if not (a == b).all():
                ok = False
                print(f"  {name}: counter {f}: gpu {a} oracle {b}")
        if len(got) != len(want):
            ok = False
            gt, wt = set(got["timestamp"].tolist()), set(want["timestamp"].tolist())
            miss, extra = sorted(wt - gt), sorted(gt - wt)
            print(f"  {name}: {len(got)} messages, oracle {len(want)}; missing {len(miss)} (first {miss[:5]}), extra {len(extra)} (first {extra[:5]})")
        else:
            try:
                helpers.assert_same_messages(got, want)
            except AssertionError as e:
                ok = False
                print(f"  {name}: {e}")
        if ok:
            try:
                helpers.assert_same_counters(cnt, wst, float_tol=0.02 if fmt else 0.0)
            except AssertionError as e:
                ok = False
                print(f"  {name}: {e}")
        fails += 0 if ok else 1
        nch = max(1, tm["n_timed_chunks"])
        print(f"{'OK  ' if ok else 'FAIL'} {name}: {len(got)} msgs, {n} samples, cand {tm['n_candidates']} rec {tm['n_records']} live {tm['n_live_records']}; "
              f"per chunk: convert {tm['convert_ms'] / nch:.3f} sweep {tm['sweep_ms'] / nch:.3f} slice {tm['slice_ms'] / nch:.3f} prescreen {tm['prescreen_ms'] / nch:.3f} ms ({nch} chunks)")
    return fails


if __name__ == "__main__":
    sys.exit(main())
