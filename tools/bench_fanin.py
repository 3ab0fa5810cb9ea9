"""Throughput of the fan-in host row (readsb_amd/host/readsb_gpu_fanin): K synthetic UC8 files in a RAM-backed directory,
demodulated concurrently on one GPU, one context per stream; prints the CLI's own summary for K = 1, 2, 4.
File reads (page cache) and PCIe uploads are inside the number — this is the PCIe-inclusive rate, not bench.py's metric.
    python tools/bench_fanin.py [--seconds 100] [--streams 1,2,4]"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=100.0)
    ap.add_argument("--streams", default="1,2,4")
    a = ap.parse_args()
    helpers.ensure_built()
    counts = [int(x) for x in a.streams.split(",")]
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        paths = []
        for k in range(max(counts)):
            iq = helpers.synth(seconds=a.seconds, seed=500 + k, rate=2000.0, threads=min(32, os.cpu_count() or 8))
            p = os.path.join(d, f"s{k}.iq")
            iq.tofile(p)
            paths.append(p)
            del iq
        cli = os.path.join(ROOT, "readsb_amd", "host", "readsb_gpu_fanin")
        for n in counts:
            args = [cli]
            for p in paths[:n]:
                args += ["--ifile", p]
            args += ["--out-prefix", os.path.join(d, "out"), "--stats", "--gpu-chunk-buffers", os.environ.get("FANIN_CHUNK", "512")]
            for rep in range(2):                      # second run: page cache and GPU clocks warm
                r = subprocess.run(args, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr
            print([ln for ln in r.stderr.splitlines() if ln.startswith("fan-in:")][-1], flush=True)


if __name__ == "__main__":
    main()
