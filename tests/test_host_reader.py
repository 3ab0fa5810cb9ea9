"""CPU: the host adapter's file reader (readsb_amd/host/demod_gpu.c — read-ahead thread, parallel pread slices, the two
page-locked chunk buffers, EOF rules of ifileRun, sdr_ifile.c:194-241) driven against a stand-in for the library that only
records what it is fed (tests/host_stub/reader_check.c).  What reaches mgpu_feed_iq must be the file, in order, in feeds of
at most cfg.max_samples, straight out of the registered buffers, followed by exactly one mgpu_finish."""
import os
import subprocess

import numpy as np
import pytest

import helpers

SRC = os.path.join(helpers.ROOT, "tests", "host_stub", "reader_check.c")
HOST = os.path.join(helpers.ROOT, "readsb_amd", "host")
B = 131072


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hoststub") / "reader_check")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-o", exe, SRC, os.path.join(HOST, "demod_gpu.c"), "-lpthread", "-lm"], check=True)
    return exe


def _stats(line):
    return {k: int(v) for k, v in (kv.split("=") for kv in line.split())}


@pytest.mark.parametrize("fmt,bps", [("UC8", 2), ("SC16", 4)])
@pytest.mark.parametrize("nbytes_of", [lambda c, bps: 0, lambda c, bps: 2 * bps, lambda c, bps: c * bps - 2, lambda c, bps: c * bps,
                                       lambda c, bps: c * bps + bps, lambda c, bps: 3 * c * bps, lambda c, bps: 3 * c * bps + 12345 * bps + 1,
                                       lambda c, bps: 7 * c * bps + 4096 * 3])
def test_file_reaches_the_demodulator_unchanged(checker, tmp_path, fmt, bps, nbytes_of):
    chunk_buffers = 3
    chunk_samples = chunk_buffers * B
    nbytes = nbytes_of(chunk_samples, bps)
    data = np.random.default_rng(nbytes).integers(0, 256, size=nbytes, dtype=np.uint8)
    src, out = tmp_path / "in.iq", tmp_path / "out.iq"
    data.tofile(src)
    r = subprocess.run([checker, str(src), fmt, str(chunk_buffers), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    st = _stats(r.stdout.strip())
    whole = nbytes // bps * bps                                  # a trailing partial sample is dropped (have / bytes_per_sample)
    # only the LAST feed may be cut short to whole samples: a partial sample can only sit at the end of the file
    assert np.array_equal(np.fromfile(out, dtype=np.uint8), data[:whole])
    assert st["samples"] == st["counted"] == whole // bps
    assert st["feeds"] == -(-(whole // bps) // chunk_samples)
    assert st["finish"] == 1 and st["bad"] == 0 and st["rc"] == 0
    assert st["registered"] == 2 and st["unregistered"] == 2      # two chunk buffers, page-locked once


def test_pipe_input(checker, tmp_path):
    """`--ifile -`: not seekable, sequential reads with short read() returns (sdr_ifile.c:221-235)."""
    nbytes = 5 * B * 2 + 777 * 2
    data = np.random.default_rng(5).integers(0, 256, size=nbytes, dtype=np.uint8)
    out = tmp_path / "out.iq"
    p = subprocess.Popen([checker, "-", "UC8", "2", str(out)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=False)
    raw = data.tobytes()
    for off in range(0, len(raw), 100003):                      # odd-sized writes
        p.stdin.write(raw[off:off + 100003])
    p.stdin.close()
    line = p.stdout.read().decode().strip()
    assert p.wait(timeout=120) == 0
    st = _stats(line)
    assert np.array_equal(np.fromfile(out, dtype=np.uint8), data)
    assert st["samples"] == nbytes // 2 and st["finish"] == 1 and st["bad"] == 0


@pytest.fixture(scope="module")
def fanin_checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hoststub") / "fanin_check")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-o", exe, os.path.join(helpers.ROOT, "tests", "host_stub", "fanin_check.c"),
                    os.path.join(HOST, "sdr_gpu_fanin.c"), os.path.join(HOST, "demod_gpu.c"), "-lpthread", "-lm"], check=True)
    return exe


def test_fanin_streams_options_and_devices(fanin_checker, tmp_path):
    """Seven files of two formats: `--iformat` applies to the `--ifile`s after it, every stream gets its own context on device
    index % devices (the stand-in reports three; `--gpu-devices 2` narrows that), common options reach every context,
    every context is fed exactly its own file and finished once."""
    rng = np.random.default_rng(1)
    sizes = [5 * B * 2 + 10, 0, 2 * B * 4, 3 * B * 2, B * 4 + 8, 12, 9 * B * 2]
    fmts = ["UC8", "UC8", "SC16", "UC8", "SC16Q11", "SC16Q11", "UC8"]
    args, datas, last = [], [], None
    for k, (n, f) in enumerate(zip(sizes, fmts)):
        d = rng.integers(0, 256, size=n, dtype=np.uint8)
        p = tmp_path / f"s{k}.iq"
        d.tofile(p)
        datas.append(d)
        if f != last:
            args += ["--iformat", f]
            last = f
        args += ["--ifile", str(p)]
    prefix = tmp_path / "fed"
    r = subprocess.run([fanin_checker, str(prefix)] + args + ["--aggressive", "--modeac", "--preamble-threshold", "40", "--gpu-devices", "2",
                                                              "--gpu-chunk-buffers", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    head = _stats(lines[0])
    assert head == {"run": 0, "streams": 7, "finish": 7, "nfix": 2, "mode_ac": 1, "thr": 40}
    code = {"UC8": 0, "SC16": 1, "SC16Q11": 2}
    for k, ln in enumerate(lines[1:]):
        st = _stats(ln)
        bps = 2 if fmts[k] == "UC8" else 4
        assert st["stream"] == k and st["device"] == k % 2 and st["format"] == code[fmts[k]]
        assert st["samples"] == st["processed"] == sizes[k] // bps
        fed = np.fromfile(f"{prefix}.{st['ctx']}", dtype=np.uint8)
        assert np.array_equal(fed, datas[k][: sizes[k] // bps * bps]), f"stream {k}"


def test_fanin_needs_an_input(fanin_checker, tmp_path):
    r = subprocess.run([fanin_checker, str(tmp_path / "x"), "--fix"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "requires an --ifile argument" in r.stderr


@pytest.fixture(scope="module")
def cli_standin(tmp_path_factory):
    """readsb_gpu_ifile (the stand-alone CLI) linked against the oracle stand-in instead of libmodes_gpu.so."""
    exe = str(tmp_path_factory.mktemp("hoststub") / "readsb_gpu_ifile_standin")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-ffp-contract=off", "-o", exe, os.path.join(HOST, "readsb_gpu_ifile.c"),
                    os.path.join(HOST, "demod_gpu.c"), os.path.join(helpers.ROOT, "tests", "host_stub", "modes_gpu_standin.c"),
                    os.path.join(helpers.ORACLE_DIR, "modes_oracle.c"), os.path.join(helpers.ORACLE_DIR, "modes_oracle_fields.c"),
                    "-lpthread", "-lm"], check=True)
    return exe


@pytest.mark.parametrize("fmt,flag,nfix,nsamples,extra", [
    (0, "--fix", 1, 9 * B + 4321, []),
    (0, "--fix", 1, 6 * B, []),                               # exact multiple: the zero-length EOF buffer (sdr_ifile.c:223-237)
    (2, "--aggressive", 2, 7 * B + 99, []),
    (0, "--fix", 1, 8 * B + 17, ["--modeac"]),
])
def test_cli_host_code_on_the_standin(cli_standin, tmp_path, fmt, flag, nfix, nsamples, extra):
    """Option parsing, the chunked reader, per-message delivery, --raw line format and the --stats block of the C host side,
    end to end on the CPU: the output must be the oracle's message list (which the goldens pin to the reference)."""
    mode_ac = 1 if extra else 0
    iq = helpers.synth(nsamples=nsamples, seed=1000 + nsamples % 97, fmt=fmt, rate=900.0 if mode_ac else 2000.0, dense=2 if mode_ac else 0)
    path = tmp_path / "cap.iq"
    iq.tofile(path)
    want, wst = helpers.oracle_run(iq, fmt, nfix, 1, 58, mode_ac=mode_ac)
    r = subprocess.run([cli_standin, "--device-type", "ifile", "--ifile", str(path), "--iformat", helpers.FMT_NAMES[fmt], flag, "--raw", "--mlat",
                        "--stats", "--gpu-chunk-buffers", "4", "--startup-time-ms", str(helpers.STARTUP_MS)] + extra,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = ["@%012X%s;" % (int(m["timestamp"]), bytes(m["msg"][: int(m["msgbits"]) // 8]).hex()) for m in want]
    assert len(lines) > 50 and r.stdout.strip().splitlines() == lines
    assert f"{int(wst['demod_preambles'])} Mode-S message preambles received" in r.stderr
    assert f"{int(wst['demod_accepted'][0])} accepted with correct CRC" in r.stderr
    assert f"{int(wst['samples_processed'])} samples processed" in r.stderr
