"""CPU tests of the product library: it loads without a GPU, exports every symbol the header
declares, refuses to run without a device (no CPU fallback), and its host-built tables equal
the reference's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers

TABLES = np.load(os.path.join(helpers.GOLDEN_DIR, "tables.npz"))


def _header_functions():
    text = open(os.path.join(helpers.ROOT, "include", "modes_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(mgpu_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if n != "mgpu_msg_signal_level"))


def test_library_exports_every_declared_symbol(built):
    import readsb_amd
    lib = C.CDLL(readsb_amd.lib_path())
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/modes_gpu.h but not exported"


def test_struct_sizes_match_the_c_header(built, tmp_path):
    """The ctypes mirrors must have exactly the layout a C caller sees (compiled with plain gcc -std=c11,
    which also proves the header is C, not C++)."""
    import subprocess
    import readsb_amd
    from readsb_amd import binding
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "modes_gpu.h"\nint main(void){printf("%zu %zu %zu %zu\\n",'
                   'sizeof(struct mgpu_config),sizeof(struct mgpu_msg),sizeof(struct mgpu_counters),sizeof(struct mgpu_timing));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(helpers.ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(binding.Config), readsb_amd.MSG_DTYPE.itemsize, C.sizeof(binding.Counters), C.sizeof(binding.Timing)]
    assert sizes[1] == 64


def test_abi_version_is_checked(built, tmp_path):
    """mgpu_create refuses a config that names another header version (a host compiled against an older, shorter struct), and the
    defaults macro a C host gets writes nothing behind the struct size it passes."""
    import subprocess
    import readsb_amd
    from readsb_amd import binding
    lib = readsb_amd.load_library()
    lib.mgpu_abi_version.restype = C.c_uint32
    header = open(os.path.join(helpers.ROOT, "include", "modes_gpu.h")).read()
    version = int(re.search(r"#define MGPU_ABI_VERSION (\d+)", header).group(1))
    assert lib.mgpu_abi_version() == version
    cfg = binding.Config()
    lib.mgpu_config_defaults(C.byref(cfg))
    assert cfg.abi_version == version
    cfg.abi_version = version - 1
    ctx = C.c_void_p()
    lib.mgpu_create.argtypes = [C.POINTER(binding.Config), C.POINTER(C.c_void_p)]
    lib.mgpu_create.restype = C.c_int
    assert lib.mgpu_create(C.byref(cfg), C.byref(ctx)) == -1 and not ctx.value            # MGPU_E_INVAL, before any device is looked for
    cfg.abi_version = version
    cfg.chunk_buffers = 1 << 20
    assert lib.mgpu_create(C.byref(cfg), C.byref(ctx)) == -1 and not ctx.value
    cfg.chunk_buffers = 0
    for thr in (0, -5, 4096, 70000):                   # the sweep's 16-bit coefficient / 32-bit accumulators: 1 .. 4095 (the reference clamps to 40 .. 400)
        cfg.preamble_threshold = thr
        assert lib.mgpu_create(C.byref(cfg), C.byref(ctx)) == -1 and not ctx.value
    assert binding.ABI_VERSION == version             # the ctypes mirrors name the header revision they were written against
    # a host whose struct ends 8 bytes earlier (round 3's): the bytes behind it stay untouched
    lib.mgpu_config_defaults_abi.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.mgpu_config_defaults_abi.restype = None
    buf = (C.c_uint8 * (C.sizeof(binding.Config) + 8))(*([0xA5] * (C.sizeof(binding.Config) + 8)))
    lib.mgpu_config_defaults_abi(buf, C.sizeof(binding.Config) - 8, 3)
    assert bytes(buf[C.sizeof(binding.Config) - 8:]) == b"\xa5" * 16
    # and the C form: the macro passes the compiling host's own size and version
    src = tmp_path / "d.c"
    src.write_text('#include <stdio.h>\n#include "modes_gpu.h"\nint main(void){struct mgpu_config c; mgpu_config_defaults(&c);'
                   'printf("%u %u %u\\n", c.abi_version, (unsigned) MGPU_ABI_VERSION, mgpu_abi_version()); return 0;}\n')
    exe = tmp_path / "d"
    libdir = os.path.dirname(readsb_amd.lib_path())
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(helpers.ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-l:" + os.path.basename(readsb_amd.lib_path()), "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == [str(version)] * 3


def test_no_cpu_fallback(built):
    """Without a GPU the product must fail loudly, never compute on the CPU."""
    import readsb_amd
    lib = readsb_amd.load_library()
    if lib.mgpu_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(readsb_amd.MgpuError, match="no usable HIP device"):
        readsb_amd.Demodulator()


def test_host_crc_matches_reference_tables(built):
    import readsb_amd
    lib = readsb_amd.load_library()
    assert (lib.mgpu_crc_table_size(1, 56), lib.mgpu_crc_table_size(1, 112)) == (51, 107)
    assert (lib.mgpu_crc_table_size(2, 56), lib.mgpu_crc_table_size(2, 112)) == (1326, 3831)
    assert (lib.mgpu_crc_table_size(0, 56), lib.mgpu_crc_table_size(0, 112)) == (0, 0)
    for nfix in (1, 2):
        for bits in (56, 112):
            gold = TABLES[f"nfix{nfix}_{bits}"]
            for syn, n, b0, b1 in gold:
                x, y = C.c_int(), C.c_int()
                assert lib.mgpu_crc_diagnose(nfix, int(syn), bits, C.byref(x), C.byref(y)) == n
                assert (x.value, y.value) == (b0, b1)
    rng = np.random.default_rng(0)
    olib = helpers.oracle_lib()
    olib.modes_oracle_crc_init(2)
    for _ in range(2000):
        syn = int(rng.integers(1, 1 << 24))
        x, y, u, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert lib.mgpu_crc_diagnose(2, syn, 112, C.byref(x), C.byref(y)) == olib.modes_oracle_diagnose(syn, 112, C.byref(u), C.byref(v))
    single = TABLES["single_bit_syndrome_112"]
    for k in range(112):
        m = np.zeros(14, dtype=np.uint8)
        m[k >> 3] = 0x80 >> (k & 7)
        assert lib.mgpu_crc_checksum(m.ctypes.data, 112) == single[k]
    for _ in range(200):
        m = rng.integers(0, 256, size=14, dtype=np.uint8)
        for bits in (56, 112):
            assert lib.mgpu_crc_checksum(m.ctypes.data, bits) == olib.modes_oracle_checksum(m.ctypes.data, bits)


def test_host_uc8_table_matches_reference(built):
    import readsb_amd
    lib = readsb_amd.load_library()
    tab = np.ctypeslib.as_array(lib.mgpu_uc8_table(), shape=(65536,))
    assert np.array_equal(tab.reshape(256, 256), TABLES["uc8_mag_by_i_q"])


WALK_CASES = [(1, 12, 256, 4, 50), (2, 12, 256, 8, 200), (3, 30, 128, 3, 20), (4, 8, 512, 6, 500), (5, 40, 64, 5, 10),
              (6, 20, 300, 7, 100), (7, 6, 512, 2, 2000), (8, 50, 32, 4, 5), (9, 3, 2400, 4, 100), (10, 4, 1500, 16, 300),
              # the pipeline's own shape (512 and 1024 buffers in 8 ranges), and streams on which a first version of the ranges'
              # handling of an expiry inside the chunk went wrong: many aircraft, the table growing behind the expiry, an address
              # taken for refreshed by an earlier range's frame that was not
              (11, 40, 512, 8, 200), (12, 40, 1024, 8, 200), (314295693908, 30, 256, 5, 3000), (172793523793, 17, 1024, 6, 1000),
              (1071970496161, 30, 512, 15, 1000), (947893337438, 12, 700, 16, 3000)]


def test_parallel_walk_equals_serial_walk(built):
    """Host logic, no GPU: the speculative buffer-range walk (real threads) must make exactly the decisions
    of the serial walk on streams whose aircraft come, expire from the ICAO filter and return (chunks shorter
    and longer than the filter's 60 s clock) — and the cases together must have exercised both outcomes of
    the speculation (range committed / batch restarted at the range)."""
    import readsb_amd
    lib = C.CDLL(readsb_amd.lib_path())
    f = lib.mgpu_selftest_walk
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    f.restype = C.c_int
    held = []
    for case in WALK_CASES:
        pm = C.c_uint32(0)
        assert f(*case, C.byref(pm)) == 0, f"parallel walk differs from the serial walk for {case}"
        held.append(pm.value)
    # permille of chunks whose ranges all committed in the first batch: some did, some needed a restart
    assert any(h > 500 for h in held) and any(h < 1000 for h in held), held


@pytest.mark.parametrize("seed,nchunks,nbuf,naircraft", [
    (11, 40, 64, 60),        # 3.5 s chunks: an expiry every 17th chunk, aircraft arriving all the time
    (12, 12, 512, 300),      # 28 s chunks, enough aircraft for the filter table to grow twice (86, 171 addresses)
    (13, 6, 1200, 40),       # chunks longer than the filter's 60 s clock: a chunk two expiries fall into is walked serially
    (14, 60, 16, 500),       # many small chunks, a large population
])
def test_device_walk_model_equals_serial_walk(built, seed, nchunks, nbuf, naircraft):
    """Host logic, no GPU: the device walk's fixed point over per-buffer walks (kernels/walk.inc restated in
    Resolver::device_walk_model) and its premise check (Resolver::apply_device_walk) make exactly the serial walk's decisions
    — and most chunks are decided by it, in at most three walks."""
    import readsb_amd
    lib = C.CDLL(readsb_amd.lib_path())
    f = lib.mgpu_selftest_device_walk
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    f.restype = C.c_int
    st = (C.c_uint64 * 4)()
    assert f(seed, nchunks, nbuf, naircraft, 3, st) == 0, f"device walk model differs from the serial walk: {list(st)}"
    by_model, serial, walks, most = list(st)
    print("model:", by_model, "serial:", serial, "walks:", walks, "most:", most)
    assert by_model + serial == nchunks and most <= 3
    if nbuf <= 512:
        assert by_model >= nchunks // 2, list(st)


def test_bench_roofline_helpers(tmp_path):
    """bench.py's informative VALU-issue figure parses an SQ counter summary collected with THIS device code (the summary
    carries the hash of the kernel sources), refuses one of other code, and degrades to None, never raises."""
    import bench
    good = tmp_path / "sq.txt"
    good.write_text("# kernel_source_sha: %s\nk_sweep\n   SQ_INSTS_VALU   14000000   (n=8)\nk_slice\n   SQ_INSTS_VALU   38000000   (n=8)\n" % bench.kernel_source_sha())
    v = bench.valu_issue("k_sweep", 0.040, 67108864, str(good))
    assert v and v["wave_insts_per_launch"] == 14000000 and 0.2 < v["frac"] < 1.0
    v2 = bench.valu_issue("k_slice", 0.123, 67108864, str(good))
    assert v2 and v2["wave_insts_per_launch"] > v["wave_insts_per_launch"]
    stale = tmp_path / "stale.txt"
    stale.write_text(good.read_text().replace(bench.kernel_source_sha(), "0123456789abcdef"))
    assert bench.valu_issue("k_sweep", 0.040, 67108864, str(stale)) is None            # counters of other code are not this run's
    assert bench.valu_issue("k_sweep", 0.040, 67108864, "/nonexistent/file") is None and bench.valu_issue("k_sweep", 0.0, 1, str(good)) is None


def test_device_index_on_its_numa_node(built, tmp_path):
    """Host logic, no GPU: the pipeline's threads pick their L3 groups by the device's place among the node's GPUs of its NUMA node
    (sysfs), not by the HIP ordinal — a fake /sys/bus/pci/devices with eight accelerators on two nodes and a few other functions."""
    import readsb_amd
    lib = C.CDLL(readsb_amd.lib_path())
    f = lib.mgpu_selftest_device_index
    f.argtypes = [C.c_char_p, C.c_char_p]
    f.restype = C.c_int
    gpus = ["0000:0a:00.0", "0000:23:00.0", "0000:5a:00.0", "0000:72:00.0", "0000:8b:00.0", "0000:a4:00.0", "0000:d9:00.0", "0000:f1:00.0"]

    def put(name, vendor, device, cpus):
        d = tmp_path / name
        d.mkdir()
        (d / "vendor").write_text(vendor + "\n")
        (d / "device").write_text(device + "\n")
        (d / "local_cpulist").write_text(cpus + "\n")

    for k, g in enumerate(gpus):
        put(g, "0x1002", "0x75a3", "0-63,128-191" if k < 4 else "64-127,192-255")
    put("0000:00:00.0", "0x1022", "0x14a4", "0-63,128-191")          # a host bridge
    put("0000:0b:00.0", "0x1002", "0x1234", "0-63,128-191")          # another function of the same vendor
    put("0000:24:00.0", "0x15b3", "0x1021", "0-63,128-191")          # a NIC
    for k, g in enumerate(gpus):
        assert f(str(tmp_path).encode(), g.encode()) == k % 4, g
    assert f(str(tmp_path).encode(), b"0000:ff:00.0") == -1
    assert f(None, b"x") == -2
