"""-m gpu: the WHOLE reference program with its demodulator replaced by the GPU library at link time.
oracle/_ref/full/readsb_full_gpu = the reference's own objects + readsb_amd/host/readsb_tree/demod_gpu_wrap.c with
`ld --wrap=demodulate2400 --wrap=demodulate2400AC --wrap=icaoFilterExpire --wrap=icaoFilterAdd` (no source change) +
libmodes_gpu.so; it is run like readsb itself (`--device-type ifile --ifile … --dump-beast …`) next to the unmodified program
on the same file, and the beast stream its network layer writes and its demodulator statistics must be identical.
tests/test_dropin.py exercises the same adapter on the CPU against a stand-in (and explains the reference program's two
start-up orders, which the adapter follows); this is the product library on the GPU.

Both binaries are built in the development container (they contain the reference's objects) and travel with oracle/_ref."""
import os

import pytest

import helpers
import test_dropin as td

GPU_EXE = os.path.join(helpers.ORACLE_DIR, "_ref", "full", "readsb_full_gpu")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(td.FULL) and os.path.exists(GPU_EXE)), reason="oracle/_ref/full binaries not present")]


@pytest.mark.parametrize("kw,opts,nfix", [
    (dict(seconds=4.0, seed=301, rate=1800.0, naircraft=24), [], 1),
    (dict(seconds=3.0, seed=302, rate=700.0, dense=2, naircraft=24), ["--aggressive"], 2),
    (dict(seconds=3.0, seed=302, rate=700.0, dense=2, naircraft=24), ["--aggressive", "--modeac"], 2),
    # 200 aircraft: the ICAO filter resizes, the program's two start-up orders give different outputs (round 1's failure:
    # the library then always flipped after buffer 0, the program on the GPU host did not)
    (dict(seconds=4.0, seed=301, rate=1800.0, naircraft=200), [], 1),
    (dict(seconds=6.0, seed=311, rate=2500.0, naircraft=400), ["--aggressive"], 2),
])
def test_reference_program_on_the_gpu_library(built, tmp_path, kw, opts, nfix):
    iq = helpers.synth(**kw)
    td.check_wrapped_against_reference(GPU_EXE, iq, "UC8", 0, opts, tmp_path, nfix, timeout=120)


def test_reference_program_with_the_gpu_converter_too(built, tmp_path):
    """READSB_GPU_CONVERT=1: the reader thread's iq_convert_fn is mgpu_convert() as well (`--wrap=init_converter`)."""
    iq = helpers.synth(seconds=3.0, seed=305, rate=1500.0, fmt=2, naircraft=24)
    env = dict(os.environ, READSB_GPU_CONVERT="1")
    td.check_wrapped_against_reference(GPU_EXE, iq, "SC16Q11", 2, ["--aggressive"], tmp_path, 2, env=env, timeout=120)
    assert "init_converter: using the GPU library" in td.check_wrapped_against_reference.wrapped_output
