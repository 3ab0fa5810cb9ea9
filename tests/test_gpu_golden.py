"""-m gpu: the HIP path against the committed golden fixtures (reference output, tests/golden/)."""
import json
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
INDEX = json.load(open(os.path.join(helpers.GOLDEN_DIR, "index.json")))


@pytest.mark.parametrize("name", sorted(INDEX))
def test_hip_path_matches_golden(built, name):
    import readsb_amd
    c = INDEX[name]
    iq = helpers.synth(**c["synth"])
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, name + ".msgs.npy"))
    d = readsb_amd.Demodulator(fmt=c["fmt"], nfix_crc=c["nfix"], fix_df=c["fixdf"], preamble_threshold=c["thr"],
                               startup_time_ms=helpers.STARTUP_MS, max_samples=32 * 131072)
    got, cnt = d.demodulate_capture(iq)
    d.close()
    helpers.assert_same_messages(got, gold)
    for f in helpers.COUNTER_FIELDS:
        assert np.asarray(cnt[f]).tolist() == c["stats"][f], f
