"""CPU: the oracle restatement (oracle/elas_oracle.cpp) against the golden vectors the compiled reference
produced (tests/golden/, made by tests/golden/make_golden.py).  Bit-exact at every stage."""
import numpy as np
import pytest

import util
from pyoracle import ElasParams

DIG = util.digests()
FAST = [k for k in DIG if k != "synth5000_4kstrip_d192"]


@pytest.mark.parametrize("name", sorted(DIG))
def test_inputs_reproducible(name):
    """Fixture images / the seeded generator give exactly the bytes the reference was run on."""
    e = DIG[name]
    L, R = util.case_images(e)
    assert list(L.shape) == e["shape"]
    assert [util.sha(L), util.sha(R)] == e["input_sha256"]


@pytest.mark.parametrize("name", sorted(DIG))
def test_oracle_matches_reference_digests(oracle, name):
    e = DIG[name]
    L, R = util.case_images(e)
    n = oracle.run_stages(util.case_params(e, ElasParams), L, R)
    assert n == e["n_support"]
    bad = [k for k in util.STAGES if util.sha(oracle.stage(k)) != e["stages"][k]]
    assert not bad, "stages differ from the reference: %s" % bad


@pytest.mark.parametrize("name", sorted(FAST))
def test_oracle_matches_reference_arrays(oracle, name):
    e = DIG[name]
    gold = util.golden_npz(name)
    if not gold:
        pytest.skip("digest-only case")
    L, R = util.case_images(e)
    oracle.run_stages(util.case_params(e, ElasParams), L, R)
    for k, g in gold.items():
        o = oracle.stage(k)
        if k.startswith("wta"):
            assert np.array_equal(o, o.astype(np.int16).astype(np.float32)), "WTA map must be integer-valued"
            o = o.astype(np.int16)
        assert o.size == g.size, k
        assert np.array_equal(o.ravel().view(np.uint8), g.ravel().view(np.uint8)), k


def test_process_seam_equals_stages(oracle):
    """orc_process (Elas::process semantics) == the staged run, and D2 is the L/R-checked right map."""
    e = DIG["kitti0_crop_d64"]
    L, R = util.case_images(e)
    p = util.case_params(e, ElasParams)
    D1, D2, _ = oracle.process(p, L, R)
    assert util.sha(D1) == e["stages"]["final1"]
    assert util.sha(D2) == e["stages"]["final2"]


def test_too_few_support_points_leaves_outputs_untouched(oracle):
    """elas.cpp:63-69: <3 support points -> early return, caller's maps untouched (zeros from the driver).
    Needs add_corners=0 (ROBOTICS); with MIDDLEBURY the six corner points alone keep the pipeline going."""
    L = np.zeros((60, 100), np.uint8)
    p = ElasParams.preset("robotics")
    p.disp_max = 63
    D1, D2, _ = oracle.process(p, L, L)
    assert not D1.any() and not D2.any()
