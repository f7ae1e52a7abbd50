"""CPU: the committed host reference of the config-3 end-to-end test (tests/golden/config3_e2e_ref.npz, tests/e2e_config3.py) is the
one the test's constants ask for, and it has the properties the id gate demands of the REFERENCE before it looks at the HIP run
(tests/test_gpu_configs.py::_id_gate): clear decisions, ids that follow the image, not one id repeated."""
import numpy as np

import e2e_config3 as R


def test_fixture_matches_the_test_constants():
    assert R.FIXTURE.exists(), "run tests/golden/make_config3_e2e.py"
    z = np.load(R.FIXTURE)
    assert np.array_equal(z["header"], R._header())          # format, sizes, seeds: a changed seed needs a new file
    assert z["e32_noise_vis"].shape == (R.NQ, R.E) and z["e32_noise_vis"].dtype == np.float32
    assert z["logits32"].shape == (1, R.VOCAB) and z["logits32"].dtype == np.float32
    assert np.isfinite(z["e32_noise_vis"]).all() and np.isfinite(z["logits32"]).all()
    assert 0.3 < float(z["aligned_rel_rms"]) < 2.0            # the second volume moves the visual tokens by tens of percent
    # only the fp32 side and the thresholds are in the file: the bf16 yardstick is computed next to the HIP run
    assert not [k for k in z.files if "16" in k]


def test_reference_side_of_the_id_gate():
    z = np.load(R.FIXTURE)
    ids = {v: z[f"ids_{v}"].tolist() for v in ("noise", "smooth")}
    mg = {v: z[f"margins_{v}"].tolist() for v in ("noise", "smooth")}
    thr = {v: float(z[f"thr_{v}"]) for v in ("noise", "smooth")}
    for v in ids:
        assert len(ids[v]) == len(mg[v]) == R.NEW and thr[v] > 0
        assert all(0 <= t < R.VOCAB for t in ids[v])
        assert sum(m > thr[v] for m in mg[v]) >= 3, (v, mg[v], thr[v])               # the reference decides clearly
    part = [t for t in range(R.NEW) if ids["noise"][t] != ids["smooth"][t] and mg["noise"][t] > thr["noise"] and mg["smooth"][t] > thr["smooth"]]
    assert part, (ids, mg, thr)                                                          # ... differently for the two volumes
    assert len(set(ids["noise"])) > 1 or len(set(ids["smooth"])) > 1                     # ... and not one id for ever
    # the first greedy id is the argmax of the stored first-step logits of the benchmark's volume
    assert int(z["logits32"][0].argmax()) == ids["noise"][0]


def test_fixture_was_made_with_this_oracle():
    """An edit of oracle/u2_oracle.py that changes what it computes invalidates the file: the fingerprint stored by
    the maker -- the oracle's whole path at a tiny size on name-seeded parameters -- must be what the oracle computes NOW.  On failure:
    `python tests/golden/make_config3_e2e.py` (the GPU test would meanwhile fall back to the live reference, e2e_config3.load)."""
    z = np.load(R.FIXTURE)
    assert "oracle_fingerprint" in z.files
    live = R.oracle_fingerprint()
    assert live.shape == (9,) and np.isfinite(live).all() and live[0] > 0.1
    assert R.fingerprint_matches(z["oracle_fingerprint"], live), (z["oracle_fingerprint"], live)
    # the fingerprint notices arithmetic: one rounding-point change (the DiffTS loop form in bf16 is not one -- fp32 here; a scaled
    # softmax temperature is) moves it far beyond the host-to-host tolerance
    import oracle.u2_oracle as O
    orig = O.diff_token_selection
    try:
        O.diff_token_selection = lambda sd, p, x, tau=1.0, loop_form=False: orig(sd, p, x, tau=1.05, loop_form=loop_form)
        assert not R.fingerprint_matches(z["oracle_fingerprint"], R.oracle_fingerprint())
    finally:
        O.diff_token_selection = orig
