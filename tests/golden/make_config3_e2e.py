"""Makes tests/golden/config3_e2e_ref.npz: the HOST reference of the config-3 end-to-end test (tests/e2e_config3.py), so that the
GPU suite does not spend a quarter of its wall time re-deriving it.  Runs on the CPU only (oracle + HF decoder):

    python tests/golden/make_config3_e2e.py            # ~10 minutes on 8 cores, ~2 on a pool box's host

Checks before it writes: (1) what e2e_config3.load() rebuilds from the file is BIT-EQUAL to what reference_live() returned, and
bf16_noise_run() -- the part the test keeps live -- returns reference_live()'s bf16 tensors bit for bit; (2) ids / margins agree
with the ones a GPU box's host recorded when the same reference ran live inside the test (profiles/r05_parity.json, if present;
the thresholds come from a bf16 run and differ between CPU generations: printed, not compared).

    python tests/golden/make_config3_e2e.py --stamp ORACLE_AT_MAKE_TIME.py

adds the oracle fingerprint (e2e_config3.oracle_fingerprint, file format 2) to a file made before the fingerprint existed WITHOUT
re-running the reference: the argument is the oracle source the file was made with (`git show <commit>:oracle/u2_oracle.py`); the
stamp is the fingerprint computed WITH THAT SOURCE, so the file is accepted afterwards only if the current oracle computes the same.
(Round 6 stamped the round-5 file this way: the round's oracle edit -- the DiffTS loop form behind a flag -- left the fingerprint
bit-identical.)"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import e2e_config3 as R  # noqa: E402


def _builders():
    """mm_config / oracle_cfg of tests/test_gpu_configs.py without importing its GPU fixtures' side effects."""
    import test_gpu_configs as T
    return T.mm_config, T.oracle_cfg


def main():
    torch.set_grad_enabled(False)
    mm_config, oracle_cfg = _builders()
    t0 = time.perf_counter()
    s = R.setup(mm_config, oracle_cfg)
    table32 = s.m.get_input_embeddings().weight.detach().clone()
    print(f"model + inputs: {time.perf_counter() - t0:.0f} s", flush=True)
    t0 = time.perf_counter()
    ref = R.reference_live(s)
    print(f"reference: {time.perf_counter() - t0:.0f} s", flush=True)
    tmp = R.FIXTURE.with_suffix(".tmp.npz")
    R.save(ref, tmp)
    # (1) round trip: load() reads the fp32 embedding table of the model -- reference_live left s.m in bf16, so hand it the copy
    live16 = R.bf16_noise_run(s)
    for k in ("e16_noise", "logits16"):
        assert live16[k].dtype == ref[k].dtype and torch.equal(live16[k], ref[k]), k
    s.m.get_input_embeddings().weight.data = table32
    back = R.load(s, tmp)
    assert back is not None
    for k in ("e32_noise", "logits32"):
        assert back[k].dtype == ref[k].dtype and back[k].shape == ref[k].shape, (k, back[k].dtype, ref[k].dtype, back[k].shape)
        assert torch.equal(back[k], ref[k]), k
    assert back["refs"] == ref["refs"] and back["thrs"] == ref["thrs"] and back["aligned_rel_rms"] == ref["aligned_rel_rms"]
    # (2) against the live run on the GPU box's host
    rec = ROOT / "profiles" / "r05_parity.json"
    if rec.exists():
        d = json.loads(rec.read_text()).get("config3_E4096_256cube_end_to_end", {})
        for v in ("noise", "smooth"):
            if f"{v}:fp32_ids" in d:
                ids, mg, thr = d[f"{v}:fp32_ids"], d[f"{v}:fp32_top2_margins"], d[f"{v}:flip_threshold"]
                print(v, "ids", ref["refs"][v][0], "recorded", ids)
                print(v, "margins", [round(x, 4) for x in ref["refs"][v][1]], "recorded", [round(x, 4) for x in mg])
                print(v, "threshold", round(ref["thrs"][v], 4), "recorded", round(thr, 4))
                assert ref["refs"][v][0] == ids
                assert max(abs(a - b) for a, b in zip(ref["refs"][v][1], mg)) < 2e-2
    tmp.replace(R.FIXTURE)
    print("wrote", R.FIXTURE, R.FIXTURE.stat().st_size, "bytes")


def stamp(old_oracle_source):
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("_oracle_at_make_time", old_oracle_source)
    old = importlib.util.module_from_spec(spec)
    sys.modules["_oracle_at_make_time"] = old      # (dataclasses looks the module up while the class body is evaluated)
    spec.loader.exec_module(old)
    cur = R.O
    try:
        R.O = old
        fp = R.oracle_fingerprint()
    finally:
        R.O = cur
    z = dict(np.load(R.FIXTURE))
    assert int(z["header"][0]) == 1 and "oracle_fingerprint" not in z, "already stamped"
    z["header"] = R._header()
    z["oracle_fingerprint"] = fp
    np.savez(R.FIXTURE, **z)
    print("stamped", R.FIXTURE, fp, "current oracle matches:", R.fingerprint_matches(fp))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--stamp":
        stamp(sys.argv[2])
    else:
        main()
