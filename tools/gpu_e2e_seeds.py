#!/usr/bin/env python
"""VERDICT r4 "weak" #2: is the default ViT attention loop (flash double pipeline, mode 7) systematically further from the fp32
reference than the other two correct HIP orderings after the config-3 chain, or was 1.31 x one draw?  For each of N seeds
(parameters, volume and ids all re-drawn): oracle fp32 + oracle bf16 on the host, then the HIP path's inputs_embeds under
{flash_mode 7 (default), flash_mode 1 (128-row units), vit_flash 0 (unfused attention)}; prints / writes, per seed and form,
rel-rms distance to the fp32 reference divided by the bf16 reference's own distance -- over the whole spliced sequence and
over the 256 aligned-token rows alone (768 of the 1024 rows are bit-exact table lookups).

    python tools/gpu_e2e_seeds.py [N=5] [out.json]
"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from helpers import err_stats  # noqa: E402
from oracle import u2_oracle as O  # noqa: E402
from test_gpu_configs import build_path, mm_config, oracle_cfg  # noqa: E402
from u2tokenizer_amd import ops, synth  # noqa: E402

bf, D = torch.bfloat16, "cuda"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dst = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "gpurun_out" / "r05_e2e_seeds.json"
    torch.set_grad_enabled(False)
    ops.device_check()
    E, vocab, S, Lt = 4096, 4096, 1024, 1024
    c = mm_config(E, [32, 256, 256])
    oc = oracle_cfg(c)
    forms = (("flash_mode7_double_pipeline (default)", None, None, None), ("flash_mode1_128_row_units", "flash_mode", 1, 0),
             ("unfused_attention", "vit_flash", 0, 1))
    rows = []
    for seed in range(101, 101 + n):
        t0 = time.time()
        path, sd32, sd16 = build_path(c, vocab, seed)
        vol = synth.synth_volume(1, 8, c["image_size"], seed=seed, dtype=torch.float16)
        ids = synth.synth_ids(1, S, S - 24, vocab, seed=seed, name="input_ids")
        qids = synth.synth_ids(1, Lt, 40, vocab, seed=seed, name="question_ids")
        e32, _ = O.prepare_inputs_for_multimodal(sd32, sd32["model.embed_tokens.weight"], ids, vol.float(), qids, oc)
        e16, _ = O.prepare_inputs_for_multimodal(sd16, sd16["model.embed_tokens.weight"], ids, vol.to(bf), qids, oc)
        o_all = err_stats(e16.float(), e32)["rel_rms"]
        o_tok = err_stats(e16.float()[:, 1:257], e32[:, 1:257])["rel_rms"]
        row = {"seed": seed, "o16_vs_o32_rel_rms": o_all, "o16_vs_o32_rel_rms_token_rows": o_tok, "forms": {}}
        tower = path.holder.vision_tower
        for name, opt, val, back in forms:
            if opt:
                ops.set_option(opt, val)
            try:
                if hasattr(tower, "invalidate_feature_cache"):
                    tower.invalidate_feature_cache()
                emb = path.prepare_inputs_for_multimodal(ids.to(D), None, None, None, None, vol.to(D), qids.to(D))[4].float().cpu()
            finally:
                if opt:
                    ops.set_option(opt, back)
            h_all = err_stats(emb, e32)["rel_rms"]
            h_tok = err_stats(emb[:, 1:257], e32[:, 1:257])["rel_rms"]
            row["forms"][name] = {"hip_vs_o32_rel_rms": h_all, "ratio": h_all / o_all, "ratio_token_rows": h_tok / o_tok}
        row["seconds"] = time.time() - t0
        rows.append(row)
        print(seed, {k: round(v["ratio"], 3) for k, v in row["forms"].items()}, "token rows", {k: round(v["ratio_token_rows"], 3) for k, v in row["forms"].items()},
              f"o16 {o_all:.5f}", flush=True)
        del path, sd32, sd16
        torch.cuda.empty_cache()
    summary = {name: {"mean_ratio": sum(r["forms"][name]["ratio"] for r in rows) / len(rows),
                      "max_ratio": max(r["forms"][name]["ratio"] for r in rows),
                      "mean_ratio_token_rows": sum(r["forms"][name]["ratio_token_rows"] for r in rows) / len(rows)} for name, *_ in forms}
    print("summary", json.dumps(summary, indent=1))
    dst.parent.mkdir(exist_ok=True)
    dst.write_text(json.dumps({"what": "config 3 (E 4096, 256^3, 4-layer lively tokenizer) inputs_embeds: rel-rms distance to the fp32 "
                               "oracle / the bf16 oracle's own distance, per seed and ViT attention form", "rows": rows, "summary": summary}, indent=1))


if __name__ == "__main__":
    main()
