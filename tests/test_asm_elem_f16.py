"""CPU: tools/asm_elem_f16.py derives the IEEE-half text of the generated asm loops from their bf16 text at build time (the f16
build of the library, csrc/Makefile).  Every rule must hit, nothing bf16 may survive, and nothing else may change."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "u2tokenizer_amd" / "csrc"
sys.path.insert(0, str(ROOT / "tools"))
import asm_elem_f16  # noqa: E402


def test_derived_text_differs_only_in_the_element_type():
    # (gemm_bt_asm.inc, round 6: the drain forms pack the held tile -- v_cvt_pk -- and the GELU form unpacks it again: 4 waves x 12 drain
    #  bodies x 8 element pairs)
    for inc, unpack_pairs, thr in (("gemm_bt_asm.inc", 384, 0), ("flash_dp2_asm.inc", 128, 72), ("tokattn_pv_asm.inc", 0, 0)):
        src = (CSRC / inc).read_text()
        out = asm_elem_f16.convert(src)
        a, b = src.splitlines(), out.splitlines()
        assert len(a) == len(b)
        n_mfma = n_cvt = n_lo = n_hi = n_thr = 0
        for x, y in zip(a, b):
            if x == y:
                continue
            if "v_mfma_f32_" in x:
                assert y == x.replace("_bf16", "_f16")
                n_mfma += 1
            elif "v_cvt_pk_bf16_f32" in x:
                assert y == x.replace("v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32")
                n_cvt += 1
            elif "v_lshlrev_b32" in x:
                m = re.search(r"v_lshlrev_b32 (v\d+), 16, (v\d+)", x)
                assert m and f"v_cvt_f32_f16 {m.group(1)}, {m.group(2)}" in y
                n_lo += 1
            elif "0xffff0000" in x:
                m = re.search(r"v_and_b32 (v\d+), 0xffff0000, (v\d+)", x)
                assert m and f"v_cvt_f32_f16_sdwa {m.group(1)}, {m.group(2)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" in y
                n_hi += 1
            elif "0x5f800000" in x:
                assert y == x.replace("0x5f800000", "0x47000000")      # 2^64 -> 2^15: half overflows at 65504
                n_thr += 1
            else:
                assert x.startswith("// GENERATED") and y.startswith("// DERIVED"), (x, y)
        assert n_mfma > 0 and n_lo == n_hi == unpack_pairs and n_thr == thr and (n_cvt > 0) == (inc != "tokattn_pv_asm.inc")
        assert "bf16" not in re.sub(r"//.*", "", out)
