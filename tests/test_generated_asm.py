"""The two inline-asm loops (flash attention KV loop, big-tile GEMM K loop) are generated: the committed .inc files must
be what the generators in tools/ produce, so that a schedule change is always made in the generator."""
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
CSRC = ROOT / "u2tokenizer_amd" / "csrc"


def _run(*args):
    return subprocess.run([sys.executable, *args], cwd=ROOT, check=True, capture_output=True, text=True).stdout


def test_gemm_bt_asm_is_generated():
    assert _run("tools/gen_gemm_bt_asm.py") == (CSRC / "gemm_bt_asm.inc").read_text()


def test_flash_dp_asm_is_generated():
    want = _run("tools/gen_flash_dp_asm.py") + _run("tools/gen_flash_dp_asm.py", "--timed")
    assert want == (CSRC / "flash_dp_asm.inc").read_text()


def test_gemm_bt_schedule_issues_every_piece_once():
    """per wave and K tile: 8 A pieces + 2 NJ B pieces, 16 NJ MFMAs, 4 (4 + NJ) fragment reads inside the loop"""
    import re
    text = (CSRC / "gemm_bt_asm.inc").read_text()
    for nj in (4, 3):
        body = re.search(rf"#define GEMM_BT_ASM_TEXT_NJ{nj} \\\n(.*?)\n#define", text, re.S).group(1)
        lines = re.findall(r'"(.*)\\n"', body)
        for w in range(4):
            i0 = next(k for k, l in enumerate(lines) if l.startswith(f".Lbt_loop{w}_"))
            i1 = next(k for k, l in enumerate(lines) if l.startswith(f"s_cbranch_scc1 .Lbt_loop{w}_"))
            loop = lines[i0:i1]
            assert sum("buffer_load_dwordx4" in l for l in loop) == 8 + 2 * nj
            assert sum(l.startswith("v_mfma") for l in loop) == 16 * nj
            assert sum(l.startswith("ds_read_b128") for l in loop) == 4 * (4 + nj)
            assert sum(l.startswith("s_barrier") for l in loop) == 1
