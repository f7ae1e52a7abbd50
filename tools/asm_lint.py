#!/usr/bin/env python3
"""Static hazard checks over the generated inline-asm loops (csrc/flash_dp_asm.inc, csrc/gemm_bt_asm.inc).

hipcc inserts the gfx950 software wait states for its own code; inside an asm block nobody does, so the generators
place them by hand.  This linter re-checks the rules those generators rely on, on the final text:

  R1  a VALU that reads the result of a transcendental (v_exp_f32, v_rcp_f32, ...) needs one wait state in between
      (an independent instruction or s_nop).
  R2  an instruction that uses M0 (buffer_load ... lds) must not directly follow the SALU write of M0.
  R3  a VALU / LDS / VMEM instruction that reads a VGPR written by an MFMA needs >= 18 wait states after a 16-pass MFMA
      (v_mfma_f32_32x32x16_bf16 is 8 passes: >= 11; 18 is what the generators use).  Checked inside straight-line
      regions by counting issued instructions (each is >= 1 wait state; s_nop N counts N + 1).  MFMA -> MFMA chaining
      on the same accumulator is exempt (hardware forwards SrcC).
  R4  v_permlane32_swap needs 2 wait states after a VALU write of one of its operands, and its results 2 before use.
  R5  every s_waitcnt lgkmcnt(N) inside an MFMA slot sequence must have N <= number of LDS reads issued since the last
      lgkmcnt(0) drain (a larger N would be a wait that can never protect anything: a generator bug).
  R6  the result of a packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two passes) must not be read by the
      very next instruction (hipcc separates such pairs with `s_nop 0`; the drain forms' GELU interleaves four chains instead).
Accumulator registers named literally (a5, a[16:31]: the drain forms) are tracked like VGPRs for R3.

    python tools/asm_lint.py u2tokenizer_amd/csrc/flash_dp_asm.inc u2tokenizer_amd/csrc/gemm_bt_asm.inc
"""
import re
import sys

TRANS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32")


def regs(tok):
    """registers named by one operand token: 'v12' -> {v12}; 'v[4:7]' -> {v4..v7}; '-%[mr0]' -> {%mr0}; else empty"""
    tok = tok.strip().lstrip("-").strip("|").split(" ")[0]
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {f"{m.group(1)}{m.group(2)}"}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"%\[(\w+)\]", tok)
    if m:
        return {"%" + m.group(1)}
    return set()


def parse(line):
    """-> (opcode, dst register set, src register set)"""
    op, _, rest = line.partition(" ")
    rest = re.sub(r"\boffset:\d+|\boffen\b|\blds\b|\boff\b|\b(op_sel_hi|op_sel|neg_lo|neg_hi):\[[\d,]+\]|\bclamp\b", "", rest)
    toks = [t for t in re.split(r",\s*", rest.strip()) if t]
    if not toks:
        return op, set(), set()
    if op.startswith(("ds_write", "buffer_store", "global_store", "s_", "buffer_load")) or op.startswith("v_cmp"):
        return op, set(), set().union(*[regs(t) for t in toks])
    if op.startswith("v_permlane32_swap"):
        r = set().union(*[regs(t) for t in toks])
        return op, r, r
    return op, regs(toks[0]), set().union(*[regs(t) for t in toks[1:]]) if len(toks) > 1 else set()


def lint(name, lines):
    errs = []
    last_trans = None            # registers written by the previous instruction if it was transcendental
    last_pk = None               # ... if it was packed fp32 math
    prev_wrote_m0 = False
    mfma_age = {}                # vgpr -> wait states since an MFMA wrote it
    valu_age = {}                # vgpr -> wait states since a VALU wrote it (for R4)
    swap_age = {}                # vgpr -> wait states since a permlane swap wrote it
    reads_since_drain = 0
    label_at = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    snaps = {}                   # label -> list of (mfma_age, valu_age, swap_age) snapshots taken at forward branches to it
    for n, line in enumerate(lines):
        if line.endswith(":"):
            # paths join: ages = the youngest over the fall-through path and every forward branch that lands here; a label
            # that is (also) a backward-branch target starts from "old enough" (the loop body ends in SALU bookkeeping)
            lab = line[:-1]
            backward = any(re.match(r"s_c?branch\w* " + re.escape(lab) + r"$", l) for l in lines[n:])
            if backward:
                mfma_age.clear(); valu_age.clear(); swap_age.clear()
            for (ma, va, sa) in snaps.get(lab, []):
                for cur, snap in ((mfma_age, ma), (valu_age, va), (swap_age, sa)):
                    for r, a in snap.items():
                        cur[r] = min(cur.get(r, 10 ** 9), a)
            last_trans = None
            last_pk = None
            prev_wrote_m0 = False
            continue
        mb = re.match(r"s_c?branch\w* (\S+)$", line)
        if mb and label_at.get(mb.group(1), -1) > n:
            snaps.setdefault(mb.group(1), []).append((dict(mfma_age), dict(valu_age), dict(swap_age)))
        if re.match(r"s_branch ", line):
            # unconditional: nothing falls through
            mfma_age = {}; valu_age = {}; swap_age = {}
            continue
        op, dst, src = parse(line)
        ws = 1
        if op == "s_nop":
            ws = int(line.split()[1]) + 1
        is_valu = op.startswith("v_") and not op.startswith("v_mfma")
        is_mem = op.startswith(("ds_", "buffer_", "global_"))
        # R1
        if last_trans and is_valu and (src & last_trans):
            errs.append(f"{name}:{n}: R1 '{line}' reads a transcendental result without a wait state")
        # R6
        if last_pk and (is_valu or is_mem or op.startswith("v_mfma")) and (src & last_pk):
            errs.append(f"{name}:{n}: R6 '{line}' reads a packed-math result in the next issue slot")
        # R2
        if prev_wrote_m0 and " lds" in line + " ":
            errs.append(f"{name}:{n}: R2 '{line}' directly follows the M0 write")
        # R3
        if is_valu or is_mem:
            for r in src:
                if r in mfma_age and mfma_age[r] < 18:
                    errs.append(f"{name}:{n}: R3 '{line}' reads {r} {mfma_age[r]} wait states after an MFMA wrote it")
                    break
        # R4
        if op.startswith("v_permlane32_swap"):
            for r in src:
                if valu_age.get(r, 99) < 2:
                    errs.append(f"{name}:{n}: R4 '{line}' swaps {r} {valu_age[r]} wait states after its VALU write")
        elif is_valu:
            for r in src:
                if swap_age.get(r, 99) < 2:
                    errs.append(f"{name}:{n}: R4 '{line}' reads {r} {swap_age[r]} wait states after the swap")
        # R5
        m = re.match(r"s_waitcnt .*lgkmcnt\((\d+)\)", line)
        if m:
            k = int(m.group(1))
            if k > reads_since_drain:
                errs.append(f"{name}:{n}: R5 '{line}' allows {k} outstanding reads but only {reads_since_drain} were issued")
            reads_since_drain = min(reads_since_drain, k)
        if op.startswith("ds_read"):
            reads_since_drain += 1
        # age bookkeeping
        for d in (mfma_age, valu_age, swap_age):
            for r in list(d):
                d[r] += ws
        if op.startswith("v_mfma"):
            for r in dst:
                mfma_age[r] = 0
        elif op.startswith("v_permlane32_swap"):
            for r in dst:
                swap_age[r] = 0
                mfma_age.pop(r, None)
        elif is_valu or op.startswith("ds_read"):
            for r in dst:
                valu_age[r] = 0
                mfma_age.pop(r, None)
        last_trans = dst if op.split("_e64")[0] in TRANS else None
        last_pk = dst if op.startswith(("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")) else None
        prev_wrote_m0 = bool(re.match(r"s_(mov_b32|add_u32) m0,", line))
    return errs


def blocks(path):
    text = open(path).read()
    for m in re.finditer(r"#define (\w+_ASM_TEXT\w*) \\\n(.*?)(?=\n#define|\n// clang-format on|\Z)", text, re.S):
        yield m.group(1), re.findall(r'"(.*)\\n"', m.group(2))


def main(paths):
    errs = []
    for p in paths:
        for name, lines in blocks(p):
            errs += lint(name, lines)
    for x in errs:
        print(x)
    return errs


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1:]) else 0)
