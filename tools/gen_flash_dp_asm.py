#!/usr/bin/env python3
"""Generates u2tokenizer_amd/csrc/flash_dp_asm.inc: the KV loop of the double-pipeline flash attention kernel
(attn.hip, flash_dpa_kernel) as ONE inline-asm block for gfx950.

Why generated assembly: the loop interleaves, per "slot", one MFMA of query block x with the 7 softmax VALU of one
score pair of query block y and an LDS fragment read 3 slots ahead.  hipcc cannot be made to keep that order AND keep
the six 16-register accumulators in place at <= 256 VGPRs (two waves per SIMD): every formulation tried through
inline-asm operands either reordered the slots, rotated the accumulators through extra tuples, or spilled (DESIGN.md,
flash study).  Here every register of the loop is fixed by hand and the compiler only sees clobbers.

Layout of the fixed registers (all clobbered): see REG below.  The block zeroes its state, runs prologue / tile loop /
epilogue, and leaves the four O^T accumulator tuples in LDS (the K/V ring is dead by then) for the C++ epilogue.

    python tools/gen_flash_dp_asm.py > u2tokenizer_amd/csrc/flash_dp_asm.inc
"""
import sys

TIMED = "--timed" in sys.argv   # diagnostics build: s_memtime deltas of the loop sections -> 5 x uint64 at %[dbg]
PF = 3          # fragment reads in flight
NSLOT = 4       # LDS ring slots (16 KB each: K tile 8 KB | V^T tile 8 KB)
AHEAD = 3       # tiles in flight

# ---- fixed VGPRs -------------------------------------------------------------------------------------------------
AB = [None, 112, 113, 114]          # ab[1..3]; ab[0] is an operand
ADR = 115
O = {(0, 0): 116, (0, 1): 132, (1, 0): 148, (1, 1): 164}   # O^T accumulators [block][nb], 16 regs each
SC = {0: 180, 1: 196}                                        # raw scores of a 32-key half, 16 regs each
PFR = {0: 212, 1: 220}                                       # packed P: 2 tuples of 4 per block
FR = 228                                                     # fragment ring: 4 tuples of 4
T0, T1, T2 = 244, 245, 246
PS = [247, 248, 249, 250]
MX, TM, TN, NEG, TA = 251, 252, 253, 254, 255
VLO, VHI = 112, 255

# ---- fixed SGPRs -------------------------------------------------------------------------------------------------
S_T, S_ISSUE, S_SLOT_T, S_SLOT_N, S_AK, S_AV, S_NV, S_A, S_B, S_KOFF, S_VOFF, S_DST = range(36, 48)
S_NOW, S_PREV, S_ACC = 48, 50, 52   # 64-bit pairs: now, prev, 5 accumulators (52..61)
SLO, SHI = 36, 61 if TIMED else 47

out = []


def e(s):
    out.append(s)


def v(n):
    return f"v{n}"


def vr(n, w):
    return f"v[{n}:{n + w - 1}]"


def s(n):
    return f"s{n}"


def qf(b, k):
    return f"%[qf{b}{k}]"


def ab(k):
    return "%[ab0]" if k == 0 else v(AB[k])


def m_run(b):
    return f"%[mr{b}]"


def l_run(b):
    return f"%[lr{b}]"


def stamp(i):
    """adds the time since the previous stamp to accumulator i (only at points where no LDS read is in flight)"""
    if not TIMED:
        return
    e(f"s_memtime s[{S_NOW}:{S_NOW + 1}]")
    e("s_waitcnt lgkmcnt(0)")
    if i is not None:
        a = S_ACC + 2 * i
        e(f"s_sub_u32 {s(S_A)}, {s(S_NOW)}, {s(S_PREV)}")
        e(f"s_subb_u32 {s(S_B)}, {s(S_NOW + 1)}, {s(S_PREV + 1)}")
        e(f"s_add_u32 {s(a)}, {s(a)}, {s(S_A)}")
        e(f"s_addc_u32 {s(a + 1)}, {s(a + 1)}, {s(S_B)}")
    e(f"s_mov_b64 s[{S_PREV}:{S_PREV + 1}], s[{S_NOW}:{S_NOW + 1}]")


def issue(label):
    """DMA of tile S_ISSUE (if < ntile) into its ring slot: 2 K pieces + 2 V^T pieces of 1 KB per wave."""
    e(f"s_cmp_ge_u32 {s(S_ISSUE)}, %[ntile]")
    e(f"s_cbranch_scc1 .Lfdp_noissue_{label}_%=")
    e(f"s_and_b32 {s(S_A)}, {s(S_ISSUE)}, {NSLOT - 1}")
    e(f"s_lshl_b32 {s(S_A)}, {s(S_A)}, 14")
    e(f"s_add_u32 {s(S_DST)}, {s(S_A)}, %[dma_base]")
    e(f"s_mul_i32 {s(S_KOFF)}, {s(S_ISSUE)}, %[ktile]")
    e(f"s_lshl_b32 {s(S_VOFF)}, {s(S_ISSUE)}, 7")
    for (off, vo, rs, so) in [(0, "%[ko0]", "%[rsk]", S_KOFF), (1024, "%[ko1]", "%[rsk]", S_KOFF),
                              (8192, "%[vo0]", "%[rsv]", S_VOFF), (9216, "%[vo1]", "%[rsv]", S_VOFF)]:
        if off:
            e(f"s_add_u32 m0, {s(S_DST)}, {off}")
        else:
            e(f"s_mov_b32 m0, {s(S_DST)}")
        e("s_nop 0")
        e(f"buffer_load_dwordx4 {vo}, {rs}, {s(so)} offen lds")
    e(f"s_add_u32 {s(S_ISSUE)}, {s(S_ISSUE)}, 1")
    e(f".Lfdp_noissue_{label}_%=:")


def softmax_pair(y, i):
    """the 7 VALU of score pair i (scores 2i, 2i+1 of block y's half) -> packed P word i, partial sum"""
    s0, s1 = SC[y] + 2 * i, SC[y] + 2 * i + 1
    pk = PFR[y] + i  # pf[i >> 2].u[i & 3] = consecutive registers
    e(f"v_fma_f32 {v(T0)}, {v(s0)}, %[scale], -{m_run(y)}")
    e(f"v_fma_f32 {v(T1)}, {v(s1)}, %[scale], -{m_run(y)}")
    e(f"v_exp_f32 {v(T0)}, {v(T0)}")
    e(f"v_exp_f32 {v(T1)}, {v(T1)}")
    e("s_nop 0")
    e(f"v_add_f32 {v(T2)}, {v(T0)}, {v(T1)}")
    e(f"v_cvt_pk_bf16_f32 {v(pk)}, {v(T0)}, {v(T1)}")
    e(f"v_add_f32 {v(PS[i & 3])}, {v(PS[i & 3])}, {v(T2)}")


def phase(x, do_q, do_p, do_s, vh):
    """MFMAs of block x interleaved with the softmax of block y = 1 - x.  K rows at LDS address S_AK, V^T tile at S_AV."""
    y = 1 - x
    nm = (4 if do_q else 0) + (4 if do_p else 0)

    def is_q(i):
        return do_q and (not do_p or i % 2 == 0)

    def idx(i):
        return i // 2 if (do_q and do_p) else i

    def read(i):
        k = idx(i)
        if is_q(i):
            e(f"v_add_u32 {v(ADR)}, {s(S_AK)}, {ab(k)}")
            off = 0
        else:
            e(f"v_add_u32 {v(ADR)}, {s(S_AV)}, {ab(vh * 2 + (k >> 1))}")
            off = 4096 * (k & 1)
        e(f"ds_read_b128 {vr(FR + 4 * (i % 4), 4)}, {v(ADR)} offset:{off}")

    if do_s:
        for p in PS:
            e(f"v_mov_b32 {v(p)}, 0")
    for i in range(min(PF, nm)):
        read(i)
    for i in range(8):
        if i < nm:
            outstanding = min(PF - 1, nm - 1 - i)
            e(f"s_waitcnt lgkmcnt({outstanding})")
            k = idx(i)
            fa = vr(FR + 4 * (i % 4), 4)
            if is_q(i):
                acc = vr(SC[x], 16)
                c = "0" if k == 0 else acc
                e(f"v_mfma_f32_32x32x16_bf16 {acc}, {fa}, {qf(x, k)}, {c}")
            else:
                acc = vr(O[(x, k & 1)], 16)
                e(f"v_mfma_f32_32x32x16_bf16 {acc}, {fa}, {vr(PFR[x] + 4 * (k >> 1), 4)}, {acc}")
            if i + PF < nm:
                read(i + PF)
        if do_s:
            softmax_pair(y, i)
    if nm < 8:
        # MFMA results are read by the VALU right below: 18 wait states (the compiler does this on its own code)
        e("s_nop 15")
        e("s_nop 3")
    if do_s:
        e(f"v_add_f32 {v(PS[0])}, {v(PS[0])}, {v(PS[1])}")
        e(f"v_add_f32 {v(PS[2])}, {v(PS[2])}, {v(PS[3])}")
        e(f"v_add_f32 {v(PS[0])}, {v(PS[0])}, {v(PS[2])}")
        e(f"v_add_f32 {l_run(y)}, {l_run(y)}, {v(PS[0])}")


def max_rescale(x, label):
    """row max of block x's fresh scores (S_NV = valid keys of that half), rare rescale of its running state"""
    sc = SC[x]
    e(f"s_cmp_ge_i32 {s(S_NV)}, 32")
    e(f"s_cbranch_scc1 .Lfdp_nomask_{label}_%=")
    e(f"v_sub_u32 {v(TN)}, {s(S_NV)}, %[hi4]")   # keys of the half this lane may use: (r&3) + 8 (r>>2) < nv_lane
    for r in range(16):
        c = (r & 3) + 8 * (r >> 2)
        e(f"v_cmp_ge_i32 vcc, {c}, {v(TN)}")
        e(f"v_cndmask_b32 {v(sc + r)}, {v(sc + r)}, {v(NEG)}, vcc")
    e(f".Lfdp_nomask_{label}_%=:")
    e(f"v_max3_f32 {v(MX)}, {v(sc)}, {v(sc + 1)}, {v(sc + 2)}")
    e(f"v_max3_f32 {v(TM)}, {v(sc + 3)}, {v(sc + 4)}, {v(sc + 5)}")
    e(f"v_max3_f32 {v(TN)}, {v(sc + 6)}, {v(sc + 7)}, {v(sc + 8)}")
    e(f"v_max3_f32 {v(TA)}, {v(sc + 9)}, {v(sc + 10)}, {v(sc + 11)}")
    e(f"v_max3_f32 {v(MX)}, {v(MX)}, {v(TM)}, {v(TN)}")
    e(f"v_max3_f32 {v(TM)}, {v(sc + 12)}, {v(sc + 13)}, {v(sc + 14)}")
    e(f"v_max3_f32 {v(TA)}, {v(TA)}, {v(TM)}, {v(sc + 15)}")
    e(f"v_max_f32 {v(MX)}, {v(MX)}, {v(TA)}")
    # both half-waves hold keys of the same 32 rows: A' = [A.lo, B.lo], B' = [A.hi, B.hi]
    e(f"v_mov_b32 {v(TM)}, {v(MX)}")
    e("s_nop 1")
    e(f"v_permlane32_swap_b32 {v(TM)}, {v(MX)}")
    e("s_nop 1")
    e(f"v_max_f32 {v(MX)}, {v(MX)}, {v(TM)}")
    e(f"v_mul_f32 {v(MX)}, %[scale], {v(MX)}")                       # in exp2 units
    e(f"v_add_f32 {v(TM)}, 0x41000000, {m_run(x)}")                  # running max + 8
    e(f"v_cmp_gt_f32 vcc, {v(MX)}, {v(TM)}")
    e(f"s_cbranch_vccz .Lfdp_norescale_{label}_%=")
    e(f"v_max_f32 {v(TM)}, {m_run(x)}, {v(MX)}")                     # new running max
    e(f"v_sub_f32 {v(TN)}, {m_run(x)}, {v(TM)}")
    e(f"v_exp_f32 {v(TN)}, {v(TN)}")                                 # alpha
    e(f"v_mov_b32 {m_run(x)}, {v(TM)}")
    e("s_nop 15")                                                    # O^T may have just been written by MFMAs
    e("s_nop 3")
    e(f"v_mul_f32 {l_run(x)}, {l_run(x)}, {v(TN)}")
    for nb in range(2):
        for r in range(16):
            e(f"v_mul_f32 {v(O[(x, nb)] + r)}, {v(O[(x, nb)] + r)}, {v(TN)}")
    e(f".Lfdp_norescale_{label}_%=:")


def gen():
    e("// GENERATED by tools/gen_flash_dp_asm.py -- do not edit")
    # ---- init
    for k in (1, 2, 3):
        e(f"v_xor_b32 {v(AB[k])}, {32 * k}, %[ab0]")  # kt_off: ab[k] = ab[0] ^ (k << 5)
    e(f"v_mov_b32 {v(NEG)}, 0xff800000")
    for key in O:
        for r in range(16):
            e(f"v_mov_b32 {v(O[key] + r)}, 0")
    for b in range(2):
        e(f"v_mov_b32 {m_run(b)}, 0xff800000")
        e(f"v_mov_b32 {l_run(b)}, 0")
    e(f"s_mov_b32 {s(S_T)}, 0")
    e(f"s_mov_b32 {s(S_ISSUE)}, 0")
    # ---- prologue DMA, wait for tile 0
    for i in range(AHEAD):
        issue(f"pro{i}")
    e("s_cmp_ge_u32 %[ntile], 3")
    e("s_cbranch_scc1 .Lfdp_w3_%=")
    e("s_cmp_eq_u32 %[ntile], 2")
    e("s_cbranch_scc1 .Lfdp_w2_%=")
    e("s_waitcnt vmcnt(0)")
    e("s_branch .Lfdp_w_%=")
    e(".Lfdp_w2_%=:")
    e("s_waitcnt vmcnt(4)")
    e("s_branch .Lfdp_w_%=")
    e(".Lfdp_w3_%=:")
    e("s_waitcnt vmcnt(8)")
    e(".Lfdp_w_%=:")
    e("s_barrier")
    # ---- prologue phases on half 0 (tile 0, slot 0)
    e(f"s_mov_b32 {s(S_AK)}, %[lds]")
    e(f"s_add_u32 {s(S_AV)}, %[lds], 8192")
    e(f"s_mov_b32 {s(S_NV)}, %[seq]")
    phase(0, True, False, False, 0)
    max_rescale(0, "p0")
    phase(1, True, False, True, 0)
    max_rescale(1, "p1")
    # ---- tile loop
    if TIMED:
        for i in range(5):
            e(f"s_mov_b64 s[{S_ACC + 2 * i}:{S_ACC + 2 * i + 1}], 0")
    stamp(None)
    e(".Lfdp_loop_%=:")
    e(f"s_and_b32 {s(S_A)}, {s(S_T)}, {NSLOT - 1}")
    e(f"s_lshl_b32 {s(S_A)}, {s(S_A)}, 14")
    e(f"s_add_u32 {s(S_SLOT_T)}, {s(S_A)}, %[lds]")
    e(f"s_add_u32 {s(S_AK)}, {s(S_SLOT_T)}, 4096")
    e(f"s_add_u32 {s(S_AV)}, {s(S_SLOT_T)}, 8192")
    # valid keys of half 2t+1:  S - (2t+1)*32
    e(f"s_lshl_b32 {s(S_A)}, {s(S_T)}, 6")
    e(f"s_sub_i32 {s(S_NV)}, %[seq], {s(S_A)}")
    e(f"s_sub_i32 {s(S_NV)}, {s(S_NV)}, 32")
    phase(0, True, True, True, 0)
    max_rescale(0, "a0")
    phase(1, True, True, True, 0)
    max_rescale(1, "a1")
    stamp(0)
    e(f"s_add_u32 {s(S_A)}, {s(S_T)}, 1")
    e(f"s_cmp_eq_u32 {s(S_A)}, %[ntile]")
    e("s_cbranch_scc1 .Lfdp_epi_%=")
    # tile t+1 must have landed; behind the barrier tile t-1 is dead and its slot takes tile t+AHEAD
    e(f"s_add_u32 {s(S_A)}, {s(S_T)}, 2")
    e(f"s_cmp_lt_u32 {s(S_A)}, %[ntile]")
    e("s_cbranch_scc1 .Lfdp_lw4_%=")
    e("s_waitcnt vmcnt(0)")
    e("s_branch .Lfdp_lw_%=")
    e(".Lfdp_lw4_%=:")
    e("s_waitcnt vmcnt(4)")
    e(".Lfdp_lw_%=:")
    stamp(1)
    e("s_barrier")
    stamp(2)
    issue("loop")
    stamp(3)
    e(f"s_add_u32 {s(S_A)}, {s(S_T)}, 1")
    e(f"s_and_b32 {s(S_A)}, {s(S_A)}, {NSLOT - 1}")
    e(f"s_lshl_b32 {s(S_A)}, {s(S_A)}, 14")
    e(f"s_add_u32 {s(S_AK)}, {s(S_A)}, %[lds]")          # K rows 0..31 of tile t+1
    # S_AV stays: V^T tile t, keys 32..63 (vh = 1)
    e(f"s_sub_i32 {s(S_NV)}, {s(S_NV)}, 32")              # half 2t+2
    phase(0, True, True, True, 1)
    max_rescale(0, "b0")
    phase(1, True, True, True, 1)
    max_rescale(1, "b1")
    stamp(4)
    e(f"s_add_u32 {s(S_T)}, {s(S_T)}, 1")
    e("s_branch .Lfdp_loop_%=")
    # ---- epilogue: the last half (tile ntile-1, keys 32..63): S_AV still points at its V^T tile
    e(".Lfdp_epi_%=:")
    phase(0, False, True, True, 1)
    phase(1, False, True, False, 1)
    # ---- leave O^T in LDS: all waves are done with the ring first
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    q = 0
    for key in [(0, 0), (0, 1), (1, 0), (1, 1)]:
        for j in range(4):
            e(f"ds_write_b128 %[dump], {vr(O[key] + 4 * j, 4)} offset:{q * 1024}")
            q += 1
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if TIMED:
        for i in range(5):
            a = S_ACC + 2 * i
            e(f"v_mov_b32 {v(T0)}, {s(a)}")
            e(f"v_mov_b32 {v(T1)}, {s(a + 1)}")
            e(f"global_store_dwordx2 %[dbg], {vr(T0, 2)}, off offset:{8 * i}")
        e("s_waitcnt vmcnt(0)")


gen()
print("// clang-format off")
print("#define FLASH_DP_ASM_TEXT" + ("_TIMED" if TIMED else "") + " \\")
body = [l for l in out if not l.startswith("//")]
for i, line in enumerate(body):
    print(f'  "{line}\\n"' + (" \\" if i + 1 < len(body) else ""))
print("// clang-format on")
clob = [f'"v{i}"' for i in range(VLO, VHI + 1)] + [f'"s{i}"' for i in range(SLO, SHI + 1)] + ['"vcc"', '"scc"', '"memory"']
print("#define FLASH_DP_ASM_CLOBBERS" + ("_TIMED" if TIMED else "") + " \\")
for i in range(0, len(clob), 12):
    tail = ", \\" if i + 12 < len(clob) else ""
    print("  " + ", ".join(clob[i:i + 12]) + tail)
print(f"// fixed registers: v[{VLO}:{VHI}], s[{SLO}:{SHI}], vcc, scc, m0; {len(out)} lines", file=sys.stderr)
