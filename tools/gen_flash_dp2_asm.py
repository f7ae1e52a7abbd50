#!/usr/bin/env python3
"""Generates u2tokenizer_amd/csrc/flash_dp2_asm.inc: the KV loop of the round-4 double-pipeline flash attention kernel
(attn.hip, flash_dp2_kernel, "mode 7") as ONE inline-asm block for gfx950.

Same work split as the round-1 loop (mode 5, tools/gen_flash_dp_asm.py, removed in round 4; history da95d2b): one wave owns two 32-row query blocks and alternates phases over
32-key half tiles in which the 8 MFMAs of block x are interleaved, slot by slot, with the softmax VALU of block y.
The round-1..3 loop ran 59 SIMD cycles per MFMA slot and was VALU-ISSUE bound (per slot: 2 v_fma 7.3 + 2 v_exp 14.9 +
2 v_add 4.2 + v_cvt_pk 4.2 + address add 2 + a share of the row max / rescale test 6 + ~12 beside the MFMA).  What
changed, all of it to take VALU instructions out of the slot:

  * the softmax scale lives in Q (the kernel pre-multiplies its Q fragments by scale * log2 e) and the running max is
    SUBTRACTED BY THE MATRIX PIPE: the first MFMA of every Q K^T chain takes C = MNEG[x], a 16-register tuple holding
    -m of the lane's query row, so the accumulator comes out as  s - m  and  p = v_exp_f32(acc)  directly: no FMA.
  * no row max in the loop.  m is set from the TRUE row max of the first 32 keys and afterwards only has to stay
    within 2^64 of the running max (softmax is shift-invariant; fp32 / bf16 keep their relative precision at any
    magnitude).  The test is on the phase's row-sum piece, which exists anyway:  sum of 16 p > 2^64  (also true for
    inf) sends the wave to an out-of-line slow path that takes the exact row max, rescales O / l / MNEG, and redoes
    the 16 exponentials from the untouched raw scores.  On real data it never runs; tests force it.
  * row sums: two independent add chains, no temporaries, no zeroing.
  * the v_exp results are consumed one slot later (no trans-use s_nop).
  * LDS fragment addresses are lane base + IMMEDIATE: the tile loop is unrolled over the 4 ring slots, so the five
    v_add_u32 per phase are gone.
  * the two query blocks of a wave read each K / V^T fragment from LDS ONCE: the phase of block 0 reads its 8 fragments
    into 8 tuples, the phase of block 1 finds them there (no ds_read, no lgkmcnt wait: 6 issues per slot).
  * the next tile's four LDS-DMA pieces go out from MFMA slots 1 / 3 / 5 / 7 of the phase behind the barrier instead of
    from a stretch of their own between barrier and phase (260 of 2950 cycles per tile and wave); past the last tile
    the pieces are aimed outside both buffers (zeros, no memory traffic), so every wait is a fixed count.
Per slot: v_exp, v_exp, s_waitcnt, v_mfma, ds_read_b128, v_add, v_add, v_cvt_pk = 8 issues (was 12-13), 6 in the phases
that reuse fragments.  Measured (profiles/r04_flash_loop_variants.log): 2950 -> 2734 cycles per 64-key tile and wave,
118.4 -> 109.7 us per ViT launch; --no-share / --no-dmaphase rebuild the loops without them.

    python tools/gen_flash_dp2_asm.py > u2tokenizer_amd/csrc/flash_dp2_asm.inc
"""
import sys

EXACT = "--exact" in sys.argv   # Q fragments as given; every score is multiplied by scale * log2 e in fp32 (2 more VALU per slot)
SHARE = "--no-share" not in sys.argv     # the two query blocks of a wave read each K / V^T fragment from LDS once (8 live tuples)
DMAPH = "--no-dmaphase" not in sys.argv  # next tile's LDS-DMA pieces issued from MFMA slots of the phase behind the barrier
TIMED = "--timed" in sys.argv   # diagnostics build: s_memtime deltas of the loop sections -> 5 x uint64 at %[dbg]
PF = 3          # fragment reads in flight
NSLOT = 4       # LDS ring slots (16 KB each: K tile 8 KB | V^T tile 8 KB)
AHEAD = 3       # tiles in flight
THR_BITS = 0x5f800000   # 2^64: a phase's row-sum piece above this sends the wave to the slow path

# ---- fixed VGPRs: v[VLO:255]; tuples start on even registers (gfx90a+ rule) ---------------------------------------
NFR = 8 if SHARE else 4                                      # fragment tuples
VLO_REST = 69 if SHARE else 85
VLO = VLO_REST - 33                                          # (even: the Q tuples start there; v[VLO_REST - 1] stays unused)
QF = {(b, k): VLO + 16 * b + 4 * k for b in range(2) for k in range(4)}   # Q fragments [block][k slice], loaded by the block
ADR = VLO_REST                                               # address temporary of the epilogue's reads
AB = [VLO_REST + 1 + i for i in range(4)]    # lane base addresses of the fragment reads (LDS base folded in)
_o = VLO_REST + 5
O = {(0, 0): _o, (0, 1): _o + 16, (1, 0): _o + 32, (1, 1): _o + 48}   # O^T accumulators [block][nb], 16 regs each
SC = {0: _o + 64, 1: _o + 80}                                # scores - m of a 32-key half, 16 regs each
MNEG = {0: _o + 96, 1: _o + 112}                             # -m of the lane's row, replicated: C operand of Q K^T
PFR = {0: _o + 128, 1: _o + 136}                             # packed P: 2 tuples of 4 per block
FR = _o + 144                                                # fragment tuples of 4
TT = [[FR + 4 * NFR, FR + 4 * NFR + 1], [FR + 4 * NFR + 2, FR + 4 * NFR + 3]]   # exp results, alternating by slot parity
PSA, PSB = FR + 4 * NFR + 4, FR + 4 * NFR + 5
assert PSB == 255 and _o % 2 == 0
# out-of-line pieces and prologue only (called at a phase end: the exp results, PSB and ADR are dead there; the fragment
# tuples are NOT when the blocks share them)
MX, TM, TN, TA = TT[1][0], TT[1][1], PSB, ADR
VHI = 255

# ---- fixed SGPRs -------------------------------------------------------------------------------------------------
S_T, S_ISSUE, S_NV, S_A, S_B, S_KOFF, S_VOFF, S_AV = range(36, 44)
S_RA = 44                            # 64-bit return address of the out-of-line pieces
S_NOW, S_PREV, S_ACC = 46, 48, 50    # TIMED: now, prev, 5 accumulators (50..59)
SLO, SHI = 36, 59 if TIMED else 45

out = []
uid = [0]
inflight = []   # fragment tuples whose ds_read has been issued and not yet waited for, oldest first (LDS returns in order)


def lds_read(text, tup):
    e(text)
    inflight.append(tup)


def lds_wait_for(tup):
    """s_waitcnt lgkmcnt(N) with the largest N that still guarantees the read into tuple `tup` has landed"""
    if tup in inflight:
        n = len(inflight) - 1 - max(i for i, t in enumerate(inflight) if t == tup)
        e(f"s_waitcnt lgkmcnt({n})")
        del inflight[:len(inflight) - n]


def lds_drained():
    del inflight[:]


def e(s):
    out.append(s)


def v(n):
    return f"v{n}"


def vr(n, w):
    return f"v[{n}:{n + w - 1}]"


def s(n):
    return f"s{n}"


def qf(b, k):
    return vr(QF[(b, k)], 4)


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
    lds_drained()
    if i is not None:
        a = S_ACC + 2 * i
        e(f"s_sub_u32 {s(S_A)}, {s(S_NOW)}, {s(S_PREV)}")
        e(f"s_subb_u32 {s(S_B)}, {s(S_NOW + 1)}, {s(S_PREV + 1)}")
        e(f"s_add_u32 {s(a)}, {s(a)}, {s(S_A)}")
        e(f"s_addc_u32 {s(a + 1)}, {s(a + 1)}, {s(S_B)}")
    e(f"s_mov_b64 s[{S_PREV}:{S_PREV + 1}], s[{S_NOW}:{S_NOW + 1}]")


def call(label):
    """branch to an out-of-line piece that returns with s_setpc_b64 s[S_RA:S_RA+1]"""
    uid[0] += 1
    n = uid[0]
    e(f"s_getpc_b64 s[{S_RA}:{S_RA + 1}]")
    e(f".Lfd2_pc{n}_%=:")
    e(f"s_add_u32 {s(S_RA)}, {s(S_RA)}, .Lfd2_ret{n}_%=-.Lfd2_pc{n}_%=")
    e(f"s_addc_u32 {s(S_RA + 1)}, {s(S_RA + 1)}, 0")
    e(f"s_branch {label}")
    e(f".Lfd2_ret{n}_%=:")


def issue(slot, label):
    """DMA of tile S_ISSUE (if < ntile) into ring slot `slot`: 2 K pieces + 2 V^T pieces of 1 KB per wave.
    DMAPH: always four pieces (past the last tile they read outside both buffers: zeros, no memory traffic)."""
    if not DMAPH:
        e(f"s_cmp_ge_u32 {s(S_ISSUE)}, %[ntile]")
        e(f"s_cbranch_scc1 .Lfd2_noissue_{label}_%=")
    e(f"s_mul_i32 {s(S_KOFF)}, {s(S_ISSUE)}, %[ktile]")
    e(f"s_lshl_b32 {s(S_VOFF)}, {s(S_ISSUE)}, 7")
    if DMAPH:
        e(f"s_cmp_ge_u32 {s(S_ISSUE)}, %[ntile]")
        e(f"s_cselect_b32 {s(S_KOFF)}, 0x7ff00000, {s(S_KOFF)}")
        e(f"s_cselect_b32 {s(S_VOFF)}, 0x7ff00000, {s(S_VOFF)}")
    for (off, vo, rs, so) in [(0, "%[ko0]", "%[rsk]", S_KOFF), (1024, "%[ko1]", "%[rsk]", S_KOFF),
                              (8192, "%[vo0]", "%[rsv]", S_VOFF), (9216, "%[vo1]", "%[rsv]", S_VOFF)]:
        e(f"s_add_u32 m0, %[dma_base], {slot * 16384 + off}")
        e("s_nop 0")
        e(f"buffer_load_dwordx4 {vo}, {rs}, {s(so)} offen lds")
    e(f"s_add_u32 {s(S_ISSUE)}, {s(S_ISSUE)}, 1")
    if not DMAPH:
        e(f".Lfd2_noissue_{label}_%=:")


def phase(x, do_q, do_p, do_s, koff, voff, vh, first=False, dyn_v=False, reuse=False, dma=None, prefetch=None,
          pre_done=False):
    """MFMAs of block x interleaved with the softmax of block y = 1 - x.
    koff / voff: immediate LDS offsets of the K half tile (32 rows) and of the V^T tile; vh: key half of the V^T tile.
    first: the Q K^T chain starts from 0 (prologue: no running max yet).  dyn_v: V^T tile address = S_AV (epilogue).
    reuse: the fragments are in their tuples already (SHARE: the other block's phase read them).  dma: ring slot whose
    4 LDS-DMA pieces (tile S_ISSUE, made harmless past the last tile) go out from slots 1 / 3 / 5 / 7.
    prefetch = (koff, voff, vh) of the NEXT reading phase: its first PF fragment reads go out from slots 5 / 6 / 7 of this
    (reusing) phase, whose tuples 0 .. PF-1 are free by then; that phase is then generated with pre_done."""
    y = 1 - x
    nm = (4 if do_q else 0) + (4 if do_p else 0)

    def is_q(i):
        return do_q and (not do_p or i % 2 == 0)

    def idx(i):
        return i // 2 if (do_q and do_p) else i

    def read(i):
        k = idx(i)
        tup = i % NFR
        fr = vr(FR + 4 * tup, 4)
        if reuse:
            return
        if is_q(i):
            lds_read(f"ds_read_b128 {fr}, {v(AB[k])} offset:{koff}", tup)
        else:
            base = AB[vh * 2 + (k >> 1)]
            if dyn_v:
                e(f"v_add_u32 {v(ADR)}, {s(S_AV)}, {v(base)}")
                lds_read(f"ds_read_b128 {fr}, {v(ADR)} offset:{4096 * (k & 1)}", tup)
            else:
                lds_read(f"ds_read_b128 {fr}, {v(base)} offset:{voff + 4096 * (k & 1)}", tup)

    def finish_pair(i):
        """row-sum adds and the pack of score pair i (its exponentials were issued one slot earlier)"""
        t0, t1 = TT[i & 1]
        if i == 0:
            e(f"v_mov_b32 {v(PSA)}, {v(t0)}")
            e(f"v_mov_b32 {v(PSB)}, {v(t1)}")
        else:
            e(f"v_add_f32 {v(PSA)}, {v(PSA)}, {v(t0)}")
            e(f"v_add_f32 {v(PSB)}, {v(PSB)}, {v(t1)}")
        e(f"v_cvt_pk_bf16_f32 {v(PFR[y] + i)}, {v(t0)}, {v(t1)}")

    def scale_pair(i):
        """EXACT: scores of pair i into exp2 units, in place, one slot before their exponentials"""
        if EXACT:
            e(f"v_mul_f32 {v(SC[y] + 2 * i)}, %[scale], {v(SC[y] + 2 * i)}")
            e(f"v_mul_f32 {v(SC[y] + 2 * i + 1)}, %[scale], {v(SC[y] + 2 * i + 1)}")

    pieces = []
    if dma is not None:
        # tile S_ISSUE -> ring slot `dma`; past the last tile the source offsets point outside both buffers (the loads
        # return zeros into a dead slot without touching memory): the same four pieces per tile, so the vmcnt waits are
        # the same counts up to the last tile
        e(f"s_mul_i32 {s(S_KOFF)}, {s(S_ISSUE)}, %[ktile]")
        e(f"s_lshl_b32 {s(S_VOFF)}, {s(S_ISSUE)}, 7")
        e(f"s_cmp_ge_u32 {s(S_ISSUE)}, %[ntile]")
        e(f"s_cselect_b32 {s(S_KOFF)}, 0x7ff00000, {s(S_KOFF)}")
        e(f"s_cselect_b32 {s(S_VOFF)}, 0x7ff00000, {s(S_VOFF)}")
        e(f"s_add_u32 {s(S_ISSUE)}, {s(S_ISSUE)}, 1")
        pieces = [(0, "%[ko0]", "%[rsk]", S_KOFF), (1024, "%[ko1]", "%[rsk]", S_KOFF),
                  (8192, "%[vo0]", "%[rsv]", S_VOFF), (9216, "%[vo1]", "%[rsv]", S_VOFF)]
    if not pre_done:
        for i in range(min(PF, nm)):
            read(i)
    if do_s:
        scale_pair(0)
    def mfma(i):
        k = idx(i)
        fa = vr(FR + 4 * (i % NFR), 4)
        if is_q(i):
            acc = vr(SC[x], 16)
            c = acc if k else ("0" if first else vr(MNEG[x], 16))
            e(f"v_mfma_f32_32x32x16_bf16 {acc}, {fa}, {qf(x, k)}, {c}")
        else:
            acc = vr(O[(x, k & 1)], 16)
            e(f"v_mfma_f32_32x32x16_bf16 {acc}, {fa}, {vr(PFR[x] + 4 * (k >> 1), 4)}, {acc}")

    for i in range(8):
        piece = pieces[i // 2] if (pieces and i % 2 == 1) else None
        if piece:
            e(f"s_add_u32 m0, %[dma_base], {dma * 16384 + piece[0]}")
        if reuse and i < nm:
            mfma(i)   # nothing to wait for: the MFMA leads the slot (and keeps the other block's fresh scores 18 states away)
        if do_s:
            t0, t1 = TT[i & 1]
            e(f"v_exp_f32 {v(t0)}, {v(SC[y] + 2 * i)}")
            e(f"v_exp_f32 {v(t1)}, {v(SC[y] + 2 * i + 1)}")
        if i < nm and not reuse:
            lds_wait_for(i % NFR)
            mfma(i)
            if i + PF < nm:
                read(i + PF)
        if piece:
            e(f"buffer_load_dwordx4 {piece[1]}, {piece[2]}, {s(piece[3])} offen lds")
        if prefetch is not None and i >= 8 - PF:
            j = i - (8 - PF)                    # read j of a full phase: even = K fragment j / 2, odd = V^T fragment j / 2
            fr = vr(FR + 4 * (j % NFR), 4)
            if j % 2 == 0:
                lds_read(f"ds_read_b128 {fr}, {v(AB[j // 2])} offset:{prefetch[0]}", j % NFR)
            else:
                kk = j // 2
                lds_read(f"ds_read_b128 {fr}, {v(AB[prefetch[2] * 2 + (kk >> 1)])} offset:{prefetch[1] + 4096 * (kk & 1)}", j % NFR)
        if do_s and i < 7:
            scale_pair(i + 1)
        if do_s and i > 0:
            finish_pair(i - 1)
    if do_s:
        finish_pair(7)
        e(f"v_add_f32 {v(PSA)}, {v(PSA)}, {v(PSB)}")
        e(f"v_cmp_lt_f32 vcc, 0x{THR_BITS:08x}, {v(PSA)}")
        uid[0] += 1
        lab = f".Lfd2_ok{uid[0]}_%="
        e(f"s_cbranch_vccz {lab}")
        call(f".Lfd2_slow{y}_%=")
        e(f"{lab}:")
        e(f"v_add_f32 {l_run(y)}, {l_run(y)}, {v(PSA)}")
    if nm < 8:
        # MFMA results are read by the VALU / LDS stores right below: 18 wait states
        e("s_nop 15")
        e("s_nop 3")


def mask_call(x):
    """keys past the end of the sequence in the half block x just scored (S_NV = valid keys of that half)"""
    uid[0] += 1
    lab = f".Lfd2_nomask{uid[0]}_%="
    e(f"s_cmp_ge_i32 {s(S_NV)}, 32")
    e(f"s_cbranch_scc1 {lab}")
    call(f".Lfd2_mask{x}_%=")
    e(f"{lab}:")


def row_max(sc):
    """MX = max of the 16 scores of this lane and of its partner lane (the other 16 keys of the same query row)"""
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


VX = MNEG[1]   # prologue only: v^T of the extra key as 16 packed bf16 pairs, parked in block 1's C tuple until that is filled
KX = SC[1]     # prologue only: the lane's 4 x 16 bytes of the extra key (dead before block 1's first Q K^T)
SX = [PFR[1], PFR[1] + 1]   # prologue only: the extra key's score for the lane's row of block 0 / 1 (block 1's P comes later)


def extra_scores():
    """score of the extra key against the lane's two query rows, on the matrix pipe: a "K tile" whose 32 rows are all the
    extra key (every lane holds the same 4 x 16 bytes of it, its half-wave's share of the head dim) times Q^T leaves the
    score of query l31 in EVERY accumulator register of both half-waves: 4 MFMAs per block, no cross-lane step.
    -inf when the call has no extra key (%[xflag] = 0)."""
    e(f"v_mov_b32 {v(SX[0])}, 0xff800000")
    e(f"v_mov_b32 {v(SX[1])}, 0xff800000")
    e("s_cmp_eq_u32 %[xflag], 0")
    e("s_cbranch_scc1 .Lfd2_nox_%=")
    e("s_waitcnt vmcnt(12)")             # Q, the extra key and its v are older than the 12 LDS-DMA pieces
    for b in range(2):
        acc = vr(SC[0], 16)
        for ks in range(4):
            e(f"v_mfma_f32_32x32x16_bf16 {acc}, {vr(KX + 4 * ks, 4)}, {qf(b, ks)}, {acc if ks else '0'}")
        e("s_nop 15")
        e("s_nop 3")
        e(f"v_mov_b32 {v(SX[b])}, {v(SC[0])}")
    e(".Lfd2_nox_%=:")


def init_max(x):
    """prologue: running max of block x := the larger of the true row max of its first 32 keys and the EXTRA key's score
    (%[sx]: -inf when the call has no extra key; raw units in the EXACT loop).  The extra key also opens the running sums:
    l = p / 2 (both half-waves carry it), O^T = p v -- what the C++ epilogue did after the loop, behind three rounds of
    global loads on the unit's critical path; here the loads fly while the first tiles arrive.  Scores and C tuple follow."""
    row_max(SC[x])
    e(f"v_max_f32 {v(MX)}, {v(MX)}, {v(SX[x])}")
    if EXACT:
        e(f"v_mul_f32 {m_run(x)}, %[scale], {v(MX)}")     # m_run in exp2 units, MNEG and the fresh scores raw
    else:
        e(f"v_mov_b32 {m_run(x)}, {v(MX)}")
    e(f"v_sub_f32 {v(TN)}, {v(SX[x])}, {v(MX)}")
    if EXACT:
        e(f"v_mul_f32 {v(TN)}, %[scale], {v(TN)}")
    e(f"v_exp_f32 {v(TN)}, {v(TN)}")                       # p of the extra key (0 without one)
    e("s_nop 0")
    e(f"v_mul_f32 {l_run(x)}, 0.5, {v(TN)}")
    for nb in range(2):
        for g in range(4):
            for w in range(2):
                src = VX + nb * 8 + g * 2 + w               # d = 32 nb + 8 g + 4 hi + 2 w + {0, 1}
                r = 4 * g + 2 * w
                e(f"v_lshlrev_b32 {v(TM)}, 16, {v(src)}")
                e(f"v_mul_f32 {v(O[(x, nb)] + r)}, {v(TN)}, {v(TM)}")
                e(f"v_and_b32 {v(TM)}, 0xffff0000, {v(src)}")
                e(f"v_mul_f32 {v(O[(x, nb)] + r + 1)}, {v(TN)}, {v(TM)}")
    for r in range(16):
        e(f"v_sub_f32 {v(MNEG[x] + r)}, 0, {v(MX)}")
    for r in range(16):
        e(f"v_sub_f32 {v(SC[x] + r)}, {v(SC[x] + r)}, {v(MX)}")


def slow_path(y):
    """out of line: some p of block y's current half exceeded 2^64.  O[y], l[y], MNEG[y] and the raw scores SC[y] are at
    the old m; nothing of this half has entered O or l yet.  Take the exact row max, move m there (never down),
    scale everything that is at the old m exactly once, redo the half's exponentials."""
    e(f".Lfd2_slow{y}_%=:")
    e("s_nop 15")   # O[y] may have been written by recent MFMAs of a short phase
    e("s_nop 3")
    row_max(SC[y])
    e(f"v_max_f32 {v(MX)}, 0, {v(MX)}")                  # d = max(row max - m, 0)
    e(f"v_sub_f32 {v(TN)}, 0, {v(MX)}")
    e(f"v_exp_f32 {v(TN)}, {v(TN)}")                     # alpha = 2^-d
    e(f"v_add_f32 {m_run(y)}, {m_run(y)}, {v(MX)}")
    e("s_nop 0")
    e(f"v_mul_f32 {l_run(y)}, {l_run(y)}, {v(TN)}")
    for nb in range(2):
        for r in range(16):
            e(f"v_mul_f32 {v(O[(y, nb)] + r)}, {v(O[(y, nb)] + r)}, {v(TN)}")
    if EXACT:   # SC[y] is already in exp2 units (all 8 pairs were scaled in the phase); the C tuple is raw
        e(f"v_mul_f32 {v(TA)}, %[rscale], {v(MX)}")
    for r in range(16):
        e(f"v_sub_f32 {v(MNEG[y] + r)}, {v(MNEG[y] + r)}, {v(TA if EXACT else MX)}")
    for r in range(16):
        e(f"v_sub_f32 {v(SC[y] + r)}, {v(SC[y] + r)}, {v(MX)}")
    for i in range(8):
        t0, t1 = TT[0]
        e(f"v_exp_f32 {v(t0)}, {v(SC[y] + 2 * i)}")
        e(f"v_exp_f32 {v(t1)}, {v(SC[y] + 2 * i + 1)}")
        e("s_nop 0")
        e(f"v_cvt_pk_bf16_f32 {v(PFR[y] + i)}, {v(t0)}, {v(t1)}")
        if i == 0:
            e(f"v_add_f32 {v(PSA)}, {v(t0)}, {v(t1)}")
        else:
            e(f"v_add_f32 {v(PSA)}, {v(PSA)}, {v(t0)}")
            e(f"v_add_f32 {v(PSA)}, {v(PSA)}, {v(t1)}")
    e(f"s_setpc_b64 s[{S_RA}:{S_RA + 1}]")


def mask_path(x):
    e(f".Lfd2_mask{x}_%=:")
    e("s_nop 15")   # SC[x] was written by the phase's last Q K^T MFMA
    e("s_nop 3")
    sc = SC[x]
    e(f"v_sub_u32 {v(TN)}, {s(S_NV)}, %[hi4]")   # keys of the half this lane may use: (r&3) + 8 (r>>2) < nv_lane
    e(f"v_mov_b32 {v(TA)}, 0xff800000")
    for r in range(16):
        c = (r & 3) + 8 * (r >> 2)
        e(f"v_cmp_ge_i32 vcc, {c}, {v(TN)}")
        e(f"v_cndmask_b32 {v(sc + r)}, {v(sc + r)}, {v(TA)}, vcc")
    e(f"s_setpc_b64 s[{S_RA}:{S_RA + 1}]")


def gen():
    e("// GENERATED by tools/gen_flash_dp2_asm.py -- do not edit")
    # ---- init
    e(f"v_add_u32 {v(AB[0])}, %[lds], %[ab0]")
    for k in (1, 2, 3):
        e(f"v_xor_b32 {v(AB[k])}, {32 * k}, {v(AB[0])}")  # kt_off: chunk ^= 2k (the LDS base is 128-byte aligned)
    # Q fragments of the lane's two query rows (B operands of every Q K^T MFMA), the extra key and its v: all issued here, in
    # front of the first tiles' LDS-DMA, so that the unit pays ONE memory latency before its first MFMA.  Every wait of the
    # prologue is a count of the 12 LDS-DMA pieces issued behind these 20 loads.  O^T and l are opened by init_max.
    for b in range(2):
        for k in range(4):
            e(f"global_load_dwordx4 {qf(b, k)}, %[qa{b}], off offset:{32 * k}")
    for nb in range(2):
        for g in range(4):
            e(f"global_load_dwordx2 {vr(VX + nb * 8 + g * 2, 2)}, %[vxa], off offset:{nb * 64 + g * 16}")
    for ks in range(4):
        e(f"global_load_dwordx4 {vr(KX + 4 * ks, 4)}, %[kxa], off offset:{ks * 32}")
    e(f"s_mov_b32 {s(S_T)}, 0")
    e(f"s_mov_b32 {s(S_ISSUE)}, 0")
    # ---- prologue DMA, wait for tile 0
    for i in range(AHEAD):
        issue(i, f"pro{i}")
    extra_scores()
    if DMAPH:
        e("s_waitcnt vmcnt(8)")
    else:
        e("s_cmp_ge_u32 %[ntile], 3")
        e("s_cbranch_scc1 .Lfd2_w3_%=")
        e("s_cmp_eq_u32 %[ntile], 2")
        e("s_cbranch_scc1 .Lfd2_w2_%=")
        e("s_waitcnt vmcnt(0)")
        e("s_branch .Lfd2_w_%=")
        e(".Lfd2_w2_%=:")
        e("s_waitcnt vmcnt(4)")
        e("s_branch .Lfd2_w_%=")
        e(".Lfd2_w3_%=:")
        e("s_waitcnt vmcnt(8)")
        e(".Lfd2_w_%=:")
    e("s_barrier")
    # ---- prologue phases on half 0 (tile 0, slot 0)
    e(f"s_mov_b32 {s(S_NV)}, %[seq]")
    phase(0, True, False, False, 0, 0, 0, first=True)
    mask_call(0)
    init_max(0)
    PRE = SHARE and "--no-prefetch" not in sys.argv
    phase(1, True, False, True, 0, 0, 0, first=True, reuse=SHARE, prefetch=(4096, 8192, 0) if PRE else None)
    mask_call(1)
    init_max(1)
    # ---- tile loop, unrolled over the ring slots
    if TIMED:
        for i in range(5):
            e(f"s_mov_b64 s[{S_ACC + 2 * i}:{S_ACC + 2 * i + 1}], 0")
    stamp(None)
    e(".p2align 6")
    e(".Lfd2_loop_%=:")
    for j in range(NSLOT):
        base = j * 16384
        nxt = ((j + 1) % NSLOT) * 16384
        # valid keys of half 2t+1:  S - (2t+1)*32
        e(f"s_lshl_b32 {s(S_A)}, {s(S_T)}, 6")
        e(f"s_sub_i32 {s(S_NV)}, %[seq], {s(S_A)}")
        e(f"s_sub_i32 {s(S_NV)}, {s(S_NV)}, 32")
        phase(0, True, True, True, base + 4096, base + 8192, 0, pre_done=PRE)
        mask_call(0)
        phase(1, True, True, True, base + 4096, base + 8192, 0, reuse=SHARE)
        mask_call(1)
        stamp(0)
        e(f"s_add_u32 {s(S_A)}, {s(S_T)}, 1")
        e(f"s_cmp_eq_u32 {s(S_A)}, %[ntile]")
        e(f"s_mov_b32 {s(S_AV)}, {base + 8192}")
        e("s_cbranch_scc1 .Lfd2_epi_%=")
        # tile t+1 must have landed; behind the barrier tile t-1 is dead and its slot takes tile t+AHEAD
        if DMAPH:
            e("s_waitcnt vmcnt(4)")                            # two tiles of four pieces are always in flight here
        else:
            e(f"s_add_u32 {s(S_A)}, {s(S_T)}, 2")
            e(f"s_cmp_lt_u32 {s(S_A)}, %[ntile]")
            e(f"s_cbranch_scc1 .Lfd2_lw4_{j}_%=")
            e("s_waitcnt vmcnt(0)")
            e(f"s_branch .Lfd2_lw_{j}_%=")
            e(f".Lfd2_lw4_{j}_%=:")
            e("s_waitcnt vmcnt(4)")
            e(f".Lfd2_lw_{j}_%=:")
        stamp(1)
        e("s_barrier")
        stamp(2)
        if not DMAPH:
            issue((j + AHEAD) % NSLOT, f"loop{j}")
        stamp(3)
        e(f"s_sub_i32 {s(S_NV)}, {s(S_NV)}, 32")              # half 2t+2
        # K rows 0..31 of tile t+1 (next slot); V^T tile t, keys 32..63 (vh = 1)
        phase(0, True, True, True, nxt, base + 8192, 1, dma=(j + AHEAD) % NSLOT if DMAPH else None)
        mask_call(0)
        # (its spare slots carry the first fragment reads of the next tile's first phase: no barrier in between.  The LDS-DMA
        # pieces stay in the reading phase above: moved here they measured 2677 -> 2714 cycles per tile)
        phase(1, True, True, True, nxt, base + 8192, 1, reuse=SHARE, prefetch=(nxt + 4096, nxt + 8192, 0) if PRE else None)
        mask_call(1)
        stamp(4)
        e(f"s_add_u32 {s(S_T)}, {s(S_T)}, 1")
    e("s_branch .Lfd2_loop_%=")
    # ---- epilogue: the last half (tile ntile-1, keys 32..63): its V^T tile is at LDS offset S_AV
    e(".Lfd2_epi_%=:")
    lds_drained()   # (entered from behind a reusing phase: nothing in flight, whatever the last generated body left)
    phase(0, False, True, True, 0, 0, 1, dyn_v=True)
    phase(1, False, True, False, 0, 0, 1, dyn_v=True, reuse=SHARE)
    # ---- leave O^T in LDS: all waves are done with the ring first (and no LDS-DMA piece is still on its way)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    lds_drained()
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
            e(f"v_mov_b32 {v(TT[0][0])}, {s(a)}")
            e(f"v_mov_b32 {v(TT[0][1])}, {s(a + 1)}")
            e(f"global_store_dwordx2 %[dbg], {vr(TT[0][0], 2)}, off offset:{8 * i}")
        e("s_waitcnt vmcnt(0)")
    # the lane id again, as an OUTPUT: what the C++ epilogue derives from it cannot be hoisted above the block and
    # kept alive across it (the compiler only has the registers below VLO there)
    e("v_mbcnt_lo_u32_b32 %[lid], -1, 0")
    e("v_mbcnt_hi_u32_b32 %[lid], -1, %[lid]")
    e("s_branch .Lfd2_end_%=")
    # ---- out-of-line pieces
    for b in range(2):
        slow_path(b)
    for b in range(2):
        mask_path(b)
    e(".Lfd2_end_%=:")


gen()
sfx = ("_X" if EXACT else "") + ("" if SHARE else "_NS") + ("" if DMAPH else "_ND") + ("_TIMED" if TIMED else "")
print("// clang-format off")
print(f"#define FLASH_DP2_ASM_TEXT{sfx} \\")
body = [l for l in out if not l.startswith("//")]
for i, line in enumerate(body):
    print(f'  "{line}\\n"' + (" \\" if i + 1 < len(body) else ""))
print("// clang-format on")
clob = [f'"v{i}"' for i in range(VLO, VHI + 1)] + [f'"s{i}"' for i in range(SLO, SHI + 1)] + ['"vcc"', '"scc"', '"memory"']
print(f"#define FLASH_DP2_ASM_CLOBBERS{sfx} \\")
for i in range(0, len(clob), 12):
    tail = ", \\" if i + 12 < len(clob) else ""
    print("  " + ", ".join(clob[i:i + 12]) + tail)
print(f"// fixed registers: v[{VLO}:{VHI}], s[{SLO}:{SHI}], vcc, scc, m0; {len(out)} lines", file=sys.stderr)
