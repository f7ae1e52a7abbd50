#!/usr/bin/env python3
"""Generates u2tokenizer_amd/csrc/gemm_bt_asm.inc: the K loop of the 256 x (64 NJ) x 64 "big tile" bf16 GEMM
(gemm_bt.hip, gemm_bt_kernel<NJ, PAIR, SPLIT>, NJ = 4 or 3; NJ = 3 also in the SwiGLU-pair form) as ONE inline-asm block for gfx950.

One workgroup = 4 waves (2 x 2), one wave per SIMD, each wave a 128 x (32 NJ) output tile = 4 x NJ accumulators of
v_mfma_f32_32x32x16_bf16 in AccVGPRs (asm operands "+a": the compiler zeroes them before and runs the epilogue after).
LDS: 2 stages x (A tile 256 rows x 128 B | B tile 64 NJ rows x 128 B) <= 128 KB, rows XOR-swizzled as in gemm_bt.hip.
Per K tile a wave issues 16 NJ MFMAs in four k16 blocks; an MFMA "slot" carries at most one LDS read and, in some
slots, one LDS-DMA piece:

    kk = 0 : first 4 + NJ slots: LDS reads of the kk = 1 fragments       + this wave's share of the DMA of K tile t+1
    kk = 1 : first 4 + NJ slots: LDS reads of the kk = 2 fragments       + ... (rest of K tile t+1)
    kk = 2 : first 4 + NJ slots: LDS reads of the kk = 3 fragments
    s_waitcnt vmcnt(0) ; s_barrier          (K tile t+1 landed; everybody has read all of K tile t)
    kk = 3 : first 4 + NJ slots: LDS reads of kk = 0 of K tile t+1       + first pieces of K tile t+2

so the matrix pipe only drains at the one barrier per K tile, every DMA piece has >= 4 NJ slots to land, and fragments
are read one k16 block ahead into the other half of a double register buffer.  The 4 x (8 + 2 NJ) DMA pieces of a K
tile are spread evenly over the 12 NJ slots of that window, each wave in different slots: the four waves run in lock
step, and pieces issued in the same slot queue behind one another in the CU's single address path (gen() docstring).
Each wave therefore has its own copy of the loop.
DMA: MUBUF buffer_load_dwordx4 ... lds, descriptor per matrix (rows past M / N read as zero), 32-bit lane offsets
relative to the tile origin (two per matrix: even / odd 8-row pieces differ in the swizzle); tile origin, K advance and
row-block advance in the scalar offset.  The K loop of one output tile runs on into the next output tile of the same
(persistent) workgroup: K tile nkt of this tile IS K tile 0 of the next one (origin nbase), so a tile switch costs no
DMA burst and no exposed latency (operand `first` = 1 only for a workgroup's first tile).

Round 4 added, on the same skeleton (gen_ring, gen_deep below, each with its own schedule notes): the RING form (256 x 128 tiles, three
LDS stages, K tile t+2 in flight behind a counted vmcnt) and the DEEP forms of the 192- and 256-wide tiles (three stages for one
operand, two for the other) -- what the ablation builds (--ablate) showed the two-stage loop to be bound by is the latency of
its own operand stream, not the matrix pipe.  Fragment waits are counted per slot (frag_waits); --drain-waits restores the one
lgkmcnt(0) per k16 step.

    python tools/gen_gemm_bt_asm.py > u2tokenizer_amd/csrc/gemm_bt_asm.inc
    python tools/gen_gemm_bt_asm.py --ablate nodma,noread   (measurement-only build, see tools/mk_ab_build.sh)
"""
import sys

# ---- fixed VGPRs (clobbered) ---------------------------------------------------------------------------------------
AA = [56, 57, 58, 59]      # LDS read addresses of the A fragments, k16 step kk, current stage
AB = [60, 61, 62, 63]      # ... of the B fragments
BUF = 64                   # fragment double buffer: buf b at BUF + 32 b: A_i at + 4 i, B_j at + 16 + 4 j
VLO, VHI = 56, 127
# ---- fixed SGPRs (clobbered) ---------------------------------------------------------------------------------------
S_KT, S_K1A, S_K2A, S_SA, S_SB, S_TMP, S_ST = 36, 37, 38, 39, 40, 41, 42
S_ROWA = [43, 44, 45, 46]
S_ROWB = [47, 48, 49, 50]
S_DA, S_DB = 51, 52          # LDS byte offsets of this wave's DMA rows in the A / B tile of stage 0
S_K1B, S_K2B, S_K0A, S_K0B = 53, 54, 55, 56   # scalar byte offsets of K tiles kt+1 / kt+2 (and 0) per matrix
SLO, SHI = 36, 56

out = []
ABL = set()   # measurement-only builds (wrong results): "nodma" / "noread" / "nobar" / "nomfma" inside the K loop
IN_LOOP = [False]


def e(s):
    if IN_LOOP[0]:
        if "nodma" in ABL and s.startswith("buffer_load"):
            return
        if "noread" in ABL and s.startswith("ds_read"):
            return
        if "nobar" in ABL and s.startswith("s_barrier"):
            return
        if "nomfma" in ABL and s.startswith("v_mfma"):
            return
    out.append(s)


def v(n):
    return f"v{n}"


def vr(n, w):
    return f"v[{n}:{n + w - 1}]"


def s(n):
    return f"s{n}"


def afrag(b, i):
    return vr(BUF + 32 * b + 4 * i, 4)


def bfrag(b, j):
    return vr(BUF + 32 * b + 16 + 4 * j, 4)


FRAG_AGPR = "--frag-agpr" in sys.argv
FIXED_ACC = [False]   # True: accumulator (i, j) is the literal tuple a[16 (NJ i + j) : +15] (the drain forms pin the accumulator file)


def acc(i, j):
    if FIXED_ACC[0]:
        return f"a[{16 * (NJ * i + j)}:{16 * (NJ * i + j) + 15}]"
    return f"%[c{i >> 1}{i & 1}{j}]"


NJ = 4                     # 32-column blocks per wave: tile = 256 x (64 NJ); set per variant by gen()


def read(b, kk, n):
    """n-th fragment read (order A0 B0 .. B(NJ-1) A1 A2 A3) of k16 step kk into buffer b"""
    order = [("a", 0)] + [("b", j) for j in range(NJ)] + [("a", 1), ("a", 2), ("a", 3)]
    m, idx = order[n]
    if m == "a":
        e(f"ds_read_b128 {afrag(b, idx)}, {v(AA[kk])} offset:{4096 * idx}")
    else:
        e(f"ds_read_b128 {bfrag(b, idx)}, {v(AB[kk])} offset:{4096 * idx}")


def dma(mat, p, s_stage, s_k):
    """piece p (8 rows x 128 B) of this wave's 64 rows of the A or B tile; s_stage = LDS base of (matrix, stage);
    s_k = (A offset register, B offset register) of the K tile"""
    s_k = s_k[0] if mat == "a" else s_k[1]
    if p:
        e(f"s_add_u32 m0, {s(s_stage)}, {1024 * p}")
    else:
        e(f"s_mov_b32 m0, {s(s_stage)}")
    row = (S_ROWA if mat == "a" else S_ROWB)[p >> 1]
    e(f"s_add_u32 {s(S_TMP)}, {s(s_k)}, {s(row)}")
    vo = f"%[v{mat}{p & 1}]"
    e(f"buffer_load_dwordx4 {vo}, %[rs{mat}], {s(S_TMP)} offen lds")


COUNTED = "--drain-waits" not in sys.argv   # fragment waits counted per slot (below); --drain-waits: one lgkmcnt(0) per k16 step


def frag_waits():
    """slot -> N of the `s_waitcnt lgkmcnt(N)` in front of that slot's MFMA.  The 4 + NJ fragments of a k16 step were read in
    the order A0 B0 .. B(NJ-1) A1 A2 A3, one per slot, during the first 4 + NJ slots of the PREVIOUS step (LDS returns in
    order); this step's slots issue the next step's reads the same way.  Slot (i, j) needs A_i and B_j: it may leave
    outstanding the older reads behind the ones it needs plus the newer reads issued so far -- the last fragment (A3, read
    in slot 3 + NJ) is waited for 3 NJ slots into the step instead of at its start, where one lgkmcnt(0) used to stall
    the matrix pipe for the LDS latency of a read issued 4 (NJ = 2: 2) slots earlier."""
    nread = 4 + NJ
    waits, have = {}, -1
    for slot in range(4 * NJ):
        i, j = slot // NJ, slot % NJ
        need = max(0 if i == 0 else NJ + i, 1 + j)
        if need > have:
            have = need
            waits[slot] = (nread - 1 - need) + min(slot * RPS, nread)
    return waits


# --reads-per-slot N (measurement, deep form only): N fragment reads behind each of the first ceil((4 + NJ) / N) MFMAs of a k16 step instead of one
RPS = int(sys.argv[sys.argv.index("--reads-per-slot") + 1]) if "--reads-per-slot" in sys.argv else 1


SWAP = [False]   # True: MFMA operands exchanged -> the accumulator tile comes out TRANSPOSED (lane = column n, registers = rows m)


def mfma_block(b, extra, a=None, bb=None, zero_c=False):
    """the 4 NJ MFMAs of one k16 step out of buffer b; extra(slot) emits the slot's companion instruction(s); zero_c: the step OPENS an
    output tile -- C = 0 instead of the accumulator (the drain forms: nobody has to zero 192 registers between two tiles)"""
    a, bb = a or afrag, bb or bfrag
    waits = frag_waits() if COUNTED else {0: 0}
    for slot in range(4 * NJ):
        i, j = slot // NJ, slot % NJ
        if slot in waits:
            e(f"s_waitcnt lgkmcnt({waits[slot]})")
        c = "0" if zero_c else acc(i, j)
        if SWAP[0]:
            e(f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {a(b, i)}, {bb(b, j)}, {c}")
        else:
            e(f"v_mfma_f32_32x32x16_bf16 {acc(i, j)}, {bb(b, j)}, {a(b, i)}, {c}")
        extra(slot)


def gen(nj, pair=False):
    """All four waves run the same MFMA / read / barrier skeleton, but each has its own copy of the loop with its
    LDS-DMA pieces in different slots: issued in the same slot by all four (lock-step) waves, the pieces queue behind
    one another in the CU's single address path and every one of them blocks its wave for ~130 cycles with the
    matrix pipe idle (measured: 8192^3 at 48 % of peak, K-step time proportional to the DMA bytes).  The 4 x 16
    pieces of K tile t+1 are spread over the 3 x 4 NJ slots between the barrier of iteration t-1 and the end of the
    kk = 1 block of iteration t (they must land by the barrier of iteration t, 4 NJ slots later)."""
    global NJ
    NJ = nj
    del out[:]
    nslot, nread, npb = 4 * NJ, 4 + NJ, 2 * NJ  # MFMA slots per k16 step, fragment reads, B pieces per wave
    npw = 8 + npb                               # DMA pieces per wave and K tile
    window = 3 * nslot
    # piece n of wave w -> window slot (blocks: 0 = kk 3 of the previous iteration, 1 = kk 0, 2 = kk 1)
    sched = {w: {} for w in range(4)}
    for n in range(npw):
        for w in range(4):
            g = (n * 4 + w) * window // (4 * npw)
            sched[w].setdefault(g, []).append(n)

    def piece(n, s_a, s_b, s_k):
        if n < 8:
            dma("a", n, s_a, s_k)
        else:
            dma("b", n - 8, s_b, s_k)

    K0, K1, K2 = (S_K0A, S_K0B), (S_K1A, S_K1B), (S_K2A, S_K2B)
    # ---- setup (common)
    for k in (1, 2, 3):
        e(f"v_xor_b32 {v(AA[k])}, {32 * k}, %[aa0]")
        e(f"v_xor_b32 {v(AB[k])}, {32 * k}, %[ab0]")
    e(f"v_mov_b32 {v(AA[0])}, %[aa0]")
    e(f"v_mov_b32 {v(AB[0])}, %[ab0]")
    e(f"s_mov_b32 {s(S_ROWA[0])}, 0")
    for q in (1, 2, 3):
        e(f"s_add_u32 {s(S_ROWA[q])}, {s(S_ROWA[q - 1])}, %[lda16]")
    if pair:
        # SwiGLU-pair form (gemm_bt.hip): the wave's 16-row groups of the B tile alternate between gate rows and up rows of
        # the weight -- their byte offsets (relative to the tile origin) are operands instead of q * 16 ldb
        for q in range((npb + 1) // 2):
            e(f"s_mov_b32 {s(S_ROWB[q])}, %[rowb{q}]")
    else:
        e(f"s_mov_b32 {s(S_ROWB[0])}, 0")
        for q in (1, 2, 3):
            e(f"s_add_u32 {s(S_ROWB[q])}, {s(S_ROWB[q - 1])}, %[ldb16]")
    e(f"s_mov_b32 {s(S_KT)}, 0")
    e(f"s_lshl_b32 {s(S_DA)}, %[wave], 13")                       # 64 rows x 128 B per wave
    e(f"s_mul_i32 {s(S_DB)}, %[wave], {2048 * NJ}")               # 16 NJ rows per wave
    e(f"s_add_u32 {s(S_DB)}, {s(S_DB)}, 0x8000")
    e(f"s_mov_b32 {s(S_ST)}, %[st0]")                             # LDS offset of the stage being computed on
    # scalar offsets of K tiles 0 / 1 / 2: the K loop runs on into the NEXT output tile of this workgroup
    # (K tile nkt of this tile = K tile 0 of the next one, whose origin is nbase), nkt >= 2
    e(f"s_mov_b32 {s(S_K0A)}, %[base_a]")
    e(f"s_mov_b32 {s(S_K0B)}, %[base_b]")
    e(f"s_add_u32 {s(S_K1A)}, %[base_a], 128")
    e(f"s_add_u32 {s(S_K1B)}, %[base_b], 128")
    e(f"s_add_u32 {s(S_K2A)}, %[base_a], 256")
    e(f"s_add_u32 {s(S_K2B)}, %[base_b], 256")
    e("s_cmp_eq_u32 %[nkt], 2")
    e(f"s_cselect_b32 {s(S_K2A)}, %[nbase_a], {s(S_K2A)}")
    e(f"s_cselect_b32 {s(S_K2B)}, %[nbase_b], {s(S_K2B)}")
    for w in (1, 2, 3):
        e(f"s_cmp_eq_u32 %[wave], {w}")
        e(f"s_cbranch_scc1 .Lbt_w{w}_%=")
    for w in range(4):
        if w:
            e(f".Lbt_w{w}_%=:")
        early = [n for g in range(nslot) for n in sched[w].get(g, [])]
        # first tile of the workgroup: K tile 0 -> this stage at once, and the part of K tile 1 that the steady state
        # issues in the kk 3 block of "iteration -1".  Later tiles: the previous tile's K loop already did both, and
        # its last barrier made K tile 0 visible.
        e("s_cmp_eq_u32 %[first], 0")
        e(f"s_cbranch_scc1 .Lbt_cont{w}_%=")
        e(f"s_add_u32 {s(S_SA)}, {s(S_DA)}, {s(S_ST)}")
        e(f"s_add_u32 {s(S_SB)}, {s(S_DB)}, {s(S_ST)}")
        for n in range(npw):
            piece(n, S_SA, S_SB, K0)
        e(f"s_xor_b32 {s(S_TMP)}, {s(S_ST)}, 0x10000")
        e(f"s_add_u32 {s(S_SA)}, {s(S_DA)}, {s(S_TMP)}")
        e(f"s_add_u32 {s(S_SB)}, {s(S_DB)}, {s(S_TMP)}")
        for n in early:
            piece(n, S_SA, S_SB, K1)
        e(f"s_waitcnt vmcnt({len(early)})")
        e("s_barrier")
        e(f".Lbt_cont{w}_%=:")
        e("s_waitcnt lgkmcnt(0)")     # (a scalar load the compiler left in flight would return out of order and spoil the counted fragment waits)
        for n in range(nread):
            read(0, 0, n)
        # ---- K loop of wave w
        e(f".Lbt_loop{w}_%=:")
        IN_LOOP[0] = True
        # targets: rest of tile kt+1 -> other stage (S_SA / S_SB); first pieces of tile kt+2 -> this stage
        e(f"s_xor_b32 {s(S_TMP)}, {s(S_ST)}, 0x10000")
        e(f"s_add_u32 {s(S_SA)}, {s(S_TMP)}, {s(S_DA)}")
        e(f"s_add_u32 {s(S_SB)}, {s(S_TMP)}, {s(S_DB)}")

        def xk(blk, rb, rkk):
            def f(slot):
                if slot < nread:
                    read(rb, rkk, slot)
                for n in sched[w].get(blk * nslot + slot, []):
                    piece(n, S_SA, S_SB, K1)
            return f

        mfma_block(0, xk(1, 1, 1))
        mfma_block(1, xk(2, 0, 2))
        mfma_block(0, lambda slot: read(1, 3, slot) if slot < nread else None)
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")
        for k in range(4):                                   # fragment addresses -> the other stage
            e(f"v_xor_b32 {v(AA[k])}, 0x10000, {v(AA[k])}")
            e(f"v_xor_b32 {v(AB[k])}, 0x10000, {v(AB[k])}")
        e(f"s_add_u32 {s(S_SA)}, {s(S_ST)}, {s(S_DA)}")       # the stage everybody just left takes tile kt+2
        e(f"s_add_u32 {s(S_SB)}, {s(S_ST)}, {s(S_DB)}")

        def x3(slot):
            if slot < nread:
                read(0, 0, slot)
            for n in sched[w].get(slot, []):
                piece(n, S_SA, S_SB, K2)

        mfma_block(1, x3)
        e(f"s_xor_b32 {s(S_ST)}, {s(S_ST)}, 0x10000")
        e(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, 1")
        # K tile offsets of the next iteration: kt+1 <- kt+2;  kt+2 <- next one, which is K tile 0 of the next output
        # tile when it reaches nkt
        e(f"s_mov_b32 {s(S_K1A)}, {s(S_K2A)}")
        e(f"s_mov_b32 {s(S_K1B)}, {s(S_K2B)}")
        e(f"s_add_u32 {s(S_K2A)}, {s(S_K2A)}, 128")
        e(f"s_add_u32 {s(S_K2B)}, {s(S_K2B)}, 128")
        e(f"s_add_u32 {s(S_TMP)}, {s(S_KT)}, 2")
        e(f"s_cmp_eq_u32 {s(S_TMP)}, %[nkt]")
        e(f"s_cselect_b32 {s(S_K2A)}, %[nbase_a], {s(S_K2A)}")
        e(f"s_cselect_b32 {s(S_K2B)}, %[nbase_b], {s(S_K2B)}")
        e(f"s_cmp_lt_u32 {s(S_KT)}, %[nkt]")
        e(f"s_cbranch_scc1 .Lbt_loop{w}_%=")
        IN_LOOP[0] = False
        if w < 3:
            e("s_branch .Lbt_done_%=")
    e(".Lbt_done_%=:")
    # the loop has staged K tile 0 of the NEXT output tile (landed, visible) and issued the first pieces of its K tile
    # 1, which stay in flight across the epilogue; its last reads fetched fragments nobody uses
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 15")                                        # MFMA results -> compiler's VALU reads: 18 wait states
    e("s_nop 7")


# ---- ring form: 256 x 128 tiles (NJ = 2), THREE LDS stages -------------------------------------------------------------
# A 256 x 128 x 64 stage is 48 KB (A 32 KB | B 16 KB), so three fit (144 KB) and a K tile can be two iterations ahead of
# its use: the pieces of K tile t+2 are issued over ALL FOUR k16 blocks of iteration t (into the stage tile t-1 left one
# barrier ago) and have to land by the barrier of iteration t+1 -- between 3 and 7 blocks later; the two-stage form gives
# a piece 1 .. 3 blocks (of 8 MFMAs at this tile width: its K loop measured 2500 cycles per K tile for 1024 cycles of
# MFMA, profiles/r03_bt_sweep.log "256x128").  The wait in front of the barrier is COUNTED: s_waitcnt vmcnt(n), n = the
# pieces of tile t+2 this wave has issued so far in the iteration; everything older -- all of tile t+1 -- has landed
# (vector memory returns in order).  The loop is unrolled over the three stages (every LDS address is a register picked at
# generation time or an immediate), entered at the stage that holds K tile 0 (operand st0 = 0 / 1 / 2) and left after
# any body; as in the two-stage form it runs on into the next output tile of the workgroup: on exit K tile 0 of the next
# tile has landed and ALL of its K tile 1 is in flight.
RSTAGE = 49152
RAA = [[56 + 4 * st + k for k in range(4)] for st in range(3)]        # v56..v67: A fragment addresses [stage][kk]
RAB = [[68 + 4 * st + k for k in range(4)] for st in range(3)]        # v68..v79: B fragment addresses
RBUF = 80                                                             # fragment double buffer: buf b at RBUF + 32 b
RVLO, RVHI = 56, 143


def gen_ring(nj=2):
    global NJ
    NJ = nj
    del out[:]
    nslot, nread, npb = 4 * NJ, 4 + NJ, 2 * NJ
    npw = 8 + npb
    window = 4 * nslot
    sched = {w: {} for w in range(4)}
    for n in range(npw):
        for w in range(4):
            g = (n * 4 + w) * window // (4 * npw)
            sched[w].setdefault(g, []).append(n)

    def rafrag(b, i):
        return vr(RBUF + 32 * b + 4 * i, 4)

    def rbfrag(b, j):
        return vr(RBUF + 32 * b + 16 + 4 * j, 4)

    def rread(b, st, kk, n):
        order = [("a", 0)] + [("b", j) for j in range(NJ)] + [("a", 1), ("a", 2), ("a", 3)]
        m, idx = order[n]
        if m == "a":
            e(f"ds_read_b128 {rafrag(b, idx)}, {v(RAA[st][kk])} offset:{4096 * idx}")
        else:
            e(f"ds_read_b128 {rbfrag(b, idx)}, {v(RAB[st][kk])} offset:{4096 * idx}")

    def rpiece(n, st, s_k):
        """piece n of this wave (A pieces 0..7, then B) of a K tile into stage st; s_k = (A, B) offset registers of the K tile"""
        mat, p = ("a", n) if n < 8 else ("b", n - 8)
        e(f"s_add_u32 m0, {s(S_DA if mat == 'a' else S_DB)}, {RSTAGE * st + 1024 * p}")
        row = (S_ROWA if mat == "a" else S_ROWB)[p >> 1]
        e(f"s_add_u32 {s(S_TMP)}, {s(s_k[0] if mat == 'a' else s_k[1])}, {s(row)}")
        e(f"buffer_load_dwordx4 %[v{mat}{p & 1}], %[rs{mat}], {s(S_TMP)} offen lds")

    def rblock(b, extra):
        mfma_block(b, extra, rafrag, rbfrag)

    K0, K1, K2 = (S_K0A, S_K0B), (S_K1A, S_K1B), (S_K2A, S_K2B)
    # ---- setup (common)
    for k in range(4):
        if k:
            e(f"v_xor_b32 {v(RAA[0][k])}, {32 * k}, %[aa0]")
            e(f"v_xor_b32 {v(RAB[0][k])}, {32 * k}, %[ab0]")
        else:
            e(f"v_mov_b32 {v(RAA[0][0])}, %[aa0]")
            e(f"v_mov_b32 {v(RAB[0][0])}, %[ab0]")
    for st in (1, 2):
        for k in range(4):
            e(f"v_add_u32 {v(RAA[st][k])}, {RSTAGE * st}, {v(RAA[0][k])}")
            e(f"v_add_u32 {v(RAB[st][k])}, {RSTAGE * st}, {v(RAB[0][k])}")
    e(f"s_mov_b32 {s(S_ROWA[0])}, 0")
    for q in (1, 2, 3):
        e(f"s_add_u32 {s(S_ROWA[q])}, {s(S_ROWA[q - 1])}, %[lda16]")
    e(f"s_mov_b32 {s(S_ROWB[0])}, 0")
    for q in (1, 2, 3):
        e(f"s_add_u32 {s(S_ROWB[q])}, {s(S_ROWB[q - 1])}, %[ldb16]")
    e(f"s_mov_b32 {s(S_KT)}, 0")
    e(f"s_lshl_b32 {s(S_DA)}, %[wave], 13")
    e(f"s_mul_i32 {s(S_DB)}, %[wave], {2048 * NJ}")
    e(f"s_add_u32 {s(S_DB)}, {s(S_DB)}, 0x8000")
    e(f"s_mov_b32 {s(S_K0A)}, %[base_a]")
    e(f"s_mov_b32 {s(S_K0B)}, %[base_b]")
    e(f"s_add_u32 {s(S_K1A)}, %[base_a], 128")
    e(f"s_add_u32 {s(S_K1B)}, %[base_b], 128")
    e(f"s_add_u32 {s(S_K2A)}, %[base_a], 256")
    e(f"s_add_u32 {s(S_K2B)}, %[base_b], 256")
    e("s_cmp_eq_u32 %[nkt], 2")
    e(f"s_cselect_b32 {s(S_K2A)}, %[nbase_a], {s(S_K2A)}")
    e(f"s_cselect_b32 {s(S_K2B)}, %[nbase_b], {s(S_K2B)}")
    for w in (1, 2, 3):
        e(f"s_cmp_eq_u32 %[wave], {w}")
        e(f"s_cbranch_scc1 .Lbr_w{w}_%=")
    for w in range(4):
        if w:
            e(f".Lbr_w{w}_%=:")
        # first tile of the workgroup (st0 = 0 then): K tiles 0 and 1 at once; tile 0 has landed when only tile 1's pieces
        # are outstanding
        e("s_cmp_eq_u32 %[first], 0")
        e(f"s_cbranch_scc1 .Lbr_cont{w}_%=")
        for n in range(npw):
            rpiece(n, 0, K0)
        for n in range(npw):
            rpiece(n, 1, K1)
        e(f"s_waitcnt vmcnt({npw})")
        e("s_barrier")
        e(f".Lbr_cont{w}_%=:")
        e("s_waitcnt lgkmcnt(0)")     # (a scalar load the compiler left in flight would return out of order and spoil the counted fragment waits)
        for st in (1, 2):
            e(f"s_cmp_eq_u32 %[st0], {st}")
            e(f"s_cbranch_scc1 .Lbr_e{w}_{st}_%=")
        for st in range(3):                      # entry st: fragments of k16 step 0 of K tile 0, then body st
            if st:
                e(f".Lbr_e{w}_{st}_%=:")
            for n in range(nread):
                rread(0, st, 0, n)
            e(f"s_branch .Lbr_b{w}_{st}_%=")
        IN_LOOP[0] = True
        for st in range(3):
            e(f".Lbr_b{w}_{st}_%=:")
            tgt, nxt = (st + 2) % 3, (st + 1) % 3

            def xk(blk, rb, rst, rkk):
                def f(slot):
                    if slot < nread:
                        rread(rb, rst, rkk, slot)
                    for n in sched[w].get(blk * nslot + slot, []):
                        rpiece(n, tgt, K2)
                return f

            rblock(0, xk(0, 1, st, 1))
            rblock(1, xk(1, 0, st, 2))
            rblock(0, xk(2, 1, st, 3))
            issued = sum(len(sched[w].get(g, [])) for g in range(3 * nslot))
            e(f"s_waitcnt vmcnt({issued})")
            e("s_barrier")
            rblock(1, xk(3, 0, nxt, 0))
            e(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, 1")
            e(f"s_add_u32 {s(S_K2A)}, {s(S_K2A)}, 128")
            e(f"s_add_u32 {s(S_K2B)}, {s(S_K2B)}, 128")
            e(f"s_add_u32 {s(S_TMP)}, {s(S_KT)}, 2")
            e(f"s_cmp_eq_u32 {s(S_TMP)}, %[nkt]")
            e(f"s_cselect_b32 {s(S_K2A)}, %[nbase_a], {s(S_K2A)}")
            e(f"s_cselect_b32 {s(S_K2B)}, %[nbase_b], {s(S_K2B)}")
            e(f"s_cmp_lt_u32 {s(S_KT)}, %[nkt]")
            if st < 2:
                e(f"s_cbranch_scc0 .Lbr_done_%=")
            else:
                e(f"s_cbranch_scc1 .Lbr_b{w}_0_%=")
        IN_LOOP[0] = False
        if w < 3:
            e("s_branch .Lbr_done_%=")
    e(".Lbr_done_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 15")
    e("s_nop 7")


# ---- deep forms: THREE stages for the operand that streams, two for the other ----------------------------------------------
# 256 x 192 (and 256 x 256) stages do not fit the LDS three times, but the two operands need not be treated alike: in the ViT the
# activations stream (25-100 MB per product) against weights that stay in the L2; in the tokenizer 100 MB of cold weights stream
# against 17 MB of activations.  The streaming ("deep") operand gets a third stage and the ring form's schedule -- its K tile
# t+2 is issued during iteration t and has until the barrier of iteration t+1 --, the other one keeps two stages and the
# two-stage deadline (K tile t+1 lands by the barrier of iteration t).  Vector memory returns in order, so inside an iteration
# the shallow operand's pieces go FIRST (k16 step 3 of the previous iteration, behind its barrier, and step 0) and the deep
# one's LAST (steps 1 and 2): the wait in front of the barrier, vmcnt(pieces of the deep operand per wave), then covers
# everything older -- the shallow tile t+1 and the deep tile t+1 issued an iteration ago.  LDS: [A stages][B stages]; the loop is
# unrolled over lcm(3, 2) = 6 stage pairs and entered at (K tiles this workgroup has consumed) mod 6 (operand st0).

def gen_deep(nj, deep):
    global NJ
    NJ = nj
    del out[:]
    SA, SB = (3, 2) if deep == "a" else (2, 3)
    L = 6
    nslot, nread, npb = 4 * NJ, 4 + NJ, 2 * NJ
    BSTG = NJ * 8192
    OFFB = 32768 * SA
    VB = 56
    XAA = [[VB + 4 * st + k for k in range(4)] for st in range(SA)]
    XAB = [[VB + 4 * SA + 4 * st + k for k in range(4)] for st in range(SB)]
    XBUF = VB + 4 * (SA + SB)
    vhi = XBUF + 2 * nread * 4 - 1

    def xa(b, i):
        if FRAG_AGPR and FIXED_ACC[0]:   # (--frag-agpr, measurement: the fragment double buffer in a[192:247], the accumulators' spare registers)
            return f"a[{192 + 4 * nread * b + 4 * i}:{192 + 4 * nread * b + 4 * i + 3}]"
        return vr(XBUF + 4 * nread * b + 4 * i, 4)

    def xb(b, j):
        if FRAG_AGPR and FIXED_ACC[0]:
            return f"a[{192 + 4 * nread * b + 16 + 4 * j}:{192 + 4 * nread * b + 16 + 4 * j + 3}]"
        return vr(XBUF + 4 * nread * b + 16 + 4 * j, 4)

    def xread(b, sa, sb, kk, n):
        order = [("a", 0)] + [("b", j) for j in range(NJ)] + [("a", 1), ("a", 2), ("a", 3)]
        m, idx = order[n]
        if m == "a":
            e(f"ds_read_b128 {xa(b, idx)}, {v(XAA[sa][kk])} offset:{4096 * idx}")
        else:
            e(f"ds_read_b128 {xb(b, idx)}, {v(XAB[sb][kk])} offset:{4096 * idx}")

    def xpiece(mat, p, st, s_k):
        if mat == "a":
            e(f"s_add_u32 m0, {s(S_DA)}, {32768 * st + 1024 * p}")
        else:
            e(f"s_add_u32 m0, {s(S_DB)}, {BSTG * st + 1024 * p}")
        row = (S_ROWA if mat == "a" else S_ROWB)[p >> 1]
        e(f"s_add_u32 {s(S_TMP)}, {s(s_k[0] if mat == 'a' else s_k[1])}, {s(row)}")
        e(f"buffer_load_dwordx4 %[v{mat}{p & 1}], %[rs{mat}], {s(S_TMP)} offen lds")

    dmat, smat = ("a", "b") if deep == "a" else ("b", "a")
    npc = {"a": 8, "b": npb}
    window = 2 * nslot
    # piece i of wave w of a matrix -> position in its two-step window.  (--shallow-window N, measurement: the SHALLOW operand's pieces in the
    # first N slots behind the barrier instead of the whole two-step window -- earlier issue, denser burst)
    swin = int(sys.argv[sys.argv.index("--shallow-window") + 1]) if "--shallow-window" in sys.argv else window
    sched = {m: {w: {} for w in range(4)} for m in "ab"}
    for m in "ab":
        for i in range(npc[m]):
            for w in range(4):
                g = (i * 4 + w) * (swin if m == smat else window) // (4 * npc[m])
                sched[m][w].setdefault(g, []).append(i)
    K0, K1, K2 = (S_K0A, S_K0B), (S_K1A, S_K1B), (S_K2A, S_K2B)
    # ---- setup (common)
    for k in range(4):
        if k:
            e(f"v_xor_b32 {v(XAA[0][k])}, {32 * k}, %[aa0]")
            e(f"v_xor_b32 {v(XAB[0][k])}, {32 * k}, %[ab0]")
        else:
            e(f"v_mov_b32 {v(XAA[0][0])}, %[aa0]")
            e(f"v_mov_b32 {v(XAB[0][0])}, %[ab0]")
    for st in range(1, SA):
        for k in range(4):
            e(f"v_add_u32 {v(XAA[st][k])}, {32768 * st}, {v(XAA[0][k])}")
    for st in range(1, SB):
        for k in range(4):
            e(f"v_add_u32 {v(XAB[st][k])}, {BSTG * st}, {v(XAB[0][k])}")
    e(f"s_mov_b32 {s(S_ROWA[0])}, 0")
    for q in (1, 2, 3):
        e(f"s_add_u32 {s(S_ROWA[q])}, {s(S_ROWA[q - 1])}, %[lda16]")
    e(f"s_mov_b32 {s(S_ROWB[0])}, 0")
    for q in (1, 2, 3):
        e(f"s_add_u32 {s(S_ROWB[q])}, {s(S_ROWB[q - 1])}, %[ldb16]")
    e(f"s_mov_b32 {s(S_KT)}, 0")
    e(f"s_lshl_b32 {s(S_DA)}, %[wave], 13")
    e(f"s_mul_i32 {s(S_DB)}, %[wave], {2048 * NJ}")
    e(f"s_add_u32 {s(S_DB)}, {s(S_DB)}, {OFFB}")
    e(f"s_mov_b32 {s(S_K0A)}, %[base_a]")
    e(f"s_mov_b32 {s(S_K0B)}, %[base_b]")
    e(f"s_add_u32 {s(S_K1A)}, %[base_a], 128")
    e(f"s_add_u32 {s(S_K1B)}, %[base_b], 128")
    e(f"s_add_u32 {s(S_K2A)}, %[base_a], 256")
    e(f"s_add_u32 {s(S_K2B)}, %[base_b], 256")
    e("s_cmp_eq_u32 %[nkt], 2")
    e(f"s_cselect_b32 {s(S_K2A)}, %[nbase_a], {s(S_K2A)}")
    e(f"s_cselect_b32 {s(S_K2B)}, %[nbase_b], {s(S_K2B)}")
    for w in (1, 2, 3):
        e(f"s_cmp_eq_u32 %[wave], {w}")
        e(f"s_cbranch_scc1 .Lbd_w{w}_%=")
    for w in range(4):
        if w:
            e(f".Lbd_w{w}_%=:")
        e("s_cmp_eq_u32 %[first], 0")
        e(f"s_cbranch_scc1 .Lbd_cont{w}_%=")
        # first tile of the workgroup (st0 = 0): K tile 0 of both operands, then what the steady state has in flight when an
        # iteration starts: all of the deep operand's K tile 1 and the part of the shallow one's that k16 step 3 issues
        for i in range(8):
            xpiece("a", i, 0, K0)
        for i in range(npb):
            xpiece("b", i, 0, K0)
        early = [i for g in range(nslot) for i in sched[smat][w].get(g, [])]
        for i in range(npc[dmat]):
            xpiece(dmat, i, 1, K1)
        for i in early:
            xpiece(smat, i, 1, K1)
        e(f"s_waitcnt vmcnt({npc[dmat] + len(early)})")
        e("s_barrier")
        e(f".Lbd_cont{w}_%=:")
        e("s_waitcnt lgkmcnt(0)")     # (a scalar load the compiler left in flight would return out of order and spoil the counted fragment waits)
        for u in range(1, L):
            e(f"s_cmp_eq_u32 %[st0], {u}")
            e(f"s_cbranch_scc1 .Lbd_e{w}_{u}_%=")
        for u in range(L):
            if u:
                e(f".Lbd_e{w}_{u}_%=:")
            for n in range(nread):
                xread(0, u % SA, u % SB, 0, n)
            e(f"s_branch .Lbd_b{w}_{u}_%=")
        for u in range(L):
            e(f".Lbd_b{w}_{u}_%=:")
            sa, sb = u % SA, u % SB
            na, nb = (u + 1) % SA, (u + 1) % SB
            st_of = {"a": sa, "b": sb}
            nstages = {"a": SA, "b": SB}
            shallow_next = (st_of[smat] + 1) % 2          # shallow tile t+1 -> its other stage
            deep_tgt = (st_of[dmat] + 2) % 3              # deep tile t+2 -> the stage tile t-1 left

            def xk(items, rb, rsa, rsb, rkk):
                """companions of a k16 step's slots: fragment reads + the DMA pieces in items: slot -> [(matrix, piece, stage, K)]"""
                def f(slot):
                    for n in range(slot * RPS, min(nread, slot * RPS + RPS)):
                        xread(rb, rsa, rsb, rkk, n)
                    for (m, i, st, s_k) in items.get(slot, []):
                        xpiece(m, i, st, s_k)
                return f

            def win(m, half, st, s_k):
                return {slot: [(m, i, st, s_k) for i in sched[m][w].get(half * nslot + slot, [])] for slot in range(nslot)}

            IN_LOOP[0] = True     # (--ablate-deep: measurement-only builds)
            mfma_block(0, xk(win(smat, 1, shallow_next, K1), 1, sa, sb, 1), xa, xb)
            mfma_block(1, xk(win(dmat, 0, deep_tgt, K2), 0, sa, sb, 2), xa, xb)
            mfma_block(0, xk(win(dmat, 1, deep_tgt, K2), 1, sa, sb, 3), xa, xb)
            e(f"s_waitcnt vmcnt({npc[dmat]})")
            e("s_barrier")
            # behind the barrier: the shallow operand's stage of tile t is free -> first half of its tile t+2
            mfma_block(1, xk(win(smat, 0, st_of[smat], K2), 0, na, nb, 0), xa, xb)
            IN_LOOP[0] = False
            e(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, 1")
            e(f"s_mov_b32 {s(S_K1A)}, {s(S_K2A)}")
            e(f"s_mov_b32 {s(S_K1B)}, {s(S_K2B)}")
            e(f"s_add_u32 {s(S_K2A)}, {s(S_K2A)}, 128")
            e(f"s_add_u32 {s(S_K2B)}, {s(S_K2B)}, 128")
            e(f"s_add_u32 {s(S_TMP)}, {s(S_KT)}, 2")
            e(f"s_cmp_eq_u32 {s(S_TMP)}, %[nkt]")
            e(f"s_cselect_b32 {s(S_K2A)}, %[nbase_a], {s(S_K2A)}")
            e(f"s_cselect_b32 {s(S_K2B)}, %[nbase_b], {s(S_K2B)}")
            e(f"s_cmp_lt_u32 {s(S_KT)}, %[nkt]")
            if u < L - 1:
                e(f"s_cbranch_scc0 .Lbd_done_%=")
            else:
                e(f"s_cbranch_scc1 .Lbd_b{w}_0_%=")
        if w < 3:
            e("s_branch .Lbd_done_%=")
    e(".Lbd_done_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 15")
    e("s_nop 7")
    return vhi


# ---- drain forms (round 6): tile i's epilogue under the K loop of tile i + 1 ------------------------------------------------
# What the K = 768 products of the ViT lose is not their K loop but the seam between two output tiles of a workgroup: the accumulators
# are the only copy of the tile, so swap -> scale / bias (-> GELU) -> pack -> store (25 MB per round of tiles, all 256 workgroups at
# once) runs with the matrix pipe idle, and the next tile's first vmcnt wait queues behind the stores (profiles/r03_bt_epilogue_*.log:
# 15 of the q|k|v product's 70 us are its three store bursts; fc1 + GELU 115 us against 80 for the plain product).  The drain forms of
# the deep 256 x 192 loop split the epilogue in two:
#   CONVERT  (exposed, ~1 us per tile): at the START of the asm statement of tile i + 1 the 192 accumulator registers of tile i are
#            read, scaled / biased in fp32 and packed to 96 registers of 16-bit elements (v132..v227, "hold"; 24 groups of 4 = 24
#            16-byte stores), in the row-major order a store wants (v_permlane32_swap on the PACKED pairs: 2 per group instead of 4);
#            the first k16 step of the tile's K loop takes C = 0, so nobody zeroes them.  A transposed tile (the q|k|v product's V tiles) packs 8 keys of one column.
#   DRAIN    (hidden): K iteration b (0..11) of tile i + 1 stores groups 2 b, 2 b + 1 from two MFMA slots; in the GELU form it first
#            runs gelu_fast2's instruction sequence (common.h; the very opcodes hipcc emits for it, four pairs interleaved) on the
#            held pre-activations from the slots of the whole iteration.  The held pre-activation is ROUNDED to the element type
#            first -- the reference's own rounding point (its nn.Linear returns bf16 before nn.GELU sees it).
# Static placement needs the iteration index in the code: the loop is unrolled over 12 "drain" bodies D0..D11 (two turns of the 6
# stage pairs) followed by the 6 plain bodies for longer K; a tile must start at stage pair 0, i.e. nkt % 6 == 0 and nkt >= 12 (the
# ViT: K = 768, 3072), and the statement is only used for the second and later tiles of a workgroup (first = 0: K tile 0 landed, K
# tile 1 in flight).  The stores ride the vector-memory counter the counted waits rely on: they are issued behind the iteration's
# last shallow-operand piece, so vmcnt(6 + 2) in front of the barrier still means "everything older has landed" (gfx9: loads and
# stores retire in order on one counter), and they have a whole iteration to complete.
# Registers: accumulators are the literal a[0:191] (operands pin them: "+{a[32 k : 32 k + 31]}"), hold v132..v227, temporaries
# v228..v255, s57..s71, parameters of the tile being drained in the pinned block s[72:87].
HOLD, TMP = 132, 228
S_CUR, S_ADJ, S_F = 57, 58, 59
SG_C4, SG_C2, SG_C1, SG_C0 = 60, 62, 64, 66
PRM = 72            # s[72:75] store descriptor, s76 byte offset of the wave tile, s77 / s78 strides per 32-row block / 32-column block,
                    # s79 alpha (bits), s80 flags: 1 = transposed tile, 2 = alpha != 1, 4 = bias
V_C3 = TMP + 24
V_AL = TMP + 26
GELU_BLOCKS = [int(c) for c in (sys.argv[sys.argv.index("--gelu-blocks") + 1] if "--gelu-blocks" in sys.argv else "12")]
# gelu_fast2 (common.h): x / (1 + 2^(x (c0 + c1 x^2 + c2 x^4 + c3 x^6 + c4 x^8))), the c_k carrying -log2 e
GELU_CONST = {SG_C4: 0xb65ad3ea, SG_C2: 0x39b9f159, SG_C1: 0xbdd77917, SG_C0: 0xc01354e9}
C3_BITS = 0x38baa019


def hold(g, k=0):
    return HOLD + 4 * g + k


def convert(vt, scale, bias):
    """accumulators -> hold, one flavour.  Group g = 2 (3 rb + ni) + t: registers 8 t .. 8 t + 7 of accumulator (rb, ni)."""
    T = [TMP + k for k in range(8)]
    B = [[TMP + 8 + k for k in range(8)], [TMP + 16 + k for k in range(8)]]

    def bias_reads(g):
        rb_ni, t = divmod(g, 2)
        ni = rb_ni % 3
        off = (32 * ni + 16 * t) * 4
        e(f"ds_read_b128 {vr(B[g & 1][0], 4)}, %[vbias] offset:{off}")
        e(f"ds_read_b128 {vr(B[g & 1][4], 4)}, %[vbias] offset:{off + 32}")

    if bias:
        bias_reads(0)
    for g in range(24):
        a0 = 16 * (g >> 1) + 8 * (g & 1)
        for k in range(8):
            e(f"v_accvgpr_read_b32 {v(T[k])}, a{a0 + k}")
        if bias and g + 1 < 24:
            bias_reads(g + 1)
        if scale and not bias:
            for k in range(0, 8, 2):
                e(f"v_pk_mul_f32 {vr(T[k], 2)}, {vr(T[k], 2)}, {vr(V_AL, 2)}")
        if bias:
            e(f"s_waitcnt lgkmcnt({2 if g + 1 < 24 else 0})")
            for k in range(0, 8, 2):
                if scale:   # acc * alpha + bias as ONE fused multiply-add: what hipcc makes of the other forms' epilogues (fp contraction)
                    e(f"v_pk_fma_f32 {vr(T[k], 2)}, {vr(T[k], 2)}, {vr(V_AL, 2)}, {vr(B[g & 1][k], 2)}")
                else:
                    e(f"v_pk_add_f32 {vr(T[k], 2)}, {vr(T[k], 2)}, {vr(B[g & 1][k], 2)}")
        for k in range(4):
            e(f"v_cvt_pk_bf16_f32 {v(hold(g, k))}, {v(T[2 * k])}, {v(T[2 * k + 1])}")
        if not vt:
            e("s_nop 1")
            e(f"v_permlane32_swap_b32 {v(hold(g, 0))}, {v(hold(g, 2))}")
            e(f"v_permlane32_swap_b32 {v(hold(g, 1))}, {v(hold(g, 3))}")


def gelu_chunk(g):
    """gelu_fast2 (common.h) on the 8 held values of group g, in place: the instruction sequence hipcc emits for it, the four pairs
    interleaved (a packed-math result is read 4 issues later: its one wait state and the transcendental's are both covered)."""
    X = [TMP + 2 * q for q in range(4)]          # pairs: x, then the result
    X2 = [TMP + 8 + 2 * q for q in range(4)]
    P = [TMP + 16 + 2 * q for q in range(4)]
    ins = []
    for q in range(4):
        ins.append(f"v_lshlrev_b32 {v(X[q])}, 16, {v(hold(g, q))}")
        ins.append(f"v_and_b32 {v(X[q] + 1)}, 0xffff0000, {v(hold(g, q))}")
    steps = [
        lambda q: f"v_pk_mul_f32 {vr(X2[q], 2)}, {vr(X[q], 2)}, {vr(X[q], 2)}",
        lambda q: f"v_pk_fma_f32 {vr(P[q], 2)}, {vr(X2[q], 2)}, s[{SG_C4}:{SG_C4 + 1}], {vr(V_C3, 2)} op_sel_hi:[1,0,0]",
        lambda q: f"v_pk_fma_f32 {vr(P[q], 2)}, {vr(P[q], 2)}, {vr(X2[q], 2)}, s[{SG_C2}:{SG_C2 + 1}] op_sel_hi:[1,1,0]",
        lambda q: f"v_pk_fma_f32 {vr(P[q], 2)}, {vr(P[q], 2)}, {vr(X2[q], 2)}, s[{SG_C1}:{SG_C1 + 1}] op_sel_hi:[1,1,0]",
        lambda q: f"v_pk_fma_f32 {vr(P[q], 2)}, {vr(P[q], 2)}, {vr(X2[q], 2)}, s[{SG_C0}:{SG_C0 + 1}] op_sel_hi:[1,1,0]",
        lambda q: f"v_pk_mul_f32 {vr(P[q], 2)}, {vr(X[q], 2)}, {vr(P[q], 2)}",
        lambda q: f"v_exp_f32 {v(P[q])}, {v(P[q])}",
        lambda q: f"v_exp_f32 {v(P[q] + 1)}, {v(P[q] + 1)}",
        lambda q: f"v_pk_add_f32 {vr(P[q], 2)}, {vr(P[q], 2)}, 1.0 op_sel_hi:[1,0]",
        lambda q: f"v_rcp_f32 {v(P[q])}, {v(P[q])}",
        lambda q: f"v_rcp_f32 {v(P[q] + 1)}, {v(P[q] + 1)}",
        lambda q: f"v_pk_mul_f32 {vr(X[q], 2)}, {vr(X[q], 2)}, {vr(P[q], 2)}",
    ]
    for st in steps:
        for q in range(4):
            ins.append(st(q))
    for q in range(4):
        ins.append(f"v_cvt_pk_bf16_f32 {v(hold(g, q))}, {v(X[q])}, {v(X[q] + 1)}")
    return ins


def store(g):
    return f"buffer_store_dwordx4 {vr(hold(g), 4)}, %[voff], s[{PRM}:{PRM + 3}], {s(S_CUR)} offen offset:{32 * (g & 1)}"


def gen_deep_drain(swap, gelu):
    global NJ
    NJ = 3
    del out[:]
    FIXED_ACC[0] = True
    SWAP[0] = swap
    SA, SB, L = 2, 3, 6
    nslot, nread, npb = 4 * NJ, 4 + NJ, 2 * NJ
    BSTG, OFFB, VB = NJ * 8192, 32768 * SA, 56
    XAA = [[VB + 4 * st + k for k in range(4)] for st in range(SA)]
    XAB = [[VB + 4 * SA + 4 * st + k for k in range(4)] for st in range(SB)]
    XBUF = VB + 4 * (SA + SB)
    assert XBUF + 2 * nread * 4 == HOLD

    def xa(b, i):
        if FRAG_AGPR and FIXED_ACC[0]:   # (--frag-agpr, measurement: the fragment double buffer in a[192:247], the accumulators' spare registers)
            return f"a[{192 + 4 * nread * b + 4 * i}:{192 + 4 * nread * b + 4 * i + 3}]"
        return vr(XBUF + 4 * nread * b + 4 * i, 4)

    def xb(b, j):
        if FRAG_AGPR and FIXED_ACC[0]:
            return f"a[{192 + 4 * nread * b + 16 + 4 * j}:{192 + 4 * nread * b + 16 + 4 * j + 3}]"
        return vr(XBUF + 4 * nread * b + 16 + 4 * j, 4)

    def xread(b, sa, sb, kk, n):
        order = [("a", 0)] + [("b", j) for j in range(NJ)] + [("a", 1), ("a", 2), ("a", 3)]
        m, idx = order[n]
        if m == "a":
            e(f"ds_read_b128 {xa(b, idx)}, {v(XAA[sa][kk])} offset:{4096 * idx}")
        else:
            e(f"ds_read_b128 {xb(b, idx)}, {v(XAB[sb][kk])} offset:{4096 * idx}")

    def xpiece(mat, p, st, s_k):
        if mat == "a":
            e(f"s_add_u32 m0, {s(S_DA)}, {32768 * st + 1024 * p}")
        else:
            e(f"s_add_u32 m0, {s(S_DB)}, {BSTG * st + 1024 * p}")
        row = (S_ROWA if mat == "a" else S_ROWB)[p >> 1]
        e(f"s_add_u32 {s(S_TMP)}, {s(s_k[0] if mat == 'a' else s_k[1])}, {s(row)}")
        e(f"buffer_load_dwordx4 %[v{mat}{p & 1}], %[rs{mat}], {s(S_TMP)} offen lds")

    dmat, smat = "b", "a"
    npc = {"a": 8, "b": npb}
    window = 2 * nslot
    sched = {m: {w: {} for w in range(4)} for m in "ab"}
    for m in "ab":
        for i in range(npc[m]):
            for w in range(4):
                g = (i * 4 + w) * window // (4 * npc[m])
                sched[m][w].setdefault(g, []).append(i)
    K1, K2 = (S_K1A, S_K1B), (S_K2A, S_K2B)
    # ---- setup (common)
    for k in range(4):
        if k:
            e(f"v_xor_b32 {v(XAA[0][k])}, {32 * k}, %[aa0]")
            e(f"v_xor_b32 {v(XAB[0][k])}, {32 * k}, %[ab0]")
        else:
            e(f"v_mov_b32 {v(XAA[0][0])}, %[aa0]")
            e(f"v_mov_b32 {v(XAB[0][0])}, %[ab0]")
    for st in range(1, SA):
        for k in range(4):
            e(f"v_add_u32 {v(XAA[st][k])}, {32768 * st}, {v(XAA[0][k])}")
    for st in range(1, SB):
        for k in range(4):
            e(f"v_add_u32 {v(XAB[st][k])}, {BSTG * st}, {v(XAB[0][k])}")
    e(f"s_mov_b32 {s(S_ROWA[0])}, 0")
    for q in (1, 2, 3):
        e(f"s_add_u32 {s(S_ROWA[q])}, {s(S_ROWA[q - 1])}, %[lda16]")
    e(f"s_mov_b32 {s(S_ROWB[0])}, 0")
    for q in (1, 2, 3):
        e(f"s_add_u32 {s(S_ROWB[q])}, {s(S_ROWB[q - 1])}, %[ldb16]")
    e(f"s_mov_b32 {s(S_KT)}, 0")
    e(f"s_lshl_b32 {s(S_DA)}, %[wave], 13")
    e(f"s_mul_i32 {s(S_DB)}, %[wave], {2048 * NJ}")
    e(f"s_add_u32 {s(S_DB)}, {s(S_DB)}, {OFFB}")
    e(f"s_add_u32 {s(S_K1A)}, %[base_a], 128")
    e(f"s_add_u32 {s(S_K1B)}, %[base_b], 128")
    e(f"s_add_u32 {s(S_K2A)}, %[base_a], 256")
    e(f"s_add_u32 {s(S_K2B)}, %[base_b], 256")
    # store offsets of the tile being drained: pair b = (32-row block b / 3, 32-column block b % 3)
    e(f"s_mov_b32 {s(S_CUR)}, {s(PRM + 4)}")
    e(f"s_lshl_b32 {s(S_ADJ)}, {s(PRM + 6)}, 1")
    e(f"s_sub_u32 {s(S_ADJ)}, {s(PRM + 5)}, {s(S_ADJ)}")
    if gelu:
        for r, bits in GELU_CONST.items():
            e(f"s_mov_b32 {s(r)}, 0x{bits:08x}")
    # ---- CONVERT: the accumulators hold the previous tile of this workgroup (18 wait states behind its last MFMA: the previous
    # statement ended with s_nop 15 / s_nop 7)
    e(f"v_mov_b32 {v(V_AL)}, {s(PRM + 7)}")
    e(f"v_mov_b32 {v(V_AL + 1)}, {s(PRM + 7)}")
    flavours = [(0, 0, 0, 0), (2, 0, 1, 0), (4, 0, 0, 1), (6, 0, 1, 1)] + ([] if gelu else [(1, 1, 0, 0), (3, 1, 1, 0)])
    for (code, vt, sc, bi) in flavours[1:]:
        e(f"s_cmp_eq_u32 {s(PRM + 8)}, {code}")
        e(f"s_cbranch_scc1 .Lcv_{code}_%=")
    for n, (code, vt, sc, bi) in enumerate(flavours):
        if n:
            e(f".Lcv_{code}_%=:")
        convert(vt, sc, bi)
        if n + 1 < len(flavours):
            e("s_branch .Lcv_done_%=")
    e(".Lcv_done_%=:")
    if gelu:
        e(f"v_mov_b32 {v(V_C3)}, 0x{C3_BITS:08x}")     # (behind CONVERT: its bias double buffer uses the register)
    for w in (1, 2, 3):
        e(f"s_cmp_eq_u32 %[wave], {w}")
        e(f"s_cbranch_scc1 .Lbd_w{w}_%=")
    for w in range(4):
        if w:
            e(f".Lbd_w{w}_%=:")
        e("s_waitcnt lgkmcnt(0)")
        for n in range(nread):
            xread(0, 0, 0, 0, n)
        st_slot = (nread + (w + 2) % 4)      # the store's slot in k16 steps 1 and 2: no fragment read, none of this wave's pieces
        for body in range(12 + L):
            drain = body < 12
            u = body % L
            e(f".Lbd_{'d' if drain else 'b'}{w}_{body if drain else u}_%=:")
            sa, sb = u % SA, u % SB
            na, nb = (u + 1) % SA, (u + 1) % SB
            st_of = {"a": sa, "b": sb}
            shallow_next = (st_of[smat] + 1) % 2
            deep_tgt = (st_of[dmat] + 2) % 3
            # drain work of this body: slot (0..47, k16 step major) -> instructions
            work = {}
            nstore = 0
            if drain:
                if gelu:
                    ins = gelu_chunk(2 * body) + gelu_chunk(2 * body + 1)
                    # Which k16 steps carry the GELU instructions (--gelu-blocks, default "12"): an iteration is as long as the latency
                    # of the SHALLOW operand's K tile, whose last pieces leave in step 0 and are waited for behind step 2 -- steps 1 and
                    # 2 run in that latency's shadow, instructions in steps 3 and 0 sit in front of the pieces and stretch the chain
                    # (measured: spread over all four steps the GELU costs what it costs exposed, profiles/r06_drain_probe.log)
                    slots = [b_ * nslot + sl for b_ in GELU_BLOCKS for sl in range(nslot)]
                    for n_, sl in enumerate(slots):
                        work[sl] = ins[n_ * len(ins) // len(slots):(n_ + 1) * len(ins) // len(slots)]
                    if body:                                   # the pair finished by the previous body
                        work[nslot + st_slot] = work.get(nslot + st_slot, []) + [store(2 * body - 2)]
                        work[2 * nslot + st_slot] = work.get(2 * nslot + st_slot, []) + [store(2 * body - 1)]
                        nstore = 2
                else:
                    work[nslot + st_slot] = [store(2 * body)]
                    work[2 * nslot + st_slot] = [store(2 * body + 1)]
                    nstore = 2

            def xk(blk, items, rb, rsa, rsb, rkk):
                def f(slot):
                    if slot < nread:
                        xread(rb, rsa, rsb, rkk, slot)
                    for (m, i, st, s_k) in items.get(slot, []):
                        xpiece(m, i, st, s_k)
                    for ins_ in work.get(blk * nslot + slot, []):
                        e(ins_)
                return f

            def win(m, half, st, s_k):
                return {slot: [(m, i, st, s_k) for i in sched[m][w].get(half * nslot + slot, [])] for slot in range(nslot)}

            IN_LOOP[0] = True
            # (D0 is always a tile's first iteration: its first k16 step opens the accumulators with C = 0)
            mfma_block(0, xk(0, win(smat, 1, shallow_next, K1), 1, sa, sb, 1), xa, xb, zero_c=(body == 0))
            mfma_block(1, xk(1, win(dmat, 0, deep_tgt, K2), 0, sa, sb, 2), xa, xb)
            mfma_block(0, xk(2, win(dmat, 1, deep_tgt, K2), 1, sa, sb, 3), xa, xb)
            e(f"s_waitcnt vmcnt({npc[dmat] + nstore})")
            e("s_barrier")
            mfma_block(1, xk(3, win(smat, 0, st_of[smat], K2), 0, na, nb, 0), xa, xb)
            IN_LOOP[0] = False
            if drain and (not gelu or body):
                e(f"s_add_u32 {s(S_CUR)}, {s(S_CUR)}, {s(PRM + 6) if ((body - 1 if gelu else body) % 3) < 2 else s(S_ADJ)}")
            if gelu and body == 11:
                e(store(22))
                e(store(23))
            e(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, 1")
            e(f"s_mov_b32 {s(S_K1A)}, {s(S_K2A)}")
            e(f"s_mov_b32 {s(S_K1B)}, {s(S_K2B)}")
            e(f"s_add_u32 {s(S_K2A)}, {s(S_K2A)}, 128")
            e(f"s_add_u32 {s(S_K2B)}, {s(S_K2B)}, 128")
            e(f"s_add_u32 {s(S_TMP)}, {s(S_KT)}, 2")
            e(f"s_cmp_eq_u32 {s(S_TMP)}, %[nkt]")
            e(f"s_cselect_b32 {s(S_K2A)}, %[nbase_a], {s(S_K2A)}")
            e(f"s_cselect_b32 {s(S_K2B)}, %[nbase_b], {s(S_K2B)}")
            if body == 11 or (not drain and u == L - 1):
                e(f"s_cmp_lt_u32 {s(S_KT)}, %[nkt]")
                e(f"s_cbranch_scc{'0' if body == 11 else '1'} .Lbd_{'done' if body == 11 else f'b{w}_0'}_%=")
            # (a tile ends behind D11 or behind the last plain body: nkt is a multiple of 6 and at least 12)
        if w < 3:
            e("s_branch .Lbd_done_%=")
    e(".Lbd_done_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 15")
    e("s_nop 7")
    FIXED_ACC[0] = False
    SWAP[0] = False


# ---- measured and removed in round 4 (history: commits "eight-wave forms ...", "L2-prefetch trial ...") ----------------------------
#  * eight-wave forms of all three tile widths (two waves per SIMD, 128 x 64 / 64 x 96 / 64 x 64 accumulators per wave): tie the
#    four-wave loops on every shape (profiles/r04_bt8_vit.log, r04_bt8_tok.log) -- the K loop is not bound by one wave's in-order issue.
#  * --ablate builds (profiles/r04_bt_kloop_ablations.log), 2048 x 12288 x 4096 cold: DMA stream alone 160 us, MFMAs alone 110, both
#    197: the stream is latency-bound (one 56 KB K tile in flight per CU = 45 GB/s per CU against ~1.25 us), which is what
#    the ring form's third stage buys back for 256 x 128 tiles; 256 x 192 / 256 x 256 stages do not fit three times.
#  * L2 prefetch (a dword per 128-byte row segment, 2 / 4 / 8 K tiles ahead, through buffer_load_dword ... lds): the scattered
#    touches cost the address path more than the latency they hide (r04_bt_l2_prefetch_*.log): ViT q|k|v 71 -> 80 us.


print("// GENERATED by tools/gen_gemm_bt_asm.py -- do not edit")
print("// clang-format off")
# --ablate a,b: measurement-only build (WRONG results) of the four-wave two-stage loops, see ABL
ABLATE = set(sys.argv[sys.argv.index("--ablate") + 1].split(",")) if "--ablate" in sys.argv else set()
# --ablate-deep a,b: the same for the deep forms (variants 24 / 26 and the transposed twin; the drain forms stay whole)
ABLATE_DEEP = set(sys.argv[sys.argv.index("--ablate-deep") + 1].split(",")) if "--ablate-deep" in sys.argv else set()
for nj, abl in ((4, ""), (3, ""), (3, "pair")):
    ABL.clear()
    ABL.update(ABLATE)
    gen(nj, pair=abl == "pair")
    print(f"#define GEMM_BT_ASM_TEXT_NJ{nj}{('_' + abl.upper()) if abl else ''} \\")
    for i, line in enumerate(out):
        print(f'  "{line}\\n"' + (" \\" if i + 1 < len(out) else ""))
ABL.clear()
gen_ring(2)
print("#define GEMM_BT_ASM_TEXT_NJ2_RING \\")
for i, line in enumerate(out):
    print(f'  "{line}\\n"' + (" \\" if i + 1 < len(out) else ""))
# deep forms: B (the weight operand) keeps three stages.  The A-deep twins of round 4 measured the same or slower on every shape and
# were never selected: gone.  "_T" = the same loop with the MFMA operands exchanged: the accumulator tile comes out
# transposed, which is how the ViT's q|k|v product writes V^T (the flash kernel's operand) straight from its V tiles -- a lane then
# holds 8 keys of one head-dim column in exactly the [0-3, 8-11 | 4-7, 12-15] order of that layout (gemm_bt.hip: vt_epilogue).
deep_hi = {}
for nj, deep, swap in ((3, "b", False), (4, "b", False), (3, "b", True)):
    ABL.clear()
    ABL.update(ABLATE_DEEP)
    SWAP[0] = swap
    deep_hi[(nj, deep)] = gen_deep(nj, deep)
    SWAP[0] = False
    print(f"#define GEMM_BT_ASM_TEXT_NJ{nj}_D{deep.upper()}{'_T' if swap else ''} \\")
    for i, line in enumerate(out):
        print(f'  "{line}\\n"' + (" \\" if i + 1 < len(out) else ""))
# round 6: the deep 256 x 192 loops on the PINNED accumulator file (a[0:191] literal) -- the first tile of a workgroup of the drain kernel --
# and the drain forms (gen_deep_drain): plain, transposed tile, GELU
for swap in (False, True):
    ABL.clear()
    FIXED_ACC[0], SWAP[0] = True, swap
    gen_deep(3, "b")
    FIXED_ACC[0], SWAP[0] = False, False
    print(f"#define GEMM_BT_ASM_TEXT_NJ3_DB_FX{'_T' if swap else ''} \\")
    for i, line in enumerate(out):
        print(f'  "{line}\\n"' + (" \\" if i + 1 < len(out) else ""))
for swap, gelu in ((False, False), (True, False), (False, True)):
    ABL.clear()
    gen_deep_drain(swap, gelu)
    print(f"#define GEMM_BT_ASM_TEXT_NJ3_DB_DRAIN{'_T' if swap else ''}{'_GELU' if gelu else ''} \\")
    for i, line in enumerate(out):
        print(f'  "{line}\\n"' + (" \\" if i + 1 < len(out) else ""))
for name, lo, hi in (("GEMM_BT_ASM_CLOBBERS", VLO, VHI), ("GEMM_BT_ASM_CLOBBERS_RING", RVLO, RVHI),
                     ("GEMM_BT_ASM_CLOBBERS_NJ3_DEEP", 56, deep_hi[(3, "b")]), ("GEMM_BT_ASM_CLOBBERS_NJ4_DEEP", 56, deep_hi[(4, "b")])):
    clob = [f'"v{i}"' for i in range(lo, hi + 1)] + [f'"s{i}"' for i in range(SLO, SHI + 1)] + ['"scc"', '"memory"']
    print(f"#define {name} \\")
    for i in range(0, len(clob), 12):
        tail = ", \\" if i + 12 < len(clob) else ""
        print("  " + ", ".join(clob[i:i + 12]) + tail)
# the drain statements: the K loop's registers, the hold and the temporaries, s36..s71 (the accumulators are pinned operands)
clob = [f'"v{i}"' for i in range(56, 256)] + [f'"s{i}"' for i in range(SLO, 72)] + ['"scc"', '"memory"']
print("#define GEMM_BT_ASM_CLOBBERS_DRAIN \\")
for i in range(0, len(clob), 12):
    print("  " + ", ".join(clob[i:i + 12]) + (", \\" if i + 12 < len(clob) else ""))
# the accumulator file of the drain kernel: named literally by its statements, which list it as clobbered (no operand carries it: the
# compiler would shuttle 192 loop-carried values through VGPRs and scratch around every statement) -- gemm_bt.hip, gemm_bt_drain_kernel
clob = [f'"a{i}"' for i in range(248 if FRAG_AGPR else 192)]
print("#define GEMM_BT_ASM_CLOBBERS_ACC192 \\")
for i in range(0, len(clob), 12):
    print("  " + ", ".join(clob[i:i + 12]) + (", \\" if i + 12 < len(clob) else ""))
print("#define GEMM_BT_ASM_TEXT_ACC192_ZERO \\")
for i in range(192):
    print(f'  "v_accvgpr_write_b32 a{i}, 0\\n"' + (" \\" if i < 191 else ""))
# read-out of the accumulator file into the pinned VGPR tuples v[56 + 16 k : 71 + 16 k] (k = 3 i + j: accumulator (i, j)); opens with the
# wait states an MFMA result needs before a VALU may read it (the K loop's own tail has them too: harmless twice)
print("#define GEMM_BT_ASM_TEXT_ACC192_READ \\")
print('  "s_nop 15\\n" \\')
print('  "s_nop 7\\n" \\')
for i in range(192):
    print(f'  "v_accvgpr_read_b32 v{56 + i}, a{i}\\n"' + (" \\" if i < 191 else ""))
print("// clang-format on")
