"""CPU: executable statement of the two LDS layouts that feed ds_read_b64_tr_b16 (no GPU needed).

What the instruction does was measured with tools/ubench/tr_read.hip on an MI355X: inside a 16-lane group, lane i receives
element i & 3 of the 8-byte segments that lanes (i >> 2) + 4 j (j = 0..3) point at.  With lane a pointing at (row a >> 2,
piece a & 3) of a [4 rows][16 columns] block this hands lane i column i of the four rows -- a 4 x 16 transpose.

Checked here, with the index arithmetic of the kernels restated in Python:
  * gemm.hip, K-major operand tiles ([64 k][BX] with rotated 16-byte chunks): every MFMA fragment lane ends up with
    k = 8 (lane >> 4) + {0..7} of column lane & 15 of its fragment, for both tile widths; the two 16-lane groups of a half wave
    touch 16 different 16-byte slots of the 256-byte bank row (conflict-free; PMC: SQ_LDS_BANK_CONFLICT = 0);
  * attn_bwd.hip, shared row-major [64][64] tiles with the bit-reversed row-pair swizzle: the transposed 32 x 32 fragment gets
    rows r0 + 4 hi + {0..3}, r0 + 8 + 4 hi + {0..3} of column d in the accumulator's k-slot order, and both access patterns
    (row fragments with ds_read_b128, transpose reads) are conflict-free, while the forward kernel's plain swizzle is not for
    the transpose reads (what profiles/r02_attn_bwd_lds_pmc.txt measured).
"""
import itertools


def tr_read(addr_of_lane, lds):
    """ds_read_b64_tr_b16 for one wave: addr_of_lane[l] = byte address; lds: dict byte address of a 2-byte element -> value.
    Returns out[l] = 4 values."""
    out = []
    for lane in range(64):
        g, i = lane & ~15, lane & 15
        vals = []
        for j in range(4):
            src = g + (i >> 2) + 4 * j
            vals.append(lds[addr_of_lane[src] + 2 * (i & 3)])
        out.append(vals)
    return out


def slots_of(addrs):
    """16-byte slots inside the 256-byte bank row touched by 8-byte accesses"""
    return [(a % 256) // 16 for a in addrs]


# ------------------------------------------------------------------------------------------------ gemm.hip K-major tiles
def kmaj_rot(cprx, k):
    return (2 * (k & 3) + 8 * ((k >> 3) & 1)) if cprx == 16 else (2 * ((k >> 1) & 1) + 4 * ((k >> 3) & 1))


def test_gemm_kmajor_tile_fragments_and_banks():
    for bx in (128, 64):
        cprx = bx // 8
        # LDS image as the DMA writes it: slot (krow, slotcol) holds global chunk cc = (slotcol - rot(krow)) mod cprx
        lds = {}
        for krow in range(64):
            for slotcol in range(cprx):
                cc = (slotcol - kmaj_rot(cprx, krow)) % cprx
                for e in range(8):
                    lds[(krow * cprx + slotcol) * 16 + 2 * e] = (krow, cc * 8 + e)      # (k, column)
        for frag_col0, kk in itertools.product(range(0, bx, 16), range(2)):
            for second in (0, 1):
                addr = []
                for lane in range(64):
                    a, kg = lane & 15, lane >> 4
                    krow = 8 * kg + (a >> 2)                                           # per-lane part (fta / ftb)
                    cc = frag_col0 // 8 + ((a & 3) >> 1)
                    off = (krow * cprx + ((cc + kmaj_rot(cprx, krow)) % cprx)) * 16 + (a & 1) * 8
                    addr.append(off + kk * 32 * cprx * 16 + second * 4 * cprx * 16)
                got = tr_read(addr, lds)
                for lane in range(64):
                    k0 = kk * 32 + 8 * (lane >> 4) + 4 * second
                    assert got[lane] == [(k0 + j, frag_col0 + (lane & 15)) for j in range(4)], (bx, frag_col0, kk, lane)
                for half in (0, 32):                                                    # conflict model: per half wave
                    touched = set()
                    for lane in range(half, half + 32):
                        if (lane & 1) == 0:                                             # two lanes share a 16-byte slot
                            touched.add((addr[lane] % 256) // 16)
                    assert len(touched) == 16, (bx, frag_col0, kk, second, half, sorted(touched))


# ------------------------------------------------------------------------------------------------ attn_bwd.hip tiles
def rev3(p):
    return ((p & 1) << 2) | (p & 2) | ((p >> 2) & 1)


def tile_off(row, chunk, swz):
    p = (row >> 1) & 7
    return row * 128 + ((chunk ^ (rev3(p) if swz == "rev" else p)) << 4)


def attn_tr_addrs(nb, r0, second, swz):
    addr = []
    for lane in range(64):
        a = lane & 15
        chunk = 4 * nb + 2 * ((lane >> 4) & 1) + ((a & 3) >> 1)
        row = r0 + 4 * (lane >> 5) + (a >> 2) + 8 * second
        addr.append(tile_off(row, chunk, swz) + (a & 1) * 8)
    return addr


def test_attention_backward_tiles_transpose_reads_and_banks():
    lds = {}
    for row in range(64):
        for chunk in range(8):
            for e in range(8):
                lds[tile_off(row, chunk, "rev") + 2 * e] = (row, chunk * 8 + e)
    for nb, r0 in itertools.product(range(2), range(0, 64, 16)):
        for second in (0, 1):
            addr = attn_tr_addrs(nb, r0, second, "rev")
            got = tr_read(addr, lds)
            for lane in range(64):
                rows = [r0 + 8 * second + 4 * (lane >> 5) + j for j in range(4)]
                assert got[lane] == [(r, nb * 32 + (lane & 31)) for r in rows], (nb, r0, second, lane)
    # k-slot order of an accumulator fed back as the B operand: slot jj <-> 16 ks2 + 8 (jj >> 2) + 4 hi + (jj & 3)
    for hi in (0, 1):
        slots = [4 * hi + j for j in range(4)] + [8 + 4 * hi + j for j in range(4)]
        assert slots == [8 * (jj >> 2) + 4 * hi + (jj & 3) for jj in range(8)]

    def conflicts(swz):
        worst = 1
        for nb, r0, second in itertools.product(range(2), range(0, 64, 16), (0, 1)):
            addr = attn_tr_addrs(nb, r0, second, swz)
            for half in (0, 32):
                banks = {}
                for lane in range(half, half + 32):
                    for w in range(2):                                                  # 8 bytes = 2 banks
                        banks.setdefault((addr[lane] // 4 + w) % 64, set()).add(addr[lane] // 4 + w)
                worst = max(worst, max(len(v) for v in banks.values()))
        return worst

    assert conflicts("rev") == 1, "transpose reads must be conflict-free with the bit-reversed swizzle"
    assert conflicts("plain") == 2, "the forward kernel's swizzle makes them 2-way conflicts (the PMC finding)"
    # row fragments (ds_read_b128, 16 consecutive rows at one chunk per 16-lane group) stay conflict-free
    for swz in ("rev", "plain"):
        for base, chunk in itertools.product(range(0, 64, 16), range(8)):
            slots = {(tile_off(base + r, chunk, swz) % 256) // 16 for r in range(16)}
            assert len(slots) == 16, (swz, base, chunk)
