"""CPU model of gemm_bt.hip's index arithmetic (the file was gemm_pp.hip in round 1) (no GPU needed): DMA piece -> LDS image (source-side swizzle),
fragment read addresses, v_mfma_f32_32x32x16_bf16 lane maps, accumulator -> C map.  Run: python tools/sim_gemm_pp.py"""
import itertools
import numpy as np


def swz(BK, row):
    return (row >> 1) & 7 if BK == 64 else (row >> 2) & 3


def run(BN, BK, NS, M=256, K=128, seed=0):
    ROWB, CPR = BK * 2, BK // 8
    RPP = 64 // CPR
    ROWS = 256 + BN
    STAGE = ROWS * ROWB
    NP = STAGE // 1024
    PG0 = (NP // 4 + 1) // 2
    PG1 = NP // 4 - PG0
    KSTEPS, NI = BK // 16, BN // 64
    rng = np.random.default_rng(seed)
    A = rng.integers(-4, 5, size=(M, K)).astype(np.float64)
    B = rng.integers(-4, 5, size=(BN, K)).astype(np.float64)
    C = np.zeros((M, BN))
    acc = {}  # (G, j, lane, mi, ni, r) -> float
    seen_pieces = set()
    for kt in range(K // BK):
        lds = np.full((STAGE // 2,), np.nan)  # bf16 elements
        for G in (0, 1):
            PG, P0 = (PG0, 0) if G == 0 else (PG1, 4 * PG0)
            for j in range(4):
                for i in range(PG):
                    p = P0 + i * 4 + j
                    seen_pieces.add(p)
                    for lane in range(64):
                        r = p * RPP + lane // CPR
                        gc = (lane % CPR) ^ swz(BK, r)
                        src = A[min(r, M - 1)] if r < 256 else B[r - 256]
                        dst = ((P0 + j) * 1024 + i * 4096 + lane * 16) // 2
                        lds[dst:dst + 8] = src[kt * BK + gc * 8: kt * BK + gc * 8 + 8]
        assert not np.isnan(lds).any()
        for G, j, lane in itertools.product((0, 1), range(4), range(64)):
            hi, l31 = lane >> 5, lane & 31
            wm2, wn2 = j >> 1, j & 1
            a_base = (G * 128 + wm2 * 64 + l31) * ROWB
            b_base = (256 + wn2 * (BN // 2) + l31) * ROWB
            for ks in range(KSTEPS):
                koff = ((ks * 2 + hi) ^ swz(BK, l31)) << 4
                for mi in range(2):
                    o = (a_base + mi * 32 * ROWB + koff) // 2
                    acc[("x", G, j, lane, ks, mi)] = lds[o:o + 8].copy()
                for ni in range(NI):
                    o = (b_base + ni * 32 * ROWB + koff) // 2
                    acc[("w", G, j, lane, ks, ni)] = lds[o:o + 8].copy()
        # MFMA: D[i][jc] += sum_k Aop[i][k] * Bop[jc][k]; Aop lane l: i = l&31, k = 8*(l>>5)+e ; Bop same with jc
        for G, j in itertools.product((0, 1), range(4)):
            wm2, wn2 = j >> 1, j & 1
            for ks, mi, ni in itertools.product(range(KSTEPS), range(2), range(NI)):
                Aop = np.zeros((32, 16)); Bop = np.zeros((32, 16))
                for lane in range(64):
                    Aop[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = acc[("w", G, j, lane, ks, ni)]
                    Bop[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = acc[("x", G, j, lane, ks, mi)]
                D = Aop @ Bop.T  # [n][m]
                for lane in range(64):
                    for r in range(16):
                        row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                        key = ("c", G, j, lane, mi, ni, r)
                        acc[key] = acc.get(key, 0.0) + D[row, lane & 31]
    assert seen_pieces == set(range(NP)), (sorted(seen_pieces), NP)
    # epilogue map: v_permlane32_swap(vdst = quad 2t, src = quad 2t+1): vdst[32..63] <-> src[0..31]
    for G, j in itertools.product((0, 1), range(4)):
        for mi, ni, t, e in itertools.product(range(2), range(NI), range(2), range(4)):
            for l in range(32):
                klo_src = ("c", G, j, l, mi, ni, 8 * t + 4 + e)
                khi_dst = ("c", G, j, l + 32, mi, ni, 8 * t + e)
                acc[khi_dst], acc[klo_src] = acc[klo_src], acc[khi_dst]
    for G, j, lane in itertools.product((0, 1), range(4), range(64)):
        hi, l31 = lane >> 5, lane & 31
        wm2, wn2 = j >> 1, j & 1
        m_base = G * 128 + wm2 * 64 + l31
        n_base = wn2 * (BN // 2) + 8 * hi
        for mi, ni, t, e in itertools.product(range(2), range(NI), range(2), range(8)):
            C[m_base + mi * 32, n_base + ni * 32 + t * 16 + e] = acc[("c", G, j, lane, mi, ni, 8 * t + e)]
    ref = A @ B.T
    ok = np.array_equal(C, ref)
    # bank-conflict check of the fragment reads (ds_read_b128: 4 groups of 16 lanes, bank = (addr/4) % 64)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups]
    worst = 1
    for ks in range(KSTEPS):
        for g in groups:
            slots = {}
            for lane in g:
                hi, l31 = lane >> 5, lane & 31
                addr = l31 * ROWB + (((ks * 2 + hi) ^ swz(BK, l31)) << 4)
                slots.setdefault((addr // 16) % 16, set()).add(addr)
            worst = max(worst, max(len(v) for v in slots.values()))
    return ok, worst, dict(NP=NP, PG0=PG0, PG1=PG1, STAGE=STAGE, LDS=NS * STAGE)


if __name__ == "__main__":
    for cfg in [(128, 64, 3), (192, 64, 2), (256, 64, 2), (256, 32, 4), (192, 32, 4)]:
        print(cfg, run(*cfg))
