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


def _bt_layout(nj):
    """LDS contents of one K tile as the DMA pieces of gemm_bt_kernel<NJ> write them (formulas of gemm_bt.hip):
    byte address -> (matrix, tile row, 16-byte K chunk)."""
    lds = {}
    for wave in range(4):
        for lane in range(64):
            pr, sw0, pc = lane >> 3, (lane >> 4) & 3, lane & 7
            for mat, rows_per_wave, base in (("a", 64, 0), ("b", 16 * nj, 32768)):
                for p in range(rows_per_wave // 8):
                    row = wave * rows_per_wave + p * 8 + pr            # va0 / va1 + (p >> 1) * 16 rows in the scalar offset
                    chunk = pc ^ sw0 ^ (4 if p & 1 else 0)              # the odd piece's lane offset carries sw0 ^ 4
                    addr = base + wave * rows_per_wave * 128 + p * 1024 + lane * 16   # M0 = wave base + 1024 p, lane-linear
                    assert addr not in lds
                    lds[addr] = (mat, row, chunk)
    return lds


def test_gemm_bt_fragment_reads_hit_the_rows_the_dma_wrote():
    """Executable spec of the big-tile GEMM's LDS layout: every ds_read_b128 of an MFMA fragment (address registers and
    immediate offsets taken from the generated asm) must land on the (row, K chunk) the MFMA operand layout wants, in
    data some DMA piece wrote, and the 16 lanes of each quarter wave must touch 16 different 16-byte bank groups."""
    import re
    text = (CSRC / "gemm_bt_asm.inc").read_text()
    for nj in (4, 3):
        body = re.search(rf"#define GEMM_BT_ASM_TEXT_NJ{nj} \\\n(.*?)\n#define", text, re.S).group(1)
        lines = re.findall(r'"(.*)\\n"', body)
        # address registers: v_xor_b32 vR, 32*k, %[aa0|ab0]; v_mov_b32 vR, %[aa0|ab0]
        areg = {}
        for l in lines:
            m = re.match(r"v_xor_b32 v(\d+), (\d+), %\[(aa0|ab0)\]", l)
            if m:
                areg[int(m.group(1))] = (m.group(3)[1], int(m.group(2)))
            m = re.match(r"v_mov_b32 v(\d+), %\[(aa0|ab0)\]", l)
            if m:
                areg[int(m.group(1))] = (m.group(2)[1], 0)
        assert sorted(x for _, x in areg.values()) == [0, 0, 32, 32, 64, 64, 96, 96]
        reads = set()
        for l in lines:
            m = re.match(r"ds_read_b128 v\[(\d+):\d+\], v(\d+) offset:(\d+)", l)
            if m:
                mat, kx = areg[int(m.group(2))]
                reads.add((mat, kx >> 5, int(m.group(3))))          # (matrix, k16 step kk, immediate offset)
        assert reads == {("a", kk, 4096 * i) for kk in range(4) for i in range(4)} | \
                        {("b", kk, 4096 * j) for kk in range(4) for j in range(nj)}
        lds = _bt_layout(nj)
        for wave in range(4):
            wm, wn = wave >> 1, wave & 1
            for (mat, kk, off) in reads:
                slots = []
                for lane in range(64):
                    hi, l31 = lane >> 5, lane & 31
                    abk0 = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4)
                    base = wm * 16384 if mat == "a" else 32768 + wn * (nj * 4096)
                    addr = (base + abk0) ^ (kk << 5)                  # aa0 / ab0 of the kernel, xor 32 kk in the asm
                    addr += off
                    want_row = (wm * 128 if mat == "a" else wn * 32 * nj) + (off // 4096) * 32 + l31
                    assert lds[addr] == (mat, want_row, kk * 2 + hi), (nj, wave, mat, kk, off, lane)
                    slots.append((addr // 16) % 16)
                for q in range(4):
                    assert len(set(slots[16 * q:16 * q + 16])) == 16  # conflict-free ds_read_b128


def _ring_lines():
    import re
    text = (CSRC / "gemm_bt_asm.inc").read_text()
    body = re.search(r"#define GEMM_BT_ASM_TEXT_NJ2_RING \\\n(.*?)\n#define", text, re.S).group(1)
    return re.findall(r'"(.*)\\n"', body)


def test_gemm_bt_ring_schedule_and_counted_waits():
    """Ring form (256 x 128 tiles, three LDS stages): every body of every wave issues the 12 pieces of ONE K tile into the stage
    two ahead of the one it computes on, 32 MFMAs, 24 fragment reads (18 from its own stage, 6 from the next one behind the
    barrier), and its one counted wait names exactly the pieces issued in front of it -- so that everything older (the whole
    of the K tile the next iteration computes on) has landed."""
    import re
    lines = _ring_lines()
    for w in range(4):
        for st in range(3):
            i0 = lines.index(f".Lbr_b{w}_{st}_%=:")
            i1 = next(k for k in range(i0, len(lines)) if lines[k].startswith("s_cmp_lt_u32"))
            body = lines[i0:i1]
            assert sum(l.startswith("v_mfma") for l in body) == 32
            assert sum(l.startswith("s_barrier") for l in body) == 1
            bar = body.index("s_barrier")
            wait = re.match(r"s_waitcnt vmcnt\((\d+)\)", body[bar - 1])
            assert wait and int(wait.group(1)) == sum("buffer_load_dwordx4" in l for l in body[:bar])
            assert sum("buffer_load_dwordx4" in l for l in body) == 12
            # DMA targets: m0 = wave base register + stage * 49152 + 1024 * piece, stage = (st + 2) % 3
            m0 = [int(re.match(r"s_add_u32 m0, s(\d+), (\d+)", l).group(2)) for l in body if l.startswith("s_add_u32 m0")]
            assert sorted(x - 49152 * ((st + 2) % 3) for x in m0) == sorted([1024 * p for p in range(8)] + [1024 * p for p in range(4)])
            # fragment reads: address registers of stage st in front of the barrier, of stage st + 1 behind it
            regs = [int(re.match(r"ds_read_b128 v\[\d+:\d+\], v(\d+)", l).group(1)) for l in body if l.startswith("ds_read")]
            nb = sum(l.startswith("ds_read") for l in body[:bar])
            assert len(regs) == 24 and nb == 18
            assert all(56 + 4 * st <= r < 60 + 4 * st or 68 + 4 * st <= r < 72 + 4 * st for r in regs[:nb])
            nx = (st + 1) % 3
            assert all(r in (56 + 4 * nx, 68 + 4 * nx) for r in regs[nb:])


def test_gemm_bt_ring_fragment_reads_hit_the_rows_the_dma_wrote():
    """Executable spec of the ring form's LDS layout: stage s = [A tile 32 KB | B tile 16 KB] at 49152 s; the address registers the
    asm derives for stage s (v_add_u32 of 49152 s to the xor-ed stage-0 address) read, for every wave and k16 step, the
    (row, K chunk) the MFMA operand wants out of bytes a DMA piece of that stage wrote; quarter waves are conflict-free."""
    import re
    lines = _ring_lines()
    nj = 2
    areg = {}
    for l in lines:
        m = re.match(r"v_xor_b32 v(\d+), (\d+), %\[(aa0|ab0)\]", l)
        if m:
            areg[int(m.group(1))] = (m.group(3)[1], int(m.group(2)), 0)
        m = re.match(r"v_mov_b32 v(\d+), %\[(aa0|ab0)\]", l)
        if m:
            areg[int(m.group(1))] = (m.group(2)[1], 0, 0)
    for l in lines:
        m = re.match(r"v_add_u32 v(\d+), (\d+), v(\d+)", l)
        if m:
            mat, kx, _ = areg[int(m.group(3))]
            areg[int(m.group(1))] = (mat, kx, int(m.group(2)))
    assert len(areg) == 24 and sorted({x for _, _, x in areg.values()}) == [0, 49152, 98304]
    reads = set()
    for l in lines:
        m = re.match(r"ds_read_b128 v\[(\d+):\d+\], v(\d+) offset:(\d+)", l)
        if m:
            mat, kx, stage = areg[int(m.group(2))]
            reads.add((mat, kx >> 5, stage, int(m.group(3))))
    assert reads == {("a", kk, 49152 * st, 4096 * i) for kk in range(4) for st in range(3) for i in range(4)} | \
                    {("b", kk, 49152 * st, 4096 * j) for kk in range(4) for st in range(3) for j in range(nj)}
    lds0 = _bt_layout(nj)          # one stage as the DMA pieces write it (same piece -> row / chunk formulas as the two-stage form)
    assert max(lds0) < 49152
    for wave in range(4):
        wm, wn = wave >> 1, wave & 1
        for (mat, kk, stage, off) in reads:
            slots = []
            for lane in range(64):
                hi, l31 = lane >> 5, lane & 31
                abk0 = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4)
                base = wm * 16384 if mat == "a" else 32768 + wn * (nj * 4096)
                addr = ((base + abk0) ^ (kk << 5)) + stage + off
                want_row = (wm * 128 if mat == "a" else wn * 32 * nj) + (off // 4096) * 32 + l31
                assert lds0[addr - stage] == (mat, want_row, kk * 2 + hi), (wave, mat, kk, stage, off, lane)
                slots.append((addr // 16) % 16)
            for q in range(4):
                assert len(set(slots[16 * q:16 * q + 16])) == 16


def test_gemm_bt_deep_forms_schedule_and_layout():
    """Deep forms (three LDS stages for the streaming operand, two for the other; 256 x 192 and 256 x 256 tiles): per wave and
    body the pieces of one K tile of each operand, issued shallow-first / deep-last so that the one counted wait (= the deep
    operand's pieces per wave) covers everything older; targets: shallow tile t+1 -> its other stage in front of the barrier,
    its tile t+2 -> the stage just left behind it, deep tile t+2 -> stage (t + 2) mod 3; fragment reads from the (t mod 3,
    t mod 2) stage pair, behind the barrier from the next pair; the fragment reads return what the DMA pieces wrote."""
    import re
    text = (CSRC / "gemm_bt_asm.inc").read_text()
    for nj, sfx in ((3, ""), (4, ""), (3, "_T")):       # ("_T": the transposed-tile twin, checked against its sibling below)
        for deep in "b":
            body = re.search(rf"#define GEMM_BT_ASM_TEXT_NJ{nj}_D{deep.upper()}{sfx} \\\n(.*?)\n#define", text, re.S).group(1)
            lines = re.findall(r'"(.*)\\n"', body)
            SA, SB = (3, 2) if deep == "a" else (2, 3)
            offb, bstg = 32768 * SA, nj * 8192
            npc = {"a": 8, "b": 2 * nj}
            nd, ns = npc[deep], npc["b" if deep == "a" else "a"]
            # address registers -> (matrix, kk * 32, stage offset)
            areg = {}
            for l in lines:
                m = re.match(r"v_xor_b32 v(\d+), (\d+), %\[(aa0|ab0)\]", l)
                if m:
                    areg[int(m.group(1))] = (m.group(3)[1], int(m.group(2)), 0)
                m = re.match(r"v_mov_b32 v(\d+), %\[(aa0|ab0)\]", l)
                if m:
                    areg[int(m.group(1))] = (m.group(2)[1], 0, 0)
            for l in lines:
                m = re.match(r"v_add_u32 v(\d+), (\d+), v(\d+)", l)
                if m:
                    mat, kx, _ = areg[int(m.group(3))]
                    areg[int(m.group(1))] = (mat, kx, int(m.group(2)))
            assert len(areg) == 20
            sda = int(re.search(r"s_lshl_b32 s(\d+), %\[wave\], 13", "\n".join(lines)).group(1))
            for w in range(4):
                for u in range(6):
                    i0 = lines.index(f".Lbd_b{w}_{u}_%=:")
                    i1 = next(k for k in range(i0, len(lines)) if lines[k].startswith("s_cmp_lt_u32"))
                    bl = lines[i0:i1]
                    assert sum(l.startswith("v_mfma") for l in bl) == 16 * nj
                    assert sum(l.startswith("s_barrier") for l in bl) == 1
                    bar = bl.index("s_barrier")
                    assert bl[bar - 1] == f"s_waitcnt vmcnt({nd})"
                    tg = [(("a" if int(m.group(1)) == sda else "b"), int(m.group(2)))
                          for m in (re.match(r"s_add_u32 m0, s(\d+), (\d+)", l) for l in bl) if m]
                    pos = [k for k, l in enumerate(bl) if l.startswith("s_add_u32 m0")]
                    assert len(tg) == nd + ns
                    stage = lambda mat, x: x // (32768 if mat == "a" else bstg)
                    sh = "b" if deep == "a" else "a"
                    front = [(m, x) for (m, x), k in zip(tg, pos) if k < bar]
                    back = [(m, x) for (m, x), k in zip(tg, pos) if k > bar]
                    # in front of the barrier: shallow pieces first, then ALL deep pieces
                    kinds = [m for m, _ in front]
                    assert kinds == sorted(kinds, key=lambda m: m == deep) and kinds.count(deep) == nd
                    assert all(stage(m, x) == ((u + 2) % 3 if m == deep else (u + 1) % 2) for m, x in front)
                    assert all(m == sh and stage(m, x) == u % 2 for m, x in back)
                    assert len(back) + kinds.count(sh) == ns
                    regs = [int(re.match(r"ds_read_b128 v\[\d+:\d+\], v(\d+)", l).group(1)) for l in bl if l.startswith("ds_read")]
                    nbr = sum(l.startswith("ds_read") for l in bl[:bar])
                    assert len(regs) == 4 * (4 + nj) and nbr == 3 * (4 + nj)
                    want = lambda uu: {"a": 32768 * (uu % SA), "b": bstg * (uu % SB)}
                    assert all(areg[r][2] == want(u)[areg[r][0]] for r in regs[:nbr])
                    assert all(areg[r][2] == want(u + 1)[areg[r][0]] and areg[r][1] == 0 for r in regs[nbr:])
            # layout: stage 0 of each operand as the DMA writes it (B tile behind the A stages)
            lds0 = {}
            for addr, (mat, row, chunk) in _bt_layout(nj).items():
                lds0[addr if mat == "a" else addr - 32768 + offb] = (mat, row, chunk)
            reads = set()
            for l in lines:
                m = re.match(r"ds_read_b128 v\[(\d+):\d+\], v(\d+) offset:(\d+)", l)
                if m:
                    mat, kx, st = areg[int(m.group(2))]
                    reads.add((mat, kx >> 5, st, int(m.group(3))))
            for wave in range(4):
                wm, wn = wave >> 1, wave & 1
                for (mat, kk, st, off) in reads:
                    for lane in range(64):
                        hi, l31 = lane >> 5, lane & 31
                        abk0 = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4)
                        base = wm * 16384 if mat == "a" else offb + wn * (nj * 4096)
                        addr = ((base + abk0) ^ (kk << 5)) + off          # (+ st: the same bytes of another stage)
                        want_row = (wm * 128 if mat == "a" else wn * 32 * nj) + (off // 4096) * 32 + l31
                        assert lds0[addr] == (mat, want_row, kk * 2 + hi)


def test_gemm_bt_transposed_tile_twin_only_exchanges_the_mfma_operands():
    """GEMM_BT_ASM_TEXT_NJ3_DB_T (the ViT's q|k|v product writes V^T from its V tiles): the deep 256 x 192 loop line for line,
    except that every MFMA takes (A fragment, B fragment) instead of (B fragment, A fragment) -- the accumulator tile comes out
    transposed, nothing else moves.  And the A-deep twins of round 4 are gone from the build."""
    import re
    text = (CSRC / "gemm_bt_asm.inc").read_text()
    grab = lambda name: re.findall(r'"(.*)\\n"', re.search(rf"#define {name} \\\n(.*?)\n#define", text, re.S).group(1))
    plain, twin = grab("GEMM_BT_ASM_TEXT_NJ3_DB"), grab("GEMM_BT_ASM_TEXT_NJ3_DB_T")
    assert len(plain) == len(twin)
    n = 0
    for a, b in zip(plain, twin):
        if a.startswith("v_mfma"):
            op, rest = a.split(" ", 1)
            acc, x, y, c = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest)]
            assert b == f"{op} {acc}, {y}, {x}, {c}", (a, b)
            n += 1
        else:
            assert a == b
    assert n == 6 * 4 * 48                      # 6 unrolled bodies x 4 waves x 48 MFMAs per K tile
    assert "_DA" not in text


def test_tokattn_pv_asm_is_generated_and_counts_its_waits(tmp_path):
    """tools/gen_tokattn_asm.py: the P V phase of tok_attn2_kernel (16 / 8 accumulator tiles, 6 transposed fragments in flight).
    The committed text is what the generator writes; every MFMA waits for exactly its own fragment (two ds_read_b64_tr_b16 per
    fragment, LDS returns in order): lgkmcnt = 2 x (fragments requested after it), and the last one drains."""
    import re
    import subprocess
    import sys
    dst = tmp_path / "pv.inc"
    subprocess.run([sys.executable, str(ROOT / "tools" / "gen_tokattn_asm.py"), str(dst)], check=True)
    text = (CSRC / "tokattn_pv_asm.inc").read_text()
    assert dst.read_text() == text
    for nb in (16, 8):
        body = re.search(rf"#define TOKATTN_PV_ASM_TEXT_{nb} \\\n(.*?)\n#define", text, re.S).group(1)
        lines = re.findall(r'"(.*)\\n"', body)
        issued = done = 0
        for l in lines:
            if l.startswith("ds_read_b64_tr_b16"):
                issued += 1
            elif l.startswith("s_waitcnt lgkmcnt"):
                allowed = int(re.search(r"\((\d+)\)", l).group(1))
                assert issued - allowed == 2 * (done + 1), (l, issued, done)   # fragment `done` (both halves) has landed
            elif l.startswith("v_mfma"):
                done += 1
        assert done == nb and issued == 2 * nb and lines[-1].startswith("s_nop")


def test_flash_dp2_asm_is_generated():
    want = "".join(_run("tools/gen_flash_dp2_asm.py", *f) for f in ((), ("--timed",), ("--exact",), ("--exact", "--timed")))
    assert want == (CSRC / "flash_dp2_asm.inc").read_text()


def test_flash_dp_fragment_reads_hit_the_rows_the_dma_wrote():
    """Same executable spec for the flash KV loop: a 64-key ring slot = K tile [64 keys][64 d] at +0 and V^T tile
    [64 d][64 keys] at +8192, both 128-byte rows with the kt_off swizzle; DMA pieces per attn.hip (flash_dp2_kernel), reads
    from the generated asm (lane base register AB[k] = lds + (ab0 ^ 32 k), ds_read_b128 .., AB[k] offset:slot * 16384 + ..)."""
    import re
    text = (CSRC / "flash_dp2_asm.inc").read_text()
    body = re.search(r"#define FLASH_DP2_ASM_TEXT \\\n(.*?)\n#define", text, re.S).group(1)
    lines = re.findall(r'"(.*)\\n"', body)
    abreg = {}
    for l in lines:
        m = re.match(r"v_add_u32 v(\d+), %\[lds\], %\[ab0\]", l)
        if m:
            abreg["v" + m.group(1)] = 0
            base = m.group(1)
    for l in lines:
        m = re.match(rf"v_xor_b32 v(\d+), (\d+), v{base}$", l)
        if m:
            abreg["v" + m.group(1)] = int(m.group(2)) >> 5
    assert sorted(abreg.values()) == [0, 1, 2, 3]
    reads = set()
    for l in lines:
        n = re.match(r"ds_read_b128 v\[\d+:\d+\], (v\d+) offset:(\d+)", l)
        if n and n.group(1) in abreg:
            reads.add((abreg[n.group(1)], int(n.group(2))))
    # every ring slot: K rows of both 32-row halves through all four k slices; V^T rows of both 32-row blocks through all
    # four (key half, k slice pair) lane bases
    assert {(k, off) for (k, off) in reads if off % 16384 < 8192} == {(k, sl * 16384 + 4096 * hf) for k in range(4) for sl in range(4) for hf in range(2)}
    assert {(k, off) for (k, off) in reads if off % 16384 >= 8192} == {(k, sl * 16384 + 8192 + 4096 * nb) for k in range(4) for sl in range(4) for nb in range(2)}
    # what the DMA pieces of the four waves put where (tile-relative byte address -> (row, 16-byte chunk))
    tile = {}
    for wv in range(4):
        for lane in range(64):
            for piece in range(2):
                prow = wv * 16 + (lane >> 3) + 8 * piece
                chunk = (lane & 7) ^ ((prow >> 1) & 7)
                addr = wv * 2048 + piece * 1024 + lane * 16
                assert addr not in tile
                tile[addr] = (prow, chunk)
    assert len(tile) == 512
    for half in range(2):                                # K rows 32 half .. : S_AK = slot + 4096 half
        for k in range(4):
            slots = []
            for lane in range(64):
                hi, l31 = lane >> 5, lane & 31
                ab = l31 * 128 + (((k * 2 + hi) ^ ((l31 >> 1) & 7)) << 4)
                assert tile[half * 4096 + ab] == (half * 32 + l31, k * 2 + hi)
                slots.append(((half * 4096 + ab) // 16) % 16)
            for q in range(4):
                assert len(set(slots[16 * q:16 * q + 16])) == 16
    for vh in range(2):                                  # V^T: d rows nb * 32 + l31, key chunk (vh * 2 + ks2) * 2 + hi
        for ks2 in range(2):
            for nb in range(2):
                for lane in range(64):
                    hi, l31 = lane >> 5, lane & 31
                    k = vh * 2 + ks2
                    ab = l31 * 128 + (((k * 2 + hi) ^ ((l31 >> 1) & 7)) << 4)
                    assert tile[nb * 4096 + ab] == (nb * 32 + l31, k * 2 + hi)


def test_generated_asm_passes_the_hazard_lint():
    """tools/asm_lint.py: the software wait states hipcc would insert for its own code (transcendental -> VALU, M0 write
    -> LDS-DMA, MFMA result -> VALU / store, permlane swap) are present in the hand-scheduled loops, and the counted
    lgkmcnt waits are consistent with the reads issued."""
    sys.path.insert(0, str(ROOT / "tools"))
    import asm_lint
    seen = 0
    for inc in ("flash_dp2_asm.inc", "gemm_bt_asm.inc"):
        for name, lines in asm_lint.blocks(str(CSRC / inc)):
            seen += 1
            assert len(lines) > 190
            assert asm_lint.lint(name, lines) == []
    # flash KV loop (exact / pre-scaled x plain / timed), GEMM K loop NJ = 4, NJ = 3, NJ = 3 SwiGLU-pair, NJ = 2 ring, two deep forms + the
    # transposed-tile twin; round 6: the deep 256 x 192 loop on literal accumulators (+ twin), three drain forms, accumulator zero / read-out
    assert seen == 18
    # the linter itself: each rule fires on a minimal violation
    bad = {
        "R1": ["v_exp_f32 v1, v1", "v_add_f32 v2, v1, v1"],
        "R2": ["s_mov_b32 m0, s4", "buffer_load_dwordx4 v1, s[4:7], s8 offen lds"],
        "R3": ["v_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], v[4:7], 0", "v_max3_f32 v50, v16, v17, v18"],
        "R4": ["v_mov_b32 v2, v3", "v_permlane32_swap_b32 v2, v3"],
        "R5": ["ds_read_b128 v[0:3], v9 offset:0", "s_waitcnt lgkmcnt(2)"],
        "R6": ["v_pk_mul_f32 v[2:3], v[2:3], v[2:3]", "v_rcp_f32 v2, v2"],
    }
    assert len(asm_lint.lint("R3a", ["v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]", "v_accvgpr_read_b32 v9, a3"])) == 1
    for rule, text in bad.items():
        errs = asm_lint.lint(rule, text)
        assert len(errs) == 1 and rule in errs[0]


# ---- round 6: the drain forms of the deep 256 x 192 loop (tools/gen_gemm_bt_asm.py: gen_deep_drain) ---------------------------------
def _grab(name):
    import re
    text = (CSRC / "gemm_bt_asm.inc").read_text()
    return re.findall(r'"(.*)\\n"', re.search(rf"#define {name} \\\n(.*?)\n#define", text, re.S).group(1))


def test_gemm_bt_literal_accumulator_twins():
    """GEMM_BT_ASM_TEXT_NJ3_DB_FX / _FX_T (the first tile of a drain-kernel workgroup): the deep 256 x 192 loops line for line, with
    accumulator (i, j) = the literal tuple a[16 (3 i + j) : + 15] instead of an operand."""
    import re
    for sfx in ("", "_T"):
        ref, fx = _grab("GEMM_BT_ASM_TEXT_NJ3_DB" + sfx), _grab("GEMM_BT_ASM_TEXT_NJ3_DB_FX" + sfx)
        assert len(ref) == len(fx)
        for a, b in zip(ref, fx):
            want = re.sub(r"%\[c(\d)(\d)(\d)\]", lambda m: "a[%d:%d]" % (16 * (3 * (2 * int(m.group(1)) + int(m.group(2))) + int(m.group(3))),
                                                                       16 * (3 * (2 * int(m.group(1)) + int(m.group(2))) + int(m.group(3))) + 15), a)
            assert b == want, (a, b)


def _drain_parts(name):
    """-> (lines before the per-wave loops, {wave: {body label: lines}})"""
    lines = _grab(name)
    head = lines[:lines.index("s_cmp_eq_u32 %[wave], 1")]
    bodies = {w: {} for w in range(4)}
    import re
    cur = None
    for l in lines:
        m = re.match(r"\.Lbd_([db])(\d)_(\d+)_%=:", l)
        if m:
            cur = (int(m.group(2)), m.group(1) + m.group(3))
            bodies[cur[0]][cur[1]] = []
        elif l.startswith(".Lbd_w") or l.startswith(".Lbd_done"):
            cur = None
        elif cur:
            bodies[cur[0]][cur[1]].append(l)
    return head, bodies


def test_gemm_bt_drain_schedule_stores_and_counted_waits():
    """Per wave: twelve drain bodies (two turns of the six stage pairs) + the six plain bodies; every body is a K iteration of the deep
    loop (48 MFMAs, 8 + 6 pieces, 28 fragment reads, one barrier).  Each of the 24 hold groups is stored exactly once, 16 bytes per lane from its
    own four registers, even groups at offset 0 and odd ones at 32 of the running scalar offset, which advances once per pair; the stores sit
    between the iteration's last shallow piece and the counted wait, which names exactly the deep pieces + the stores in front of it (so
    that "everything older has landed" still holds on the one in-order counter).  The GELU form stores pair b from body b + 1 (its
    instructions fill body b) and the last pair behind D11."""
    import re
    for name, gelu in (("GEMM_BT_ASM_TEXT_NJ3_DB_DRAIN", False), ("GEMM_BT_ASM_TEXT_NJ3_DB_DRAIN_T", False),
                       ("GEMM_BT_ASM_TEXT_NJ3_DB_DRAIN_GELU", True)):
        head, bodies = _drain_parts(name)
        sda = int(re.search(r"s_lshl_b32 s(\d+), %\[wave\], 13", "\n".join(head)).group(1))
        for w in range(4):
            assert list(bodies[w]) == [f"d{b}" for b in range(12)] + [f"b{u}" for u in range(6)]
            stored = []
            for lab, bl in bodies[w].items():
                assert sum(l.startswith("v_mfma") for l in bl) == 48 and sum(l.startswith("ds_read_b128") for l in bl) == 28
                assert sum(l.startswith("buffer_load_dwordx4") for l in bl) == 14 and bl.count("s_barrier") == 1
                bar = bl.index("s_barrier")
                st = [k for k, l in enumerate(bl) if l.startswith("buffer_store_dwordx4")]
                in_front = [k for k in st if k < bar]
                assert bl[bar - 1] == f"s_waitcnt vmcnt({6 + len(in_front)})"
                # the stores in front of the wait come behind the last SHALLOW (A) piece in front of it
                a_pieces = [k for k, l in enumerate(bl[:bar]) if re.match(rf"s_add_u32 m0, s{sda}, ", l)]
                assert not in_front or min(in_front) > max(a_pieces)
                for k in st:
                    m = re.match(r"buffer_store_dwordx4 v\[(\d+):(\d+)\], %\[voff\], s\[72:75\], s57 offen offset:(\d+)", bl[k])
                    g, rem = divmod(int(m.group(1)) - 132, 4)
                    assert rem == 0 and int(m.group(2)) == int(m.group(1)) + 3 and int(m.group(3)) == 32 * (g & 1)
                    stored.append((lab, g, k > bar))
                if lab.startswith("b"):
                    assert not st
                # the running store offset advances once per stored pair: + the column-block stride twice, then the row-block adjustment
                adv = [l for l in bl if l.startswith("s_add_u32 s57, s57,")]
                b = int(lab[1:]) if lab.startswith("d") else None
                pair = None if b is None else (b - 1 if gelu else b)
                if pair is not None and pair >= 0:
                    assert adv == [f"s_add_u32 s57, s57, {'s78' if pair % 3 < 2 else 's58'}"], (lab, adv)
                else:
                    assert adv == []
            assert sorted(g for _, g, _ in stored) == list(range(24))
            # a tile's accumulators are OPENED by the first k16 step of D0 (C = 0: CONVERT zeroes nothing), accumulated into everywhere else
            for lab, bl in bodies[w].items():
                mf = [l for l in bl if l.startswith("v_mfma")]
                opened = [l.endswith(", 0") for l in mf]
                assert opened == ([True] * 12 + [False] * 36 if lab == "d0" else [False] * 48), lab
            for lab, g, behind in stored:
                b = int(lab[1:])
                assert g // 2 == (b - 1 if gelu and not (b == 11 and behind) else b) or (gelu and b == 11 and behind and g // 2 == 11), (lab, g)
        if gelu:       # every hold register of pair b is rewritten (packed GELU results) inside body b, before its stores one body later
            for w in range(4):
                for b in range(12):
                    bl = bodies[w][f"d{b}"]
                    cv = [int(re.match(r"v_cvt_pk_bf16_f32 v(\d+),", l).group(1)) for l in bl if l.startswith("v_cvt_pk_bf16_f32")]
                    assert sorted(cv) == list(range(132 + 8 * b, 132 + 8 * b + 8))
                    assert sum(l.startswith("v_rcp_f32") for l in bl) == 16


def test_gemm_bt_drain_convert_puts_every_element_where_its_store_writes_it():
    """Executable spec of CONVERT + the drain stores: the accumulator layout of v_mfma_f32_32x32x16 (lane = row / column, register r =
    (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the other index) is pushed symbolically through every flavour's instruction stream --
    v_accvgpr_read, packed math (element-wise), v_cvt_pk (pairs), v_permlane32_swap (upper half of the first operand <-> lower half of the
    second), the bias reads -- and through the store addressing (lane offset + running scalar offset + immediate).  Row-major flavours must
    deliver C[m][n] for all 128 x 96 elements of the wave tile with bias[n] added to column n; the transposed flavour the V^T layout of
    vt_epilogue (gemm_bt.hip): element (m, n) at n * vt_ld + position, position = 16-key group order [0-3, 8-11, 4-7, 12-15]."""
    import re
    head, bodies = _drain_parts("GEMM_BT_ASM_TEXT_NJ3_DB_DRAIN")
    i0 = next(k for k, l in enumerate(head) if l.startswith("s_cmp_eq_u32 s80,"))
    conv = head[i0:]
    # flavour code -> its straight-line instruction list
    labels = {int(m.group(1)): k for k, l in enumerate(conv) for m in [re.match(r"\.Lcv_(\d+)_%=:", l)] if m}
    first = next(k for k, l in enumerate(conv) if l.startswith("v_accvgpr_read_b32"))
    starts = dict(labels)
    starts[0] = first - 1
    ldc, vt_ld = 1000, 5000
    for code in (0, 2, 4, 6, 1, 3):
        vt, bias = bool(code & 1), bool(code & 4)
        k = starts[code] + 1
        V = {}                                       # (vgpr, lane) -> symbolic value
        B = {}                                       # pending bias reads: handled immediately (LDS returns in order, waits are the lint's job)
        read = set()
        while not (conv[k].startswith("s_branch") or conv[k].startswith(".Lcv_")):
            l = conv[k]
            k += 1
            m = re.match(r"v_accvgpr_read_b32 v(\d+), a(\d+)", l)
            if m:
                a = int(m.group(2))
                read.add(a)
                i, j, r = a // 48, (a // 16) % 3, a % 16
                for lane in range(64):
                    other = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                    V[int(m.group(1)), lane] = {"m": 32 * i + (other if vt else (lane & 31)), "n": 32 * j + ((lane & 31) if vt else other), "bias": None}
                continue
            m = re.match(r"ds_read_b128 v\[(\d+):\d+\], %\[vbias\] offset:(\d+)", l)
            if m:
                for q in range(4):
                    for lane in range(64):
                        V[int(m.group(1)) + q, lane] = ("biasval", int(m.group(2)) // 4 + 4 * (lane >> 5) + q)   # vbias carries 16 hi bytes
                continue
            m = re.match(r"v_pk_(mul|add|fma)_f32 v\[(\d+):\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\](?:, v\[(\d+):\d+\])?", l)
            if m:
                d, a_, b_ = int(m.group(2)), int(m.group(3)), int(m.group(4))
                breg = {"add": b_, "fma": int(m.group(5)) if m.group(5) else None, "mul": None}[m.group(1)]
                for q in range(2):
                    for lane in range(64):
                        x = dict(V[a_ + q, lane])
                        if breg is not None:
                            tag, n = V[breg + q, lane]
                            assert tag == "biasval" and x["bias"] is None
                            x["bias"] = n
                        V[d + q, lane] = x
                continue
            m = re.match(r"v_cvt_pk_bf16_f32 v(\d+), v(\d+), v(\d+)", l)
            if m:
                for lane in range(64):
                    V[int(m.group(1)), lane] = (V[int(m.group(2)), lane], V[int(m.group(3)), lane])
                continue
            m = re.match(r"v_permlane32_swap_b32 v(\d+), v(\d+)", l)
            if m:
                a_, b_ = int(m.group(1)), int(m.group(2))
                for lane in range(32):
                    V[a_, lane + 32], V[b_, lane] = V[b_, lane], V[a_, lane + 32]
                continue
            assert l.startswith(("s_waitcnt lgkmcnt", "s_nop")), l
        assert read == set(range(192))
        # the stores of wave 0's drain bodies: pair b at scalar offset s_base + rb * S_rb + ni * S_ni (S: strides of the flavour), + 32 t
        S_rb, S_ni = (64, 2 * 32 * vt_ld) if vt else (2 * 32 * ldc, 64)
        seen = {}
        for b in range(12):
            for l in bodies[0][f"d{b}"]:
                m = re.match(r"buffer_store_dwordx4 v\[(\d+):\d+\],.* offset:(\d+)", l)
                if not m:
                    continue
                soff = (b // 3) * S_rb + (b % 3) * S_ni + int(m.group(2))
                for lane in range(64):
                    l31, hi = lane & 31, lane >> 5
                    voff = 2 * ((l31 * vt_ld + 8 * hi) if vt else (l31 * ldc + 8 * hi))
                    for q in range(4):
                        for half in range(2):
                            e_ = V[int(m.group(1)) + q, lane][half]
                            addr = (voff + soff) // 2 + 2 * q + half           # element index relative to the wave tile's origin
                            assert addr not in seen
                            seen[addr] = e_
        assert len(seen) == 128 * 96
        for addr, e_ in seen.items():
            if vt:
                n, pos = divmod(addr, vt_ld)
                grp, p = divmod(pos, 16)
                key = 16 * grp + [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15][p]
                assert (e_["m"], e_["n"], e_["bias"]) == (key, n, None), (code, addr, e_)
            else:
                m_, n = divmod(addr, ldc)
                assert (e_["m"], e_["n"]) == (m_, n) and e_["bias"] == (n if bias else None), (code, addr, e_)


def test_gemm_bt_drain_kernel_keeps_the_compiler_out_of_the_accumulator_file(tmp_path):
    """gemm_bt_drain_kernel names a0 .. a191 literally and lists them as clobbers of its statements -- no C++ object holds the
    accumulators, so nothing but an audit of the compiled kernel can show that hipcc leaves them alone between the statements (a spill into the
    accumulator file, a v_accvgpr_* of its own): between the zeroing statement and the read-out, every instruction that touches an AGPR, and
    every scratch access, must sit inside an asm statement.  (cdna_hip_programming.md section 5.7, item 4.)"""
    import re
    out = tmp_path / "gemm_bt.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", str(CSRC / "gemm_bt.hip"),
                    "-o", str(out)], check=True, capture_output=True, cwd=CSRC, timeout=900)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN2u220gemm_bt_drain_kernel\w+):.*?s_endpgm", text, re.S | re.M)
    assert len(kernels) == 6
    for k in kernels:
        body = text[text.index("\n" + k + ":"):]
        body = body[:body.index("s_endpgm")].splitlines()
        marks = [i for i, l in enumerate(body) if "#ASMSTART" in l or "#ASMEND" in l]
        assert len(marks) % 2 == 0 and len(marks) >= 12
        inside = [False] * len(body)
        for a, b in zip(marks[::2], marks[1::2]):
            for i in range(a, b + 1):
                inside[i] = True
        # the read-out statement is the last one that names an accumulator register
        last_acc_stmt = max(b for a, b in zip(marks[::2], marks[1::2]) if any("v_accvgpr_read_b32" in l for l in body[a:b]))
        for i in range(marks[0], last_acc_stmt):
            if not inside[i]:
                code = body[i].split(";")[0]
                touched = [int(x) for x in re.findall(r"\ba(\d+)\b", code)] + [int(x) for pr in re.findall(r"\ba\[(\d+):(\d+)\]", code) for x in pr]
                # (a192 and up are the compiler's: it parks VGPRs there around the statements, which clobber v56 .. v255)
                assert all(a >= 192 for a in touched) and "scratch_" not in code, (k, i, body[i])
