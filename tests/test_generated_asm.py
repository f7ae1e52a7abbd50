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
            assert len(lines) > 200
            assert asm_lint.lint(name, lines) == []
    assert seen == 11  # flash KV loop (exact / pre-scaled x plain / timed), GEMM K loop NJ = 4, NJ = 3, NJ = 3 SwiGLU-pair, NJ = 2 ring, two deep forms + the transposed-tile twin
    # the linter itself: each rule fires on a minimal violation
    bad = {
        "R1": ["v_exp_f32 v1, v1", "v_add_f32 v2, v1, v1"],
        "R2": ["s_mov_b32 m0, s4", "buffer_load_dwordx4 v1, s[4:7], s8 offen lds"],
        "R3": ["v_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], v[4:7], 0", "v_max3_f32 v50, v16, v17, v18"],
        "R4": ["v_mov_b32 v2, v3", "v_permlane32_swap_b32 v2, v3"],
        "R5": ["ds_read_b128 v[0:3], v9 offset:0", "s_waitcnt lgkmcnt(2)"],
    }
    for rule, text in bad.items():
        errs = asm_lint.lint(rule, text)
        assert len(errs) == 1 and rule in errs[0]
