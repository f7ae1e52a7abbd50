"""Which GEMM shapes does one step of the bench workload launch?  (U2TOK_GEMM_TRACE=1: gemm.hip prints one line per
product)   U2TOK_GEMM_TRACE=1 python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-roofline 2> trace; python tools/gemm_shapes.py trace"""
import collections
import re
import sys

c = collections.Counter()
for line in open(sys.argv[1]):
    m = re.match(r"gemm M=(\d+) N=(\d+) K=(\d+) nz=(\d+) flags=(0x[0-9a-f]+)", line)
    if m:
        c[tuple(m.groups())] += 1
tot = 0
for (M, N, K, nz, fl), n in sorted(c.items(), key=lambda kv: -int(kv[0][0]) * int(kv[0][1]) * int(kv[0][2]) * int(kv[0][3]) * kv[1]):
    gf = 2 * int(M) * int(N) * int(K) * int(nz) * n / 1e9
    tot += gf
    print(f"{n:4d} x  M={M:>6} N={N:>6} K={K:>5} nz={nz:>3} flags={fl}  {gf:9.1f} GF")
print(f"total {tot:.1f} GF in {sum(c.values())} launches")
