#!/usr/bin/env python3
"""Register / LDS usage of every kernel in compiled objects (reads the code-object metadata; no GPU needed).

    python tools/kernel_regs.py u2tokenizer_amd/csrc/build/gemm.o [...]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

B = "/opt/rocm/lib/llvm/bin/"
KEYS = ("name", "vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
        "vgpr_spill_count")


def kernels(obj):
    with tempfile.TemporaryDirectory() as t:
        fat, dev = Path(t) / "fat.bin", Path(t) / "dev.o"
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, str(fat)], check=True)
        subprocess.run([B + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"], check=True)
        notes = subprocess.run([B + "llvm-readelf", "--notes", str(dev)], capture_output=True, text=True).stdout
    cur, res = {}, []
    for line in notes.splitlines() + ["  - .end"]:
        if re.match(r"\s+- \.", line):
            if "name" in cur and "vgpr_count" in cur:
                res.append(cur)
            cur = {}
        m = re.search(r"\.(%s):\s+(\S+)" % "|".join(KEYS), line)
        if m and not (m.group(1) == "name" and not m.group(2).startswith("_Z")):
            cur[m.group(1)] = m.group(2)
    return res


if __name__ == "__main__":
    for obj in sys.argv[1:]:
        for k in kernels(obj):
            n = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
            n = re.sub(r"\(.*", "", n).replace("void u2::", "")
            g = lambda key: str(k.get(key, "-"))  # noqa: E731
            print(f"{n[:64]:64s} vgpr={g('vgpr_count'):>3s} agpr={g('agpr_count'):>3s} sgpr={g('sgpr_count'):>3s} "
                  f"lds={g('group_segment_fixed_size'):>6s} scratch={g('private_segment_fixed_size')} "
                  f"spill={g('vgpr_spill_count')}")
