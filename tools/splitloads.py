#!/usr/bin/env python
"""Find vector loads that hipcc split into scalars: compile every csrc/*.hip for gfx950 with -save-temps and report kernels in
which `global_load_dword` instructions follow one another off the same address register at offsets +4 / +8 / +12 (what a
predicated `cond ? *(const float4*)p : zero` turned into in the round-4 FIR-noise kernel: 64 loads per lane instead of 16).
CPU only:  python tools/splitloads.py [file.hip ...]"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neural-waveshaping-synthesis_amd", "csrc")


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with tempfile.TemporaryDirectory() as tmp:
        for src in files:
            base = os.path.splitext(os.path.basename(src))[0]
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                            "-c", src, "-o", os.path.join(tmp, base + ".o"), "-save-temps=obj"], check=True, stderr=subprocess.DEVNULL)
            asm = glob.glob(os.path.join(tmp, base + "-hip-amdgcn-amd-amdhsa-gfx950.s"))[0]
            cur, stats = None, collections.Counter()
            for line in open(asm):
                m = re.match(r"^(_Z\S+):\s", line)
                if m:
                    cur = m.group(1)
                m = re.match(r"\s+global_load_dword (v\d+), (v\[\d+:\d+\]), off(?: offset:(\d+))?", line)
                if m and cur and int(m.group(3) or 0) in (4, 8, 12):
                    stats[cur] += 1
            for k, v in stats.items():
                if v >= 3:
                    print(f"{base}: {k[:100]}: {v} dword loads at +4 / +8 / +12 of a shared base")


if __name__ == "__main__":
    main()
