#!/usr/bin/env python3
"""tools/kernel_regs.py -- register / spill summary of every kernel in a gfx950 assembly listing.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o k.s file.hip && python tools/kernel_regs.py k.s
"""
import re
import subprocess
import sys


def main():
    text = open(sys.argv[1]).read()
    want = ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size")
    for block in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            pass
        name = re.sub(r"\(anonymous namespace\)::|mhx::|\(mhx::.*", "", name)
        vals = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1)) for k in want if re.search(rf"\.{k}:\s+(\d+)", block)}
        v = vals.get("vgpr_count", 0)
        waves = 512 // (((v + 7) // 8) * 8) if v else 8
        print(f"vgpr={v:4d} (waves/SIMD {min(waves, 8)}) sgpr_spill={vals.get('sgpr_spill_count', 0):4d} vgpr_spill={vals.get('vgpr_spill_count', 0):3d} scratch={vals.get('private_segment_fixed_size', 0):4d}  {name[:100]}")


if __name__ == "__main__":
    main()
