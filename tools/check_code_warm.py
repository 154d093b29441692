#!/usr/bin/env python3
"""Post-link check of libqn_hip.so: the kernels that read their own code as DATA at entry (qn_kernels.hip.h CODE WARM-UP:
32 KB from the program counter on, unconditionally) must lie at least that far in front of the end of the device code object's .text -- with
XNACK off a read past the loaded image is a memory fault on a default path.  Run by csrc/Makefile after every link; fails the build otherwise.
    usage: check_code_warm.py <libqn_hip.so>"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
WARMED = ("s2_eval_kernel", "s2_evalr_kernel", "s2_vec_kernel")  # every kernel with a qn_code_warm_issue / pc0 read in it
REACH = 256 * 128  # bytes read from the (128-byte aligned) program counter on


def main(lib):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={co}"])
        sec = subprocess.check_output([f"{LLVM}/llvm-readelf", "-S", "-W", co], text=True)
        sym = subprocess.check_output([f"{LLVM}/llvm-readelf", "-s", "-W", co], text=True)
    m = re.search(r"\]\s+\.text\s+PROGBITS\s+([0-9a-f]+)\s+[0-9a-f]+\s+([0-9a-f]+)", sec)
    if not m:
        sys.exit("check_code_warm: no .text in the device code object")
    text_end = int(m.group(1), 16) + int(m.group(2), 16)
    worst, n = None, 0
    for line in sym.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC" and any(f"{len(k)}{k}" in f[7] for k in WARMED):
            start, size = int(f[1], 16), int(f[2])
            room = text_end - (start + size)  # the program counter is somewhere inside the kernel: its END + the reach must stay inside .text
            n += 1
            if worst is None or room < worst[0]:
                worst = (room, f[7])
    if n == 0:
        sys.exit("check_code_warm: none of the warmed kernels found -- has a kernel been renamed?")
    if worst[0] < REACH:
        sys.exit(f"check_code_warm: {worst[1]} ends {worst[0]} bytes in front of the end of .text, the code warm-up reads up to {REACH} past its program "
                 "counter: move a kernel that does not warm its code behind it (definition order), or pad")
    print(f"check_code_warm: {n} warmed kernels, the closest ends {worst[0]} bytes in front of the end of .text (>= {REACH} required)")


if __name__ == "__main__":
    main(sys.argv[1])
