#!/usr/bin/env python3
"""VGPR / SGPR / scratch of the kernels in one object of libislands_amd.so:
    python tools/kernel_resources.py islands_amd/lib/obj/search.o [name-filter]"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
obj, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as td:
    subprocess.check_call(["objcopy", "--dump-section", f".hip_fatbin={td}/fat.bin", obj])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={td}/fat.bin",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={td}/k.co"])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", f"{td}/k.co"], capture_output=True, text=True).stdout
for b in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if flt not in name:
        continue
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", b).group(1))
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", "").replace("void ", "").replace("(SearchParams)", "")
    print(f"{dn:60s} vgpr {g('vgpr_count'):4d} sgpr {g('sgpr_count'):4d} scratch {g('private_segment_fixed_size'):5d} "
          f"lds {g('group_segment_fixed_size')}")
