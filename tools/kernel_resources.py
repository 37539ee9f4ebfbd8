#!/usr/bin/env python3
"""Register / LDS / scratch footprint of every kernel of the built extension, read from the code objects (no GPU needed)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
llvm = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as tmp:
    for tu in sorted(f[:-6] for f in os.listdir(os.path.join(ROOT, "jxl_coder_amd", "build")) if f.endswith(".hip.o")):
        obj = os.path.join(ROOT, "jxl_coder_amd", "build", tu + ".hip.o")
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.o")
        if subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "copy.o")], capture_output=True).returncode:      # (an output file: without one objcopy rewrites `obj` in place and its new mtime hides a stale object from jxl_coder_amd.build)
            continue
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            if pat in name:
                print(f"{tu:16s} {name[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")
