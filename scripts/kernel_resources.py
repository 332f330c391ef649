#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in the built objects (video-quierer_amd/lib/obj/*.o), from the code
objects' metadata notes:  python scripts/kernel_resources.py [name-substring ...].  A kernel with scratch > 0 spills."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as td:
        co, fb = os.path.join(td, "dev.co"), os.path.join(td, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fb}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda key: (re.search(rf"\.{key}:\s+(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        yield dict(name=name, vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                   scratch=g("private_segment_fixed_size"), spill_v=g("vgpr_spill_count"), spill_s=g("sgpr_spill_count"))


if __name__ == "__main__":
    want = sys.argv[1:]
    for obj in sorted(glob.glob(os.path.join(ROOT, "video-quierer_amd", "lib", "obj", "*.o"))):
        for k in kernels(obj):
            if want and not any(w in k["name"] for w in want):
                continue
            print(f'{os.path.basename(obj):16s} vgpr {k["vgpr"]:>3} agpr {k["agpr"]:>3} sgpr {k["sgpr"]:>3} lds {k["lds"]:>6} scratch {k["scratch"]:>4} '
                  f'spills v{k["spill_v"]} s{k["spill_s"]}  {k["name"][:150]}')
