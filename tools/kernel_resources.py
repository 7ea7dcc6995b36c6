"""Register / scratch / LDS use of every gfx950 kernel in the built library, read from the code objects' metadata notes
(llvm-readelf --notes on the ELF embedded in each nlopt_amd/lib/obj/*.o): what `.private_segment_fixed_size: 0` claims are checked with.
usage: python tools/kernel_resources.py [substring-of-kernel-name]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(objdir=os.path.join(ROOT, "nlopt_amd", "lib", "obj")):
    out = []
    with tempfile.TemporaryDirectory() as td:
        for f in sorted(os.listdir(objdir)):
            if not f.endswith(".o"):
                continue
            fat, elf = os.path.join(td, "fat.bin"), os.path.join(td, "k.elf")
            r = subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, os.path.join(objdir, f), os.path.join(td, "copy.o")],
                               capture_output=True)
            if r.returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue
            r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--output=" + elf], capture_output=True)
            if r.returncode:
                continue
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
                out.append(dict(file=f, name=g("name"), vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                                lds=g("group_segment_fixed_size"), vgpr_spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count")))
            os.remove(fat)
    return out


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    print("%-34s %5s %5s %5s %8s %7s %7s  %s" % ("object", "vgpr", "agpr", "sgpr", "scratch", "lds", "vspill", "kernel"))
    for k in kernels():
        if pat in k["name"]:
            print("%-34s %5s %5s %5s %8s %7s %7s  %s" % (k["file"], k["vgpr"], k["agpr"], k["sgpr"], k["scratch"], k["lds"], k["vgpr_spill"], k["name"][:90]))
