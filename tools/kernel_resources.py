#!/usr/bin/env python3
"""Register / scratch / LDS use of every gfx950 kernel in the library's objects (build/obj/*.o), read from the code objects' metadata:
llvm-objcopy dumps each object's .hip_fatbin, clang-offload-bundler unbundles the gfx950 code object, llvm-readelf --notes lists
.vgpr_count / .vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size / .group_segment_fixed_size per kernel.

    python tools/kernel_resources.py            # table of every kernel, spilling ones flagged
    python tools/kernel_resources.py --spills   # only kernels with spilled VGPRs or scratch

tests/test_kernel_resources.py holds the library to "no spilled vector register, no scratch" with this module."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def demangle(names):
    out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else ["c++filt"],
                         input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def kernels_of_object(obj):
    """[{name, vgpr_count, ...}] for every kernel of the gfx950 code object bundled in `obj`."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fatbin"), os.path.join(d, "co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    # the note's body is a YAML document between '---' and '...'
    import yaml
    lines = notes.splitlines()
    start = next(i for i, l in enumerate(lines) if l.strip() == "---")
    end = next((i for i in range(start + 1, len(lines)) if lines[i].strip() == "..."), len(lines))
    meta = yaml.safe_load("\n".join(lines[start + 1:end]))
    return [{"name": k[".name"], **{f: int(k.get("." + f, 0)) for f in FIELDS}} for k in meta["amdhsa.kernels"]]


def all_kernels(objdir=None):
    objdir = objdir or os.path.join(ROOT, "build", "obj")
    out = []
    for f in sorted(os.listdir(objdir)):
        if f.endswith(".o"):
            for k in kernels_of_object(os.path.join(objdir, f)):
                k["object"] = f
                out.append(k)
    for k, d in zip(out, demangle([k["name"] for k in out])):
        k["demangled"] = re.sub(r"\(.*$", "", d).replace("void ", "").replace("hns::", "")
    return out


if __name__ == "__main__":
    ks = all_kernels()
    only = "--spills" in sys.argv
    print(f"{'object':22s} {'VGPR':>4s} {'AGPR':>4s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'LDS':>6s}  kernel")
    bad = 0
    for k in ks:
        flag = k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)
        bad += bool(flag)
        if only and not flag:
            continue
        print(f"{k['object']:22s} {k.get('vgpr_count', 0):4d} {k.get('agpr_count', 0):4d} {k.get('vgpr_spill_count', 0):6d} {k.get('sgpr_spill_count', 0):6d} "
              f"{k.get('private_segment_fixed_size', 0):7d} {k.get('group_segment_fixed_size', 0):6d}  {k['demangled']}")
    print(f"{len(ks)} kernels, {bad} with spilled vector registers or scratch")
