#!/usr/bin/env python3
"""Register, LDS and scratch use of every kernel in a built libmpc.so, read from the code objects' metadata (no GPU
needed).  usage: python tools/kernel_registers.py [library] [substring of the kernel name]
The library is copied to a temporary directory first: llvm-objdump --offloading writes the bundles it extracts next to
its input."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(library=None):
    """-> {kernel name (demangled where c++filt is installed): {"vgpr": n, "agpr": n, "sgpr": n, "lds": bytes, "scratch": bytes}}"""
    library = library or os.path.join(REPO, "compression_amd", "libmpc.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(library, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], cwd=tmp, check=True, capture_output=True)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, name)],
                                   capture_output=True, text=True, check=True).stdout
            for block in notes.split("  - .agpr_count:")[1:]:
                fields = dict(re.findall(r"\.(\w+):\s+(\S+)", ".agpr_count:" + block.split("\n  - .agpr_count:")[0]))
                if "name" not in fields or "vgpr_count" not in fields:
                    continue
                out[fields["name"]] = dict(vgpr=int(fields["vgpr_count"]), agpr=int(fields["agpr_count"]), sgpr=int(fields["sgpr_count"]),
                                           lds=int(fields.get("group_segment_fixed_size", 0)),
                                           scratch=int(fields.get("private_segment_fixed_size", 0)))
    if out:
        names = list(out)
        filt = shutil.which("c++filt")
        if filt:
            demangled = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
            out = {(d or n): out[n] for n, d in zip(names, demangled)}
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else None
    want = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    for name, k in sorted(kernels(lib).items(), key=lambda kv: -kv[1]["vgpr"]):
        if want in name:
            print(f"{k['vgpr']:4d} vgpr {k['agpr']:4d} agpr {k['sgpr']:4d} sgpr {k['lds']:7d} B lds {k['scratch']:5d} B scratch  {name[:110]}")
