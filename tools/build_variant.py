#!/usr/bin/env python
"""Experiment builds: tools/build_variant.py NAME file.hip[,file2.hip] -DFLAG [-DFLAG2=3 ...]
compiles the named translation units with the extra flags, links them with the stock objects of the
other units into tantivy_amd/lib/variants/libtantivy_amd_NAME.so; TQ_LIB_PATH=<that file> makes
tantivy_amd.binding load it (A/B runs of kernel variants inside one gpurun call)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tantivy_amd import build as B  # noqa: E402


def main():
    name, units, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    B.build()
    vdir = os.path.join(B.HERE, "lib", "variants")
    odir = os.path.join(B.OBJ_DIR, "variants", name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    objs = []
    procs = []
    for src in B.SOURCES:
        base = os.path.basename(src)
        if base in units:
            obj = os.path.join(odir, base + ".o")
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + B.FLAGS + flags + ["-c", src, "-o", obj]))
        else:
            obj = os.path.join(B.OBJ_DIR, base + ".o")
        objs.append(obj)
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("compile failed")
    out = os.path.join(vdir, "libtantivy_amd_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    print(out)


if __name__ == "__main__":
    main()
