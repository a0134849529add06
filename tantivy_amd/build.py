"""Builds tantivy_amd/lib/libtantivy_amd.so (HIP kernels + C ABI + C++ host mirror) for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off: BM25 is add / IEEE divide / multiply with
no fused multiply-add, like the reference."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "lib", "libtantivy_amd.so")
SOURCES = [
    os.path.join(HERE, "csrc", "tq_and.hip"),
    os.path.join(HERE, "csrc", "tq_union.hip"),
    os.path.join(HERE, "csrc", "tq_ushare.hip"),
    os.path.join(HERE, "csrc", "tq_ashare.hip"),
    os.path.join(HERE, "csrc", "tq_count.hip"),
    os.path.join(HERE, "csrc", "tq_tree.hip"),
    os.path.join(HERE, "csrc", "tq_count.cpp"),
    os.path.join(HERE, "csrc", "tq_xunion.hip"),
    os.path.join(HERE, "csrc", "tq_phrase.hip"),
    os.path.join(HERE, "csrc", "tq_misc.hip"),
    os.path.join(HERE, "csrc", "tq_encode.hip"),
    os.path.join(HERE, "csrc", "tq_prepare.hip"),
    os.path.join(HERE, "csrc", "tq_api.cpp"),
    os.path.join(HERE, "csrc", "tq_terms.cpp"),
    os.path.join(HERE, "csrc", "tq_plan_chunks.cpp"),
    os.path.join(HERE, "csrc", "tq_plan_share.cpp"),
    os.path.join(HERE, "csrc", "tq_plan_misc.cpp"),
    os.path.join(HERE, "csrc", "tq_search.cpp"),
    os.path.join(HERE, "csrc", "tq_submit.cpp"),
    os.path.join(HERE, "csrc", "tq_comm.cpp"),
    os.path.join(HERE, "host", "searcher.cpp"),
    os.path.join(HERE, "host", "host_capi.cpp"),
    os.path.join(HERE, "host", "term_info_store.cpp"),
]
HEADERS = [
    os.path.join(HERE, "csrc", "tq_common.hpp"),
    os.path.join(HERE, "csrc", "tq_device.h"),
    os.path.join(HERE, "csrc", "tq_launch.h"),
    os.path.join(HERE, "csrc", "tq_prepare.h"),
    os.path.join(HERE, "csrc", "tq_internal.hpp"),
    os.path.join(HERE, "host", "searcher.hpp"),
    os.path.join(HERE, "host", "bm25.hpp"),
    os.path.join(HERE, "host", "term_info_store.hpp"),
    os.path.join(os.path.dirname(HERE), "include", "tantivy_amd.h"),
    os.path.join(os.path.dirname(HERE), "include", "tantivy_amd_host.h"),
]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall"]
OBJ_DIR = os.path.join(HERE, "lib", "obj")


def csrc_hash():
    """sha256 over the kernel / C-ABI sources: ties a profile (profiles/traffic.json entries carry
    the hash of the tree they were measured on) to the code bench.py is running."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(HERE, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


# which translation unit a scan kernel of a rocprofv3 trace comes from
KERNEL_FILES = {"and_kernel": "tq_and.hip", "union_kernel_small": "tq_union.hip", "union_kernel": "tq_union.hip",
                "or_kernel": "tq_union.hip", "ushare_kernel": "tq_ushare.hip", "ashare_kernel": "tq_ashare.hip", "count_bitmap_kernel": "tq_count.hip", "tree_kernel": "tq_tree.hip", "xunion_kernel": "tq_xunion.hip",
                "phrase_sweep_kernel": "tq_phrase.hip", "phrase_kernel": "tq_phrase.hip"}


def kernel_hash(kernel_names, read=None):
    """sha256 over the sources of the named scan kernels (their .hip files + the device headers all of
    them share): what a PMC entry of profiles/traffic.json is valid for.  Finer than csrc_hash(): the
    host planner or another kernel family changing does not invalidate the entry.  `read(name)`
    returns a csrc file's bytes (default: the working tree; tools/ pass `git show` of a commit)."""
    import hashlib
    import re

    files = {"tq_common.hpp", "tq_device.h"}
    for k in kernel_names:
        m = re.search(r"(\w+)<", k)
        base = m.group(1) if m else k
        if base in KERNEL_FILES:
            files.add(KERNEL_FILES[base])
    if read is None:
        def read(name):
            with open(os.path.join(HERE, "csrc", name), "rb") as f:
                return f.read()
    h = hashlib.sha256()
    for name in sorted(files):
        h.update(name.encode())
        h.update(read(name))
    return h.hexdigest()[:16]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    m = os.path.getmtime(LIB)
    return all(os.path.getmtime(p) <= m for p in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """One hipcc -c per translation unit, in parallel (the kernel families dominate: ~45 s each),
    then one link."""
    if not force and up_to_date():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = [os.path.join(OBJ_DIR, os.path.basename(src) + ".o") for src in SOURCES]

    newest_header = max(os.path.getmtime(h) for h in HEADERS)

    def compile_one(pair):
        src, obj = pair
        # incremental: an object newer than its source and every header is kept
        if not force and os.path.exists(obj) and \
                os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header):
            return
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, zip(SOURCES, objs)))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
