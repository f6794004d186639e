"""Builds libplonkit_amd.so (HIP kernels + C ABI + host side) for gfx950 with hipcc, in-tree.

`python -m plonkit_amd.build` or plonkit_amd.build.build().  hipcc cross-compiles without a GPU.
The shared object lands in plonkit_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
SO = os.path.join(LIBDIR, "libplonkit_amd.so")
CLI = os.path.join(LIBDIR, "plonkit")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result", "-DPLK_BUILD"]
# experiments only (tools/ab_flags.sh): extra compiler flags for every source, or for the sources named in PLK_HIPCC_EXTRA_FILES
_EXTRA, _EXTRA_FILES = os.environ.get("PLK_HIPCC_EXTRA", "").split(), [f for f in os.environ.get("PLK_HIPCC_EXTRA_FILES", "").split(",") if f]
if not _EXTRA_FILES:
    FLAGS += _EXTRA
# per-source flags: the bucket accumulation is scheduled for ILP (-2.2 % on that kernel, same-box A/B in
# profiles/r04_msm_accumulate_sched_ab.txt; the same flag on the whole library makes a proof 0.3 ms slower)
FILE_FLAGS = {"msm_accumulate.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              "g1ntt.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}         # dump-lagrange: 192.0 -> 188.8 ms at 2^20, same file


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")) and not f.startswith("cli_"))


def _headers_digest():
    h = hashlib.sha1()
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith(".h"):
                h.update(open(os.path.join(root, f), "rb").read())
    h.update((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items())) + repr((_EXTRA, _EXTRA_FILES))).encode())
    return h.hexdigest()[:16]


def _compile(src, digest, verbose):
    obj = os.path.join(OBJDIR, src + "." + digest + ".o")
    srcp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(srcp):
        return obj
    cmd = ["hipcc"] + FLAGS + FILE_FLAGS.get(src, []) + (_EXTRA if src in _EXTRA_FILES else []) + ["-x", "hip", "-c", srcp, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build(verbose=False, force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    digest = _headers_digest()
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = _sources()
    for f in os.listdir(OBJDIR):                                   # objects of earlier header generations are dead weight
        if f.endswith(".o") and ("." + digest + ".o") not in f:
            os.remove(os.path.join(OBJDIR, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, digest, verbose), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    cli_src = os.path.join(CSRC, "cli_main.cpp")
    if os.path.exists(cli_src) and (force or not os.path.exists(CLI) or os.path.getmtime(CLI) < max(os.path.getmtime(cli_src), os.path.getmtime(SO))):
        cmd = ["hipcc", "-O2", "-std=c++17", "-pthread", cli_src, "-o", CLI, "-L" + LIBDIR, "-lplonkit_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
