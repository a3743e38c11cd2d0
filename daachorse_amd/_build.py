"""Builds daachorse_amd/lib/libdaachorse_amd.so for gfx950 with hipcc (in-tree, so the .so
travels with the repository snapshot to the GPU box)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdaachorse_amd.so")

SOURCES = ["pma.cpp", "repack.cpp", "gram.cpp", "gram2.cpp", "gram4.cpp", "gram2w.cpp", "pfx.cpp", "builder.cpp", "charwise.cpp", "charwise_builder.cpp", "api_upload.hip", "api_scan.hip", "api_select.hip", "api_iter.hip", "api_options.hip", "scan_kernels.hip",
           "gram_kernels.hip", "gram2_kernels.hip", "gram4_kernels.hip", "pfx_kernels.hip", "emit3_kernels.hip", "find3_kernels.hip", "left3_kernels.hip", "gram2w_kernels.hip", "restart_kernels.hip", "charwise_kernels.hip", "synth.hip"]
HEADERS = ["api_internal.hpp", "pma.hpp", "repack.hpp", "gram.hpp", "gram2.hpp", "gram4.hpp", "gram2w.hpp", "pfx.hpp", "charwise.hpp", "build_common.hpp", "device_tables.hpp", "chain_scan.hpp", os.path.join("..", "..", "include", "daachorse_amd.h"),
           os.path.join("..", "..", "include", "daac_synth.h")]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def source_hash():
    """sha256 over the sources the library is built from (names + bytes, in SOURCES + HEADERS order): the stamp tools/profile_round.sh
    writes into profiles/hbm_traffic.json and bench.py compares (`roofline.traffic_stale`) — counters of other kernels than the ones
    that ran are worth saying so."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        path = os.path.join(CSRC, f)
        h.update(os.path.basename(f).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in files)


def build(force=False, verbose=False):
    """One hipcc -c per source, in parallel (gram_kernels.hip with its template instances is the long pole),
    objects under daachorse_amd/build/ (git-ignored), then one link."""
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(_HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]

    def compile_one(src):
        obj = os.path.join(obj_dir, src + ".o")
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
