"""Build recipe for the native library (hipcc, gfx950 only, in-tree so it travels to the GPU box)."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libhomer_gpu.so")
SOURCES = ["tables.cpp", "context.cpp", "dropin.cpp", "cmdlist.cpp", "k_pixel.hip", "k_transform.hip", "k_intra.hip", "k_interp.hip", "k_loop.hip", "k_motion.hip", "k_tuchain.hip", "k_intrasearch.hip", "k_tree.hip", "k_chromasearch.hip", "k_saooffsets.hip", "k_subpel.hip", "k_probe.hip", "k_primtest.hip", "k_encode.hip"]
# -ffp-contract=off: the few double-precision cost terms must round exactly like the reference's x87-free SSE2 code
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip"] + (["-DHENC_PROFILE"] if os.environ.get("HENC_PROFILE") else []) + os.environ.get("HENC_EXTRA_FLAGS", "").split()


STAMP = LIB_PATH + ".flags"   # the flags the library was built with: a profiling / debug-info build is never mistaken for the product build


def _stale():
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP) or open(STAMP).read() != " ".join(FLAGS):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs] + [os.path.join(PKG_DIR, "..", "include", "homer_gpu.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def source_digest():
    """A short digest of everything the library is built from (csrc/ and the public header): the GPU box gets the tree without .git, so this - not a commit - is how
    a measurement file says which build it belongs to (bench.py `build`, tools/pmc_kernels.py `source_digest`)."""
    import hashlib
    h = hashlib.sha1()
    paths = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs if f.endswith((".h", ".hip", ".cpp", ".inc")))
    for path in paths + [os.path.join(PKG_DIR, "..", "include", "homer_gpu.h")]:
        h.update(os.path.relpath(path, PKG_DIR).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:12]


def build_native(force=False, verbose=False):
    """Compile every HIP source into homerhevc_amd/libhomer_gpu.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # one hipcc per source, four at a time, then one link: the TU-chain file alone takes most of a minute
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory(prefix="homer_gpu_build_") as td:
        objs = [os.path.join(td, os.path.splitext(src)[0] + ".o") for src in SOURCES]

        def compile_one(pair):
            src, obj = pair
            cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)

        order = sorted(zip(SOURCES, objs), key=lambda p: -os.path.getsize(os.path.join(CSRC, p[0])))     # the long compiles first
        with ThreadPoolExecutor(max_workers=4) as pool:
            list(pool.map(compile_one, order))
        link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB_PATH + ".tmp"]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
        os.replace(LIB_PATH + ".tmp", LIB_PATH)
        with open(STAMP, "w") as f:
            f.write(" ".join(FLAGS))
    return LIB_PATH
