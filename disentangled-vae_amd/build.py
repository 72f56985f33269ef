"""Build libdvae_hip.so (gfx950) in-tree with hipcc.

    python disentangled-vae_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the
GPU box with the gpurun snapshot.  Rebuilds only when a source is newer.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdvae_hip.so")
DIAG_LIB = os.path.join(HERE, "libdvae_hip_diag.so")     # --diag: product kernels + the measured-slower alternates (csrc/common.hpp: DVAE_DIAG)
ARCH = "gfx950"
# per-file flags (the reason stands at the top of the file named)
FILE_FLAGS = {"mcem_mstep.hip": ["-fno-slp-vectorize"], "train_rows3.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _newer_than_lib(paths, LIB=LIB):
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in paths)


def build(force=False, verbose=True, diag=False):
    """diag=True: the diagnostic library (-DDVAE_DIAG) as libdvae_hip_diag.so beside the product one; load it with DVAE_LIB=<path>."""
    if diag:
        return _build(DIAG_LIB, os.path.join(HERE, "build", "diag"), ["-DDVAE_DIAG"], force, verbose)
    return _build(LIB, os.path.join(HERE, "build"), [], force, verbose)


def _build(LIB, objdir, extra, force, verbose):
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if not force and not _newer_than_lib(deps, LIB):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(objdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                [os.path.getmtime(src)] + [os.path.getmtime(p) for p in deps if p.endswith((".hpp", ".h", ".inc"))]):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function"] + FILE_FLAGS.get(os.path.basename(src), []) + extra + os.environ.get("DVAE_CFLAGS", "").split()
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv))
