"""Builds za_amd/libza_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libza_amd.so")
LIB_ZA2C = os.path.join(HERE, "libza2c.so")     # the outer drop-in boundary: za's C binding symbols (csrc/za2c.cpp)
DRIVER = os.path.join(HERE, "za-gpu")           # `za setup|prove` shaped command-line driver (csrc/za_gpu_main.cpp)
LIB_SYNTH = os.path.join(HERE, "libza_amd_synth.so")   # synthetic workloads + self-test hooks (include/za_amd_synth.h): NOT the product
SOURCES = ["msm.hip", "ntt.hip", "heval.hip", "formats.hip", "keyload.hip", "prover.hip", "setup.hip", "pairing.hip", "outer.hip"]
SYNTH_SOURCES = ["synth.hip"]
HEADERS = ["verifier_sol.inc", "internal_api.h", "fp.h", "fd.h", "ec.h", "ecd.h", "nttd.h", "common.h", "formats.h", "handles.h", os.path.join("..", "..", "include", "za_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-fno-gpu-rdc"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES + SYNTH_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs if s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        # no C++ symbol leaves the library: the C ABI of include/za_amd.h (+ the private table of csrc/internal_api.h)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [
            "-Wl,--version-script=" + os.path.join(CSRC, "libza_amd.map")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # the test / bench scaffolding as a library of its own, on top of the product library's (C++) internals
    sobjs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs if s in SYNTH_SOURCES]
    if force or jobs or _stale(LIB_SYNTH, sobjs + [LIB]):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_SYNTH] + sobjs + ["-L" + HERE, "-lza_amd",
               "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("libza_amd_synth link failed:\n%s\n%s" % (r.stdout, r.stderr))
    # libza2c.so (verbose / setup / prove / verify over the C ABI) and the za-gpu driver: plain host C++
    za2c_src = os.path.join(CSRC, "za2c.cpp")
    if force or jobs or _stale(LIB_ZA2C, [za2c_src, LIB, os.path.join(HERE, "..", "include", "za_amd.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", LIB_ZA2C, za2c_src, "-L" + HERE, "-lza_amd",
               "-Wl,-rpath,$ORIGIN", "-Wl,--version-script=" + os.path.join(CSRC, "libza2c.map")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("libza2c build failed:\n%s\n%s" % (r.stdout, r.stderr))
    main_src = os.path.join(CSRC, "za_gpu_main.cpp")
    if force or _stale(DRIVER, [main_src, LIB_ZA2C]):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", DRIVER, main_src, "-L" + HERE, "-lza2c", "-lza_amd",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + HERE]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("za-gpu build failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
