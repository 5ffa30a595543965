"""In-tree build of the native pieces: libctu_engine.so (hipcc, gfx950) and the `ctucopy` CLI."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(HERE, "libctu_engine.so")
CLI = os.path.join(ROOT, "bin", "ctucopy")
SYNTH_LIB = os.path.join(HERE, "libctu_synth.so")  # the input generator alone (include/ctu_synth.h): no HIP runtime behind it

ENGINE_SRCS = ["engine.hip", "opts.cc", "design.cc", "synth.cc"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(dirpath):
    return [os.path.join(dirpath, f) for f in os.listdir(dirpath) if f.endswith((".hip", ".cc", ".h"))]


def build_engine(force=False, verbose=False):
    deps = _deps(CSRC) + [os.path.join(ROOT, "include", h) for h in ("ctu_engine.h", "ctu_synth.h")]
    if force or _newer(LIB, deps):
        # -fno-slp-vectorize: packed f32 VALU (v_pk_*) plus the moves it needs is slower than scalar f32 on gfx950
        cmd = [HIPCC, "-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-pthread",
               *[os.path.join(CSRC, s) for s in ENGINE_SRCS], "-o", LIB]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.run(cmd, check=True)
    return LIB


def build_synth(force=False):
    """The synthetic-set generator as a library of its own (g++ only): processes that never touch the GPU - the CPU
    baseline's workers, fixture scripts - load this instead of the HIP-linked engine library."""
    src = os.path.join(CSRC, "synth.cc")
    if force or _newer(SYNTH_LIB, [src, os.path.join(ROOT, "include", "ctu_synth.h")]):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"), src,
                        "-o", SYNTH_LIB], check=True)
    return SYNTH_LIB


def build_cli(force=False):
    if not os.path.isdir(HOST):
        return None
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith(".cc")]
    if not srcs:
        return None
    build_engine()
    deps = srcs + _deps(HOST) + [LIB]
    if force or _newer(CLI, deps):
        os.makedirs(os.path.dirname(CLI), exist_ok=True)
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", CSRC, *srcs,
               "-o", CLI, "-L", HERE, "-lctu_engine", "-Wl,-rpath," + HERE]
        subprocess.run(cmd, check=True)
    return CLI


def build_all(force=False):
    build_engine(force)
    build_synth(force)
    build_cli(force)


if __name__ == "__main__":
    build_all(force=True)
