"""Builds librectify_amd/librectify_amd.so (HIP kernels + C ABI) for gfx950 with hipcc.

Usage: python -m librectify_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librectify_amd.so")
SOURCES = [
    "kernels_filter.hip",
    "kernels_seeds.hip",
    "kernels_flood.hip",
    "kernels_fit.hip",
    "kernels_ransac.hip",
    "kernels_groups.hip",
    "context.hip",
    "vp_host.cpp",
    "api.cpp",
]
# -ffp-contract=off: every FMA in the canonical arithmetic is an explicit fmaf(); division and
# sqrt stay correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
# -fno-slp-vectorize: left alone, clang packs neighbouring scalar f32 FMAs into v_pk_fma_f32, which on gfx950
# issues slower than the two v_fma_f32 it replaces (MI355X_MICROARCH.md, "price of one filler").
# -fvisibility=hidden: the library exports the six reference symbols and the lr_* extensions, nothing else (the headers
# under include/ push default visibility around their declarations).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-Wno-unused-function", "-fno-slp-vectorize", "-fvisibility=hidden", "-fvisibility-inlines-hidden"] + os.environ.get("LR_EXTRA_FLAGS", "").split()


# kernels_filter.hip: inputs are finite by contract (the reference divides/compares them freely as well), so the
# max-chains need no sNaN quieting: without the IEEE mode bit v_max_f32 is a single instruction.
PER_FILE_FLAGS = {"kernels_filter.hip": os.environ.get("LR_FILTER_FLAGS", "-fno-honor-nans -mno-amdgpu-ieee").split()}


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: librectify_amd needs the ROCm toolchain (no CPU fallback exists)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.abspath(__file__),
        os.path.join(HERE, "..", "include", "librectify.h"),
        os.path.join(HERE, "..", "include", "librectify_amd.h"),
    ]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        extra = PER_FILE_FLAGS.get(src, [])
        cmd = [hipcc] + FLAGS + extra + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out.decode()))
        elif verbose and out.strip():
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("librectify_amd build failed")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
                           "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
