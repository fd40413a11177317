"""Build the native pieces in-tree with hipcc for gfx950 (no cmake, no JIT cache).

  libStarFlashAttention.so   HIP kernels + C ABI (include/star_flash_attn.h) + the C++ template
                             surface of src/flash_attn.h           -> starflashattention_amd/lib/
  star_flash_attn*.so        pybind11/ATen binding (src/flash_api.cpp), same module name and
                             function as the reference's extension  -> repo root

hipcc cross-compiles without a GPU, so this runs in the authoring container; the built .so
files travel to the GPU box with the tree (they are git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "starflashattention_amd", "csrc")
LIBDIR = os.path.join(ROOT, "starflashattention_amd", "lib")
OBJDIR = os.path.join(ROOT, "build", "obj")
LIB_NAME = "libStarFlashAttention.so"
ARCH = "gfx950"

EXTRA_FLAGS = {}      # per-source extra hipcc flags

# (the longest compiles first: the pool below takes the sources in this order)
KERNEL_SOURCES = ["prefill_w4_kernel.hip", "prefill_w4_kernel_p1.hip", "prefill_w4_kernel_p2.hip", "prefill_w4_kernel_p3.hip",
                  "prefill_w4d_kernel.hip", "decode_kernel.hip", "decode_gqa_kernel.hip", "decode_gqa_mfma_kernel.hip", "prefill_d256_kernel.hip",
                  "prefill_kernel.hip", "prefill_kernel_bm128.hip", "prefill_dispatch.hip",
                  "aux_kernels.hip", "c_api.hip", "cxx_surface.hip"]
# Earlier kernel generations kept for A/B runs (tools/prefill_ab.py, pytest -m variants).  They are never an
# auto choice of launch_prefill, so the shipping library does not carry them: build_lib(variants=True)
# compiles them (and -DSFA_WITH_VARIANTS) into a second library, libStarFlashAttention_ab.so.
VARIANT_SOURCES = ["prefill_kernel16.hip", "prefill_baseline.hip", "prefill_w4r2_kernel.hip"]
AB_LIB_NAME = "libStarFlashAttention_ab.so"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0 with gfx950 support)")
    return exe


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _stamp(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(CSRC, "prefill_w4_kernel.hip")]        # included by prefill_w4_kernel_p1..3.hip
    hs += [os.path.join(ROOT, "include", "star_flash_attn.h")]
    hs += [os.path.join(ROOT, "src", f) for f in ("params.h", "traits.h", "flash_attn.h")]
    return [h for h in hs if os.path.exists(h)]


def build_lib(force=False, verbose=False, extra_flags=(), variants=False):
    """hipcc -> starflashattention_amd/lib/libStarFlashAttention.so; returns its path.
    variants=True builds the A/B library (earlier kernel generations included) next to it instead."""
    objdir = OBJDIR + ("_ab" if variants else "")
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(LIBDIR, AB_LIB_NAME if variants else LIB_NAME)
    names = KERNEL_SOURCES + (VARIANT_SOURCES if variants else [])
    srcs = [os.path.join(CSRC, s) for s in names]
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + ROOT,
             "-Wall", "-Wno-unused-function"] + (["-DSFA_WITH_VARIANTS=1"] if variants else []) + list(extra_flags)
    stamp = _stamp(srcs + _headers(), " ".join(flags) + repr(sorted(EXTRA_FLAGS.items())))
    stamp_file = out + ".stamp"
    if (not force and os.path.exists(out) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return out
    cc = hipcc()

    hdrs = _headers()

    def compile_one(src):
        # an object is rebuilt only when its source, a header or its flags changed (the 4-wave prefill kernels take
        # minutes each; a one-line change elsewhere should not recompile them)
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        mine = flags + EXTRA_FLAGS.get(os.path.basename(src), [])
        ostamp = _stamp([src] + hdrs, " ".join(mine))
        ostamp_file = obj + ".stamp"
        if (not force and os.path.exists(obj) and os.path.exists(ostamp_file) and open(ostamp_file).read() == ostamp):
            return obj
        log = _run([cc] + mine + ["-c", src, "-o", obj])
        if verbose and log.strip():
            print(log)
        with open(ostamp_file, "w") as f:
            f.write(ostamp)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 4, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    _run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return out


def build_pybind(force=False, verbose=False):
    """Compile src/flash_api.cpp into the top-level `star_flash_attn` extension module."""
    import torch
    from torch.utils import cpp_extension

    lib = build_lib()
    src = os.path.join(ROOT, "src", "flash_api.cpp")
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(ROOT, "star_flash_attn" + suffix)
    stamp = _stamp([src] + _headers(), torch.__version__)
    stamp_file = os.path.join(ROOT, "build", "star_flash_attn.stamp")
    if (not force and os.path.exists(out) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return out
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    inc = cpp_extension.include_paths(device_type="cuda")     # torch-ROCm: adds the HIP include dirs
    tlib = cpp_extension.library_paths(device_type="cuda")
    cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=star_flash_attn",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-Wno-deprecated-declarations", "-I" + ROOT, "-I" + sysconfig.get_paths()["include"]]
    cmd += ["-I" + i for i in inc]
    cmd += [src, "-o", out]
    cmd += ["-L" + p for p in tlib] + ["-L" + LIBDIR]
    cmd += ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
            "-lStarFlashAttention", "-Wl,-rpath,$ORIGIN/starflashattention_amd/lib"]
    cmd += ["-Wl,-rpath," + p for p in tlib]
    log = _run(cmd)
    if verbose and log.strip():
        print(log)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return out


def build_examples(force=False, verbose=False):
    """hipcc the torch-free C++ harness against the C++ surface (src/flash_attn.h)."""
    lib = build_lib()
    src = os.path.join(ROOT, "examples", "cpp", "flash_decoder_harness.cc")
    out = os.path.join(ROOT, "build", "flash_decoder_harness")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if (not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(lib))):
        return out
    log = _run([hipcc(), "-O2", "-std=c++17", f"--offload-arch={ARCH}", "-I" + ROOT, src, "-L" + LIBDIR,
                "-lStarFlashAttention", "-L/opt/rocm/lib", "-lroctx64",
                "-Wl,-rpath,$ORIGIN/../starflashattention_amd/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    if verbose and log.strip():
        print(log)
    return out


def build_all(force=False, verbose=False):
    return build_lib(force, verbose), build_pybind(force, verbose), build_examples(force, verbose)


if __name__ == "__main__":
    force = "--force" in sys.argv
    only_lib = "--lib-only" in sys.argv
    print(build_lib(force, verbose=True))
    if "--variants" in sys.argv:
        print(build_lib(force, verbose=True, variants=True))
    if not only_lib:
        print(build_pybind(force, verbose=True))
        print(build_examples(force, verbose=True))
