"""Build libvdn_render.so (gfx950) from csrc/*.hip with hipcc, in-tree.

hipcc cross-compiles without a GPU. An object is rebuilt only when its source or one of the headers it includes
(hipcc -MD dependency files next to the objects) is newer.
Usage: python -m vdn_hip.build [--force] [-j N]
"""
import concurrent.futures as cf
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
# VDN_BUILD_VARIANT="name:-DFLAG=..": a side copy of the whole library built with extra flags (objects in _build_<name>,
# libvdn_render_<name>.so) for same-box A/B runs of compile-time switches through VDN_LIB (development only)
_VARIANT = os.environ.get("VDN_BUILD_VARIANT", "")
_VNAME, _VFLAGS = (_VARIANT.split(":", 1) + [""])[:2] if _VARIANT else ("", "")
OBJDIR = os.path.join(HERE, "_build" + ("_" + _VNAME if _VNAME else ""))
LIB = os.path.join(HERE, "libvdn_render%s.so" % ("_" + _VNAME if _VNAME else ""))
ARCH = "gfx950"
BASE_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC] + _VFLAGS.split()
PER_FILE_FLAGS = {"rays.hip": ["-ffp-contract=off"], "train_rays.hip": ["-ffp-contract=off"],
                  # k_sdf_fwd2.h: no SLP packing of the epilogue into v_pk_*_f32, MFMA accumulators in arch VGPRs
                  "sdf_bf16.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"],
                  "shade_bf16.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"],
                  "sdf_tail_bf16.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"],
                  "train_sdf_color_bf16.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    m = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _deps_mtime(depfile, hm):
    """Newest mtime among the headers an object really includes (hipcc -MD wrote them next to the object); without a
    dependency file, the newest header of the tree."""
    try:
        toks = open(depfile).read().replace("\\\n", " ").split()
    except OSError:
        return hm
    m = 0.0
    for t in toks[1:]:
        if t.endswith(".h") and not t.startswith(("/opt/", "/usr/")):
            q = t if os.path.isabs(t) else os.path.join(PKG, t)
            try:
                m = max(m, os.path.getmtime(q))
            except OSError:
                return hm           # a header was removed or renamed: rebuild
    return m


def _compile(src, force, hm):
    obj = os.path.join(OBJDIR, src[:-4] + ".o")
    dep = obj[:-2] + ".d"
    sp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), _deps_mtime(dep, hm)):
        return src, 0.0, ""
    t = time.time()
    cmd = ["hipcc"] + BASE_FLAGS + PER_FILE_FLAGS.get(src, []) + ["-MD", "-MF", dep, "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=PKG)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    return src, time.time() - t, r.stderr


def scan_m0_writers(lib_path=None):
    """Post-link gate (every library variant: VDN_BUILD_VARIANT too). csrc/vdn_common.h glds16_imm*: a wave's LDS-DMA pieces of one
    chunk share ONE write of M0 (the LDS destination base), made by the first piece's statement and read by the others up to a
    chunk step later; the later statements cannot declare that dependency to the compiler. It holds as long as nothing else
    writes M0 in between: every instruction with M0 as destination, in every kernel that issues LDS-DMA, must be the head of one
    of the DMA statements (`s_mov_b32 / s_add_u32 m0` - `s_nop 0` - `global_load_lds_dwordx4`). A foreign writer (a compiler
    upgrade that starts using M0, a new build flag) would land weights in the wrong LDS slot silently - so the build fails
    instead. -> (kernels scanned, M0 writes seen); raises RuntimeError on a violation; (0, 0) when llvm-objdump is missing."""
    import re
    import shutil
    import tempfile
    lib_path = lib_path or LIB
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        return 0, 0
    kernels = writes = 0
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(lib_path, os.path.join(tmp, "lib.so"))
        subprocess.run([objdump, "--offloading", os.path.join(tmp, "lib.so")], check=True, capture_output=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            dis = subprocess.run([objdump, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for fn in re.split(r"\n(?=[0-9a-f]{16} <)", dis):          # functions, at the symbol labels
                if "global_load_lds_dwordx4" not in fn:
                    continue
                kernels += 1
                name = fn.split("\n", 1)[0]
                lines = [ln.split("//")[0].strip() for ln in fn.splitlines()[1:] if ln.strip()]
                for i, ln in enumerate(lines):
                    parts = ln.replace(",", " ").split()
                    if len(parts) >= 2 and parts[1] == "m0" and not parts[0].startswith(("s_cmp", "s_bitcmp")):     # M0 as the destination
                        writes += 1
                        if parts[0] not in ("s_mov_b32", "s_add_u32") or not (
                                i + 2 < len(lines) and lines[i + 1].startswith("s_nop") and lines[i + 2].startswith("global_load_lds_dwordx4")):
                            raise RuntimeError("%s: M0 is written outside an LDS-DMA statement in %s: %r" % (os.path.basename(lib_path), name, lines[i:i + 3]))
    return kernels, writes


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    hm = _headers_mtime()
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    rebuilt = False
    with cf.ThreadPoolExecutor(jobs) as ex:
        for src, dt, _ in ex.map(lambda s: _compile(s, force, hm), srcs):
            if dt > 0:
                rebuilt = True
                if verbose:
                    print("[vdn_hip.build] %-24s %.1fs" % (src, dt), flush=True)
    objs = [os.path.join(OBJDIR, s[:-4] + ".o") for s in srcs]
    if rebuilt or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        try:
            k, w = scan_m0_writers(LIB)
        except RuntimeError:
            os.replace(LIB, LIB + ".rejected")          # never leave a library that failed the gate where lib.load() finds it
            raise
        if verbose:
            print("[vdn_hip.build] linked %s (M0 gate: %d LDS-DMA kernels, %d M0 writes, all inside DMA statements)" % (LIB, k, w), flush=True)
    return LIB


if __name__ == "__main__":
    j = None
    if "-j" in sys.argv:
        j = int(sys.argv[sys.argv.index("-j") + 1])
    build(force="--force" in sys.argv, jobs=j)
