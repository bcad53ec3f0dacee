"""Build recipes for the native pieces (gfx950 only, in-tree outputs)."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zopfli_amd", "csrc")
LIB = os.path.join(ROOT, "zopfli_amd", "libzopfli_amd.so")
# the reference's shared library is libzopfli.so.1 (Makefile:45): the same file under that name, so that
# a program linked against the reference's library picks this one up from LD_LIBRARY_PATH
LIB_SONAME = os.path.join(ROOT, "zopfli_amd", "libzopfli.so.1")
DATAGEN = os.path.join(CSRC, "tools", "libzopfli_datagen.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources():
    host = os.path.join(CSRC, "host")
    dev = os.path.join(CSRC, "device")
    cc = sorted(os.path.join(host, f) for f in os.listdir(host) if f.endswith(".cc"))
    hdr = [os.path.join(d, f) for d in (host, dev) for f in os.listdir(d) if f.endswith(".h")]
    hdr.append(os.path.join(ROOT, "include", "zopfli_amd.h"))
    hdr.append(os.path.join(CSRC, "libzopfli_amd.map"))
    return os.path.join(dev, "zmx_hip.hip"), cc, hdr


def build_product(force=False):
    """hipcc --offload-arch=gfx950: HIP device layer + C++ host code -> libzopfli_amd.so."""
    hip, cc, hdr = _sources()
    if not force and not _newer(LIB, [hip] + cc + hdr) and os.path.exists(LIB_SONAME):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fvisibility=hidden", "-Wl,-soname,libzopfli.so.1",
           "-Wl,--version-script=" + os.path.join(CSRC, "libzopfli_amd.map"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(CSRC, "host"),
           "-I" + os.path.join(CSRC, "device"), hip] + cc + ["-o", LIB, "-lpthread", "-ldl"]
    subprocess.check_call(cmd)
    shutil.copyfile(LIB, LIB_SONAME)   # (a copy, not a symlink: it has to survive the snapshot to the GPU box)
    return LIB


def build_datagen(force=False):
    src = os.path.join(CSRC, "tools", "datagen.c")
    if force or _newer(DATAGEN, [src]):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", DATAGEN, src])
    return DATAGEN


def build_oracle():
    """Test infrastructure: the C restatement and (when /root/reference exists) the real reference."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


def build_hosttest():
    """Test infrastructure: product host sources + oracle-backed zmx layer (CPU-only checks)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostlib")])


REF_CLI = os.path.join(ROOT, "tests", "_build", "zopfli_ref_cli_amd")


def build_ref_cli():
    """Test infrastructure: the REFERENCE's own CLI (src/zopfli/zopfli_bin.c, compiled where it lies
    under /root/reference, nothing copied) linked against libzopfli_amd.so instead of libzopfli —
    the drop-in claim of INTEGRATION.md section 1.  Only possible where /root/reference exists; the
    binary travels to the GPU box in tests/_build/."""
    src = "/root/reference/src/zopfli/zopfli_bin.c"
    if not os.path.exists(src):
        return None
    os.makedirs(os.path.dirname(REF_CLI), exist_ok=True)
    if _newer(REF_CLI, [src, LIB]):
        subprocess.check_call(["gcc", "-O2", "-w", src, "-I/root/reference/src/zopfli",
                               "-L" + os.path.dirname(LIB), "-lzopfli_amd",
                               "-Wl,-rpath," + os.path.dirname(LIB), "-o", REF_CLI])
    return REF_CLI


REF_SO_DIR = os.path.join(ROOT, "tests", "_build", "reflib")
REF_CLI_DYN = os.path.join(ROOT, "tests", "_build", "zopfli_ref_cli_dyn")


def build_ref_dynamic():
    """Test infrastructure: the reference as ITS OWN shared library (tests/_build/reflib/libzopfli.so.1,
    soname libzopfli.so.1 as in the reference's Makefile:45) and the reference's CLI linked against it
    with no rpath.  LD_LIBRARY_PATH decides at run time whose libzopfli.so.1 it gets — the drop-in
    switch of an existing dynamically linked user."""
    zsrc = "/root/reference/src/zopfli"
    if not os.path.isdir(zsrc):
        return None
    os.makedirs(REF_SO_DIR, exist_ok=True)
    so = os.path.join(REF_SO_DIR, "libzopfli.so.1")
    lib_c = [os.path.join(zsrc, f) for f in sorted(os.listdir(zsrc)) if f.endswith(".c") and f != "zopfli_bin.c"]
    if _newer(so, lib_c):
        subprocess.check_call(["gcc", "-O2", "-w", "-fPIC", "-shared", "-Wl,-soname,libzopfli.so.1"] + lib_c +
                              ["-lm", "-o", so])
        link = os.path.join(REF_SO_DIR, "libzopfli.so")
        if os.path.lexists(link):
            os.remove(link)
        shutil.copyfile(so, link)
    if _newer(REF_CLI_DYN, [os.path.join(zsrc, "zopfli_bin.c"), so]):
        subprocess.check_call(["gcc", "-O2", "-w", os.path.join(zsrc, "zopfli_bin.c"), "-I" + zsrc,
                               "-L" + REF_SO_DIR, "-lzopfli", "-o", REF_CLI_DYN])
    return REF_CLI_DYN


PNG_AMD = os.path.join(ROOT, "tests", "_build", "zopflipng_amd")
PNG_REF = os.path.join(ROOT, "tests", "_build", "zopflipng_ref")


def build_zopflipng():
    """Test infrastructure: the reference's zopflipng (src/zopflipng/*.cc + its vendored LodePNG, compiled
    where they lie under /root/reference) twice — once with the reference's own zopfli objects and once
    against libzopfli_amd.so (CustomPNGDeflate -> ZopfliDeflate, zopflipng_lib.cc:47-66).  INTEGRATION.md
    section 3; both binaries travel to the GPU box in tests/_build/."""
    ref = "/root/reference/src"
    if not os.path.isdir(os.path.join(ref, "zopflipng")):
        return None
    os.makedirs(os.path.dirname(PNG_AMD), exist_ok=True)
    png = [os.path.join(ref, "zopflipng", f) for f in ("zopflipng_bin.cc", "zopflipng_lib.cc")]
    png += [os.path.join(ref, "zopflipng", "lodepng", f) for f in ("lodepng.cpp", "lodepng_util.cpp")]
    if _newer(PNG_AMD, png + [LIB]):
        subprocess.check_call(["g++", "-O2", "-w"] + png + ["-L" + os.path.dirname(LIB), "-lzopfli_amd",
                              "-Wl,-rpath," + os.path.dirname(LIB), "-o", PNG_AMD])
    if _newer(PNG_REF, png):
        zsrc = os.path.join(ref, "zopfli")
        zc = [os.path.join(zsrc, f) for f in sorted(os.listdir(zsrc)) if f.endswith(".c") and f != "zopfli_bin.c"]
        objdir = os.path.join(ROOT, "tests", "_build", "zref_obj")
        os.makedirs(objdir, exist_ok=True)
        objs = []
        for c in zc:
            o = os.path.join(objdir, os.path.basename(c)[:-2] + ".o")
            subprocess.check_call(["gcc", "-O2", "-w", "-c", c, "-o", o])
            objs.append(o)
        subprocess.check_call(["g++", "-O2", "-w"] + png + objs + ["-lm", "-o", PNG_REF])
    return PNG_AMD


PNG_LIB = os.path.join(ROOT, "zopfli_amd", "libzopflipng_amd.so")
PNG_AMD2 = os.path.join(ROOT, "tests", "_build", "zopflipng_amd2")
PNG_FILTER_REF = os.path.join(ROOT, "tests", "_build", "libpng_filter_ref.so")


def lodepng_dir():
    """Where LodePNG's sources are (lodepng.cpp, lodepng_util.cpp and their headers): the third-party library
    zopflipng is built on; the reference vendors it under src/zopflipng/lodepng."""
    return os.environ.get("LODEPNG_DIR", "/root/reference/src/zopflipng/lodepng")


def build_png_lib(force=False):
    """libzopflipng_amd.so: zopflipng's optimiser library — csrc/png/zopflipng_amd.cc with the ABI of
    the reference's zopflipng_lib.h — on top of libzopfli_amd.so, LodePNG compiled in from LODEPNG_DIR.  Skipped
    (None) where LodePNG is not at hand; the built library travels to the GPU box."""
    lp = lodepng_dir()
    src = os.path.join(CSRC, "png", "zopflipng_amd.cc")
    lode = [os.path.join(lp, f) for f in ("lodepng.cpp", "lodepng_util.cpp")]
    if not all(os.path.exists(f) for f in lode):
        return PNG_LIB if os.path.exists(PNG_LIB) else None
    hdr = [os.path.join(ROOT, "include", "zopflipng_amd.h"), os.path.join(ROOT, "include", "zopfli_amd.h"), LIB]
    if force or _newer(PNG_LIB, [src] + lode + hdr):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", src] + lode +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + lp, "-L" + os.path.dirname(LIB), "-lzopfli_amd",
                               "-Wl,-rpath,$ORIGIN", "-lpthread", "-o", PNG_LIB])
    return PNG_LIB


def build_png_tests():
    """Test infrastructure: the REFERENCE's zopflipng command line (src/zopflipng/zopflipng_bin.cc, compiled where it
    lies) linked against libzopflipng_amd.so instead of the reference's zopflipng_lib.cc, and LodePNG's own scanline
    filter (`filter`, a static function of lodepng.cpp) behind a C symbol — the oracle of zmx_png_filter_types."""
    ref = "/root/reference/src/zopflipng"
    if not os.path.isdir(ref) or not build_png_lib():
        return None
    os.makedirs(os.path.dirname(PNG_AMD2), exist_ok=True)
    cli = os.path.join(ref, "zopflipng_bin.cc")
    if _newer(PNG_AMD2, [cli, PNG_LIB]):
        subprocess.check_call(["g++", "-O2", "-w", cli, "-I" + ref, "-L" + os.path.dirname(PNG_LIB), "-lzopflipng_amd",
                               "-lzopfli_amd", "-Wl,-rpath," + os.path.dirname(PNG_LIB), "-o", PNG_AMD2])
    helper = os.path.join(ROOT, "tests", "hostlib", "png_filter_ref.cc")
    if _newer(PNG_FILTER_REF, [helper, os.path.join(ref, "lodepng", "lodepng.cpp")]):
        subprocess.check_call(["g++", "-O2", "-w", "-fPIC", "-shared", helper, "-I" + os.path.join(ref, "lodepng"), "-o", PNG_FILTER_REF])
    return PNG_AMD2
