"""Build recipes for the native pieces (gfx950 only, in-tree outputs)."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zopfli_amd", "csrc")
LIB = os.path.join(ROOT, "zopfli_amd", "libzopfli_amd.so")
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
    return os.path.join(dev, "zmx_hip.hip"), cc, hdr


def build_product(force=False):
    """hipcc --offload-arch=gfx950: HIP device layer + C++ host code -> libzopfli_amd.so."""
    hip, cc, hdr = _sources()
    if not force and not _newer(LIB, [hip] + cc + hdr):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(CSRC, "host"),
           "-I" + os.path.join(CSRC, "device"), hip] + cc + ["-o", LIB, "-lpthread"]
    subprocess.check_call(cmd)
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
