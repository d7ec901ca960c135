"""Build the HIP extension in-tree: cet_pick_amd/libcetpick_hip.so (gfx950 code objects only).

    python -m cet_pick_amd.build          # incremental
    python -m cet_pick_amd.build --force

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libcetpick_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def source_sha16(prefixes=None):
    """Hash of the kernel sources whose file name starts with one of `prefixes` (all of csrc/*.hip, csrc/*.h and the
    C-ABI header when None): profiles/ files carry it so a reader (and bench.py) can tell whether a measurement belongs
    to the kernels that are in the tree now."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    if prefixes is None:
        files.append(os.path.join(os.path.dirname(HERE), "include", "cetpick_hip.h"))
    else:
        files = [f for f in files if os.path.basename(f).startswith(tuple(prefixes))]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "cetpick_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _extra_flags(path):
    """Per-file compiler flags: a source line `// hipcc-flags: <flags>` (e.g. -fno-slp-vectorize for a kernel whose
    scalar FMA chains the SLP vectoriser would turn into register-pair shuffles)."""
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("// hipcc-flags:"):
                out += line.split(":", 1)[1].split()
    return out


def _compile(src, force):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(sp), _headers_mtime())):
        return obj, False
    cmd = [HIPCC] + FLAGS + _extra_flags(sp) + ["-c", sp, "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [r[0] for r in res]
    if any(r[1] for r in res) or not os.path.exists(LIB):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
