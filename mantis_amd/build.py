"""Build libmantis_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m mantis_amd.build            # incremental (per-source object cache under mantis_amd/csrc/_obj)
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmantis_hip.so")
SOURCES = ["pack", "norm", "act", "rope", "ce", "vit", "gemm", "gemm176", "gemm_fp8", "attn", "attn_fwd64", "optim"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
# attn_fwd64: the hand-placed instruction stream wants one VALU instruction per source operation (no v_pk_* packing of f32 pairs)
EXTRA_FLAGS = {"attn_fwd64": ["-fno-slp-vectorize"]}
HEADERS = ["common.h", "attn_common.h", "gemm_ring.h"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(path, name):
    h = hashlib.sha1()
    for p in (path, *(os.path.join(CSRC, x) for x in HEADERS)):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(name, [])).encode())
    return h.hexdigest()


def _compile(name):
    src = os.path.join(CSRC, name + ".hip")
    obj = os.path.join(OBJ, name + ".o")
    stamp = obj + ".sha1"
    dig = _digest(src, name)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(name, []), "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}.hip:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res) or not os.path.exists(LIB)
    if changed:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[mantis_amd.build] {'built' if changed else 'up to date'}: {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build()
