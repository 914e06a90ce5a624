"""Builds the HIP kernels + C ABI into speechain_amd/lib/libspeechain_amd.so (in-tree, gfx950 only).

    python -m speechain_amd.build            # incremental
    python -m speechain_amd.build --force

hipcc cross-compiles for gfx950 without a GPU; the resulting .so travels to the GPU box with the tree.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libspeechain_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("SC_EXTRA_HIPCC_FLAGS", "").split()  # (e.g. -DSC_ATTN_TIMING: probes)
_SCRATCH = re.compile(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", re.S)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    if force or _stale(obj, [path] + headers):
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        # a kernel that spills to scratch (e.g. an accumulator array that stopped being unrolled) is a performance bug:
        # refuse to build it silently
        spills = [(k, int(n)) for k, n in _SCRATCH.findall(r.stderr) if int(n) > 0]
        if spills and not os.environ.get("SC_ALLOW_SCRATCH"):  # (the override is for instrumented debug builds only)
            os.remove(obj)
            raise RuntimeError(f"{src}: kernels using scratch memory: {spills}")
        other = [l for l in r.stderr.splitlines() if "warning:" in l or "error:" in l]
        if other:
            sys.stderr.write("\n".join(other) + "\n")
    return obj


def build(force: bool = False, jobs: int = 4) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), sources()))
    if force or _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
