"""Build libvnface.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m vn_celeb_face_recognition_amd.build [--force]

The shared object lands next to this file so it travels with a repository snapshot; it is
git-ignored.  hipcc cross-compiles gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libvnface.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=off"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "vnface.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hm = _headers_mtime()
    jobs = []
    objs = []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hm):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op]
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for src, rc, out in ex.map(run, jobs):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
    if jobs or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print("built", LIB, "(%d objects, %d recompiled)" % (len(objs), len(jobs)))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
