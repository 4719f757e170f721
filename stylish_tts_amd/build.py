"""Build libstylish_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstylish_hip.so")
SOURCES = ["api.hip", "conv1d.hip", "convnext.hip", "norms.hip", "attn.hip", "source.hip", "misc.hip", "conv2d.hip",
           "frontend.hip", "bwd.hip", "wgrad.hip", "attn_bwd.hip", "train.hip", "optim.hip", "convnext_bwd.hip", "predictors.hip", "conv32p.hip", "convp16.hip", "wgradb.hip", "disc.hip", "cfdisc.hip", "convk1.hip", "attn16.hip", "convq.hip", "comm.hip", "convnext16.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file additions.  convnext16.hip (the bf16-mode instantiations of csrc/convnext_kernel.h, which says what was measured): the SLP
# vectoriser pairs the depthwise taps / AdaLN of the fused block into v_pk_fma_f32, whose operands cannot be the SGPR weights
FILE_FLAGS = {"convnext16.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def hipcc_version():
    """first line of `hipcc --version` that names the HIP version (what compiled the shipped library), or None"""
    try:
        out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return None
    for ln in out.splitlines():
        if "HIP version" in ln:
            return ln.strip()
    return out.splitlines()[0].strip() if out else None


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "stylish_hip.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
