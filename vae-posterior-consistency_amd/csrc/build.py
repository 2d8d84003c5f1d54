"""Build libvpc_hip.so (gfx950 only) in-tree with hipcc.  Usage: python build.py [--force] [--asm]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = ["vpc_enc.hip", "vpc_dec.hip", "vpc_dec8.hip", "vpc_step.hip", "vpc_small.hip", "vpc_misc.hip", "vpc_reward.hip", "vpc_gemm.hip", "vpc_nm.hip", "vpc_nmdec.hip", "vpc_eddi.hip", "vpc_rccl.hip"]
HDRS = ["vpc_device.h", "vpc_layout.h", "vpc_abi_internal.h", "vpc_dec_args.h", "vpc_bf16.h", "vpc_rng.h", "vpc_adam.h", "../../include/vpc.h"]
LIB = os.path.join(HERE, "libvpc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = os.environ.get("VPC_EXTRA_FLAGS", "").split() + ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, asm=False):
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HDRS]
    if _newer(obj, deps):
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj]
        if asm:
            cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if asm:
            with open(obj + ".resource.txt", "w") as f:
                f.write(r.stderr)
    return obj


def build_ablate():
    """Diagnostic library with the VPC_DEBUG ablation switches / phase stamps compiled in (never loaded by the product;
    VPC_LIB points a process at it).  Objects go to build_ablate/, compiled in parallel."""
    out = os.path.join(HERE, "libvpc_hip_ablate.so")
    bdir = os.path.join(HERE, "build_ablate")
    os.makedirs(bdir, exist_ok=True)

    def one(src):
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HDRS]
        if _newer(obj, deps):
            r = subprocess.run([HIPCC] + FLAGS + ["-DVPC_ABLATE", "-c", os.path.join(HERE, src), "-o", obj], cwd=HERE,
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, SRCS))
    r = subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out] + objs + ["-ldl"], cwd=HERE,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


def build(force=False, asm=False):
    if force:
        for s in SRCS:
            o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, asm), SRCS))
    if _newer(LIB, objs):
        r = subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-ldl"], cwd=HERE,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    if "--ablate" in sys.argv:
        print(build_ablate())
    else:
        print(build(force="--force" in sys.argv, asm="--asm" in sys.argv))
