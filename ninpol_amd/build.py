"""Build libninpol_amd.so in-tree: hand-written HIP for gfx950 + the host grid builder.

`python -m ninpol_amd.build` (or __graft_entry__.build()).  hipcc cross-compiles without a GPU.
The shared object lives next to this file so it travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libninpol_amd.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
ARCH = "gfx950"

# (source, compiler, extra flags)
UNITS = [
    # host connectivity: g++ + libgomp; no FMA contraction (float32 normals must match the reference)
    ("grid_host.cpp", "g++", ["-fopenmp", "-ffp-contract=off"]),
    # native table packing; diff_mag must be the reference's value bit for bit: no contraction either
    ("pack_host.cpp", "g++", ["-fopenmp", "-ffp-contract=off"]),
    # IDW / LS: contraction off so results are the reference's bit for bit
    ("kernels_idw_ls.hip", "hipcc", ["-ffp-contract=off"]),
    ("kernels_gls.hip", "hipcc", []),
    ("kernels_gls_block.hip", "hipcc", []),
    # no atomic optimizer: it turns the work-queue atomicAdd into mbcnt + readfirstlane of the returned value right
    # behind the atomic, i.e. a full round-trip wait at the top of every pass; left alone the value is first read at
    # the end of the pass (kernels_gls_hex8mf.hip, grab_issue / grab_value)
    ("kernels_gls_hex8mf.hip", "hipcc", ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]),
    ("kernels_gls_mfw.hip", "hipcc", []),
    ("kernels_gls_mfx.hip", "hipcc", []),
    ("kernels_gls_mfg.hip", "hipcc", []),
    ("kernels_gls_quad4.hip", "hipcc", []),
    ("kernels_csr.hip", "hipcc", []),
    # device-side grid build: no contraction, like grid_host.cpp (float32 normals must match the reference)
    ("grid_device.hip", "hipcc", ["-ffp-contract=off"]),
    ("abi.hip", "hipcc", ["-Wno-unknown-pragmas"]),
    # the peer-to-peer exchange of the multi-GPU path (nin_exchange_*): HIP runtime calls only, no kernel
    ("exchange.hip", "hipcc", []),
]


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _hipcc():
    return shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp"))]
    srcs += [os.path.join(os.path.dirname(HERE), "include", "ninpol_amd.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose_resources=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    for src, cc, extra in UNITS:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        common = ["-O3", "-fPIC", "-std=c++17", "-c", os.path.join(CSRC, src), "-o", obj]
        if cc == "hipcc":
            cmd = [_hipcc(), f"--offload-arch={ARCH}"] + extra + os.environ.get("NIN_EXTRA_HIPCC_FLAGS", "").split() + common
            if verbose_resources:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        else:
            cmd = [cc, "-I", os.path.join(ROCM, "include")] + extra + common
        _run(cmd)
        objs.append(obj)
    _run(["g++", "-shared", "-o", LIB] + objs +
         ["-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-lgomp", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    build_tools()
    return LIB


def build_tools():
    """Stand-alone device test programs (tests/test_gpu_parity.py runs them on the GPU box): tools/_bin/test_xstrip."""
    tools = os.path.join(os.path.dirname(HERE), "tools")
    os.makedirs(os.path.join(tools, "_bin"), exist_ok=True)
    _run([_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-I", CSRC, os.path.join(tools, "test_xstrip.hip"), "-o",
          os.path.join(tools, "_bin", "test_xstrip")])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose_resources="--resources" in sys.argv))
