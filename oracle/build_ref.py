#!/usr/bin/env python3
"""Build recipe for oracle/_ref: the reference's OWN Grid + IDW/LS/GLS, compiled here.

TEST INFRASTRUCTURE ONLY.  Nothing in ninpol_amd/ may import anything built by this file.

What it does
------------
The reference (daviyan5/ninpol, mounted read-only at /root/reference) is Cython.  Its hot path
lives in four extension modules (`ninpol/_interpolator/grid.pyx`, `ninpol/_methods/{idw,ls,gls}.pyx`)
plus the tiny `logger.pyx` / `utils/common.py` they import.  This script runs the `cython`
compiler that ships in the image on those files *where they lie* (no copy of any reference source
is made), writes the generated C/C++ into `oracle/_ref/build/` and links the shared objects into
`oracle/_ref/ninpol/...` (PEP-420 namespace packages: no `__init__.py` is written).  Flags mirror
the reference's `setup.py:90-108` (`-O3 -fopenmp`, boundscheck/wraparound off, cdivision on) but the
reference's setup.py itself is never executed.

`interpolator.pyx` is deliberately NOT built: it does `import meshio` at module level
(`interpolator.pyx:8`) and meshio is not in the image; no stand-in is written for it.  The L3
glue it holds (table packing, COO->CSR) is restated in `oracle/ninpol_oracle.py` and pinned by the
reference's published accuracy numbers instead.

Finally `oracle/ninpol_ref_driver.pyx` (OUR code) is compiled against the reference's .pxd files; it
exposes `build_grid(...)` and `run_method(...)` to Python so tests can drive the real `Grid` and the
real `prepare()` plugins, which are `cdef` and otherwise unreachable without `Interpolator`.

Outputs go only to oracle/_ref/, which is git-ignored AND listed in .gpurunignore: the compiled reference is a
dev-container tool (validating the C restatement, generating tests/golden, the CPU calibration in
profiles/cpu_calibration.json) and never travels to the GPU box.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NINPOL_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
BUILD = os.path.join(OUT, "build")

# (source relative to REF, qualified module name, c++?)
MODULES = [
    ("ninpol/utils/common.py", "ninpol.utils.common", False),
    ("ninpol/_interpolator/logger.pyx", "ninpol._interpolator.logger", False),
    ("ninpol/_interpolator/grid.pyx", "ninpol._interpolator.grid", True),
    ("ninpol/_methods/idw.pyx", "ninpol._methods.idw", False),
    ("ninpol/_methods/ls.pyx", "ninpol._methods.ls", False),
    ("ninpol/_methods/gls.pyx", "ninpol._methods.gls", True),
]

DIRECTIVES = ("boundscheck=False,wraparound=False,nonecheck=False,"
              "initializedcheck=False,cdivision=True,language_level=3")


def sh(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def ext_suffix():
    return sysconfig.get_config_var("EXT_SUFFIX")


def compile_module(src, qualname, cplus, include_dirs, extra_cython=()):
    import numpy as np
    stem = qualname.replace(".", "_")
    gen = os.path.join(BUILD, stem + (".cpp" if cplus else ".c"))
    os.makedirs(BUILD, exist_ok=True)
    cy = [sys.executable, "-m", "cython", "-3", "-X", DIRECTIVES, "-o", gen]
    for inc in include_dirs:
        cy += ["-I", inc]
    if cplus:
        cy.append("--cplus")
    cy += list(extra_cython) + [src]
    sh(cy)
    parts = qualname.split(".")
    outdir = os.path.join(OUT, *parts[:-1])
    os.makedirs(outdir, exist_ok=True)
    so = os.path.join(outdir, parts[-1] + ext_suffix())
    cc = ["g++" if cplus else "gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-w",
          "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
          "-I", sysconfig.get_paths()["include"], "-I", np.get_include(),
          "-I", os.path.join(REF, "ninpol", "utils"),
          gen, "-o", so]
    sh(cc)
    return so


def build(force=False):
    """Build oracle/_ref if the reference tree is present. Returns True when _ref is usable."""
    driver_so = os.path.join(OUT, "ninpol_ref_driver" + ext_suffix())
    if not os.path.isdir(os.path.join(REF, "ninpol")):
        return False                            # no reference tree (the GPU box): nothing to build, nothing to use
    if os.path.exists(driver_so) and not force:
        newest_src = max(os.path.getmtime(p) for p in
                         [os.path.join(HERE, "ninpol_ref_driver.pyx"), os.path.abspath(__file__)])
        if os.path.getmtime(driver_so) >= newest_src:
            return True
    for rel, qual, cplus in MODULES:
        compile_module(os.path.join(REF, rel), qual, cplus, [REF])
    compile_module(os.path.join(HERE, "ninpol_ref_driver.pyx"), "ninpol_ref_driver", True, [REF, HERE])
    # the generated C quotes the reference's source in comments: keep only the shared objects
    import shutil
    shutil.rmtree(BUILD, ignore_errors=True)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref ready" if ok else "reference tree absent and no prebuilt oracle/_ref")
