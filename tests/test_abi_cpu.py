"""CPU checks of the C-ABI boundary: the shared library builds/loads without a GPU, exports every symbol that
include/ov2slam_hip.h declares, the ctypes PODs have the C layout, and the product path fails loudly (never falls
back to the CPU) when no device is present.  No compute calls are made here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ov2slam_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from ov2slam_amd import _lib
    return _lib.load()


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ov2_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from ov2slam_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libov2hip.so does not export {n}"
        assert n in _lib.SIGNATURES, f"ov2slam_amd/_lib.py does not bind {n}"
    assert set(_lib.SIGNATURES) <= set(names)


def test_pod_layout_matches_c():
    """compile a tiny C program against the public header and compare sizeof/offsetof with the ctypes mirrors"""
    from ov2slam_amd import ba_types as T
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "ov2slam_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(ov2_ba_problem), sizeof(ov2_ba_options), sizeof(ov2_ba_result), sizeof(ov2_ba_iter));
  printf("%zu %zu %zu %zu\n", offsetof(ov2_ba_problem, pose), offsetof(ov2_ba_problem, res_sigma),
         offsetof(ov2_ba_options, jacobi_scaling), offsetof(ov2_ba_result, log));
  return 0; }'''
    exe = "/tmp/ov2_layout_check"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[:4] == [C.sizeof(T.BaProblemC), C.sizeof(T.BaOptionsC), C.sizeof(T.BaResultC), C.sizeof(T.BaIterC)]
    assert sizes[4:] == [T.BaProblemC.pose.offset, T.BaProblemC.res_sigma.offset, T.BaOptionsC.jacobi_scaling.offset,
                         T.BaResultC.log.offset]


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    st = lib.ov2_ctx_create(0, C.byref(h))
    assert st != 0 and not h.value
    from ov2slam_amd import frontend, _lib
    with pytest.raises(_lib.Ov2Error):
        frontend.Context(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ov2slam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                # comments may cite the oracle as the restated spec; code must not import, include, link or dlopen it
                assert not re.search(r"^\s*(from|import)\s+oracle|oracle_py|libov2oracle|#include\s+\"[^\"]*oracle", txt, re.M), f


def test_no_kernel_uses_scratch():
    """every gfx950 kernel embedded in libov2hip.so must have private_segment_fixed_size == 0: on this runtime a
    dispatch that needs scratch (register spills, dynamically indexed local arrays) stalls for milliseconds in
    queue-scratch management -- a silent 5-10x slowdown of the BA path when it happened."""
    import re
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(ROOT, "ov2slam_amd", "lib", "libov2hip.so")
    if not (os.path.exists(lib) and os.path.exists(os.path.join(llvm, "clang-offload-bundler"))):
        pytest.skip("library or LLVM tools not present")
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
        assert offs, "no offload bundle found in the library"
        n_kernels, bad = 0, []
        for k, o in enumerate(offs):
            b, co = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"b{k}.co")
            open(b, "wb").write(data[o:offs[k + 1] if k + 1 < len(offs) else len(data)])
            subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={b}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            name = None
            for line in notes.splitlines():
                m = re.search(r"\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
                if m:
                    n_kernels += 1
                    if int(m.group(1)) > 0:
                        bad.append((name, int(m.group(1))))
        assert n_kernels > 40
        assert not bad, f"kernels using scratch: {bad}"
