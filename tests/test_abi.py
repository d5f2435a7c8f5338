"""CPU suite, part 3: the C-ABI library builds for gfx950, loads, exports every symbol include/zkhip.h declares, and
refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import zkevm_circuits_amd as zk
from zkevm_circuits_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ffi.lib()
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert set(names) == set(ffi.EXPORTS)
    assert b"gfx950" in lib.zkhip_version()


def test_rust_sys_crate_declares_exactly_the_header():
    """rust/zkhip-sys/src/lib.rs (source only: no Rust toolchain here) declares one `pub fn` per function of the header,
    no more and no fewer, and its three #[repr(C)] structs list the header's fields in the header's order."""
    rs = open(os.path.join(ROOT, "rust", "zkhip-sys", "src", "lib.rs")).read()
    rs_names = sorted(set(re.findall(r"pub fn (zkhip_[a-z0-9_]+)\s*\(", rs)))
    assert rs_names == declared_symbols()
    shim = open(os.path.join(ROOT, "rust", "zkhip", "src", "lib.rs")).read()
    used = set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", shim))
    assert used <= set(rs_names) and {"zkhip_msm", "zkhip_ntt", "zkhip_evaluate_h", "zkhip_coeff_to_extended"} <= used
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "zkhip.h")).read(), flags=re.S)

    def c_fields(struct):
        end = header.index("} " + struct + ";")
        body = header[header.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                out.append(re.findall(r"([A-Za-z_0-9]+)\s*(?:\[\d+\])?\s*$", part.strip())[0])
        return out

    def rs_fields(struct):
        body = re.search(r"pub struct " + struct + r" \{(.*?)\n\}", rs, flags=re.S).group(1)
        return re.findall(r"pub ([a-z_0-9]+):", body)
    for struct in ("zkhip_graph", "zkhip_lookup", "zkhip_evalh_desc"):
        assert rs_fields(struct) == c_fields(struct), struct


def test_struct_layout_matches_header():
    # sizes computed from the C declarations (LP64): catches a drifted ctypes mirror
    assert C.sizeof(ffi.Graph) == 48
    assert C.sizeof(ffi.Lookup) == 48 + 24
    assert C.sizeof(ffi.EvalhDesc) == 6 * 4 + 4 * 8 + 4 * 32 + 3 * 8 + 8 + 8 + 8 + 8 + 3 * 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8
    # and what the C compiler makes of the header itself
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc"):
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "sz.c")
            with open(src, "w") as f:
                f.write('#include <stdio.h>\n#include "%s"\nint main(void) { printf("%%zu %%zu %%zu\\n", sizeof(zkhip_graph), '
                        'sizeof(zkhip_lookup), sizeof(zkhip_evalh_desc)); return 0; }\n' % os.path.join(ROOT, "include", "zkhip.h"))
            subprocess.check_call(["gcc", src, "-o", os.path.join(tmp, "sz")])
            sizes = [int(v) for v in subprocess.check_output([os.path.join(tmp, "sz")]).split()]
        assert sizes == [C.sizeof(ffi.Graph), C.sizeof(ffi.Lookup), C.sizeof(ffi.EvalhDesc)]


def test_code_object_targets_gfx950():
    data = open(ffi.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"ntt_pass_kernel" in data and b"accumulate_kernel" in data and b"evalh_interp_kernel" in data


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present; the refusal path is exercised on CPU-only hosts")
    with pytest.raises(zk.ZkhipError):
        zk.Context(0)
    # argument checks that need no device
    lib = ffi.lib()
    assert lib.zkhip_create(C.c_int(0), None) == -1
    assert lib.zkhip_g1_to_affine(None, None) == -1
    assert lib.zkhip_domain_k(None) == 0
