"""Builds libzkhip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

Every HIP source goes through the assembler by hand (`_compile_trimmed`): the multiplier pins
its chains of dependent multiply-adds with EMPTY inline-asm statements (csrc/wide.hip.h: `chain`, `carry_out`, `veil`), and
the compiler's hazard recogniser, which cannot see into an asm statement, assumes that it wrote its output register with a
partial-register (dst_sel) write and puts one wait state (`s_nop 0`) in front of the next instruction that touches the
register -- 2 500 of the 6 500 instructions of the MSM's inner loop.  The statements are empty, so nothing was written and
the wait state protects nothing; `trim_wait_states` removes exactly those (an `s_nop 0` directly behind an empty asm
statement, between two plain vector-ALU instructions of a fixed list -- never in front of a DPP, lane-access or
transcendental instruction, whose real hazards the recogniser may have folded into the same `s_nop`) and the object is
assembled from the result.  Measured on an MI355X: accumulate_kernel 545 -> 516 us, a lone 2^19 MSM 0.831 -> 0.802 ms,
the 2^19 -> 2^21 coset NTT 0.251 -> 0.244 ms, evaluate_h 24.1 -> 23.2 ms, the MSM's merge 107 -> 98 us and a 2^19 batch
inversion 0.190 -> 0.172 ms once their multiplier uses the chain form too; every parity test passes on the trimmed code.
If any step of that route fails the source is compiled the plain way (with the wait states) and a warning says so.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzkhip.so")
SOURCES = ["api.hip", "ntt.hip", "domain.hip", "msm.hip", "evalh.hip", "poly_ops.hip", "lookup_permute.hip", "g1_fft.hip", "host_math.cpp"]
TRIMMED = ("msm.hip", "ntt.hip", "evalh.hip", "poly_ops.hip", "domain.hip", "api.hip", "g1_fft.hip", "lookup_permute.hip")
ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result",
          # the field multiplier is ~250 instructions inline: loops over a lane's eight NTT elements must still
          # unroll fully, or their register arrays fall into scratch memory
          "-mllvm", "-pragma-unroll-threshold=1000000"]

# plain vector-ALU instructions between which gfx950 has no data hazard (no DPP / SDWA / lane access / transcendental /
# carry-in forms): the producers and consumers around an empty asm statement in the multiplier and the carry passes
_PLAIN = {"v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_ashrrev_i64", "v_ashrrev_i32_e32",
          "v_lshrrev_b32_e32", "v_lshlrev_b32_e32", "v_lshl_add_u32", "v_lshl_add_u64", "v_lshl_or_b32", "v_and_b32_e32",
          "v_or_b32_e32", "v_or3_b32", "v_and_or_b32", "v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_add3_u32",
          "v_mov_b32_e32", "v_mov_b64_e32", "v_alignbit_b32", "v_bfe_u32", "v_bfe_i32"}
_MODIFIERS = ("dpp", "sdwa", "quad_perm", "row_", "op_sel", "wave_")


def _instruction(line: str):
    t = line.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        return None
    return t


def _is_plain(ins: str) -> bool:
    return ins.split()[0] in _PLAIN and not any(m in ins for m in _MODIFIERS)


def trim_wait_states(text: str):
    """(assembly without the wait states that only an empty asm statement asked for, how many were removed)"""
    lines = text.split("\n")
    keep = [True] * len(lines)
    removed = 0
    for i, line in enumerate(lines):
        if line.strip() != "s_nop 0" or i < 2:
            continue
        if lines[i - 1].strip() != ";;#ASMEND" or lines[i - 2].strip() != ";;#ASMSTART":
            continue
        # the producer: the instruction in front of the (possibly several) empty asm statements
        k = i - 3
        while k >= 1 and lines[k].strip() == ";;#ASMEND" and lines[k - 1].strip() == ";;#ASMSTART":
            k -= 2
        producer = _instruction(lines[k]) if k >= 0 else None
        consumer = _instruction(lines[i + 1]) if i + 1 < len(lines) else None
        if producer and consumer and _is_plain(producer) and _is_plain(consumer):
            keep[i] = False
            removed += 1
    return "\n".join(l for l, k in zip(lines, keep) if k), removed


def _llvm_tool(name: str) -> str:
    for base in ("/opt/rocm/lib/llvm/bin", "/opt/rocm/llvm/bin"):
        p = os.path.join(base, name)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


def _compile_trimmed(hipcc: str, src: str, obj: str, verbose: bool) -> int:
    """device code -> assembly -> trim -> code object -> bundle -> host object that embeds it (the steps hipcc runs for
    `-c`, with the trim between its compiler and its assembler); returns the number of wait states removed"""
    stem = obj[:-2]
    inc = ["-I", os.path.join(HERE, "..", "include")]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL if not verbose else None,
                       stderr=subprocess.DEVNULL if not verbose else None)

    run([hipcc, f"--offload-arch={ARCH}", *COMMON, "--cuda-device-only", "-S", "-x", "hip", src, "-o", stem + ".s", *inc])
    with open(stem + ".s") as f:
        text, removed = trim_wait_states(f.read())
    with open(stem + ".trim.s", "w") as f:
        f.write(text)
    run([_llvm_tool("clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", stem + ".trim.s", "-o",
         stem + ".dev.o"])
    run([_llvm_tool("lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", stem + ".hsaco",
         stem + ".dev.o"])
    run([_llvm_tool("clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}", "-input=/dev/null",
         f"-input={stem}.hsaco", f"-output={stem}.hipfb"])
    run([hipcc, f"--offload-arch={ARCH}", *COMMON, "--cuda-host-only", "-c", "-x", "hip", src, "-o", obj, *inc,
         "-Xclang", "-fcuda-include-gpubinary", "-Xclang", stem + ".hipfb"])
    for ext in (".s", ".trim.s", ".dev.o", ".hsaco", ".hipfb"):
        if not verbose and os.path.exists(stem + ext):
            os.remove(stem + ext)
    return removed


def _compile(hipcc: str, src: str, obj: str, verbose: bool):
    if os.path.basename(src) in TRIMMED and os.environ.get("ZKHIP_BUILD_PLAIN") != "1":
        try:
            removed = _compile_trimmed(hipcc, src, obj, verbose)
            if verbose:
                print(f"{os.path.basename(src)}: {removed} wait states behind empty asm statements removed", file=sys.stderr)
            return
        except (subprocess.CalledProcessError, OSError) as exc:
            print(f"[zkhip build] {os.path.basename(src)}: the trimmed route failed ({exc}); compiling it the plain way",
                  file=sys.stderr)
    cmd = [hipcc, f"--offload-arch={ARCH}", *COMMON, "-c", "-x", "hip", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)


def build(force: bool = False, verbose: bool = False) -> str:
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h",))]
    deps.append(os.path.join(HERE, "..", "include", "zkhip.h"))
    deps.append(os.path.abspath(__file__))
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    todo = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, deps):
            todo.append((s, o))
    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as pool:
            for s, fut in [(s, pool.submit(_compile, hipcc, s, o, verbose)) for s, o in todo]:
                try:
                    fut.result()
                except subprocess.CalledProcessError as exc:
                    raise RuntimeError(f"hipcc failed on {s}") from exc
    if force or todo or _newer(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
