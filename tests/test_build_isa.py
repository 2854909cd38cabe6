"""Build-time ISA check (CPU, needs only the built objects).

Round 4 traced wrong sums in the diffusion stage's timestep MLP (linear_nk_kernel) to `v_pk_fma_f32` executing while ANOTHER process's MFMA waves shared the
GPU (profiles/r4_two_process_determinism.txt) and rewrote that kernel without packed f32 arithmetic: this test keeps it that way. The other kernels keep their
packed forms in the default build — two engine processes on one GPU are an unsupported form, refused by the CLI and bench.py unless --allow-shared-device — and
`make PK="-Xclang -target-feature -Xclang -packed-fp32-ops"` builds the whole library without them (0.9 % slower, profiles/r5_packed_f32_ab.txt); with
TTS_EXPECT_NO_PACKED_F32=1 the test asserts that for every kernel. It disassembles the gfx950 code object embedded in every csrc/*.o (objcopy +
clang-offload-bundler + llvm-objdump, all part of the ROCm image)."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def device_disassembly(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    if os.path.getsize(fat) == 0:
        return ""
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def test_packed_f32_arithmetic_stays_out_of_the_time_mlp(tmp_path):
    if not (shutil.which("objcopy") and os.path.exists(LLVM + "/clang-offload-bundler") and os.path.exists(LLVM + "/llvm-objdump")):
        pytest.skip("binutils / ROCm llvm tools not present")
    objs = sorted(glob.glob(os.path.join(ROOT, "tortoise.cpp_amd", "csrc", "*.o")))
    if not objs:
        pytest.skip("library not built")
    kernels, mfma, seen_linear = 0, 0, False
    for obj in objs:
        asm = device_disassembly(obj, str(tmp_path))
        name, bad = None, {}
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                name = m.group(1)
                kernels += 1
            elif re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
                bad[name] = bad.get(name, 0) + 1
            elif "v_mfma_" in line:
                mfma += 1
        claimed = {k: v for k, v in bad.items() if "linear_nk_kernel" in (k or "")}
        assert not claimed, "packed f32 arithmetic is back in %s" % claimed
        if os.environ.get("TTS_EXPECT_NO_PACKED_F32"):
            assert not bad, "%s: packed f32 arithmetic in %s" % (os.path.basename(obj), sorted(bad.items(), key=lambda kv: -kv[1])[:8])
        seen_linear = seen_linear or "linear_nk_kernel" in asm
    assert seen_linear and kernels > 50 and mfma > 1000  # the disassembly really is the device code (gfx950 kernels with their MFMA bodies)


@pytest.mark.parametrize("defines,src", [(["-DTTS_DEBUG_CHECKSUM"], "diffusion.hip"), (["-DTTS_DEBUG_CHECKSUM", "-DTTS_DEBUG_NO_TIME_GUARD"], "diffusion.hip"),
                                         (["-DTTS_DEC_TRACE"], "ar.hip")])
def test_developer_ifdef_builds_still_compile(defines, src):
    """The trace / checksum #ifdef paths of csrc (developer builds: tools/build_debug_lib.sh, tools/dec_bench.hip) are not part of the product build, so nothing
    else keeps them compiling: host + device syntax check, a few seconds each."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    d = os.path.join(ROOT, "tortoise.cpp_amd")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Icsrc", "-I../include", "-fsyntax-only"] + defines + ["csrc/" + src],
                       cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
