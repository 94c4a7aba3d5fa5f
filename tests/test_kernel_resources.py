"""Register and scratch use of the built kernels, read from the code objects inside libispa_hip.so.

A kernel whose register arrays end up in scratch memory still computes the right values, a hundred times
slower, so no parity test notices: this one reads the kernel descriptors' metadata (no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pose_estimation_amd", "libispa_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernel_metadata(tmp_path):
    tools = {t: os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")}
    for t, path in tools.items():
        if not os.path.exists(path):
            found = shutil.which(t)
            if not found:
                pytest.skip(f"{t} not available")
            tools[t] = found
    if not os.path.exists(LIB):
        pytest.skip("libispa_hip.so not built")
    fat = tmp_path / "fat.bin"
    subprocess.check_call([tools["llvm-objcopy"], f"--dump-section=.hip_fatbin={fat}", LIB, str(tmp_path / "copy.so")])
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in .hip_fatbin"
    kernels = {}
    for n, at in enumerate(starts):
        end = starts[n + 1] if n + 1 < len(starts) else len(blob)
        piece = tmp_path / f"bundle{n}.bin"
        piece.write_bytes(blob[at:end])
        elf = tmp_path / f"code{n}.elf"
        subprocess.check_call([tools["clang-offload-bundler"], "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={piece}", f"--output={elf}"])
        notes = subprocess.run([tools["llvm-readelf"], "--notes", str(elf)], check=True, capture_output=True,
                               text=True).stdout
        name, pending_lds = None, 0
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)", line)
            if not m:
                continue
            key, val = m.groups()
            if key == "group_segment_fixed_size":            # listed in front of the kernel's name
                pending_lds = int(val)
            elif key == "name" and val.startswith("_Z"):
                name = val
                kernels[name] = {"group_segment_fixed_size": pending_lds}
            elif name and key in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count"):
                kernels[name][key] = int(val)
    return kernels


def test_no_kernel_spills_or_uses_scratch(tmp_path):
    kernels = _kernel_metadata(tmp_path)
    assert len(kernels) > 100, f"only {len(kernels)} kernels found in the library"
    assert any("nn_filter_bf16_t16_kernel" in k for k in kernels) and any("zb_tile_kernel" in k for k in kernels)
    # every kernel of the library keeps its arrays in registers: no scratch (private segment), no spilled vector
    # registers (scalar registers parked in lanes of a vector register cost next to nothing and are allowed)
    bad = {k: v for k, v in kernels.items()
           if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0)}
    assert not bad, f"kernels with scratch memory or spilled vector registers: {bad}"


def test_two_workgroups_per_cu_kernels_fit_their_registers(tmp_path):
    """The 128-channel filter kernels run two 4-wave workgroups per CU (two waves per SIMD: 256 registers per lane
    each); one register more and the second workgroup no longer fits, which halves the matrix-pipe utilisation
    without changing a result."""
    kernels = _kernel_metadata(tmp_path)

    def regs(fragment):
        hits = [v["vgpr_count"] for k, v in kernels.items() if fragment in k]
        assert len(hits) == 1, (fragment, len(hits))
        return hits[0]

    assert regs("nn_filter_bf16_t16_kernelILi4ELi4E") <= 256
    assert regs("nn_filter_split_kernelILi4E") <= 256
    assert regs("nn_filter_f32_qreg_kernelILi8ELi1E") <= 256
    assert regs("nn_filter_bf16_kernelILi8ELi5ELi2E") <= 256        # top-5, 128 channels
    # the float64 batched pass keeps two waves per SIMD as well
    assert regs("nn_exact_top1_kernelIfE") <= 256 and regs("nn_exact_top1_kernelItE") <= 256


def test_tiled_zbuffer_keeps_two_workgroups_per_cu(tmp_path):
    """zb_tile_kernel overlaps the phases of two resident workgroups; its static LDS (key tile, chunk list, pixel
    cache) must leave room for two in a CU's 160 KiB, and its 512-thread workgroups for two in the register file."""
    kernels = _kernel_metadata(tmp_path)
    hits = {k: v for k, v in kernels.items() if "zb_tile_kernel" in k}
    assert len(hits) == 2, list(hits)
    for name, meta in hits.items():
        assert 0 < meta["group_segment_fixed_size"] <= 160 * 1024 // 2, (name, meta)
        assert meta["vgpr_count"] <= 128, (name, meta)
