"""Compile-time guards on the gfx950 kernels (hipcc cross-compiles here): the MFMA conv kernels must
not spill to scratch memory (a silent 20 % slowdown when an array was demoted to private memory)
and must leave room for two 4-wave blocks per CU (<= 256 VGPRs)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_conv_kernels_have_no_scratch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "sbb_textline_detection_amd", "csrc", "kernels.hip")
    asm = tmp_path / "k.s"
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(asm),
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    text = asm.read_text()
    blocks = re.split(r"remark: [^\n]*Function Name: ", res.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if not any(k in name for k in ("conv_igemm_mfma", "stem_conv_pairs", "dec_tail_fused", "conv3x3_c64_direct", "bottleneck_fused")):
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgprs = int(re.search(r" VGPRs: (\d+)", b).group(1))
        spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        # no VGPR spills, and no instruction that touches private memory: the backend may still reserve a few dozen bytes
        # of frame for SGPR-spill bookkeeping that the final code never accesses (seen: 36-68 B) -- an array demoted to
        # private memory would show up as scratch_/buffer accesses and a bigger frame
        body = re.search(r"^%s:.*?\n(.*?)\n\s*s_endpgm" % re.escape(name), text, re.S | re.M).group(1)
        touches = [ln.strip() for ln in body.split("\n") if re.match(r"\s*(scratch_|buffer_(load|store)_dword[^\n]*\bs\[0:3\])", ln)]
        assert spill == 0 and not touches and scratch <= 128, f"{name}: {spill} VGPR spills, frame {scratch} B/lane, private accesses {touches[:3]}"
        assert vgprs <= 256, f"{name}: {vgprs} VGPRs"
    assert seen >= 6


def _asm_of(hipcc, src, tmp_path):
    asm = tmp_path / (os.path.basename(src) + ".s")
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(asm),
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    return asm.read_text(), res.stderr


def _kernel_bodies(text):
    """{mangled kernel name: its lines} of a device assembly listing"""
    out, name, body = {}, None, []
    for ln in text.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if ln.startswith(".Lfunc_end"):
            out[name] = body
            name = None
            continue
        body.append(ln)
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
@pytest.mark.parametrize("source,hand_counted", [("stem_pool_x3.hip", True), ("dec_halo_x3.hip", True), ("dec_halo_f16.hip", True),
                                                  ("expand_reduce_x3.hip", True), ("conv3_expand_reduce.hip", True), ("block_x3.hip", False)])
def test_tile_loops_are_not_drained_by_the_compiler(tmp_path, source, hand_counted):
    """The round-4 kernels keep loads in flight across MFMA phases; hipcc's own s_waitcnt insertion drains them (vmcnt(0) inside the
    tile loop) when a one-time load is first used inside the loop, when loads are pending at the loop entry, around conditional
    vector-memory operations and behind the LDS-DMA builtin (DESIGN.md 'The compiler's waits').  Guard: no `s_waitcnt vmcnt(0)`
    of the COMPILER's sits in a loop that holds MFMAs -- in the hand-counted kernels (all loop loads / stores in inline asm) no
    compiler vmcnt wait at all -- and nothing touches scratch memory (a scratch access is a vector-memory operation the counts miss)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    text, remarks = _asm_of(hipcc, os.path.join(ROOT, "sbb_textline_detection_amd", "csrc", source), tmp_path)
    bodies = _kernel_bodies(text)
    assert bodies, source
    checked = 0
    for name, body in bodies.items():
        if "block_x3ILb0ELb0E" in name:
            continue                                        # the in-place A/B form of the identity block (conv variant bit 20) is not the default
        # pass 1: which loops (by the header label hipcc prints on every block) contain MFMAs -- the tile loops; the one-time copy loops in
        # front of them (weights / constants to LDS) may wait as they like.  (The compiler rotates loops: layout order is not program order.)
        header, mfma_loops = None, set()
        for ln in body:
            if ln.startswith(".LBB"):
                m = re.search(r"Header=(BB\d+_\d+)", ln)
                header = m.group(1) if m else (ln.split(":")[0].lstrip(".L") if "Loop Header" in ln else None)
            if "v_mfma" in ln and header:
                mfma_loops.add(header)
        assert mfma_loops, name
        in_asm, depth, header = False, 0, None
        for ln in body:
            if "ASMSTART" in ln:
                in_asm = True
            elif "ASMEND" in ln:
                in_asm = False
            if ln.startswith(".LBB"):
                m = re.search(r"Depth=(\d+)", ln)
                depth = int(m.group(1)) if m else 0
                m = re.search(r"Header=(BB\d+_\d+)", ln)
                header = m.group(1) if m else (ln.split(":")[0].lstrip(".L") if "Loop Header" in ln else None)
            assert "scratch_" not in ln, f"{name}: {ln.strip()}"
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", ln)
            if m and not in_asm and depth >= 1 and (header in mfma_loops or depth >= 2):
                assert not hand_counted, f"{name}: compiler wait inside the hand-counted loop: {ln.strip()}"
                assert int(m.group(1)) != 0, f"{name}: the compiler drains the queue inside the tile loop: {ln.strip()}"
        checked += 1
    assert checked >= 1
    for b in re.split(r"remark: [^\n]*Function Name: ", remarks)[1:]:
        if "block_x3ILb0ELb0E" in b.split()[0]:
            continue
        assert int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0, b.split()[0]
