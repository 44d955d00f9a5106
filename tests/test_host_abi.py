"""CPU: the C-ABI library loads and exports every symbol include/mcquic_hip.h declares; host-side logic."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mcquic_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mcquic_amd import _lib
    lib = _lib.load()
    declared = _declared()
    assert declared, "no declarations parsed"
    assert sorted(_lib.SYMBOLS) == declared
    for name in declared:
        assert getattr(lib, name) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (mcq_[a-z0-9_]+)", out))
    assert set(declared) <= exported
    assert lib.mcq_version().decode().startswith("mcquic_hip")


def test_size_queries_run_without_a_gpu():
    from mcquic_amd import _lib
    lib = _lib.load()
    # the operand stream exists once per tile height (128 / 64 / 32 rows = 4 / 2 / 1 bands of 64 lanes), each copy with
    # 32 zero tail steps (16 until ABI 8); 128 -> 128 3x3 has 576 k-steps
    def sections(cout, steps):
        return sum(((cout + 32 * b - 1) // (32 * b) * steps + 32) * 64 * b for b in (4, 2, 1))
    # 3x3 layers with 64 / 128 input channels and Cout % 16 == 0 (>= 32): + the 16-row copy of the small-launch kernel
    # (csrc/conv_t16.h), Cin / 4 channel quads x 9 taps = 288 k-steps of four channels per 16-row tile, no tail
    def small(cout, cin):
        return cout // 16 * (cin // 4) * 9 * 64
    assert lib.mcq_packed_conv_weight_floats(128, 128, 3) == sections(128, 576) + small(128, 128)
    assert lib.mcq_packed_conv_weight_floats(512, 128, 3) == sections(512, 576) + small(512, 128)
    assert lib.mcq_packed_conv_weight_floats(128, 256, 3) == sections(128, 1152)       # (256 input channels: not taken)
    assert lib.mcq_conv2d_small_launch(8, 128, 4, 4, 128, 3, 1, 0, 2) == 1             # two 8 x 128 x 4 x 4 problems: 128 tiles
    assert lib.mcq_conv2d_small_launch(8, 128, 8, 8, 128, 3, 1, 0, 4) == 0             # four 8 x 8 problems fill the chip: general kernel
    assert lib.mcq_conv2d_small_launch(1, 128, 24, 16, 128, 3, 1, 0x108, 4) == 1       # residual + twin epilogue, batch 1
    assert lib.mcq_conv2d_small_launch(1, 128, 24, 16, 128, 3, 1, 0x1, 1) == 0         # SiLU prologue: general kernel
    assert lib.mcq_conv2d_small_launch(8, 128, 4, 4, 128, 1, 1, 0, 1) == 0             # 1x1
    assert lib.mcq_packed_conv_weight_floats(128, 3, 3) == sections(128, 18)         # 2 channel pairs x 9 taps
    assert lib.mcq_packed_conv_weight_floats(128, 8, 1) == sections(128, 16)         # 1x1: 4 pairs padded to one 16-deep ring
    # <= 16 output channels, 3x3: + the 16-row copy of the image-head kernel (32 four-channel groups x 9 taps + 16 tail steps)
    assert lib.mcq_packed_conv_weight_floats(12, 128, 3) == sections(12, 576) + (32 * 9 + 16) * 64
    assert lib.mcq_packed_conv_weight_floats(128, 128, 5) == 0                   # unsupported kernel size
    assert lib.mcq_packed_codebook_floats(2, 8192, 64) == (2 * 64 * 32 + 8) * 256 + 2 * 65 * 256      # + the 8-step ring tail


def test_ms_ssim_host_side():
    """Window taps equal the oracle's float32 window (= the reference's _fspecial_gauss_1d); shape rules."""
    import ctypes
    from mcquic_amd import _lib
    from oracle import metrics_ref as M
    lib = _lib.load()
    buf = (ctypes.c_float * 11)()
    lib.mcq_ms_ssim_window(buf)
    assert list(buf) == M.gauss_window().tolist()
    assert lib.mcq_ms_ssim_workspace_bytes(1, 3, 160, 512) == 0          # sides must exceed 160 (metrics.py:163-166)
    assert lib.mcq_ms_ssim_workspace_bytes(1, 3, 512, 160) == 0
    nbytes = lib.mcq_ms_ssim_workspace_bytes(2, 3, 161, 161)
    assert nbytes > 0 and nbytes % 4 == 0
    # pooled pyramid of a 768x512 pair: 4 levels x 2 images of floats, plus partial sums and level results
    planes, pyr = 32 * 3, sum((768 >> l) * (512 >> l) for l in range(1, 5))
    assert lib.mcq_ms_ssim_workspace_bytes(32, 3, 768, 512) >= 2 * planes * pyr * 4
    assert lib.mcq_ms_ssim_u8(None, None, None, None, 1, 3, 256, 256, None) == _lib.MCQ_EINVAL
    assert lib.mcq_sqdiff_sum_u8(None, None, None, 10, 1, None) == _lib.MCQ_EINVAL


def test_invalid_arguments_return_einval():
    from mcquic_amd import _lib
    lib = _lib.load()
    assert lib.mcq_conv2d_f32(None, None) == _lib.MCQ_EINVAL
    d = _lib.ConvDesc()
    assert lib.mcq_conv2d_f32(d, None) == _lib.MCQ_EINVAL
    assert lib.mcq_add_f32(None, None, None, None, 4, None) == _lib.MCQ_EINVAL
    assert lib.mcq_vq_assign_f32(None, None, None, 1, 2, 64, 4, 4, 512, None) == _lib.MCQ_EINVAL


def test_cpu_tensors_are_rejected_not_silently_computed():
    from mcquic_amd import Compressor
    model = Compressor(8, 2, [32, 16, 8]).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.encode(torch.zeros(1, 3, 128, 128))


def test_api_surface_and_errors():
    from mcquic_amd import Compressor
    from mcquic_amd.modules.compressor import AlignedPadding
    model = Compressor(8, 2, [32, 16, 8])
    assert model.QuantizationParameter == "-1"
    model.QuantizationParameter = "2"
    assert model.QuantizationParameter == "2"
    assert [tuple(c.shape) for c in model.Codebooks] == [(2, 32, 4), (2, 16, 4), (2, 8, 4)]
    assert [tuple(f.shape) for f in model.NormalizedFreq] == [(2, 32), (2, 16), (2, 8)]
    assert float(model.CodeUsage) == 1.0
    assert model.eval()(torch.zeros(1, 3, 8, 8)) is None           # eval-mode forward returns None like the reference
    with pytest.raises(RuntimeError):
        model.encode(torch.zeros(3, 8, 8))
    with pytest.raises(RuntimeError):
        model._quantizer._entropyCoder._checkShape([])
    with pytest.raises(RuntimeError):
        model._quantizer._entropyCoder._checkShape([torch.zeros(1, 2, 4, 4), torch.zeros(1, 3, 2, 2)])
    pad = AlignedPadding()
    assert tuple(pad(torch.zeros(1, 3, 200, 136)).shape) == (1, 3, 256, 256)
    x = torch.arange(2 * 3 * 130 * 250, dtype=torch.float32).reshape(2, 3, 130, 250)
    from oracle import mcquic_ref as R
    assert torch.equal(pad(x), R.aligned_padding(x))
    assert pad(torch.zeros(1, 3, 768, 512)).shape[-2:] == (768, 512)


def test_mcq_container_round_trip_and_version_rules():
    from mcquic_amd.utils import CodeSize, File, FileHeader, ImageSize, versionCheck
    header = FileHeader("0.1.40", "2", CodeSize([2, 2, 2], [48, 24, 12], [32, 16, 8], [8192, 2048, 512]), ImageSize(768, 512, 3))
    f = File(header, [b"\x01\x02\x03", b"abc", b"\xff" * 10])
    blob = f.serialize()
    import msgpack
    doc = msgpack.unpackb(blob, raw=False)
    assert list(doc) == ["fileHeader", "contents"] and list(doc["fileHeader"]) == ["qp", "version", "codeSize", "imageSize"]
    g = File.deserialize(blob)
    assert g.FileHeader.QuantizationParameter == "2" and g.FileHeader.CodeSize.k == [8192, 2048, 512]
    assert g.Content == f.Content and g.FileHeader.ImageSize.Pixels == 768 * 512
    assert abs(g.BPP - 16 * 8 / (768 * 512)) < 1e-12 and g.size() == 16
    with pytest.raises(ValueError, match="too new"):
        versionCheck("0.2.0")
    with pytest.raises(ValueError, match="too new"):
        FileHeader("1.0.0", "2", header.codeSize, header.imageSize)
    with pytest.warns(UserWarning, match="Minor version mismatch"):
        assert versionCheck("0.0.9")
    with pytest.raises(ValueError):
        File.deserialize(msgpack.packb({"nope": 1}))


def test_oversized_planes_are_rejected_before_any_launch():
    """A per-image plane set of 2 GiB or more cannot be addressed through one buffer descriptor: MCQ_ETOOLARGE."""
    from mcquic_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.x = d.w_packed = d.y = 1                      # never dereferenced: the size check comes first
    d.N, d.Cin, d.H, d.W, d.Cout, d.ksize, d.stride = 1, 128, 4096, 1024, 128, 3, 1
    assert lib.mcq_conv2d_f32(d, None) == _lib.MCQ_ETOOLARGE
    # the limit includes the 8 channels the operand rings may read past the last one (their offsets are 32-bit too):
    # 120 channels of 4096 x 1024 floats are 1.875 GiB, with the over-read exactly 2 GiB
    d.Cin = 120
    assert lib.mcq_conv2d_f32(d, None) == _lib.MCQ_ETOOLARGE
    assert lib.mcq_vq_assign_f32(1, 1, 1, 1, 2, 64, 4096, 2048, 512, None) == _lib.MCQ_ETOOLARGE


def test_wgrad_rows_kernel_declines_other_shapes():
    from mcquic_amd import _lib
    lib = _lib.load()
    assert lib.mcq_conv2d_wgrad_nchw_workspace_floats(4, 128, 12, 16, 128) == 0      # H not a multiple of 8 (and > 512 pixels in all)
    assert lib.mcq_conv2d_wgrad_nchw_workspace_floats(4, 128, 16, 12, 128) == 0      # W not a multiple of 8
    assert lib.mcq_conv2d_wgrad_nchw_workspace_floats(2, 128, 12, 16, 128) == 1      # <= 512 pixels: the one-pass 16x16-tile kernel (csrc/wgrad_t16.h), no workspace
    assert lib.mcq_conv2d_wgrad_nchw_workspace_floats(8, 128, 4, 4, 128) == 1        # small map: no workspace
    assert lib.mcq_conv2d_wgrad_nchw_workspace_floats(8, 40, 4, 4, 24) == 1          # channels not in sixteens: the LDS lane-per-weight kernel
    assert lib.mcq_conv2d_wgrad1x1_nchw_workspace_floats(8, 128, 8, 8, 128) == 1
    assert lib.mcq_conv2d_wgrad_nchw_workspace_floats(64, 128, 256, 256, 128) == 0   # a tensor of 2 GiB
    assert lib.mcq_conv2d_wgrad_nchw_workspace_floats(8, 128, 16, 16, 128) > 1       # the strip walk and its partial sums


def test_winograd_instance_keeps_its_accumulators_to_itself():
    """The 128-row Winograd instance of conv_mfma_kernel (opt-in path) addresses its 256 accumulator registers a0..a255 by
    NUMBER from inline asm (csrc/conv_mfma.hip, `WASM`): correct only while the compiler's own code never touches an AGPR
    and never spills.  Checked on the disassembly of the built object: in that kernel every AGPR operand belongs to one of
    the three hand-written instruction forms, and there is no scratch access."""
    import re
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "mcquic_amd", "_obj", "conv_mfma.o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("needs the built object mcquic_amd/_obj/conv_mfma.o and the ROCm llvm tools")
    tmp = tempfile.mkdtemp()
    try:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        asm = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    bodies, cur = {}, None
    for line in asm.splitlines():
        if line.endswith(">:"):
            cur = line if re.search(r"conv_mfma_kernelILi(4ELi2ELi\dELi12E|1ELi4ELi\dELi16E)", line) else None
        elif cur:
            bodies.setdefault(cur, []).append(line.split("//")[0])
    assert len(bodies) == 14, f"expected the 7 + 7 hand-numbered Winograd instances (F(2,3) 128-row and F(2x2,3x3); one per epilogue flag set), found {len(bodies)}"
    ok = re.compile(r"^\s*(v_mfma_f32_32x32x2_f32 a\[\d+:\d+\], v\d+, v\d+, a\[\d+:\d+\]|v_accvgpr_read_b32 v\d+, a\d+|v_accvgpr_write_b32 a\d+, 0)\s*$")
    body = []
    for name, lines in bodies.items():
        assert len(lines) > 1000
        agpr = [ln for ln in lines if re.search(r"\ba(\d+|\[\d+:\d+\])", ln)]
        assert len(agpr) >= 32 + 256 + 256, name
        stray = [ln for ln in agpr if not ok.match(ln)]
        assert not stray, f"compiler-generated AGPR use in {name}: {stray[:3]}"
        assert not [ln for ln in lines if "scratch_" in ln], f"{name} spills"
        body += lines + ["s_nop 0", "s_nop 0", "s_nop 0"]
    # The hazard recogniser does not see inside inline asm: a VALU result consumed by the very next MFMA came out wrong on
    # the GPU (the input transform therefore runs one group ahead).  No MFMA source may be written by a VALU instruction in
    # the three instructions before it.
    ins = [ln.strip() for ln in body if ln.strip()]
    for i, ln in enumerate(ins):
        m = re.match(r"v_mfma_f32_32x32x2_f32 a\[\d+:\d+\], (v\d+), (v\d+),", ln)
        if not m:
            continue
        for back in (1, 2, 3):
            prev = ins[i - back]
            w = re.match(r"(v_\w+)\s+(v\d+)", prev)
            assert not (w and not prev.startswith(("v_mfma", "v_accvgpr")) and w.group(2) in m.groups()), \
                f"VALU result feeds an inline-asm MFMA {back} instruction(s) later: {prev!r} -> {ln!r}"


def test_models_pickle_and_abi_version_is_checked():
    """no local lambdas on the modules (torch.save(model), multiprocessing spawn); the binding refuses a library
    built from another revision of the header."""
    import pickle
    from mcquic_amd import Compressor, Neon, _lib
    assert pickle.loads(pickle.dumps(Compressor(8, 2, [32, 16, 8])))._qp == "-1"
    pickle.dumps(Neon(32, 256, [8, 4, 2, 2]))
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mcquic_hip.h")).read()
    declared = int(re.search(r"#define\s+MCQ_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.mcq_abi_version() == declared == _lib.ABI_VERSION
