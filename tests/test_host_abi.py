"""CPU-side checks of the boundary: the C-ABI library loads, exports every
symbol include/unflow_hip.h declares (no compute calls without a GPU), host-only
entry points behave like the reference's OP_REQUIRES checks, and the product
path refuses to run without the HIP library or on CPU tensors."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "unflow_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(unflow_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from unflow_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libunflow_hip.so does not export " + s
    # and the Python signature table covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms


def test_no_cuda_or_compat_symbols_in_sources():
    csrc = os.path.join(ROOT, "unflow_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            text = open(os.path.join(csrc, fn)).read()
            assert "__HIP_PLATFORM" not in text and "cuda_runtime" not in text and "hipify" not in text, fn


def test_geometry_and_errors_host_only():
    from unflow_amd import ops
    shp = ops.correlation_output_shape((8, 256, 48, 160))
    assert shp == {"correlation": (8, 441, 48, 160), "padded": (8, 88, 200, 256)}
    assert ops.correlation_output_shape((1, 1, 4, 4), kernel_size=1, max_displacement=0, pad=0,
                                        stride_2=1)["correlation"] == (1, 1, 4, 4)
    # stride_1 = 2 : ceil((H + 2p - 2*border)/s1)
    assert ops.correlation_output_shape((1, 3, 11, 13), 3, 4, 3, 2, 2)["correlation"] == (1, 25, 4, 5)
    with pytest.raises(ValueError, match="kernel_size must be odd"):
        ops.correlation_output_shape((1, 1, 4, 4), kernel_size=2)
    with pytest.raises(ValueError, match="Invalid correlation settings"):
        ops.correlation_output_shape((1, 1, 4, 4), max_displacement=20, pad=0)


def test_ops_refuse_cpu_tensors():
    from unflow_amd import losses, ops
    from unflow_amd.image_warp import image_warp
    im, fl = torch.zeros(1, 4, 4, 3), torch.zeros(1, 4, 4, 2)
    for fn, args in ((ops.backward_warp, (im, fl)), (image_warp, (im, fl)), (ops.forward_warp, (fl,)),
                     (ops.downsample, (im, 2)), (ops.correlation, (torch.zeros(1, 2, 4, 4),) * 2),
                     (losses.ternary_loss, (im, im, torch.ones(1, 4, 4, 1)))):
        with pytest.raises(RuntimeError, match="no CPU kernels"):
            fn(*args)


def test_argument_validation_before_any_launch():
    from unflow_amd import ops
    # validation that needs no device work happens on the host first
    with pytest.raises(TypeError):
        ops.backward_warp([[0.0]], torch.zeros(1, 1, 1, 2))


def test_round4_entry_points_validate_on_the_host():
    """unflow_downsample_pyramid, unflow_scale_stack_* and unflow_fb_occlusion_*losses reject bad arguments before any
    device work (plain C calls with null / dummy pointers: nothing is launched), with the reference's message where the
    reference has one (ops/downsample_op.cc:37-40)."""
    import ctypes
    from unflow_amd import _lib, ops
    lib = _lib.load()
    P = ctypes.c_void_p
    one = (P * 5)(*[P(16)] * 5)
    cases = [
        (lambda: lib.unflow_downsample_pyramid(P(16), None, one, None, 6, 1, 64, 64, 3, None), "1 to 5 levels"),
        (lambda: lib.unflow_downsample_pyramid(P(16), None, one, None, 2, 1, 64, 64, 2, None), "1 or 3 channels"),
        (lambda: lib.unflow_downsample_pyramid(P(16), None, one, None, 3, 1, 64, 40, 3, None), "divisible by scale"),
        (lambda: lib.unflow_downsample_pyramid(P(20), None, one, None, 1, 1, 64, 64, 3, None), "16-byte aligned"),
        (lambda: lib.unflow_downsample_pyramid(P(16), None, one, one, 1, 1, 64, 64, 3, None), "needs two tensors"),
        (lambda: lib.unflow_scale_stack_fwd(one, one, one, one, (ctypes.c_longlong * 1)(8), (ctypes.c_float * 1)(1.0), 0, None), "1 to 8 levels"),
        (lambda: lib.unflow_scale_stack_fwd(one, one, one, one, (ctypes.c_longlong * 1)(7), (ctypes.c_float * 1)(1.0), 1, None), r"\[B,h,w,2\]"),
        (lambda: lib.unflow_scale_stack_bwd(one, None, one, (P * 1)(None), (ctypes.c_longlong * 1)(8), (ctypes.c_float * 1)(1.0), 1, None), "null tensor"),
        (lambda: lib.unflow_fb_occlusion_losses(P(16), P(16), None, None, P(16), P(16), None, P(16), 0, None, P(16), 0, 1, 4, 4, None), "are required"),
        (lambda: lib.unflow_fb_occlusion_warp_losses(P(16), P(16), P(16), None, P(16), P(16), None, None, 0, None, P(16), 0, 1, 4, 4, None), "are required"),
    ]
    for call, pattern in cases:
        with pytest.raises(ValueError, match=pattern):
            _lib.check(call())
    assert lib.unflow_downsample_pyramid(P(16), None, one, None, 1, 0, 64, 64, 3, None) == 0        # an empty batch: nothing to do
    with pytest.raises(RuntimeError, match="no CPU kernels"):
        ops.downsample_pyramid(torch.zeros(1, 64, 64, 3))


def test_missing_library_fails_loudly(tmp_path):
    """Importing works, the first use raises: never a silent fallback."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import unflow_amd._lib as L\n"
        "L.LIB_PATH = %r\n"
        "L._lib = None\n"
        "try:\n"
        "    L.load()\n"
        "except ImportError as e:\n"
        "    print('RAISED', e)\n" % (ROOT, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "RAISED" in out.stdout and "no CPU fallback" in out.stdout, out.stdout + out.stderr


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "unflow_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in text and "from oracle" not in text, fn


def test_downsample_dispatcher_mirrors_reference():
    from unflow_amd import util
    with pytest.raises(RuntimeError, match="no CPU kernels"):
        util.downsample(torch.zeros(1, 5, 4, 1), 2)     # odd extent: resize_area (util.py:28), also HIP only
    with pytest.raises(ValueError, match="must be positive"):
        util.resize_area(torch.zeros(1, 5, 4, 1), (0, 2))
    with pytest.raises(RuntimeError, match="no CPU kernels"):
        util.downsample(torch.zeros(1, 4, 4, 1), 2)     # even extent reaches the op


def test_matrix_kernels_keep_their_accumulators_in_registers(tmp_path):
    """A lambda that captures the accumulator array of the correlation kernels and picks a ring slot at
    run time makes the compiler keep the whole array in scratch memory (352 B per lane, 10x slower --
    it happened twice in round 2 and only cycle stamps showed it).  Compile the matrix-core TUs for gfx950 with
    the resource-usage remarks on and hold every kernel to a few spilled scalars at most."""
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    names, scratch, vgprs = [], [], []
    for tu in ("correlation_mfma", "correlation_bwd4"):
        src = os.path.join(ROOT, "unflow_amd", "csrc", tu + ".hip")
        out = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Wno-pass-failed", "-Wno-inline-asm",
                              "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", str(tmp_path / "x.o"), src],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        names += re.findall(r"Function Name: (\S+)", out.stderr)
        scratch += [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
        vgprs += [int(v) for v in re.findall(r" VGPRs: (\d+)", out.stderr)]
    assert len(names) == len(scratch) == len(vgprs) >= 8, names
    worst = {n: s for n, s in zip(names, scratch) if s > 0}
    assert all(s <= 128 for s in worst.values()), worst
    # the shipped backward keeps its accumulators in fixed registers behind clobber lists: the compiler must
    # count them (256 VGPRs, two waves per SIMD) and keep its own values out of scratch
    b4 = [(v, s) for n, v, s in zip(names, vgprs, scratch) if "bwd4" in n]
    assert b4 == [(256, 0), (256, 0)], b4


@pytest.mark.parametrize("gen_name,inc_name", [("gen_corr_bwd4_asm.py", "corr_bwd4_asm.inc"), ("gen_corr_fwd_asm.py", "corr_fwd2_asm.inc")])
def test_the_instruction_blocks_are_what_the_generator_writes(tmp_path, gen_name, inc_name):
    """unflow_amd/csrc/corr_bwd4_asm.inc and corr_fwd2_asm.inc are generated (tools/gen_corr_*_asm.py, no knob set) and
    committed: generator and include must agree."""
    import shutil
    gen = os.path.join(ROOT, "tools", gen_name)
    inc = os.path.join(ROOT, "unflow_amd", "csrc", inc_name)
    work = tmp_path / "repo"
    (work / "tools").mkdir(parents=True)
    (work / "unflow_amd" / "csrc").mkdir(parents=True)
    shutil.copy(gen, work / "tools" / gen_name)
    env = {k: v for k, v in os.environ.items() if not k.startswith("FWD2_")}
    subprocess.run([sys.executable, str(work / "tools" / gen_name)], check=True, capture_output=True, env=env)
    assert open(work / "unflow_amd" / "csrc" / inc_name).read() == open(inc).read()


def test_forward_instruction_stream_kernel_resources_and_m0_discipline(tmp_path):
    """corr_fwd2_kernel keeps accumulators and fragments in fixed registers behind one asm statement: the compiler must
    come out at 128 VGPRs (two workgroups per CU) with nothing in scratch.  And in both TUs with hand-written blocks
    every LDS-DMA instruction -- the blocks' own and the compiler's -- has its write of M0 in its own stretch of code
    (tools/check_store_hazard.py scan_m0: the blocks write M0 and the compiler does not model that)."""
    import importlib.util
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    total = 0
    for tu in ("correlation_fwd2", "correlation_bwd4", "correlation_mfma"):
        src = os.path.join(ROOT, "unflow_amd", "csrc", tu + ".hip")
        lst = str(tmp_path / (tu + ".s"))
        out = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Wno-pass-failed", "-Wno-inline-asm",
                              "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-S", "-o", lst, src],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        found, n = chk.scan_m0(open(lst).read().split("\n"), tu)
        assert found == [], "\n".join(found)
        total += n
        if tu == "correlation_fwd2":
            names = re.findall(r"Function Name: (\S+)", out.stderr)
            vgprs = [int(v) for v in re.findall(r" VGPRs: (\d+)", out.stderr)]
            scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
            assert [(v, s) for nm, v, s in zip(names, vgprs, scratch) if "corr_fwd2_kernel" in nm] == [(128, 0)], list(zip(names, vgprs, scratch))
    assert total > 100          # the scan saw the LDS-DMA instructions


def test_m0_scanner_flags_a_dma_behind_a_foreign_block():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    ok = ["_Zk:", "\ts_mov_b32 m0, s4", "\tv_add_u32 v1, v2, v3", "\tbuffer_load_dwordx4 v1, s[4:7], 0 offen lds",
          "\t;;#ASMSTART", "\ts_add_u32 m0, s9, 16", "\ts_nop 0", "\tbuffer_load_dwordx4 v2, s[4:7], s3 offen lds", "\t;;#ASMEND"]
    assert chk.scan_m0(ok, "t") == ([], 2)
    bad = ["_Zk:", "\ts_mov_b32 m0, s4", "\t;;#ASMSTART", "\ts_add_u32 m0, s9, 16", "\tbuffer_load_dwordx4 v2, s[4:7], s3 offen lds", "\t;;#ASMEND",
           "\tbuffer_load_dwordx4 v1, s[4:7], 0 offen lds"]
    found, n = chk.scan_m0(bad, "t")
    assert n == 2 and len(found) == 1 and "end of an asm block" in found[0]
    loop = ["_Zk:", "\ts_mov_b32 m0, s4", ".LBB0_1:", "\tbuffer_load_dword v1, s[4:7], 0 offen lds"]
    assert len(chk.scan_m0(loop, "t")[0]) == 1


def test_the_shipped_library_is_not_a_diagnostic_build():
    """unflow_build_flags() lists the UNFLOW_TUNE_* knobs the kernels were compiled with (tools/variants.sh builds
    timing-only variants, some with wrong results); the shipped build has none, and the loader refuses one that has."""
    import ctypes
    from unflow_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.unflow_build_flags.restype = ctypes.c_char_p
    assert lib.unflow_build_flags() == b""
    assert "UNFLOW_ALLOW_DIAGNOSTIC_BUILD" in open(_lib.__file__).read()
