"""tools/e64.py (the VOP1 / VOP2 -> VOP3 re-encoder of the build experiments of round 3): which instructions it touches."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("e64", os.path.join(ROOT, "tools", "e64.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


ASM = """\
_ZN4lerf5fused9s1_kernelILb0EEEv: ; @s1
\tv_add_u32_e32 v1, v2, v3
\tv_add_u32_e32 v1, 0x276c0, v3
\tv_mul_f32_e32 v4, 2.0, v5
\tv_mul_f32_e32 v4, 0x4f7ffffe, v5
\tv_cndmask_b32_e32 v1, v2, v3, vcc
\tv_cvt_f32_ubyte0_e32 v8, v22
\tv_add_u32_sdwa v3, v1, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD
\tv_fma_f32 v3, v8, s87, 2.0
_ZN4lerf5other6kernelEv: ; @other
\tv_add_u32_e32 v1, v2, v3
"""


def test_simple_ops_are_reencoded_and_literals_left_alone():
    m = _load()
    stats = {}
    out = "".join(m.convert(ASM.splitlines(True), set(m.SIMPLE), None, stats))
    assert "v_add_u32_e64 v1, v2, v3" in out
    assert "v_add_u32_e32 v1, 0x276c0, v3" in out                 # a 32-bit literal has no VOP3 form on gfx9
    assert "v_mul_f32_e64 v4, 2.0, v5" in out                     # an inline constant has
    assert "v_mul_f32_e32 v4, 0x4f7ffffe, v5" in out
    assert "v_cndmask_b32_e32 v1, v2, v3, vcc" in out             # implicit VCC: left alone
    assert "v_cvt_f32_ubyte0_e32 v8, v22" in out                  # not in the simple set
    assert "v_add_u32_sdwa" in out and "v_fma_f32 v3, v8, s87, 2.0" in out
    assert stats == {"v_add_u32": 2, "v_mul_f32": 1}


def test_all_ops_and_the_kernel_filter():
    m = _load()
    stats = {}
    out = "".join(m.convert(ASM.splitlines(True), m.SIMPLE | m.COMPLEX, r"s1_kernel", stats))
    assert "v_cvt_f32_ubyte0_e64 v8, v22" in out
    assert out.rstrip().endswith("v_add_u32_e32 v1, v2, v3")       # the second kernel does not match the filter
    assert stats["v_add_u32"] == 1
