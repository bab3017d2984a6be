"""The one-wave-per-SIMD forward (fa_fwd_w4_gfx950.hip) names its registers literally; hipcc must stay out of them and must
not spill.  tools/audit_w4.py compiles the file to assembly (no GPU needed) and checks that: no scratch, no VGPR spill, no
compiler-made v_accvgpr_* / scratch_* instruction, no compiler instruction on a VGPR at or above the generator's budget,
and a plain tile step free of v_readlane / v_writelane / s_waitcnt vmcnt (a spill reload's vmcnt(0) would drain the LDS-DMA
queue in the middle of the tile loop).  The generated streams must also be what the generator writes today."""
import os
import subprocess
import sys

from conftest import ROOT


def test_generated_streams_are_current(tmp_path):
    out = tmp_path / "fa_fwd_w4_asm.inc"
    env = dict(os.environ, W4_OUT=str(out))
    env.pop("W4_X", None)
    env.pop("W4_SCALE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_w4.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = open(os.path.join(ROOT, "aule-attention_amd", "csrc", "fa_fwd_w4_asm.inc")).read()
    assert out.read_text() == committed, "fa_fwd_w4_asm.inc is stale: run python tools/gen_w4.py"


def test_compiled_kernel_keeps_out_of_the_literal_registers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_w4.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("clean"), r.stdout[-4000:] + r.stderr[-2000:]
