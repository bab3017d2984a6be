"""The one-wave-per-SIMD forward (fa_fwd_w4_gfx950.hip) names its registers literally; hipcc must stay out of them and must
not spill.  tools/audit_w4.py compiles the file to assembly (no GPU needed) and checks that: no scratch, no VGPR spill, no
compiler-made v_accvgpr_* / scratch_* instruction, no compiler instruction on a VGPR at or above the generator's budget,
and a plain tile step free of v_readlane / v_writelane / s_waitcnt vmcnt (a spill reload's vmcnt(0) would drain the LDS-DMA
queue in the middle of the tile loop).  The generated streams must also be what the generator writes today."""
import os
import subprocess
import sys

import pytest

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


def test_hazard_lint_catches_what_it_is_for():
    """audit_w4.lint_blocks on hand-made statements: the three hazards inline asm must keep by itself are flagged, a legal
    order is not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_w4 as a
    bad = a.lint_blocks([
        ["v_exp_f32 v52, v52", "v_add_f32 v56, v56, v52"],
        ["v_mfma_f32_32x32x16_bf16 v[64:79], a[200:203], a[132:135], v[64:79]", "v_fma_f32 v52, v64, s17, v61"],
        ["s_add_u32 m0, s48, 0", "buffer_load_dwordx4 v4, s[24:27], s77 offen lds"],
        ["v_cvt_pk_bf16_f32 v144, v52, v53", "v_mfma_f32_32x32x16_bf16 a[0:15], v[192:195], v[144:147], a[0:15]"],
    ])
    assert len(bad) == 4 and "v_exp_f32" in bad[0] and "v_mfma" in bad[1] and "M0" in bad[2] and "v_cvt_pk" in bad[3]
    ok = a.lint_blocks([
        ["v_exp_f32 v52, v52", "v_exp_f32 v53, v53", "v_add_f32 v56, v56, v52",
         "v_mfma_f32_32x32x16_bf16 v[64:79], a[200:203], a[132:135], v[64:79]", "v_fma_f32 v54, v100, s17, v61",
         "s_add_u32 m0, s48, 0", "v_add_f32 v57, v57, v53", "buffer_load_dwordx4 v4, s[24:27], s77 offen lds"],
    ])
    assert ok == []


def test_audit_flags_compiler_code_in_the_literal_scalar_registers(tmp_path):
    """the embedded-request forward keeps descriptors, cursors and the LDS base in literal s[NS:NS+13] across statements; only
    amdgpu_num_sgpr keeps hipcc below NS.  The audit must flag a compiler-made instruction that touches s >= NS -- and must not flag the
    statements themselves."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_w4 as a
    ns = a.generator_ns()
    name = "_ZN8aule_hip12_GLOBAL__N_121fa_fwd_w4_kernel_d128INS_10Bf16TraitsELb1ELb0EEEvNS0_11FwdW4ParamsE"
    body = ["s_mov_b32 s%d, 0" % (ns + 2), ";;#ASMSTART", "s_add_u32 m0, s%d, 1024" % (ns + 12), "v_mfma_f32_32x32x16_bf16 a[0:15], a[128:131], a[192:195], a[0:15]",
            ";;#ASMEND", "s_add_i32 s%d, s%d, 1" % (ns - 1, ns - 2), "s_load_dwordx4 s[%d:%d], s[0:1], 0x0" % (ns - 2, ns + 1), "s_endpgm"]
    text = name + ":\n" + "\n".join("\t" + l for l in body) + "\n  - .name:           " + name + "\n    .private_segment_fixed_size: 0\n    .vgpr_spill_count: 0\n"
    f = tmp_path / "fake.s"
    f.write_text(text)
    probs = a.audit(str(f), verbose=False)
    hits = [p for p in probs if ">= NS" in p]
    assert len(hits) == 2 and ("s%d" % (ns + 2)) in hits[0] and ("s%d" % (ns + 1)) in hits[1], probs
    assert not any("m0, s%d" % (ns + 12) in p for p in probs)


@pytest.mark.parametrize("gen,env,inc,hip", [("gen_bw4.py", "BW4_OUT", "fa_bwd_dkv4_asm.inc", "fa_bwd_dkv4_gfx950.hip"),
                                              ("gen_dq4.py", "DQ4_OUT", "fa_bwd_dq4_asm.inc", "fa_bwd_dq4_gfx950.hip")])
def test_backward_streams_current_and_kernels_clean(tmp_path, gen, env, inc, hip):
    """The one-wave-per-SIMD backward kernels (dK/dV: fa_bwd_dkv4_gfx950.hip, streams from tools/gen_bw4.py; dQ: fa_bwd_dq4_gfx950.hip,
    tools/gen_dq4.py): the committed streams are what the generator writes, and the compiled kernels have no scratch, no spills, no
    compiler-made accumulator access and stay below the generator's VGPR budget outside the statements."""
    import re
    out = tmp_path / inc
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen)], env=dict(os.environ, **{env: str(out)}), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_text() == open(os.path.join(ROOT, "aule-attention_amd", "csrc", inc)).read(), f"{inc} is stale: run python tools/{gen}"
    # hipcc's VGPR budget per head dim (the struct of every (type, D) states its NV; the D = 64 kernels carry _d64 in their names)
    nvs = {int(d): int(n) for d, n in re.findall(r"struct \w+<\w+, (\d+)> \{\n    static constexpr int NV = (\d+)", out.read_text())}
    assert set(nvs) == {64, 128}, nvs
    kt0 = {128: 224, 64: 240}            # dQ kernel: first register of the transposed K fragments
    s = tmp_path / "kernel.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-inline-asm", "-S", "--cuda-device-only", "-o", str(s),
                        os.path.join(ROOT, "aule-attention_amd", "csrc", hip)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = s.read_text()
    # dK/dV: every (dtype, D, causal) instance twice -- plain and SPILL (the 5-matmul backward) -- plus (round 6) the four D = 64 instances with two key blocks per wave
    nk = 20 if gen == "gen_bw4.py" else 8
    if gen == "gen_bw4.py":
        nv2 = {int(n) for n in re.findall(r"struct Bw4Asm2<\w+> \{\n    static constexpr int NV = (\d+)", out.read_text())}
        assert nv2 == {40}, nv2
    assert len(re.findall(r"\.private_segment_fixed_size: 0\n", text)) == nk and ".private_segment_fixed_size: " in text
    for key in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
        assert set(re.findall(r"\." + key + r":\s+(\d+)", text)) == {"0"}, key
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_w4 as a
    # the code of each kernel: from its label to s_endpgm
    kernels = re.findall(r"^(_ZN\S*kernel\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    assert len(kernels) == nk, [k for k, _ in kernels]
    if gen == "gen_bw4.py":
        # a SPILL instance = its plain twin + two buffer_store_dwordx4 per iteration statement (the packed dS registers)
        nst = sorted(body.count("buffer_store_dwordx4") for _, body in kernels)
        assert nst[:12] == [0] * 12 and all(n > 0 and n % 2 == 0 for n in nst[12:]), nst
    nits = 0
    for name, body in kernels:
        D = 64 if "_d64" in name else 128
        nv = 40 if "_d64k2" in name else nvs[D]      # (two key blocks per wave: the D = 128 budget)
        inasm, problems, blocks, cur = False, [], [], []
        for l in body.split("\n"):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                inasm, cur = True, []
            elif t.startswith(";;#ASMEND"):
                inasm = False
                blocks.append(cur)
            elif inasm:
                cur.append(t)
            elif t and not t.startswith(";") and not t.startswith("."):
                if "v_accvgpr" in t or t.startswith("scratch_"):
                    problems.append(t)
                for x in re.findall(r"\bv(\d+)\b", t) + [y for p, q in re.findall(r"\bv\[(\d+):(\d+)\]", t) for y in (p, q)]:
                    if int(x) >= nv:
                        problems.append(t)
        assert not problems, (name, problems[:5])
        assert a.lint_blocks(blocks) == [], name
        if gen == "gen_dq4.py":
            # the dQ kernel's iterations are single statements with generator-inserted counted LDS waits: re-derive them from the
            # compiled text (statements with MFMAs; at most the four fragment reads of the previous statement out at entry)
            its = [b for b in blocks if any(t.startswith("v_mfma") for t in b)]
            nits += len(its)
            for b in its:
                assert a.lint_lds_waits(b, entry_pending=4) == [], a.lint_lds_waits(b, entry_pending=4)[:3]
                # ... and the one thing counts cannot see: a transposed K fragment (v224 .. / v240 ..) is requested BEFORE the MFMA that reads it
                seen = set()
                for t in b:
                    parts = t.replace(",", " ").split()
                    if parts and parts[0].startswith("ds_read"):
                        seen |= a._regs(parts[1])
                    elif parts and parts[0].startswith("v_mfma"):
                        need = {r for r in a._regs(parts[2]) if r[0] == "v" and r[1] >= kt0[D]}
                        assert need <= seen, (t, sorted(need - seen)[:4])
    assert gen != "gen_dq4.py" or nits >= 60


def test_lds_wait_lint_catches_a_missing_wait():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_w4 as a
    good = ["ds_read_b128 v[56:59], v1 offset:0", "ds_read_b128 v[48:51], v1 offset:8704", "s_waitcnt lgkmcnt(1)",
            "v_mfma_f32_32x32x16_bf16 v[160:175], v[56:59], a[128:131], 0", "s_waitcnt lgkmcnt(0)",
            "v_mfma_f32_32x32x16_bf16 v[96:111], v[48:51], a[192:195], 0"]
    assert a.lint_lds_waits(good) == []
    bad = [good[0], good[1], good[2], good[3], good[5]]          # the second read is never waited for
    assert a.lint_lds_waits(bad) and "v48" in a.lint_lds_waits(bad)[0]
    late = ["v_mfma_f32_32x32x16_bf16 a[0:15], v[224:227], v[64:67], a[0:15]", "ds_read_b64_tr_b16 v[224:225], v2 offset:0"]
    assert a.lint_lds_waits(late) == []                          # (a read BEHIND its would-be reader is a placement bug the counts cannot see ...)
    assert a.lint_lds_waits(["ds_read_b64_tr_b16 v[224:225], v2 offset:0"] + late[:1])   # ... this order is what they catch
