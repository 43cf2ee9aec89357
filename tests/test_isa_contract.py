"""The hand-scheduled inline-asm parts of the kernels rely on two things the compiler does not check for us
(it neither tracks loads issued from inline asm nor inserts hazard wait states inside asm blocks).  This test
compiles the device code to assembly and verifies them on the generated ISA:

* FIR window reads (dsp_core.hpp: dh_lds_read2 / dh_fir_arrived): between an asm `ds_read2_b32` and the next
  `s_waitcnt lgkmcnt(0)` no instruction may touch the destination registers (a register-allocator copy or spill
  there would read data that has not arrived yet);
* every DPP instruction is at least two instructions (or an s_nop) away from the VALU write of its source;
* the exact kernels contain no fused multiply-add in the FIR (bit-exactness contract) and
  the headline kernel uses no scratch.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "engine.s")
    import __graft_entry__ as g
    flags = [f for f in g.HIP_FLAGS if f not in ("-fPIC", "-shared")]
    subprocess.run([HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(g.CSRC, "engine.hip"), "-o", out],
                   check=True, cwd=g.CSRC, stderr=subprocess.DEVNULL)
    text = open(out).read()
    ks = {}
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n(.*?)^\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        ks[m.group(1)] = (m.group(2).splitlines(), m.group(3))
    assert ks
    return ks


def _regs(operand_text):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(r) for r in re.findall(r"\bv(\d+)\b", operand_text))
    return regs


def _insts(lines):
    for l in lines:
        l = l.split(";")[0].strip()
        if l and not l.startswith(".") and not l.endswith(":"):
            yield l


def _insts_asm(lines):
    """(instruction, came from an inline-asm block) -- the compiler tracks its own loads; only asm ones need this test"""
    in_asm = False
    for l in lines:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        t = l.split(";")[0].strip()
        if t and not t.startswith(".") and not t.endswith(":"):
            yield t, in_asm


def test_fir_asm_loads_are_not_touched_before_their_wait(kernels):
    checked = 0
    for name, (lines, _) in kernels.items():
        if "k_chain" not in name and "k_rrc_demod" not in name and "k_rrc_tile" not in name:
            continue
        pairs = list(_insts_asm(lines))
        insts = [a for a, _ in pairs]
        pk = [i for i, l in enumerate(insts) if l.startswith("v_pk_mul_f32") or l.startswith("v_pk_fma_f32")]
        if not pk:
            continue
        lo, hi = pk[0] - 40, pk[-1] + 1
        pending = {}                                  # register -> index of the load that writes it
        for i in range(max(lo, 0), hi):
            l = insts[i]
            op, _, rest = l.partition(" ")
            if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
                pending.clear()
                continue
            touched = _regs(rest)
            bad = touched & set(pending)
            assert not bad, "%s: `%s` touches v%s while the ds_read2 at #%d is in flight" % (name, l, sorted(bad), pending[min(bad)])
            if op == "ds_read2_b32" and pairs[i][1]:
                for r in _regs(rest.split(",")[0]):
                    pending[r] = i
                checked += 1
    assert checked > 500                              # 80+ reads in each of the FIR kernels


def test_dpp_sources_have_their_wait_states(kernels):
    seen = 0
    for name, (lines, _) in kernels.items():
        insts = list(_insts(lines))
        for i, l in enumerate(insts):
            if "_dpp" not in l.split(" ")[0]:
                continue
            seen += 1
            ops = l.partition(" ")[2].split(",")
            src = _regs(ops[1])                       # src0 is the operand the DPP network reads
            wait = 0
            for back in range(i - 1, max(i - 3, -1), -1):
                p = insts[back]
                op, _, rest = p.partition(" ")
                if op == "s_nop":
                    wait += int(rest.strip()) + 1
                    continue
                if wait >= 2:
                    break
                if op.startswith("v_") and _regs(rest.split(",")[0]) & src:
                    raise AssertionError("%s: `%s` reads v%s %d wait state(s) after `%s`" % (name, l, sorted(src), wait, p))
                wait += 1
    assert seen > 20


def _blocks(lines):
    blocks, cur = [], []
    for l in lines:
        t = l.split(";")[0].strip()
        if t.endswith(":") and t.startswith(".LBB"):
            blocks.append(cur); cur = []
        elif t and not t.startswith("."):
            cur.append(t)
    blocks.append(cur)
    return blocks


def test_exact_kernels_keep_their_two_firs_apart_and_the_hot_one_in_registers(kernels):
    """The exact chain kernels (DMR, YSF, NXDN) carry two FIR bodies (dsp_core.hpp, "Error-bounded FIR"):
    * the reference's arithmetic -- packed multiplies and packed adds, rounded one by one: NOTHING fused in that block;
    * the error-bounded one, which runs in (nearly) every pass: no scratch access in that block.  For the wide filter
      (DMR, YSF) it is the split-f16 product on the matrix cores: 36 v_mfma_f32_16x16x32_f16 per pass and no f32 FIR
      arithmetic beside them; for the narrow filter (NXDN) packed FMAs.
    A handful of spills elsewhere (rare paths, the YSF decoder half) is what a fourth wavefront per SIMD costs."""
    exact = [n for n in kernels if "k_chain" in n and ("ILi80ELb0E" in n or "ILi160ELb0E" in n)]
    assert len(exact) == 8                    # DMR, YSF, NXDN (161 taps) at sps 20 and at a run-time sps, each as launch PART 0 and PART 1 (DH_FLAG_OVERLAP_PUSHES)
    for name in exact:
        lines, meta = kernels[name]
        narrow = "ILi160ELb0E" in name
        blocks = _blocks(lines)
        ref_fir = max(blocks, key=lambda b: sum(i.startswith("v_pk_mul_f32") for i in b))
        assert sum(i.startswith("v_pk_mul_f32") for i in ref_fir) >= 600 and sum(i.startswith("v_pk_add_f32") for i in ref_fir) >= 600     # 81 (161) taps x 8 pairs
        assert not [i for i in ref_fir if i.startswith(("v_pk_fma_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mfma"))], name + " fuses inside the reference FIR"
        hot = max(blocks, key=lambda b: sum(i.startswith("v_mfma_f32_16x16x32_f16") for i in b))
        assert sum(i.startswith("v_mfma_f32_16x16x32_f16") for i in hot) == (72 if narrow else 36) and hot is not ref_fir      # 4 tiles x 3 K-steps x 3 products (6 K-steps narrow)
        assert sum(i.startswith(("v_pk_fma_f32", "v_pk_mul_f32")) for i in hot) <= 40, name + ": f32 FIR arithmetic beside the MFMAs"
        assert not [i for b in blocks for i in b if i.startswith("v_mfma_f32_16x16x4_f32")]
        hot_spills = [i for i in hot if "scratch_" in i]
        runtime_sps = "ELi3ELi0E" in name                    # the narrow filter at a run-time sps (no pipe of the reference uses it): a few 16-byte spill pairs tolerated
        assert len(hot_spills) <= (8 if runtime_sps else 0), name + " spills inside the hot FIR"
        spills = [i for b in blocks for i in b if "scratch_" in i]
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        # (static counts: with the branch weights of the hot loop the allocator puts its spill code into the rare paths --
        # the reference-order FIR, the exact evaluations, the ordered timing chain -- where there is more of it than before)
        ysf = "ELi2ELi10E" in name                           # (template arguments NZ, FAST, PROTO, SPS: protocol 2 = YSF)
        assert len(spills) <= (144 if narrow else 128 if ysf else 112) and vgprs <= (168 if narrow else 128), (name, len(spills), vgprs)      # (static count over the whole kernel text, decoder half included: round 5's YSF decode-ahead has 114; what matters is hot_spills above and tools/asm_census.py)
