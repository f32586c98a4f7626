#!/usr/bin/env python3
"""tools/asm_cost.py -- static VALU cost breakdown of one kernel in a hipcc -S listing, by the issue-cost classes
measured with tools/micro/issue_rate.hip on MI355X (cycles one SIMD needs per wave64 instruction, >= 2 waves):
  fast  2.5   v_add/sub/mul/fma/fmac_f32, v_mov, v_and/or/xor, v_add/sub_u32, v_cndmask (vcc form), all-VGPR / inline / literal operands
  slow  4.25  the same with an SGPR source operand; v_min/max(3), v_mul_lo/hi, v_pk_*, v_lshl*, v_cvt*, v_rndne, v_bfe, v_mbcnt,
              v_mad_*, v_cndmask_b32_e64 (SGPR-pair mask), everything not listed elsewhere
  cmp   4.9   v_cmp* / v_cmpx*
  lane  5.0   v_readlane / v_readfirstlane / v_writelane
  trans 8.2   v_rcp / v_rsq / v_sqrt / v_sin / v_cos / v_exp / v_log
Static counts (every instruction of the function once, or only between two `; MARK` lines), not a dynamic profile.
    python3 tools/asm_cost.py tracer_amd/csrc/kernels.s k_queueILb1 [mark_from mark_to]"""
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_cndmask_b32", "v_mac_f32", "v_mad_f32"}
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_", "v_exp_", "v_log_")
COST = {"fast": 2.5, "slow": 4.25, "cmp": 4.9, "lane": 5.0, "trans": 8.2}
SGPR = re.compile(r"\b(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|ttmp\d+)\b")


def classify(line):
    toks = line.split(None, 1)
    op = toks[0]
    args = [a.strip() for a in toks[1].split(",")] if len(toks) > 1 else []
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith(("v_cmp", "v_cmpx")):
        return "cmp", "cmp"
    if base.startswith(TRANS):
        return "trans", "trans"
    if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
        return "lane", "lane"
    srcs = args[1:]
    if base == "v_cndmask_b32":
        if len(srcs) == 3 and SGPR.search(srcs[2]) and not srcs[2].startswith("vcc"):
            return "slow", "cndmask sgpr mask"
        if any(SGPR.search(a) for a in srcs[:2]):
            return "slow", "cndmask sgpr data"
        return "fast", "cndmask vcc"
    if base in FAST:
        if any(SGPR.search(a) for a in srcs):
            return "slow", "sgpr operand"
        if op.endswith("_dpp") or "row_" in line or "quad_perm" in line:
            return "slow", "dpp"
        return "fast", "fast"
    return "slow", base


def main():
    path, kern = sys.argv[1], sys.argv[2]
    m_from = sys.argv[3] if len(sys.argv) > 3 else None
    m_to = sys.argv[4] if len(sys.argv) > 4 else None
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if kern in l and l.startswith("_ZN") and l.rstrip().split(":")[0].endswith("E") or
             (kern in l and l.startswith("_ZN") and ":" in l and not l.startswith("\t"))][0]
    end = [i for i, l in enumerate(lines[start:]) if l.strip().startswith(".Lfunc_end")][0] + start
    on = m_from is None
    cls = {}; detail = {}
    for l in lines[start:end]:
        s = l.strip()
        mm = re.match(r"; MARK (\S+)", s)
        if mm:
            if m_from and mm.group(1) == m_from: on = True
            if m_to and mm.group(1) == m_to: on = False
        if not on or not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if not s.startswith("v_"):
            continue
        c, d = classify(s)
        cls[c] = cls.get(c, 0) + 1
        detail[d] = detail.get(d, 0) + 1
    tot = sum(cls.values()); cyc = sum(COST[c] * n for c, n in cls.items())
    print(f"{kern}: {tot} VALU instructions (static), {cyc:.0f} issue cycles at the measured class costs, {cyc / max(tot, 1):.2f} cycles / instruction")
    for c in ("fast", "slow", "cmp", "lane", "trans"):
        print(f"  {c:6s} {cls.get(c, 0):6d}  x {COST[c]:.2f} = {COST[c] * cls.get(c, 0):8.0f}")
    print("  detail:", ", ".join(f"{k} {v}" for k, v in sorted(detail.items(), key=lambda kv: -kv[1])[:24]))


if __name__ == "__main__":
    main()
