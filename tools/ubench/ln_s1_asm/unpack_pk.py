#!/usr/bin/env python3
"""Rewrite classes of v_pk_{add,mul,fma}_f32 in ONE kernel of a gfx950 assembly file into their two scalar VOP3 instructions.

    unpack_pk.py in.s out.s KERNEL_SUBSTRING CLASSES [FROM:TO]

CLASSES is a comma list out of
    add      v_pk_add_f32 without modifiers                      (the DPP reductions' sums, the running sums)
    addneg   v_pk_add_f32 with op_sel_hi + neg modifiers         (x - mean: the mean broadcast from the low half and negated)
    addsel   v_pk_add_f32 with op_sel + op_sel_hi, no neg        (the cross-half adds lo + hi' / hi + lo' of the running sums)
    addmod   both of the above
    mul      v_pk_mul_f32 on VGPR pairs                          (y * y, x * Fh)
    muls     v_pk_mul_f32 with an SGPR source                    (requotient: -(x * s))
    fmas     v_pk_fma_f32 with an SGPR source                    (requotient: fma(x, s, -(x * s)))
    fma      v_pk_fma_f32 on VGPR pairs                          (requotient: the correction step, op_sel_hi broadcast of 1 / s)
    all / none
FROM:TO (optional) restricts the rewrite to the FROM-th .. (TO-1)-th packed instruction of the kernel, counted over the classes
selected — the second level of the bisection.  Prints how many instructions of each class it met and how many it rewrote.
"""
import re
import sys


def parse_list(text, key, n, default):
    m = re.search(key + r":\[([01,]+)\]", text)
    return [int(v) for v in m.group(1).split(",")] if m else [default] * n


def elem(op, idx):
    """Register `idx` (0 / 1) of a 64-bit operand: v[a:b], s[a:b]; a literal / inline constant is the same in both halves."""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", op)
    if not m:
        return op
    return "%s%d" % (m.group(1), int(m.group(2)) + idx)


def classify(mn, srcs, mods):
    sgpr = any(s.startswith("s[") for s in srcs)
    if mn == "v_pk_add_f32":
        return ("addneg" if "neg_" in mods else "addsel") if mods else "add"
    if mn == "v_pk_mul_f32":
        return "muls" if sgpr else "mul"
    return "fmas" if sgpr else "fma"


def rewrite(mn, dst, srcs, mods):
    n = len(srcs)
    op_sel = parse_list(mods, r"op_sel", n, 0) if re.search(r"op_sel:\[", mods) else [0] * n
    op_sel_hi = parse_list(mods, r"op_sel_hi", n, 1)
    neg_lo = parse_list(mods, r"neg_lo", n, 0)
    neg_hi = parse_list(mods, r"neg_hi", n, 0)
    scalar = {"v_pk_add_f32": "v_add_f32_e64", "v_pk_mul_f32": "v_mul_f32_e64", "v_pk_fma_f32": "v_fma_f32"}[mn]
    dlo, dhi = elem(dst, 0), elem(dst, 1)
    lo_src = [("-" if neg_lo[i] else "") + elem(srcs[i], op_sel[i]) for i in range(n)]
    hi_src = [("-" if neg_hi[i] else "") + elem(srcs[i], op_sel_hi[i]) for i in range(n)]
    lo = "\t%s %s, %s" % (scalar, dlo, ", ".join(lo_src))
    hi = "\t%s %s, %s" % (scalar, dhi, ", ".join(hi_src))
    hi_reads_dlo = any(s.lstrip("-") == dlo for s in hi_src)
    lo_reads_dhi = any(s.lstrip("-") == dhi for s in lo_src)
    if not hi_reads_dlo:
        return [lo, hi]
    if not lo_reads_dhi:
        return [hi, lo]
    if mn != "v_pk_fma_f32" and sorted(lo_src) == sorted(hi_src):      # a + b and b + a: one value, both halves
        return [lo, "\tv_mov_b32_e32 %s, %s" % (dhi, dlo)]
    # a true cross (lo reads the old hi, hi reads the old lo): keep the old lo in the scratch register main() reserved
    hi_t = [("-" if t.startswith("-") else "") + SCRATCH if t.lstrip("-") == dlo else t for t in hi_src]
    return ["\tv_mov_b32_e32 %s, %s" % (SCRATCH, dlo), lo, "\t%s %s, %s" % (scalar, dhi, ", ".join(hi_t))]


SCRATCH = None


def main():
    global SCRATCH
    src, out, kernel, classes = sys.argv[1:5]
    lo_i, hi_i = (int(v) for v in sys.argv[5].split(":")) if len(sys.argv) > 5 else (0, 1 << 30)
    want = set(classes.split(","))
    lines = open(src).read().split("\n")
    # one VGPR above the kernel's allocation, for the rewrites that need a temporary: next_free_vgpr / accum_offset / .vgpr_count + 4
    in_desc = False
    for i, ln in enumerate(lines):
        if ln.strip().startswith(".amdhsa_kernel"):
            in_desc = kernel in ln
        m = re.match(r"(\s*\.amdhsa_(next_free_vgpr|accum_offset)\s+)(\d+)", ln)
        if in_desc and m:
            if m.group(2) == "next_free_vgpr":
                SCRATCH = "v%d" % int(m.group(3))
            lines[i] = "%s%d" % (m.group(1), int(m.group(3)) + 4)
        if in_desc and ln.strip().startswith(".end_amdhsa_kernel"):
            in_desc = False
    in_md = False
    for i, ln in enumerate(lines):
        if ln.strip().startswith(".name:"):
            in_md = kernel in ln
        m = re.match(r"(\s*\.vgpr_count:\s+)(\d+)", ln)
        if in_md and m:
            lines[i] = "%s%d" % (m.group(1), int(m.group(2)) + 4)
    assert SCRATCH, "kernel descriptor not found"
    inside = False
    met, done = {}, {}
    seq = 0
    res = []
    pat = re.compile(r"^\s*(v_pk_(?:add|mul|fma)_f32)\s+([^,]+),\s*(.*)$")
    for ln in lines:
        if re.match(r"^_Z\S*:", ln):
            inside = kernel in ln
        m = pat.match(ln) if inside else None
        if not m:
            res.append(ln)
            continue
        mn, dst, rest = m.group(1), m.group(2).strip(), m.group(3)
        rest = rest.split(";")[0].strip()
        mm = re.search(r"\s(op_sel|op_sel_hi|neg_lo|neg_hi):", " " + rest)
        ops, mods = (rest[: mm.start()].strip(), rest[mm.start():].strip()) if mm else (rest, "")
        srcs = [s.strip() for s in ops.split(",")]
        cls = classify(mn, srcs, mods)
        met[cls] = met.get(cls, 0) + 1
        if "all" in want or cls in want or ("addmod" in want and cls in ("addneg", "addsel")):
            if lo_i <= seq < hi_i:
                res.extend(rewrite(mn, dst, srcs, mods))
                done[cls] = done.get(cls, 0) + 1
                seq += 1
                continue
            seq += 1
        res.append(ln)
    open(out, "w").write("\n".join(res))
    print("packed fp32 in %s: met %s, rewrote %s" % (kernel, dict(sorted(met.items())), dict(sorted(done.items()))))


if __name__ == "__main__":
    main()
