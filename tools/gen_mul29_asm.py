#!/usr/bin/env python3
"""Writes zkevm-circuits_amd/csrc/mul29_asm.hip.hpp (and, for tools/ubench.hip, the same text as tools/mul29_asm.hip.hpp): the 9 x 29-bit
Montgomery products of ff29.hip.hpp as ONE asm statement each (round 5).  Same column sums, same order, same results as the C forms
(ff29.hip.hpp: mul29_c, mul29_ub_c, sqr29_c, mul2add29_c) -- what changes is what the compiler adds around them: with one statement per
column part (ZK_MAD_CHAIN = 2) a product carries 162 multiply-adds in ~260 instructions (register copies for the tied accumulator
operand, an s_nop behind every statement, 64-bit shifts split in two); written out whole it is 162 + 9 (v_mul_lo) + 26 (v_and) +
17 (v_lshrrev_b64) + 1.

The 64-bit column accumulator is the fixed pair v[AL:AH] (inline asm cannot name the low half of a 64-bit operand, and
v_mul_lo_u32 / v_and_b32 read exactly that); it is declared clobbered.

    python tools/gen_mul29_asm.py > zkevm-circuits_amd/csrc/mul29_asm.hip.hpp
"""
AL, AH = 30, 31
ACC = f"v[{AL}:{AH}]"
LO = f"v{AL}"


def product(kind):
    """kind: 'vv' (mul29), 'vs' (mul29_ub: b in scalar registers), 'sq' (sqr29: a and the doubled limbs), '2' (mul2add29)."""
    # operand numbering: 0..8 = t (out, early clobber; t_j holds m_j until column j + 8), then the inputs
    n = 9
    ops = {}
    cur = 9
    def block(name, cnt):
        nonlocal cur
        ops[name] = list(range(cur, cur + cnt))
        cur += cnt
    block("a", 9)
    if kind in ("vv", "vs"):
        block("b", 9)
    elif kind == "sq":
        block("a2", 9)
    else:
        block("b", 9); block("c", 9); block("d", 9)
    block("M", 9)
    block("INV", 1)
    block("MASK", 1)
    t = list(range(9))
    lines = []
    first = True
    def mad(x, y):
        nonlocal first
        src2 = "0" if first else ACC
        first = False
        lines.append(f"v_mad_u64_u32 {ACC}, vcc, %{x}, %{y}, {src2}")
    for k in range(17):
        lo, hi = (0, k) if k < 9 else (k - 8, 8)
        if kind in ("vv", "vs", "2"):
            for i in range(lo, hi + 1):
                mad(ops["a"][i], ops["b"][k - i])
            if kind == "2":
                for i in range(lo, hi + 1):
                    mad(ops["c"][i], ops["d"][k - i])
        else:
            i = lo
            while 2 * i < k:
                mad(ops["a2"][i], ops["a"][k - i])
                i += 1
            if k % 2 == 0:
                mad(ops["a"][k // 2], ops["a"][k // 2])
        # m_i * M_(k-i): i from lo (k < 9: 0 .. k-1; k >= 9: k-8 .. 8)
        for i in range(lo, (k - 1 if k < 9 else 8) + 1):
            mad(t[i], ops["M"][k - i])
        if k < 9:
            lines.append(f"v_mul_lo_u32 %{t[k]}, {LO}, %{ops['INV'][0]}")
            lines.append(f"v_and_b32 %{t[k]}, %{ops['MASK'][0]}, %{t[k]}")
            mad(t[k], ops["M"][0])
        else:
            lines.append(f"v_and_b32 %{t[k - 9]}, %{ops['MASK'][0]}, {LO}")
        lines.append(f"v_lshrrev_b64 {ACC}, 29, {ACC}")
    lines.append(f"v_mov_b32 %{t[8]}, {LO}")
    return lines, ops


def product_inplace(kind, on):
    """The 'vv' / 'vs' product with the RESULT IN THE REGISTERS OF ONE FACTOR (round 6, the evaluator's interpreter): limb j of that factor is
    last read in column j + 8 and result limb j is written in column j + 9, so the result can take its place -- x <- x * y without a
    copy at the end, which is what lets an interpreter keep its top-of-stack in FIXED registers across the cases of its switch.  The
    Montgomery multipliers m_j (alive from column j to j + 8) get nine scratch registers.  Same column sums in the same order.
    Operands: 0..8 = x (in / out), 9..17 = scratch, then the other factor, the modulus, INV, MASK.  on = 'a' or 'b': which factor x is."""
    x = list(range(9))
    m = list(range(9, 18))
    o = list(range(18, 27))
    a, b = (x, o) if on == "a" else (o, x)
    M = list(range(27, 36))
    INV, MASK = 36, 37
    lines = []
    first = True
    def mad(p, q):
        nonlocal first
        src2 = "0" if first else ACC
        first = False
        lines.append(f"v_mad_u64_u32 {ACC}, vcc, %{p}, %{q}, {src2}")
    for k in range(17):
        lo, hi = (0, k) if k < 9 else (k - 8, 8)
        for i in range(lo, hi + 1):
            mad(a[i], b[k - i])
        for i in range(lo, (k - 1 if k < 9 else 8) + 1):
            mad(m[i], M[k - i])
        if k < 9:
            lines.append(f"v_mul_lo_u32 %{m[k]}, {LO}, %{INV}")
            lines.append(f"v_and_b32 %{m[k]}, %{MASK}, %{m[k]}")
            mad(m[k], M[0])
        else:
            lines.append(f"v_and_b32 %{x[k - 9]}, %{MASK}, {LO}")
        lines.append(f"v_lshrrev_b64 {ACC}, 29, {ACC}")
    lines.append(f"v_mov_b32 %{x[8]}, {LO}")
    return lines


def emit_inplace(name, kind, on):
    lines = product_inplace(kind, on)
    body = "\\n\\t\"\n        \"".join(lines)
    oc = "s" if kind == "vs" else "v"
    out = []
    out.append(f"template <class P>\n__device__ __forceinline__ void {name}(F29<P>& x, const F29<P>& y) {{")
    out.append("    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;")
    out.append(f"    asm(\"{body}\"")
    out.append("        : " + ", ".join(f"\"+v\"(x.l[{i}])" for i in range(9)) + ",")
    out.append("          " + ", ".join(f"\"=&v\"(m{i})" for i in range(9)))
    out.append("        : " + ", ".join(f"\"{oc}\"(y.l[{i}])" for i in range(9)) + ",")
    out.append("          " + ", ".join(f"\"s\"(P::M({i}))" for i in range(9)) + ", \"s\"(P::INV), \"s\"(MASK29)")
    out.append(f"        : \"vcc\", \"v{AL}\", \"v{AH}\");")
    out.append("}")
    return "\n".join(out)


def emit(name, kind, sig, ins):
    lines, ops = product(kind)
    body = "\\n\\t\"\n        \"".join(lines)
    out = []
    out.append(f"template <class P>\n__device__ __forceinline__ F29<P> {name}({sig}) {{")
    out.append("    F29<P> t;")
    out.append(f"    asm(\"{body}\"")
    out.append("        : " + ", ".join(f"\"=&v\"(t.l[{i}])" for i in range(9)))
    out.append("        : " + ",\n          ".join(ins) + ",")
    out.append("          " + ", ".join(f"\"s\"(P::M({i}))" for i in range(9)) + ", \"s\"(P::INV), \"s\"(MASK29)")
    out.append(f"        : \"vcc\", \"v{AL}\", \"v{AH}\");")
    out.append("    return t;\n}")
    return "\n".join(out)


def vlist(v, c="v"):
    return ", ".join(f"\"{c}\"({v}[{i}])" for i in range(9))


def main():
    print("// GENERATED by tools/gen_mul29_asm.py -- do not edit.  The Montgomery products of ff29.hip.hpp as one asm statement each")
    print("// (gfx950): the same column sums in the same order as mul29_c / mul29_ub_c / sqr29_c / mul2add29_c of ff29.hip.hpp, bit-identical results.")
    print(f"// Column accumulator: the fixed pair v[{AL}:{AH}] (clobbered).  Included by ff29.hip.hpp inside namespace zk (device compilation, ZK_MUL_ASM).")
    print("#pragma once\n")
    print(emit("mul29_asm", "vv", "const F29<P>& a, const F29<P>& b", [vlist("a.l"), vlist("b.l")]))
    print()
    print(emit("mul29_ub_asm", "vs", "const F29<P>& a, const F29<P>& b", [vlist("a.l"), vlist("b.l", "s")]))
    print()
    print(emit("sqr29_asm_", "sq", "const F29<P>& a, const uint32_t (&a2)[9]", [vlist("a.l"), vlist("a2")]))
    print("template <class P>\n__device__ __forceinline__ F29<P> sqr29_asm(const F29<P>& a) {\n    uint32_t a2[9];\n#pragma unroll\n    for (int i = 0; i < 9; ++i) a2[i] = a.l[i] << 1;\n    return sqr29_asm_<P>(a, a2);\n}")
    print()
    print(emit("mul2add29_asm", "2", "const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d", [vlist("a.l"), vlist("b.l"), vlist("c.l"), vlist("d.l")]))
    print()
    print("// In-place forms (the evaluator's interpreter, csrc/quotient.hip): the result takes the registers of one factor.")
    print("// x <- mul29(x, y): x is the FIRST factor (limbs below 2^31), y normalised")
    print(emit_inplace("mul29_ipa_asm", "vv", "a"))
    print()
    print("// x <- mul29(y, x): x is the SECOND factor (normalised), y the first")
    print(emit_inplace("mul29_ipb_asm", "vv", "b"))
    print()
    print("// x <- mul29_ub(x, y): y wave-uniform, in scalar registers")
    print(emit_inplace("mul29_ub_ipa_asm", "vs", "a"))


if __name__ == "__main__":
    main()
