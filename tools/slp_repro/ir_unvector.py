"""Rewrite SELECTED kinds of vector instructions of an LLVM IR file (text) into scalar ones: each  %r = op <N x T> a, b  becomes
N extractelements per operand, N scalar ops and an insertelement chain whose last link keeps the name %r (so the numbering of the
unnamed values is unchanged).  Used by relower.py to find WHICH vector construct instruction selection gets wrong: `opt -passes=scalarizer`
rewrites all of them at once (and the result is correct), this rewrites one kind at a time.

    kinds: select icmp fneg fbin(fdiv) ibin(add mul lshr shl xor and or) zext bitcast shuffle farith(fadd fsub fmul)
"""
import re
import sys

OP = r"(%[\w.]+|zeroinitializer|poison|undef|splat \([^)]*\)|<[^<>]*>)"
NAME = r"(%[\w.]+)"
FLAGS = r"((?:(?:nuw|nsw|nneg|exact|disjoint|contract|fast|nnan|ninf|nsz|arcp|afn|reassoc) )*)"


def tmp(name):
    """a NAMED temporary derived from the (usually numbered) result name: %123 -> %u123"""
    return "%u" + name[1:]


def elems(name, n, ty, op, tag, out, indent):
    """scalar values of the vector operand `op`"""
    r = []
    for i in range(n):
        v = f"{tmp(name)}.{tag}{i}"
        out.append(f"{indent}{v} = extractelement <{n} x {ty}> {op}, i32 {i}")
        r.append(v)
    return r


def build(name, n, ty, vals, out, indent):
    prev = "poison"
    for i, v in enumerate(vals):
        dst = name if i == n - 1 else f"{tmp(name)}.i{i}"
        out.append(f"{indent}{dst} = insertelement <{n} x {ty}> {prev}, {ty} {v}, i32 {i}")
        prev = dst


def rewrite(lines, kinds, keep=None):
    """keep = (first, last): the float-arithmetic instructions number first .. last-1 (in file order) are LEFT vectorized"""
    out, count = [], {}
    nth = 0
    for line in lines:
        m = None
        ind = re.match(r"\s*", line).group(0)
        if keep is not None and re.match(rf"\s*{NAME} = (fadd|fsub|fmul) {FLAGS}<(\d+) x (\w+)> {OP}, {OP}\s*$", line):
            nth += 1
            if keep[0] <= nth - 1 < keep[1]:
                out.append(line)
                count["kept"] = count.get("kept", 0) + 1
                continue
        if "select" in kinds and (m := re.match(rf"\s*{NAME} = select {FLAGS}<(\d+) x i1> {OP}, <\d+ x (\w+)> {OP}, <\d+ x \w+> {OP}\s*$", line)):
            r, fl, n, c, ty, a, b = m.groups()
            n = int(n)
            cs, as_, bs = elems(r, n, "i1", c, "c", out, ind), elems(r, n, ty, a, "a", out, ind), elems(r, n, ty, b, "b", out, ind)
            vals = []
            for i in range(n):
                out.append(f"{ind}{tmp(r)}.s{i} = select {fl}i1 {cs[i]}, {ty} {as_[i]}, {ty} {bs[i]}")
                vals.append(f"{tmp(r)}.s{i}")
            build(r, n, ty, vals, out, ind)
            kind = "select"
        elif "icmp" in kinds and (m := re.match(rf"\s*{NAME} = (icmp|fcmp) {FLAGS}(\w+) <(\d+) x (\w+)> {OP}, {OP}\s*$", line)):
            r, opc, fl, pred, n, ty, a, b = m.groups()
            n = int(n)
            as_, bs = elems(r, n, ty, a, "a", out, ind), elems(r, n, ty, b, "b", out, ind)
            vals = []
            for i in range(n):
                out.append(f"{ind}{tmp(r)}.s{i} = {opc} {fl}{pred} {ty} {as_[i]}, {bs[i]}")
                vals.append(f"{tmp(r)}.s{i}")
            build(r, n, "i1", vals, out, ind)
            kind = "icmp"
        elif "fneg" in kinds and (m := re.match(rf"\s*{NAME} = fneg {FLAGS}<(\d+) x (\w+)> {OP}\s*$", line)):
            r, fl, n, ty, a = m.groups()
            n = int(n)
            as_ = elems(r, n, ty, a, "a", out, ind)
            vals = []
            for i in range(n):
                out.append(f"{ind}{tmp(r)}.s{i} = fneg {fl}{ty} {as_[i]}")
                vals.append(f"{tmp(r)}.s{i}")
            build(r, n, ty, vals, out, ind)
            kind = "fneg"
        elif (m := re.match(rf"\s*{NAME} = (fdiv|fadd|fsub|fmul|add|sub|mul|lshr|ashr|shl|xor|and|or) {FLAGS}<(\d+) x (\w+)> {OP}, {OP}\s*$", line)) and (
                ("fbin" in kinds and m.group(2) == "fdiv") or ("farith" in kinds and m.group(2) in ("fadd", "fsub", "fmul")) or
                ("ibin" in kinds and m.group(2) in ("add", "sub", "mul", "lshr", "ashr", "shl", "xor", "and", "or"))):
            r, opc, fl, n, ty, a, b = m.groups()
            n = int(n)
            as_, bs = elems(r, n, ty, a, "a", out, ind), elems(r, n, ty, b, "b", out, ind)
            vals = []
            for i in range(n):
                out.append(f"{ind}{tmp(r)}.s{i} = {opc} {fl}{ty} {as_[i]}, {bs[i]}")
                vals.append(f"{tmp(r)}.s{i}")
            build(r, n, ty, vals, out, ind)
            kind = "fbin" if opc == "fdiv" else ("farith" if opc[0] == "f" else "ibin")
        elif ("zext" in kinds or "bitcast" in kinds) and (m := re.match(rf"\s*{NAME} = (zext|sext|bitcast|uitofp|sitofp) {FLAGS}<(\d+) x (\w+)> {OP} to <(\d+) x (\w+)>\s*$", line)) and (
                m.group(4) == m.group(7)) and (("bitcast" in kinds and m.group(2) == "bitcast") or ("zext" in kinds and m.group(2) != "bitcast")):
            r, opc, fl, n, ty, a, n2, ty2 = m.groups()
            n = int(n)
            as_ = elems(r, n, ty, a, "a", out, ind)
            vals = []
            for i in range(n):
                out.append(f"{ind}{tmp(r)}.s{i} = {opc} {fl}{ty} {as_[i]} to {ty2}")
                vals.append(f"{tmp(r)}.s{i}")
            build(r, n, ty2, vals, out, ind)
            kind = "bitcast" if opc == "bitcast" else "zext"
        elif "shuffle" in kinds and (m := re.match(rf"\s*{NAME} = shufflevector <(\d+) x (\w+)> {OP}, <\d+ x \w+> {OP}, <(\d+) x i32> {OP}\s*$", line)):
            r, n, ty, a, b, nm, mask = m.groups()
            n, nm = int(n), int(nm)
            if mask == "zeroinitializer":
                idx = [0] * nm
            elif mask.startswith("splat"):
                idx = [int(re.search(r"i32 (-?\d+)", mask).group(1))] * nm
            else:
                idx = [None if "poison" in t or "undef" in t else int(t.split()[-1]) for t in mask.strip("<>").split(",")]
            prev, last = "poison", max(i for i, x in enumerate(idx) if x is not None)
            for i, x in enumerate(idx):
                if x is None:
                    continue
                src, lane = (a, x) if x < n else (b, x - n)
                out.append(f"{ind}{tmp(r)}.e{i} = extractelement <{n} x {ty}> {src}, i32 {lane}")
                dst = r if i == last else f"{tmp(r)}.i{i}"
                out.append(f"{ind}{dst} = insertelement <{nm} x {ty}> {prev}, {ty} {tmp(r)}.e{i}, i32 {i}")
                prev = dst
            kind = "shuffle"
        else:
            out.append(line)
            continue
        count[kind] = count.get(kind, 0) + 1
    return out, count


if __name__ == "__main__":
    src, dst, kinds = sys.argv[1], sys.argv[2], set(sys.argv[3].split(","))
    only = sys.argv[4] if len(sys.argv) > 4 else None  # restrict to the function whose define line contains this text
    keep = tuple(int(x) for x in sys.argv[5].split(":")) if len(sys.argv) > 5 else None
    lines = open(src).read().split("\n")
    if only:
        start = next(i for i, l in enumerate(lines) if l.startswith("define") and only in l)
        end = next(i for i in range(start, len(lines)) if lines[i] == "}")
        body, count = rewrite(lines[start:end], kinds, keep)
        lines = lines[:start] + body + lines[end:]
    else:
        lines, count = rewrite(lines, kinds, keep)
    open(dst, "w").write("\n".join(lines))
    print(count)
