#!/usr/bin/env python3
"""unifdef for this tree's compile-time knobs: fixes the given macros at a value and removes the preprocessor branches that
depend on them (measured experiments become the code or go).

    python tools/resolve_knobs.py file.hip NAME=1 OTHER=0 UNDEFINED= [--plain KEEP1,KEEP2]

NAME=v     every  #if NAME / #if !NAME / #if NAME >= n / #ifdef NAME / #ifndef NAME ... [#else ...] #endif  is replaced by the branch
           that v selects, and the  #ifndef NAME / #define NAME v / #endif  default block is deleted (NAME= : the macro is undefined)
--plain    default blocks of these macros lose their #ifndef guard and stay as plain #define constants
"""
import re
import sys


def evaluate(expr, vals):
    expr = expr.split("//")[0].strip()
    m = re.fullmatch(r"(!?)\s*(\w+)", expr)
    if m and m.group(2) in vals:
        v = vals[m.group(2)]
        t = bool(int(v)) if v != "" else False
        return (not t) if m.group(1) else t
    m = re.fullmatch(r"(\w+)\s*(>=|<=|==|>|<|!=)\s*(\d+)", expr)
    if m and m.group(1) in vals:
        return eval(f"{int(vals[m.group(1)] or 0)} {m.group(2)} {m.group(3)}")
    return None


def main():
    path = sys.argv[1]
    vals, plain = {}, set()
    args = sys.argv[2:]
    while args:
        a = args.pop(0)
        if a == "--plain":
            plain = set(args.pop(0).split(","))
        else:
            k, v = a.split("=", 1)
            vals[k] = v
    lines = open(path).read().split("\n")
    out = []
    # stack entries: [resolved?, taking?, seen_else] ; unresolved blocks pass through
    stack = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        emit = all(t for r, t, _ in stack if r)
        m = re.match(r"#\s*(if|ifdef|ifndef)\b(.*)", s)
        if m:
            kind, rest = m.group(1), m.group(2)
            # default block  #ifndef NAME / #define NAME ... / #endif
            name = rest.split("//")[0].strip() if kind == "ifndef" else None
            if name and i + 2 < len(lines) and re.match(r"#\s*define\s+%s\b" % re.escape(name), lines[i + 1].strip()) and lines[i + 2].strip().startswith("#endif"):
                if name in vals:
                    i += 3
                    continue
                if name in plain:
                    if emit:
                        out.append(lines[i + 1])
                        tail = lines[i + 2].strip()[len("#endif"):].strip()
                        if tail:
                            out.append(" " * 0 + tail)
                    i += 3
                    continue
            if kind == "if":
                r = evaluate(rest, vals)
            else:
                nm = rest.split("//")[0].strip()
                r = None
                if nm in vals:
                    defined = vals[nm] != ""
                    r = defined if kind == "ifdef" else not defined
            if r is None:
                stack.append([False, True, False])
                if emit:
                    out.append(ln)
            else:
                stack.append([True, r, False])
            i += 1
            continue
        if re.match(r"#\s*else\b", s) and stack:
            top = stack[-1]
            if top[0]:
                top[1] = not top[1]
            elif emit:
                out.append(ln)
            i += 1
            continue
        if re.match(r"#\s*elif\b", s) and stack and stack[-1][0]:
            raise SystemExit(f"{path}:{i + 1}: #elif on a resolved condition is not supported")
        if re.match(r"#\s*endif\b", s) and stack:
            top = stack.pop()
            if not top[0] and all(t for r, t, _ in stack if r):
                out.append(ln)
            i += 1
            continue
        if emit:
            out.append(ln)
        i += 1
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
